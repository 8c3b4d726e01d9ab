/* Stand-in for R's R_ext/Rdynload.h (see tests/r_stub/README.md; implemented by rstub.c). */
#ifndef R_STUB_RDYNLOAD_H
#define R_STUB_RDYNLOAD_H
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const void *cRoutines, const R_CallMethodDef *callRoutines, const void *fortranRoutines,
                       const void *externalRoutines);
int R_useDynamicSymbols(DllInfo *info, int value);
#endif
