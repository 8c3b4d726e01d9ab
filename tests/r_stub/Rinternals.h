/* Stand-in for R's Rinternals.h — the slice of R's C API that r/src/chicdiff_hip_shim.c uses, written from "Writing R
 * Extensions" (sections 5 and 6).  Since round 5 the declarations are backed by a small FUNCTIONAL implementation (rstub.c) so
 * that the shim can be linked and every .Call routine executed on the GPU box: see tests/r_stub/README.md for what that does
 * and does not show.  It pins nothing about R itself. */
#ifndef R_STUB_RINTERNALS_H
#define R_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define NILSXP 0
#define SYMSXP 1
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol;
extern double R_NaReal;
extern int R_NaInt;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt
int TYPEOF(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int *INTEGER(SEXP x);
double *REAL(SEXP x);
SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_install(const char *name);
SEXP Rf_mkChar(const char *s);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP Rf_getAttrib(SEXP x, SEXP name);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_length(SEXP x);
Rboolean Rf_isInteger(SEXP x);
Rboolean Rf_isReal(SEXP x);
Rboolean Rf_isNull(SEXP x);
Rboolean Rf_isString(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
const char *CHAR(SEXP x);
SEXP Rf_ScalarReal(double x);
SEXP Rf_ScalarInteger(int x);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
SEXP R_ExternalPtrTag(SEXP s);
SEXP R_ExternalPtrProtected(SEXP s);
void R_ClearExternalPtr(SEXP s);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
#endif
