/* Stand-in for R's R.h (see tests/r_stub/README.md; implemented by rstub.c). */
#ifndef R_STUB_R_H
#define R_STUB_R_H
#include <math.h>
#include <stddef.h>
void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
int R_IsNA(double x);
char *R_alloc(size_t n, int size); /* R_ext/Memory.h: freed by R at the end of the .Call */
#define ISNA(x) R_IsNA(x)
#define ISNAN(x) (isnan(x) != 0)
#endif
