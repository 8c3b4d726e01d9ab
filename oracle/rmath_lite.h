/*
 * oracle/rmath_lite.h — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * Scalar special functions the reference path evaluates through R's nmath
 * (dnbinom_mu, pnorm, digamma, trigamma, lgamma).  R is absent from
 * /root/reference and from this image, so these are restatements of the
 * published algorithms (Loader 2000 saddle-point binomial; Cody 1969 normal
 * CDF; Amos-style asymptotic psi).  PARITY UNPINNED at the DESeq2 boundary:
 * see oracle/README.md.
 */
#ifndef ORACLE_RMATH_LITE_H
#define ORACLE_RMATH_LITE_H

#ifdef __cplusplus
extern "C" {
#endif

/* log dnbinom(x; size, mu) — R nmath dnbinom_mu(x, size, mu, give_log=TRUE). */
double oracle_dnbinom_mu_log(double x, double size, double mu);
/* 2*pnorm(-|z|) — DESeq2 nbinomWaldTest: 2*pnorm(abs(stat), lower.tail=FALSE). */
double oracle_pnorm_two_sided(double z);
/* pnorm(z, lower.tail=TRUE) */
double oracle_pnorm(double z);
double oracle_digamma(double x);
double oracle_trigamma(double x);
double oracle_lgamma(double x);
/* Loader's pieces, exported for unit tests */
double oracle_stirlerr(double n);
double oracle_bd0(double x, double np);

#ifdef __cplusplus
}
#endif
#endif
