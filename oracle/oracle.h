/*
 * oracle/oracle.h — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * CPU restatement of Chicdiff's differential-testing core (SURVEY.md §8a rows
 * a1-a9).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (chicdiff_amd/, include/chicdiff_hip.h) never
 * does.  PARITY UNPINNED at the DESeq2 boundary — see oracle/README.md.
 *
 * Matrix layout everywhere: column-major n x S (= sample-major: element (i, j) at
 * [j*n + i]), i.e. exactly R's INTEGER(mat)/REAL(mat) for the matrices built at
 * chicdiff.R:1551-1553 and :1583.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double minDisp;      /* 1e-8   estimateDispersions default                        */
    double dispTol;      /* 1e-6                                                    */
    double kappa0;       /* 1.0                                                     */
    int32_t maxit;       /* 100    fitDisp line-search iterations                     */
    int32_t betaMaxit;   /* 100    fitBeta IRLS iterations                            */
    double betaTol;      /* 1e-8                                                    */
    double minmu;        /* 0.5                                                     */
    double outlierSD;    /* 2.0                                                     */
    double dispPriorVar; /* NaN = estimate (closed form when m-p > 3; see .c)        */
    int32_t nthreads;    /* OpenMP threads over rows (1 = DESeq2-like single thread)  */
    int32_t _pad;
    double trendCoef[2]; /* NaN = fit; else alpha(mu) = c0 + c1/mu as given (DESeq2: dispersionFunction<-) */
    int32_t fitType;     /* 0 parametric (a failed fit is replaced by the local one, as DESeq2 does); 1 "mean": dispFit =
                          * mean(dispGeneEst[> 10 minDisp], trim = .001); 2 "local": locfit (locfit_oracle.c) */
    int32_t noLocalSubstitute; /* 1: a failed parametric fit is reported (ORACLE_ST_TREND_FAILED) and the coefficients reached are kept */
    double varLogDispEsts; /* NaN = estimate (mad^2 of the log residuals); else as given: lets a SLICE of a larger fit be checked with
                              all three global scalars (trend, prior variance, this) pinned to the whole fit's */
    const double *dispFitIn; /* NULL = fit the trend; else the fitted dispersions themselves, one per row (DESeq2: mcols(dds)$dispFit given):
                              lets everything downstream of a LOCAL trend be checked under the trend of the fit it is compared with */
    double xim;            /* NaN = estimate (mean_j 1 / colMeans(nf)_j, momentsDispEstimate); else as given — the fourth global scalar: it
                              enters every row's start value alpha_init = min(roughDisp, momentsDisp), hence where the search stops */
} oracle_nbglm_opts;

void oracle_nbglm_default_opts(oracle_nbglm_opts *o);

/* status bits returned in out->status */
#define ORACLE_ST_TREND_FAILED 1     /* parametric trend fit failed (DESeq2 would switch to locfit) */
#define ORACLE_ST_PRIORVAR_MC 2      /* m-p <= 3: dispPriorVar matched by simulation as DESeq2 does (set.seed(2)) */
#define ORACLE_ST_BETA_NONCONV 4     /* some rows hit betaMaxit (DESeq2 would call optim)           */
#define ORACLE_ST_ALLZERO_ROWS 8     /* some rows are all-zero (NA outputs)                         */
#define ORACLE_ST_TREND_LOCAL 16      /* dispFit comes from the local regression (fitType "local", or DESeq2's substitute for a failed parametric fit) */

typedef struct {
    /* per-row, length n (any pointer may be NULL) */
    double *baseMean, *baseVar;
    int32_t *allZero;
    double *dispInit;      /* alpha_init (A2.4)                                   */
    double *dispGeneEst;
    int32_t *dispGeneIter;
    double *dispFit, *dispMAP, *dispersion;
    int32_t *dispIter, *dispOutlier;
    double *beta0, *beta1; /* log2 scale: Intercept, condition_B_vs_A (beta1 NaN for ~1) */
    double *se0, *se1;
    double *stat, *pvalue; /* Wald statistic / p of the last coefficient           */
    double *deviance;
    int32_t *betaConv, *betaIter;
    double *maxCooks;      /* NaN unless some group has >= 3 samples               */
    int32_t *cooksArgmax;  /* which.max(cooks[i,]) - 1 over all samples; -1 if not computed */
    double *mu;            /* n x S column-major, fitted mean of the Wald fit      */
    /* scalars */
    double trendCoef[2];   /* asymptDisp, extraPois                              */
    double varLogDispEsts, dispPriorVar, sumDeviance;
    int32_t trendOuterIter, status;
} oracle_nbglm_out;

/* A1 / a5: DESeq2 estimateSizeFactorsForMatrix (median of ratios). Returns 0 or <0 on error. */
int oracle_size_factors(const int32_t *counts, int64_t n, int32_t S, double *sf);

/* a6 + a7: estimateDispersions + nbinomWaldTest for design ~group (two levels) or ~1 (all group==0). */
int oracle_nbglm_fit(const int32_t *counts, const double *nf, int64_t n, int32_t S,
                     const int32_t *group, const oracle_nbglm_opts *opts, oracle_nbglm_out *out);

/* binary128 twin of the dispersion line search (A2.6-A2.7, A4) for the listed rows: the referee when the GPU and the
 * double-precision restatement disagree on a noise-decided row.  stage 0 = gene-wise (needs dispInit), stage 1 = MAP
 * (needs dispGene, dispFit, dispPriorVar). */
int oracle_arbitrate_disp(const int32_t *counts, const double *nf, int64_t n, int32_t S, const int32_t *group,
                          const int64_t *rows, int64_t nrows, int32_t stage, const double *dispInit, const double *dispGene,
                          const double *dispFit, double dispPriorVar, const oracle_nbglm_opts *opts, double *out);

/* One row's IRLS iterates and conv_test values, without the stopping rule (the referee for rows whose two sides stop after a
 * different number of steps; see nbglm_oracle.c). */
int oracle_irls_trace(const int32_t *counts, const double *nf, int64_t n, int32_t S, const int32_t *group, int64_t row,
                      double alpha, int32_t steps, double *b0_out, double *b1_out, double *conv_out);

/* pieces exported for unit tests ------------------------------------------------------------- */
double oracle_log_posterior(double log_alpha, const double *y, const double *mu, const int32_t *group,
                            int32_t S, int32_t p, double prior_mean, double prior_sigmasq, int32_t use_prior);
double oracle_dlog_posterior(double log_alpha, const double *y, const double *mu, const int32_t *group,
                             int32_t S, int32_t p, double prior_mean, double prior_sigmasq, int32_t use_prior);
/* Gamma-identity trend fit on (means, disps) pairs; returns 0 ok / nonzero failed */
int oracle_parametric_dispersion_fit(const double *means, const double *disps, int64_t n, double coefs[2],
                                     int32_t *outer_iter);
double oracle_median(double *x, int64_t n); /* sorts x in place */

/* a2: window sums. frag_* are long-form per-fragment values, column-major nfrag x S;
 * region_ptr[n+1] CSR offsets into the fragment axis (fragments of one region contiguous).
 * N: int32 sums (bit exact); FullMean: fp64 NA(NaN)-propagating sums in fragment order.
 * chicdiff.R:1540-1547. */
int oracle_window_sums(const int32_t *fragN, const double *fragFullMean, int64_t nfrag, int32_t S,
                       const int64_t *region_ptr, int64_t n, int32_t *N, double *FullMean);

/* a4: offsets. chicdiff.R:1583-1589 (M3), :1614-1615 (nsf), :1635-1638 / :1666-1669 (theta mix).
 * out = sc(theta) (n x S).  theta = 0 reproduces normFactorsM3 renormalised (a no-op). */
int oracle_offsets(const double *FullMean, const double *sizeFactors, int64_t n, int32_t S, double theta,
                   double *out);

/* a1: count join. keys sorted ascending (baitID<<32 | otherEndID), one table per sample.
 * chicdiff.R:843-858: left join RU x chinput on (baitID, otherEndID), NA -> 0. */
int oracle_count_join(const int32_t *ru_bait, const int32_t *ru_oe, int64_t nru, const int64_t *keys,
                      const int32_t *vals, int64_t nkeys, int32_t *out);

/* a3: per-fragment background (offset ingredients), chicdiff.R:628-703 + Chicago .estimateBMean /
 * .distFun (Appendix B): for every RU row (bait, oe) and replicate s
 *   distSign = round(((start+end)_oe - (start+end)_bait)/2)                       (:648)
 *   Bmean = s_j[bait] * s_i[oe] * f(|distSign|), s_i NA -> 1, NA when s_j NA      (:659-672, :701-702)
 *   Tmean = T[tblb[bait]][tlb[oe]]; tlb NA & tblb known -> min over tlb; else NA  (:676-692)
 *   FullMean = Bmean + Tmean                                                      (:896)
 * Dense lookup tables over fragment ids [id_min, id_min+nid): midsum (int64), and per replicate
 * sj, si (NaN = absent), tblb, tlb (-1 = NA), T (ntblb x ntlb row-major, NaN = missing).
 * distfun per replicate: cubic[4], head[2], tail[2], obs_min, obs_max (10 doubles).  Outputs [S][nru]. */
int oracle_fragment_background(const int32_t *bait, const int32_t *oe, int64_t nru, int32_t id_min, int32_t nid,
                               const int64_t *midsum, int32_t S, const double *sj, const double *si,
                               const int32_t *tblb, const int32_t *tlb, const double *T, int32_t ntblb, int32_t ntlb,
                               const double *distfun, double *bmean, double *tmean, double *fullmean);

/* a9 helpers: BH adjustment (p.adjust(method="BH") on the non-NaN entries; NaN stays NaN) */
int oracle_bh_adjust(const double *p, int64_t n, double *padj);

/* A4, residual d.f. <= 3: simulation-matched prior variance (prior_mc_oracle.c) */
int oracle_prior_mc_bin(double x);
double oracle_prior_var_mc(const double *obs_counts /*[40]*/, int df);
int oracle_prior_mc_table(int df, double *out /*[200*40]*/);
/* loess(y ~ x, span, degree = 2, family = "gaussian", surface = "interpolate", cell) on sorted distinct x[n],
 * predicted at z[nz] (inside the range of x); returns the number of k-d tree vertices or < 0 */
int oracle_loess_interp(const double *x, const double *y, int n, double span, double cell, const double *z, int nz,
                        double *out);

/* R's default RNG chain (r_rng.c): set.seed + Mersenne-Twister + inversion normals + exp_rand + rgamma */
struct oracle_r_rng;
size_t oracle_r_rng_size(void);
void oracle_r_set_seed(struct oracle_r_rng *r, uint32_t seed);
double oracle_r_unif_rand(struct oracle_r_rng *r);
double oracle_r_norm_rand(struct oracle_r_rng *r);
double oracle_r_exp_rand(struct oracle_r_rng *r);
double oracle_r_rgamma(struct oracle_r_rng *r, double a, double scale);
double oracle_r_rchisq(struct oracle_r_rng *r, double df);
double oracle_r_qnorm(double p);
void oracle_r_runif(uint32_t seed, int64_t n, double *out);
void oracle_r_rnorm(uint32_t seed, int64_t n, double *out);
void oracle_r_rexp(uint32_t seed, int64_t n, double *out);
void oracle_r_rgamma_vec(uint32_t seed, double shape, double scale, int64_t n, double *out);

/* f3: IHW application side, chicdiff.R:2038-2049 (see chicdiff_oracle.c) */
int oracle_ihw_apply(const double *avDist, const double *pvalue, int64_t n, const double *breaks, const double *avWeights,
                     int32_t ngroups, int32_t *group, double *weight, double *wp, double *wpadj);

/* f4: getRegionUniverse window mode, chicdiff.R:353-426 (see chicdiff_oracle.c) */
int64_t oracle_region_universe(const int32_t *bait, const int32_t *oe, int64_t n, int32_t s, const int32_t *chr_of,
                               int32_t maxfrag, int64_t *region_ptr, int32_t *ru_bait, int32_t *ru_region, int32_t *ru_oe);

/* IHWcorrection's covariate: per-region mean of distSign, chicdiff.R:1965-1967 + :868-882 (see chicdiff_oracle.c) */
int oracle_region_avdist(const int32_t *ru_bait, const int32_t *ru_oe, const int64_t *region_ptr, int64_t n, int32_t id_min,
                         int32_t nid, const int64_t *midsum, const int32_t *chr, double *avDist);

/* a1 without chinput files: Reduce(merge) over the replicates' Chicago tables, chicdiff.R:774-807 (see chicdiff_oracle.c) */
int oracle_count_join_inner(const int32_t *ru_bait, const int32_t *ru_oe, int64_t nru, int32_t S, const int64_t *const *keys,
                            const int32_t *const *vals, const int64_t *nkeys, int32_t *out);

#ifdef __cplusplus
}
#endif
/* locfit_oracle.c: DESeq2 localDispersionFit (locfit defaults: alpha 0.7, deg 2, tricube, rbox(cut 0.8), Hermite) */
#define ORACLE_LOCFIT_MAXV 100 /* locfit's maxk */
typedef struct {
    int32_t nv, _pad;
    double x[ORACLE_LOCFIT_MAXV], h[ORACLE_LOCFIT_MAXV], f[ORACLE_LOCFIT_MAXV], d[ORACLE_LOCFIT_MAXV]; /* vertices, ascending x */
} oracle_locfit;
int oracle_locfit_build(const double *x, const double *y, const double *w, int64_t n, double alpha, double cut, oracle_locfit *fit);
double oracle_locfit_eval(const oracle_locfit *fit, double x);
int oracle_local_dispersion_fit(const double *means, const double *disps, int64_t n, oracle_locfit *fit);

#endif
