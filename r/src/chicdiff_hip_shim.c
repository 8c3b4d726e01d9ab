/*
 * chicdiff_hip_shim.c — the `.Call` shim between R and libchicdiff_hip.so (include/chicdiff_hip.h).
 *
 * R is absent from the authoring image and the GPU box, so this file is NOT built or run there;
 * tests/test_r_shim.py only checks that it compiles against a declaration-only stand-in for R's headers
 * (tests/r_stub/, a syntax / prototype check that pins nothing).  Build where R exists:
 *     R CMD SHLIB chicdiff_hip_shim.c -I../../include -L../../chicdiff_amd/lib -lchicdiff_hip
 *
 * One routine per step of DESeq2Wrap() (chicdiff.R:1494-1777) that moved to the GPU; r/R/DESeq2Wrap_hip.R calls them:
 *   chicdiff_hip_open          —                                   one context per R process and device
 *   chicdiff_hip_upload / _download / _release                      device-resident vectors (external pointers)
 *   chicdiff_hip_window_sums   chicdiff.R:1540-1556                 N / FullMean window sums -> n x S matrices
 *   chicdiff_hip_size_factors  chicdiff.R:1561-1562                 estimateSizeFactors
 *   chicdiff_hip_offsets       chicdiff.R:1583-1589, 1635-1638      normFactorsM3 / sc(theta) / size factors
 *   chicdiff_hip_theta_grid    chicdiff.R:1619-1662                 deviances of the design ~1 fits
 *   chicdiff_hip_wald_test     chicdiff.R:1557-1674 + 1721/1730/1739  size factors -> sc -> estimateDispersions ->
 *                                                                   nbinomWaldTest -> results() in one call
 *   chicdiff_hip_fit           chicdiff.R:1573-1574 etc.            estimateDispersions + nbinomWaldTest on host matrices
 *   chicdiff_hip_padj          DESeq2 results(): independent filtering + BH
 *   chicdiff_hip_ihw_apply     chicdiff.R:2038-2049
 *   chicdiff_hip_region_universe chicdiff.R:376-401
 *
 * Conventions: R matrices are column-major = the library's sample-major layout, so INTEGER()/REAL() pass through
 * untransposed.  Every device allocation is owned by an external pointer with a finalizer from the moment it
 * exists, so an Rf_error() (a longjmp) anywhere cannot leak device memory; temporaries are released eagerly on the
 * normal path.  NA_real_ is a NaN, which is what the library treats as NA; all-zero rows come back NaN and the R
 * wrapper turns them into NA.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "chicdiff_hip.h"

/* ---- context ------------------------------------------------------------------------------------------------ */
static void ctx_finalizer(SEXP p) {
    chicdiff_hip_ctx *c = (chicdiff_hip_ctx *)R_ExternalPtrAddr(p);
    if (c) chicdiff_hip_destroy(c);
    R_ClearExternalPtr(p);
}
static chicdiff_hip_ctx *ctx_of(SEXP p) {
    if (TYPEOF(p) != EXTPTRSXP || R_ExternalPtrTag(p) != Rf_install("chicdiff_hip_ctx")) Rf_error("chicdiff_hip: not a context");
    chicdiff_hip_ctx *c = (chicdiff_hip_ctx *)R_ExternalPtrAddr(p);
    if (!c) Rf_error("chicdiff_hip: the context has been closed");
    return c;
}
/* .Call(chicdiff_hip_open, device) -> context (one R process per GPU: pass that process's device index) */
SEXP chicdiff_hip_open(SEXP device) {
    chicdiff_hip_ctx *c = NULL;
    if (chicdiff_hip_create(&c, Rf_asInteger(device))) Rf_error("chicdiff_hip: %s", chicdiff_hip_last_error(NULL));
    SEXP p = PROTECT(R_MakeExternalPtr(c, Rf_install("chicdiff_hip_ctx"), R_NilValue));
    R_RegisterCFinalizerEx(p, ctx_finalizer, TRUE);
    UNPROTECT(1);
    return p;
}
SEXP chicdiff_hip_close(SEXP ctx) {
    if (TYPEOF(ctx) == EXTPTRSXP) ctx_finalizer(ctx);
    return R_NilValue;
}

/* ---- device vectors ----------------------------------------------------------------------------------------- */
typedef struct {
    void *d;
    int type;       /* INTSXP or REALSXP */
    R_xlen_t len;
} devbuf;
static void devbuf_finalizer(SEXP p) {
    devbuf *b = (devbuf *)R_ExternalPtrAddr(p);
    if (!b) return;
    SEXP ctx = R_ExternalPtrProtected(p); /* keeps the context alive as long as one of its buffers is */
    chicdiff_hip_ctx *c = TYPEOF(ctx) == EXTPTRSXP ? (chicdiff_hip_ctx *)R_ExternalPtrAddr(ctx) : NULL;
    if (c && b->d) chicdiff_hip_free(c, b->d);
    free(b);
    R_ClearExternalPtr(p);
}
static size_t elt_size(int type) { return type == INTSXP ? 4 : 8; }
/* a new device vector owned by an external pointer (returned PROTECTed: the caller counts it) */
static SEXP devbuf_new(SEXP ctx, int type, R_xlen_t len) {
    chicdiff_hip_ctx *c = ctx_of(ctx);
    devbuf *b = (devbuf *)calloc(1, sizeof(devbuf));
    if (!b) Rf_error("chicdiff_hip: out of memory");
    b->type = type;
    b->len = len;
    SEXP p = PROTECT(R_MakeExternalPtr(b, Rf_install("chicdiff_hip_buf"), ctx));
    R_RegisterCFinalizerEx(p, devbuf_finalizer, TRUE);
    if (len > 0 && chicdiff_hip_malloc(c, (uint64_t)len * elt_size(type), &b->d)) Rf_error("chicdiff_hip: %s", chicdiff_hip_last_error(c));
    return p;
}
static devbuf *devbuf_of(SEXP p, int type, R_xlen_t len, const char *what) {
    if (TYPEOF(p) != EXTPTRSXP || R_ExternalPtrTag(p) != Rf_install("chicdiff_hip_buf")) Rf_error("chicdiff_hip: %s is not a device vector", what);
    devbuf *b = (devbuf *)R_ExternalPtrAddr(p);
    if (!b) Rf_error("chicdiff_hip: %s has been released", what);
    if (b->type != type || (len >= 0 && b->len != len)) Rf_error("chicdiff_hip: %s has the wrong type or length", what);
    return b;
}
/* an argument that may be an R vector (uploaded into a temporary) or a device vector; returned PROTECTed */
static SEXP as_device(SEXP ctx, SEXP x, int type, R_xlen_t len, const char *what) {
    if (TYPEOF(x) == EXTPTRSXP) {
        devbuf_of(x, type, len, what);
        PROTECT(x);
        return x;
    }
    if (TYPEOF(x) != type || (len >= 0 && XLENGTH(x) != len)) Rf_error("chicdiff_hip: %s has the wrong type or length", what);
    SEXP p = devbuf_new(ctx, type, XLENGTH(x));
    devbuf *b = (devbuf *)R_ExternalPtrAddr(p);
    const void *src = type == INTSXP ? (const void *)INTEGER(x) : (const void *)REAL(x);
    if (b->len > 0 && chicdiff_hip_memcpy_h2d(ctx_of(ctx), b->d, src, (uint64_t)b->len * elt_size(type)))
        Rf_error("chicdiff_hip: %s", chicdiff_hip_last_error(ctx_of(ctx)));
    return p;
}
static void *dptr(SEXP p) { return ((devbuf *)R_ExternalPtrAddr(p))->d; }
/* eager release of a temporary made by as_device()/devbuf_new() (no-op for a caller-owned device vector) */
static void release_if_temp(SEXP p, SEXP original) {
    if (p != original) devbuf_finalizer(p);
}
static void check_rc(SEXP ctx, int rc, const char *where) {
    if (rc) Rf_error("%s: %s", where, chicdiff_hip_last_error(ctx_of(ctx)));
}
/* a fresh R vector with the contents of a device buffer (returned PROTECTed) */
static SEXP to_host(SEXP ctx, const void *d, int type, R_xlen_t len) {
    SEXP v = PROTECT(Rf_allocVector(type, len));
    void *dst = type == INTSXP ? (void *)INTEGER(v) : (void *)REAL(v);
    if (len > 0) check_rc(ctx, chicdiff_hip_memcpy_d2h(ctx_of(ctx), dst, d, (uint64_t)len * elt_size(type)), "chicdiff_hip download");
    return v;
}

/* .Call(chicdiff_hip_upload, ctx, x) -> device vector (integer or double) */
SEXP chicdiff_hip_upload(SEXP ctx, SEXP x) {
    if (TYPEOF(x) != INTSXP && TYPEOF(x) != REALSXP) Rf_error("chicdiff_hip_upload: integer or double vector expected");
    SEXP p = as_device(ctx, x, TYPEOF(x), -1, "x");
    UNPROTECT(1);
    return p;
}
SEXP chicdiff_hip_download(SEXP buf) {
    if (TYPEOF(buf) != EXTPTRSXP) Rf_error("chicdiff_hip_download: not a device vector");
    devbuf *b = (devbuf *)R_ExternalPtrAddr(buf);
    if (!b) Rf_error("chicdiff_hip_download: released");
    SEXP v = to_host(R_ExternalPtrProtected(buf), b->d, b->type, b->len);
    UNPROTECT(1);
    return v;
}
SEXP chicdiff_hip_release(SEXP buf) {
    if (TYPEOF(buf) == EXTPTRSXP && R_ExternalPtrTag(buf) == Rf_install("chicdiff_hip_buf")) devbuf_finalizer(buf);
    return R_NilValue;
}

static SEXP named_list(int n, const char **names) {
    SEXP l = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(l, R_NamesSymbol, nm);
    UNPROTECT(2);
    return l;
}
static void set_opts(chicdiff_nbglm_opts *o, SEXP dispPriorVar, SEXP fitType) {
    chicdiff_hip_default_opts(o);
    if (Rf_length(dispPriorVar) == 1 && !ISNAN(Rf_asReal(dispPriorVar))) o->dispPriorVar = Rf_asReal(dispPriorVar);
    o->fitType = Rf_asInteger(fitType); /* 0 "parametric", 1 "mean" (DESeq2 estimateDispersions(fitType = )) */
}
static void check_group(SEXP group, int S) {
    if (!Rf_isInteger(group) || LENGTH(group) != S) Rf_error("chicdiff_hip: group must be an integer vector with one 0/1 entry per sample");
}

/* ---- a2: window sums, chicdiff.R:1540-1556 ------------------------------------------------------------------ */
/* .Call(chicdiff_hip_window_sums, ctx, fragN (integer nfrag x S), fragFullMean (double nfrag x S or NULL),
 *       region_ptr (double n + 1: 0-based offsets of each region's first fragment row), S)
 * -> list(N = device integer n x S, FullMean = device double n x S or NULL).  Fragment rows in (regionID,
 * otherEndID) order, one column per sample. */
SEXP chicdiff_hip_window_sums(SEXP ctx, SEXP fragN, SEXP fragFM, SEXP region_ptr, SEXP nsamples) {
    const int S = Rf_asInteger(nsamples);
    if (!Rf_isReal(region_ptr) || XLENGTH(region_ptr) < 2 || S < 1) Rf_error("chicdiff_hip_window_sums: bad arguments");
    const R_xlen_t n = XLENGTH(region_ptr) - 1;
    const R_xlen_t nfrag = (R_xlen_t)REAL(region_ptr)[n];
    const int have_fm = !Rf_isNull(fragFM);
    chicdiff_hip_ctx *c = ctx_of(ctx);
    int np = 0;
    SEXP dN = as_device(ctx, fragN, INTSXP, nfrag * S, "fragN"); np++;
    SEXP dF = R_NilValue;
    if (have_fm) { dF = as_device(ctx, fragFM, REALSXP, nfrag * S, "fragFullMean"); np++; }
    /* region_ptr as int64 on the device: staged through an 8-byte double vector's storage */
    SEXP ptr64 = PROTECT(Rf_allocVector(REALSXP, n + 1)); np++;
    int64_t *h = (int64_t *)REAL(ptr64);
    for (R_xlen_t i = 0; i <= n; i++) {
        const double v = REAL(region_ptr)[i];
        if (!(v >= 0) || v > (double)nfrag || (i > 0 && v < REAL(region_ptr)[i - 1])) Rf_error("chicdiff_hip_window_sums: region_ptr must be ascending offsets");
        h[i] = (int64_t)v;
    }
    SEXP dP = devbuf_new(ctx, REALSXP, n + 1); np++;
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dP), h, 8 * (uint64_t)(n + 1)), "chicdiff_hip_window_sums");
    static const char *names[] = {"N", "FullMean"};
    SEXP out = PROTECT(named_list(2, names)); np++;
    SEXP oN = devbuf_new(ctx, INTSXP, n * S); np++;
    SET_VECTOR_ELT(out, 0, oN);
    SEXP oF = R_NilValue;
    if (have_fm) { oF = devbuf_new(ctx, REALSXP, n * S); np++; SET_VECTOR_ELT(out, 1, oF); }
    check_rc(ctx, chicdiff_hip_window_sums_dev(c, (const int32_t *)dptr(dN), have_fm ? (const double *)dptr(dF) : NULL, (int64_t)nfrag, S,
                                               (const int64_t *)dptr(dP), (int64_t)n, (int32_t *)dptr(oN), have_fm ? (double *)dptr(oF) : NULL),
             "chicdiff_hip_window_sums");
    release_if_temp(dN, fragN);
    if (have_fm) release_if_temp(dF, fragFM);
    devbuf_finalizer(dP);
    UNPROTECT(np);
    return out;
}

/* ---- a5: estimateSizeFactors, chicdiff.R:1561-1562 ----------------------------------------------------------- */
/* .Call(chicdiff_hip_size_factors, ctx, counts (integer n x S, host or device), n, S) -> double S */
SEXP chicdiff_hip_size_factors(SEXP ctx, SEXP counts, SEXP nrow, SEXP nsamples) {
    const R_xlen_t n = (R_xlen_t)Rf_asReal(nrow);
    const int S = Rf_asInteger(nsamples);
    SEXP dK = as_device(ctx, counts, INTSXP, n * S, "counts");
    SEXP sf = PROTECT(Rf_allocVector(REALSXP, S));
    check_rc(ctx, chicdiff_hip_size_factors_dev(ctx_of(ctx), (const int32_t *)dptr(dK), (int64_t)n, S, REAL(sf)), "chicdiff_hip_size_factors");
    release_if_temp(dK, counts);
    UNPROTECT(2);
    return sf;
}

/* ---- a4: offsets, chicdiff.R:1583-1589 (M3), 1614-1615 (nsf), 1635-1638 / 1666-1669 (theta mix) ---------------- */
/* .Call(chicdiff_hip_offsets, ctx, fullMean (double n x S, host or device, or NULL = size factors only),
 *       sizeFactors, theta (NA = normFactorsM3), n, S) -> device double n x S */
SEXP chicdiff_hip_offsets(SEXP ctx, SEXP fullMean, SEXP sf, SEXP theta, SEXP nrow, SEXP nsamples) {
    const R_xlen_t n = (R_xlen_t)Rf_asReal(nrow);
    const int S = Rf_asInteger(nsamples);
    if (!Rf_isReal(sf) || LENGTH(sf) != S) Rf_error("chicdiff_hip_offsets: one size factor per sample expected");
    const int have_fm = !Rf_isNull(fullMean);
    int np = 0;
    SEXP dF = R_NilValue;
    if (have_fm) { dF = as_device(ctx, fullMean, REALSXP, n * S, "fullMean"); np++; }
    SEXP out = devbuf_new(ctx, REALSXP, n * S); np++;
    const double th = ISNAN(Rf_asReal(theta)) ? NAN : Rf_asReal(theta);
    check_rc(ctx, chicdiff_hip_offsets_dev(ctx_of(ctx), have_fm ? (const double *)dptr(dF) : NULL, REAL(sf), (int64_t)n, S, th, (double *)dptr(out)),
             "chicdiff_hip_offsets");
    if (have_fm) release_if_temp(dF, fullMean);
    UNPROTECT(np);
    return out;
}

/* ---- a8: theta grid, chicdiff.R:1619-1662 --------------------------------------------------------------------- */
/* .Call(chicdiff_hip_theta_grid, ctx, counts, fullMean, sizeFactors, thetas, n, S) -> double length(thetas):
 * sum(mcols(ddsTest)$deviance) of the design ~1 fit under sc(theta) (NaN when a row is all zero, as sum() without na.rm) */
SEXP chicdiff_hip_theta_grid(SEXP ctx, SEXP counts, SEXP fullMean, SEXP sf, SEXP thetas, SEXP nrow, SEXP nsamples) {
    const R_xlen_t n = (R_xlen_t)Rf_asReal(nrow);
    const int S = Rf_asInteger(nsamples);
    if (!Rf_isReal(sf) || LENGTH(sf) != S || !Rf_isReal(thetas) || LENGTH(thetas) < 1) Rf_error("chicdiff_hip_theta_grid: bad arguments");
    SEXP dK = as_device(ctx, counts, INTSXP, n * S, "counts");
    SEXP dF = as_device(ctx, fullMean, REALSXP, n * S, "fullMean");
    SEXP dev = PROTECT(Rf_allocVector(REALSXP, LENGTH(thetas)));
    check_rc(ctx, chicdiff_hip_theta_grid_dev(ctx_of(ctx), (const int32_t *)dptr(dK), (const double *)dptr(dF), REAL(sf), (int64_t)n, S, REAL(thetas),
                                              LENGTH(thetas), NULL, REAL(dev)),
             "chicdiff_hip_theta_grid");
    release_if_temp(dK, counts);
    release_if_temp(dF, fullMean);
    UNPROTECT(3);
    return dev;
}

/* ---- a5 + a4 + a6 + a7 + a9 in one call ------------------------------------------------------------------------- */
static const char *k_fit_names[] = {"baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj", "dispGeneEst", "dispFit",
                                    "dispersion", "deviance", "maxCooks", "betaConv", "dispOutlier", "sizeFactors", "trendCoef",
                                    "varLogDispEsts", "dispPriorVar", "sumDeviance", "status", "nCooksOutliers", "filterThreshold",
                                    "filterTheta"};
enum { F_baseMean, F_lfc, F_lfcSE, F_stat, F_pvalue, F_padj, F_dispGene, F_dispFit, F_disp, F_dev, F_maxCooks, F_betaConv, F_outlier,
       F_sf, F_trend, F_varLog, F_priorVar, F_sumDev, F_status, F_nCooks, F_fthr, F_ftheta, F_COUNT };

/* runs results() on device-resident fit columns and copies everything to the R list */
static SEXP finish_fit(SEXP ctx, SEXP dK, R_xlen_t n, int S, SEXP group, SEXP cols[/*11 double*/], SEXP icols[/*3 int*/],
                       const chicdiff_nbglm_scalars *sc, const double *sf, SEXP cooksCutoff, SEXP alpha) {
    chicdiff_hip_ctx *c = ctx_of(ctx);
    SEXP out = PROTECT(named_list(F_COUNT, k_fit_names));
    int64_t n_out = 0;
    const double cutoff = Rf_asReal(cooksCutoff);
    if (!ISNAN(cutoff)) /* DESeq2 applies the Cook's cutoff only when a group has >= 3 replicates: the wrapper passes NA otherwise */
        check_rc(ctx, chicdiff_hip_cooks_filter_dev(c, (const int32_t *)dptr(dK), (int64_t)n, S, INTEGER(group), (const double *)dptr(cols[F_maxCooks]),
                                                    (const int32_t *)dptr(icols[2]), cutoff, (double *)dptr(cols[F_pvalue]), &n_out),
                 "chicdiff_hip results()");
    chicdiff_results_info info;
    memset(&info, 0, sizeof info);
    check_rc(ctx, chicdiff_hip_independent_filtering_dev(c, (const double *)dptr(cols[F_baseMean]), (const double *)dptr(cols[F_pvalue]), (int64_t)n,
                                                         Rf_asReal(alpha), (double *)dptr(cols[F_padj]), &info),
             "chicdiff_hip results()");
    for (int k = 0; k <= F_maxCooks; k++) {
        SET_VECTOR_ELT(out, k, to_host(ctx, dptr(cols[k]), REALSXP, n));
        UNPROTECT(1);
    }
    SET_VECTOR_ELT(out, F_betaConv, to_host(ctx, dptr(icols[0]), INTSXP, n));
    UNPROTECT(1);
    SET_VECTOR_ELT(out, F_outlier, to_host(ctx, dptr(icols[1]), INTSXP, n));
    UNPROTECT(1);
    SEXP v = Rf_allocVector(REALSXP, S);
    SET_VECTOR_ELT(out, F_sf, v);
    for (int j = 0; j < S; j++) REAL(v)[j] = sf ? sf[j] : NA_REAL;
    v = Rf_allocVector(REALSXP, 2);
    SET_VECTOR_ELT(out, F_trend, v);
    REAL(v)[0] = sc->trendCoef[0];
    REAL(v)[1] = sc->trendCoef[1];
    SET_VECTOR_ELT(out, F_varLog, Rf_ScalarReal(sc->varLogDispEsts));
    SET_VECTOR_ELT(out, F_priorVar, Rf_ScalarReal(sc->dispPriorVar));
    SET_VECTOR_ELT(out, F_sumDev, Rf_ScalarReal(sc->sumDeviance));
    SET_VECTOR_ELT(out, F_status, Rf_ScalarInteger(sc->status));
    SET_VECTOR_ELT(out, F_nCooks, Rf_ScalarReal((double)n_out));
    SET_VECTOR_ELT(out, F_fthr, Rf_ScalarReal(info.filterThreshold));
    SET_VECTOR_ELT(out, F_ftheta, Rf_ScalarReal(info.filterTheta));
    UNPROTECT(1);
    return out;
}
static void fit_columns(SEXP ctx, R_xlen_t n, SEXP cols[], SEXP icols[], chicdiff_nbglm_out *o, int *np) {
    for (int k = 0; k <= F_maxCooks; k++) { cols[k] = devbuf_new(ctx, REALSXP, n); (*np)++; }
    for (int k = 0; k < 3; k++) { icols[k] = devbuf_new(ctx, INTSXP, n); (*np)++; }
    memset(o, 0, sizeof *o);
    o->baseMean = (double *)dptr(cols[F_baseMean]);
    o->log2FoldChange = (double *)dptr(cols[F_lfc]);
    o->lfcSE = (double *)dptr(cols[F_lfcSE]);
    o->stat = (double *)dptr(cols[F_stat]);
    o->pvalue = (double *)dptr(cols[F_pvalue]);
    o->dispGeneEst = (double *)dptr(cols[F_dispGene]);
    o->dispFit = (double *)dptr(cols[F_dispFit]);
    o->dispersion = (double *)dptr(cols[F_disp]);
    o->deviance = (double *)dptr(cols[F_dev]);
    o->maxCooks = (double *)dptr(cols[F_maxCooks]);
    o->betaConv = (int32_t *)dptr(icols[0]);
    o->dispOutlier = (int32_t *)dptr(icols[1]);
    o->cooksArgmax = (int32_t *)dptr(icols[2]);
}

/* .Call(chicdiff_hip_wald_test, ctx, counts (integer n x S, host or device), fullMean (double n x S, host or device, or
 *       NULL: norm = "standard"), group (integer S, 0/1), theta (NA: norm = "fullmean"), dispPriorVar (NA = estimate),
 *       fitType (0 "parametric", 1 "mean"), cooksCutoff (NA = no Cook's filtering), alpha, n, S)
 *   ->  named list of host vectors / scalars.
 * estimateSizeFactors -> sc(theta) -> estimateDispersions -> nbinomWaldTest -> results(), resident on the device. */
SEXP chicdiff_hip_wald_test(SEXP ctx, SEXP counts, SEXP fullMean, SEXP group, SEXP theta, SEXP dispPriorVar, SEXP fitType,
                            SEXP cooksCutoff, SEXP alpha, SEXP nrow, SEXP nsamples) {
    const R_xlen_t n = (R_xlen_t)Rf_asReal(nrow);
    const int S = Rf_asInteger(nsamples);
    check_group(group, S);
    const int have_fm = !Rf_isNull(fullMean);
    int np = 0;
    SEXP dK = as_device(ctx, counts, INTSXP, n * S, "counts"); np++;
    SEXP dF = R_NilValue;
    if (have_fm) { dF = as_device(ctx, fullMean, REALSXP, n * S, "fullMean"); np++; }
    SEXP cols[F_maxCooks + 1], icols[3];
    chicdiff_nbglm_out o;
    fit_columns(ctx, n, cols, icols, &o, &np);
    chicdiff_nbglm_opts opts;
    set_opts(&opts, dispPriorVar, fitType);
    chicdiff_nbglm_scalars sc;
    double sf[64];
    if (S > 64) Rf_error("chicdiff_hip_wald_test: at most 64 samples");
    const double th = ISNAN(Rf_asReal(theta)) ? NAN : Rf_asReal(theta);
    check_rc(ctx, chicdiff_hip_wald_test_dev(ctx_of(ctx), (const int32_t *)dptr(dK), have_fm ? (const double *)dptr(dF) : NULL, (int64_t)n, S,
                                             INTEGER(group), th, &opts, &o, &sc, sf),
             "chicdiff_hip_wald_test");
    SEXP out = PROTECT(finish_fit(ctx, dK, n, S, group, cols, icols, &sc, sf, cooksCutoff, alpha)); np++;
    for (int k = 0; k <= F_maxCooks; k++) devbuf_finalizer(cols[k]);
    for (int k = 0; k < 3; k++) devbuf_finalizer(icols[k]);
    release_if_temp(dK, counts);
    if (have_fm) release_if_temp(dF, fullMean);
    UNPROTECT(np);
    return out;
}

/* .Call(chicdiff_hip_fit, ctx, counts, nf (normalizationFactors, double n x S, host or device), group, dispPriorVar,
 *       fitType, cooksCutoff, alpha, n, S): estimateDispersions + nbinomWaldTest + results() for given normalisation factors
 * (chicdiff.R:1573-1574, 1602-1603, 1673-1674) */
SEXP chicdiff_hip_fit(SEXP ctx, SEXP counts, SEXP nf, SEXP group, SEXP dispPriorVar, SEXP fitType, SEXP cooksCutoff, SEXP alpha,
                      SEXP nrow, SEXP nsamples) {
    const R_xlen_t n = (R_xlen_t)Rf_asReal(nrow);
    const int S = Rf_asInteger(nsamples);
    check_group(group, S);
    int np = 0;
    SEXP dK = as_device(ctx, counts, INTSXP, n * S, "counts"); np++;
    SEXP dF = as_device(ctx, nf, REALSXP, n * S, "nf"); np++;
    SEXP cols[F_maxCooks + 1], icols[3];
    chicdiff_nbglm_out o;
    fit_columns(ctx, n, cols, icols, &o, &np);
    chicdiff_nbglm_opts opts;
    set_opts(&opts, dispPriorVar, fitType);
    chicdiff_nbglm_scalars sc;
    check_rc(ctx, chicdiff_hip_nbglm_fit_dev(ctx_of(ctx), (const int32_t *)dptr(dK), (const double *)dptr(dF), (int64_t)n, S, INTEGER(group), &opts, &o,
                                             &sc),
             "chicdiff_hip_fit");
    SEXP out = PROTECT(finish_fit(ctx, dK, n, S, group, cols, icols, &sc, NULL, cooksCutoff, alpha)); np++;
    for (int k = 0; k <= F_maxCooks; k++) devbuf_finalizer(cols[k]);
    for (int k = 0; k < 3; k++) devbuf_finalizer(icols[k]);
    release_if_temp(dK, counts);
    release_if_temp(dF, nf);
    UNPROTECT(np);
    return out;
}

/* ---- a9 alone: independent filtering + BH (DESeq2 pvalueAdjustment as results() runs it) ------------------------- */
/* .Call(chicdiff_hip_padj, ctx, baseMean, pvalue, alpha) -> list(padj, filterThreshold, filterTheta, numRej) */
SEXP chicdiff_hip_padj(SEXP ctx, SEXP baseMean, SEXP pvalue, SEXP alpha) {
    if (!Rf_isReal(baseMean) || !Rf_isReal(pvalue) || XLENGTH(baseMean) != XLENGTH(pvalue)) Rf_error("chicdiff_hip_padj: bad arguments");
    const R_xlen_t n = XLENGTH(pvalue);
    static const char *names[] = {"padj", "filterThreshold", "filterTheta", "numRej"};
    SEXP out = PROTECT(named_list(4, names));
    SEXP dB = as_device(ctx, baseMean, REALSXP, n, "baseMean"), dP = as_device(ctx, pvalue, REALSXP, n, "pvalue");
    SEXP dQ = devbuf_new(ctx, REALSXP, n);
    chicdiff_results_info info;
    check_rc(ctx, chicdiff_hip_independent_filtering_dev(ctx_of(ctx), (const double *)dptr(dB), (const double *)dptr(dP), (int64_t)n, Rf_asReal(alpha),
                                                         (double *)dptr(dQ), &info),
             "chicdiff_hip_padj");
    SET_VECTOR_ELT(out, 0, to_host(ctx, dptr(dQ), REALSXP, n));
    UNPROTECT(1);
    SET_VECTOR_ELT(out, 1, Rf_ScalarReal(info.filterThreshold));
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal(info.filterTheta));
    SEXP nr = Rf_allocVector(REALSXP, 50);
    SET_VECTOR_ELT(out, 3, nr);
    memcpy(REAL(nr), info.numRej, sizeof info.numRej);
    devbuf_finalizer(dB);
    devbuf_finalizer(dP);
    devbuf_finalizer(dQ);
    UNPROTECT(4);
    return out;
}

/* ---- f3: application side of IHWcorrection, chicdiff.R:2038-2049 --------------------------------------------------- */
/* .Call(chicdiff_hip_ihw_apply, ctx, avDist, pvalue, breaks, avWeights) -> list(group, weight, weighted_pvalue, weighted_padj) */
SEXP chicdiff_hip_ihw_apply(SEXP ctx, SEXP avDist, SEXP pvalue, SEXP breaks, SEXP avWeights) {
    if (!Rf_isReal(avDist) || !Rf_isReal(pvalue) || !Rf_isReal(breaks) || !Rf_isReal(avWeights) ||
        XLENGTH(avDist) != XLENGTH(pvalue) || LENGTH(breaks) != LENGTH(avWeights) + 1)
        Rf_error("chicdiff_hip_ihw_apply: bad arguments");
    const R_xlen_t n = XLENGTH(avDist);
    static const char *names[] = {"group", "weight", "weighted_pvalue", "weighted_padj"};
    SEXP out = PROTECT(named_list(4, names));
    SEXP dA = as_device(ctx, avDist, REALSXP, n, "avDist"), dP = as_device(ctx, pvalue, REALSXP, n, "pvalue");
    SEXP dG = devbuf_new(ctx, INTSXP, n), dW = devbuf_new(ctx, REALSXP, n), dWP = devbuf_new(ctx, REALSXP, n), dWQ = devbuf_new(ctx, REALSXP, n);
    check_rc(ctx, chicdiff_hip_ihw_apply_dev(ctx_of(ctx), (const double *)dptr(dA), (const double *)dptr(dP), (int64_t)n, REAL(breaks), REAL(avWeights),
                                             LENGTH(avWeights), (int32_t *)dptr(dG), (double *)dptr(dW), (double *)dptr(dWP), (double *)dptr(dWQ)),
             "chicdiff_hip_ihw_apply");
    SET_VECTOR_ELT(out, 0, to_host(ctx, dptr(dG), INTSXP, n)); /* INT32_MIN is NA_integer_ */
    SET_VECTOR_ELT(out, 1, to_host(ctx, dptr(dW), REALSXP, n));
    SET_VECTOR_ELT(out, 2, to_host(ctx, dptr(dWP), REALSXP, n));
    SET_VECTOR_ELT(out, 3, to_host(ctx, dptr(dWQ), REALSXP, n));
    UNPROTECT(4);
    devbuf_finalizer(dA); devbuf_finalizer(dP); devbuf_finalizer(dG); devbuf_finalizer(dW); devbuf_finalizer(dWP); devbuf_finalizer(dWQ);
    UNPROTECT(7);
    return out;
}

/* ---- f4: getRegionUniverse window mode, chicdiff.R:376-401 ---------------------------------------------------------- */
/* .Call(chicdiff_hip_region_universe, ctx, baitID, oeID, RUexpand, chr_of) -> list(baitID, regionID, otherEndID) in
 * (regionID, otherEndID) order; chr_of[ID + 1] = chromosome code of rmap ID (length maxfrag + 1, -1 = not on the map) */
SEXP chicdiff_hip_region_universe(SEXP ctx, SEXP baitID, SEXP oeID, SEXP RUexpand, SEXP chr_of) {
    if (!Rf_isInteger(baitID) || !Rf_isInteger(oeID) || !Rf_isInteger(chr_of) || XLENGTH(baitID) != XLENGTH(oeID))
        Rf_error("chicdiff_hip_region_universe: bad arguments");
    const R_xlen_t n = XLENGTH(baitID);
    const int maxfrag = LENGTH(chr_of) - 1, s = Rf_asInteger(RUexpand);
    chicdiff_hip_ctx *c = ctx_of(ctx);
    SEXP dB = as_device(ctx, baitID, INTSXP, n, "baitID"), dO = as_device(ctx, oeID, INTSXP, n, "oeID");
    SEXP dC = as_device(ctx, chr_of, INTSXP, maxfrag + 1, "chr_of"), dP = devbuf_new(ctx, REALSXP, n + 1); /* int64 offsets */
    int64_t total = 0;
    check_rc(ctx, chicdiff_hip_region_universe_count_dev(c, (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)n, s, (const int32_t *)dptr(dC),
                                                         maxfrag, (int64_t *)dptr(dP), NULL, NULL, &total),
             "chicdiff_hip_region_universe");
    static const char *names[] = {"baitID", "regionID", "otherEndID"};
    SEXP out = PROTECT(named_list(3, names));
    SEXP rows[3];
    for (int k = 0; k < 3; k++) rows[k] = devbuf_new(ctx, INTSXP, (R_xlen_t)total);
    if (total > 0)
        check_rc(ctx, chicdiff_hip_region_universe_fill_dev(c, (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)n, s, (const int32_t *)dptr(dC),
                                                            maxfrag, (const int64_t *)dptr(dP), (int32_t *)dptr(rows[0]), (int32_t *)dptr(rows[1]),
                                                            (int32_t *)dptr(rows[2])),
                 "chicdiff_hip_region_universe");
    for (int k = 0; k < 3; k++) {
        SET_VECTOR_ELT(out, k, to_host(ctx, dptr(rows[k]), INTSXP, (R_xlen_t)total));
        UNPROTECT(1);
    }
    devbuf_finalizer(dB); devbuf_finalizer(dO); devbuf_finalizer(dC); devbuf_finalizer(dP);
    for (int k = 0; k < 3; k++) devbuf_finalizer(rows[k]);
    UNPROTECT(8);
    return out;
}

static const R_CallMethodDef call_methods[] = {{"chicdiff_hip_open", (DL_FUNC)&chicdiff_hip_open, 1},
                                               {"chicdiff_hip_close", (DL_FUNC)&chicdiff_hip_close, 1},
                                               {"chicdiff_hip_upload", (DL_FUNC)&chicdiff_hip_upload, 2},
                                               {"chicdiff_hip_download", (DL_FUNC)&chicdiff_hip_download, 1},
                                               {"chicdiff_hip_release", (DL_FUNC)&chicdiff_hip_release, 1},
                                               {"chicdiff_hip_window_sums", (DL_FUNC)&chicdiff_hip_window_sums, 5},
                                               {"chicdiff_hip_size_factors", (DL_FUNC)&chicdiff_hip_size_factors, 4},
                                               {"chicdiff_hip_offsets", (DL_FUNC)&chicdiff_hip_offsets, 6},
                                               {"chicdiff_hip_theta_grid", (DL_FUNC)&chicdiff_hip_theta_grid, 7},
                                               {"chicdiff_hip_wald_test", (DL_FUNC)&chicdiff_hip_wald_test, 11},
                                               {"chicdiff_hip_fit", (DL_FUNC)&chicdiff_hip_fit, 10},
                                               {"chicdiff_hip_padj", (DL_FUNC)&chicdiff_hip_padj, 4},
                                               {"chicdiff_hip_ihw_apply", (DL_FUNC)&chicdiff_hip_ihw_apply, 5},
                                               {"chicdiff_hip_region_universe", (DL_FUNC)&chicdiff_hip_region_universe, 5},
                                               {NULL, NULL, 0}};

void R_init_chicdiffhip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
