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
 *   chicdiff_hip_upload / _alloc / _download / _release             device-resident vectors (external pointers)
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
 *   chicdiff_hip_chinput_table chicdiff.R:828-831, 849              fread(chinput), RU baits only, keyed (baitID, otherEndID)
 *   chicdiff_hip_count_join    chicdiff.R:843-858                   N per RU row and replicate, 0 where unobserved
 *   chicdiff_hip_count_join_multi   chicdiff.R:843-858 (the whole replicate loop)   the same for all replicates from one read of the RU rows
 *   chicdiff_hip_bait_flags    chicdiff.R:775                       sort(unique(RU$baitID)) as a device flag table, built once
 *   chicdiff_hip_count_table / chicdiff_hip_count_join_inner  chicdiff.R:742-747, 774-807   the same without chinput files
 *   chicdiff_hip_region_avdist chicdiff.R:868-882, 1965             avDist = mean(distSign) by region for IHWcorrection()
 *   chicdiff_hip_fragment_background chicdiff.R:628-703, 894-896    Bmean, Tmean, FullMean per RU row and replicate
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
/* .Call(chicdiff_hip_alloc, ctx, "integer" | "double", length) -> uninitialised device vector (a matrix another routine fills column by column) */
SEXP chicdiff_hip_alloc(SEXP ctx, SEXP type, SEXP length) {
    if (!Rf_isString(type) || LENGTH(type) != 1) Rf_error("chicdiff_hip_alloc: type must be \"integer\" or \"double\"");
    const char *t = CHAR(STRING_ELT(type, 0));
    const double len = Rf_asReal(length);
    if (!(len >= 0) || (strcmp(t, "integer") && strcmp(t, "double"))) Rf_error("chicdiff_hip_alloc: bad arguments");
    SEXP p = devbuf_new(ctx, strcmp(t, "integer") ? REALSXP : INTSXP, (R_xlen_t)len);
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
    o->fitType = Rf_asInteger(fitType); /* 0 "parametric" (local fit substituted on failure), 1 "mean", 2 "local" (DESeq2 estimateDispersions(fitType = )) */
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

/* ---- f2 + a1: chinput -> count table -> per-replicate fragment counts, chicdiff.R:811-858 ------------------------------ */
/* one byte per fragment ID 0..max_id (non-zero = a bait of the region universe), kept in a 4-byte device vector whose
 * length fixes max_id = 4 * words - 1 (the padding bytes are zero = "not a bait") */
static SEXP bait_flags_new(SEXP ctx, SEXP baits) {
    if (!Rf_isInteger(baits)) Rf_error("chicdiff_hip: baits must be integer IDs");
    int32_t max_id = -1;
    for (R_xlen_t i = 0; i < XLENGTH(baits); i++) {
        const int b = INTEGER(baits)[i];
        if (b == NA_INTEGER || b < 0) Rf_error("chicdiff_hip: NA or negative bait ID");
        if (b > max_id) max_id = b;
    }
    const R_xlen_t words = ((R_xlen_t)max_id + 1 + 3) / 4 + 1;
    unsigned char *map = (unsigned char *)R_alloc((size_t)words, 4);
    memset(map, 0, (size_t)words * 4);
    for (R_xlen_t i = 0; i < XLENGTH(baits); i++) map[INTEGER(baits)[i]] = 1;
    SEXP dMap = devbuf_new(ctx, INTSXP, words); /* PROTECTed */
    check_rc(ctx, chicdiff_hip_memcpy_h2d(ctx_of(ctx), dptr(dMap), map, (uint64_t)words * 4), "chicdiff_hip_bait_flags");
    return dMap;
}
/* .Call(chicdiff_hip_bait_flags, ctx, baits (integer: sort(unique(RU$baitID)), chicdiff.R:775)) -> device flag table,
 * built and uploaded once and handed to chicdiff_hip_chinput_table / chicdiff_hip_count_table for every replicate */
SEXP chicdiff_hip_bait_flags(SEXP ctx, SEXP baits) {
    SEXP p = bait_flags_new(ctx, baits);
    UNPROTECT(1);
    return p;
}
/* `baits`: NULL (keep every row), integer IDs (a temporary flag table is made), or a chicdiff_hip_bait_flags table;
 * returned PROTECTed (R_NilValue counts too); *max_id = last ID the table covers */
static SEXP flags_arg(SEXP ctx, SEXP baits, int32_t *max_id) {
    SEXP d = R_NilValue;
    *max_id = -1;
    if (Rf_isNull(baits)) {
        PROTECT(d);
    } else if (TYPEOF(baits) == EXTPTRSXP) {
        devbuf_of(baits, INTSXP, -1, "baits");
        d = baits;
        PROTECT(d);
    } else {
        d = bait_flags_new(ctx, baits);
    }
    if (d != R_NilValue) *max_id = (int32_t)(((devbuf *)R_ExternalPtrAddr(d))->len * 4 - 1);
    return d;
}
static SEXP key_table_result(SEXP dK, SEXP dV, int64_t nkeys, int64_t nrows) {
    static const char *names[] = {"keys", "vals", "nkeys", "nrows"};
    SEXP out = PROTECT(named_list(4, names));
    SET_VECTOR_ELT(out, 0, dK);
    SET_VECTOR_ELT(out, 1, dV);
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal((double)nkeys));
    SET_VECTOR_ELT(out, 3, Rf_ScalarReal((double)nrows));
    UNPROTECT(1);
    return out;
}
/* .Call(chicdiff_hip_chinput_table, ctx, path, baits (integer sort(unique(RU$baitID)), a chicdiff_hip_bait_flags table, or
 *       NULL = keep every row))
 * -> list(keys = device (baitID << 32 | otherEndID, ascending; carried in an 8-byte vector), vals = device integer N,
 *         nkeys, nrows): `x <- fread(chinput); setkey(x, baitID); x <- x[J(baits), ]` (:828-831) and the
 * `setkey(temp, baitID, otherEndID)` of :849, as one table per replicate.  A header without data rows gives nkeys = 0
 * (fread's empty table: every RU row then gets N = 0 from the left join). */
SEXP chicdiff_hip_chinput_table(SEXP ctx, SEXP path, SEXP baits) {
    if (!Rf_isString(path) || LENGTH(path) != 1) Rf_error("chicdiff_hip_chinput_table: one file name expected");
    chicdiff_hip_ctx *c = ctx_of(ctx);
    int32_t max_id;
    SEXP dMap = flags_arg(ctx, baits, &max_id);
    int64_t nrows = 0, nkeys = 0;
    check_rc(ctx, chicdiff_hip_chinput_read(c, CHAR(STRING_ELT(path, 0)), 0, &nrows), "chicdiff_hip_chinput_table");
    SEXP dK = devbuf_new(ctx, REALSXP, (R_xlen_t)nrows);
    SEXP dV = devbuf_new(ctx, INTSXP, (R_xlen_t)nrows);
    check_rc(ctx, chicdiff_hip_chinput_table_dev(c, dMap == R_NilValue ? NULL : (const uint8_t *)dptr(dMap), max_id, (int64_t *)dptr(dK),
                                                 (int32_t *)dptr(dV), &nkeys),
             "chicdiff_hip_chinput_table");
    SEXP out = PROTECT(key_table_result(dK, dV, nkeys, nrows));
    if (dMap != R_NilValue) release_if_temp(dMap, baits);
    UNPROTECT(4);
    return out;
}

/* .Call(chicdiff_hip_count_table, ctx, baitID, otherEndID, N (integer vectors of one Chicago data set), baits as above)
 * -> the same list as chicdiff_hip_chinput_table.  The branch without chinput files: tempForCounts[[i]] <- x[, c("baitID",
 * "otherEndID", "N")]; setkey(x, baitID, otherEndID) (chicdiff.R:742-747) and countData[[i]][J(baits), ] (:782-787) */
SEXP chicdiff_hip_count_table(SEXP ctx, SEXP bait, SEXP oe, SEXP N, SEXP baits) {
    if (!Rf_isInteger(bait) || !Rf_isInteger(oe) || !Rf_isInteger(N) || XLENGTH(oe) != XLENGTH(bait) || XLENGTH(N) != XLENGTH(bait))
        Rf_error("chicdiff_hip_count_table: three integer vectors of one length expected");
    chicdiff_hip_ctx *c = ctx_of(ctx);
    const R_xlen_t n = XLENGTH(bait);
    int32_t max_id;
    SEXP dMap = flags_arg(ctx, baits, &max_id);
    SEXP dB = as_device(ctx, bait, INTSXP, n, "baitID"), dO = as_device(ctx, oe, INTSXP, n, "otherEndID"), dN = as_device(ctx, N, INTSXP, n, "N");
    SEXP dK = devbuf_new(ctx, REALSXP, n);
    SEXP dV = devbuf_new(ctx, INTSXP, n);
    int64_t nkeys = 0;
    if (n > 0)
        check_rc(ctx, chicdiff_hip_count_table_dev(c, (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (const int32_t *)dptr(dN), (int64_t)n,
                                                   dMap == R_NilValue ? NULL : (const uint8_t *)dptr(dMap), max_id, (int64_t *)dptr(dK),
                                                   (int32_t *)dptr(dV), &nkeys),
                 "chicdiff_hip_count_table");
    SEXP out = PROTECT(key_table_result(dK, dV, nkeys, (int64_t)n));
    devbuf_finalizer(dB); devbuf_finalizer(dO); devbuf_finalizer(dN);
    if (dMap != R_NilValue) release_if_temp(dMap, baits);
    UNPROTECT(7);
    return out;
}

/* the replicates' key tables (a list of S list(keys, vals, nkeys, ...) as chicdiff_hip_count_table / chicdiff_hip_chinput_table return
 * them) against the RU rows: inner = 1 the no-chinput branch (Reduce(merge) first), 0 the chinput branch (S independent left joins
 * from one read of the RU rows) -> device integer nru x S */
static SEXP join_over_tables(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP tables, int inner, const char *who) {
    if (TYPEOF(tables) != VECSXP || LENGTH(tables) < 1 || LENGTH(tables) > 64) Rf_error("%s: a list of 1..64 key tables expected", who);
    const int S = LENGTH(tables);
    const R_xlen_t nru = TYPEOF(ru_bait) == EXTPTRSXP ? devbuf_of(ru_bait, INTSXP, -1, "ru_bait")->len : XLENGTH(ru_bait);
    const int64_t *keys[64];
    const int32_t *vals[64];
    int64_t nkeys[64];
    for (int s = 0; s < S; s++) {
        SEXP t = VECTOR_ELT(tables, s);
        if (TYPEOF(t) != VECSXP || LENGTH(t) < 3) Rf_error("%s: table %d does not come from chicdiff_hip_count_table", who, s + 1);
        devbuf *k = devbuf_of(VECTOR_ELT(t, 0), REALSXP, -1, "table$keys"), *v = devbuf_of(VECTOR_ELT(t, 1), INTSXP, -1, "table$vals");
        nkeys[s] = (int64_t)Rf_asReal(VECTOR_ELT(t, 2));
        if (nkeys[s] < 0 || nkeys[s] > k->len || nkeys[s] > v->len) Rf_error("%s: table$nkeys does not fit table %d", who, s + 1);
        keys[s] = (const int64_t *)k->d;
        vals[s] = (const int32_t *)v->d;
    }
    SEXP dB = as_device(ctx, ru_bait, INTSXP, nru, "ru_bait"), dO = as_device(ctx, ru_oe, INTSXP, nru, "ru_oe");
    SEXP res = devbuf_new(ctx, INTSXP, nru * S);
    check_rc(ctx, inner ? chicdiff_hip_count_join_inner_dev(ctx_of(ctx), (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)nru, S, keys, vals, nkeys,
                                                            (int32_t *)dptr(res))
                        : chicdiff_hip_count_join_multi_dev(ctx_of(ctx), (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)nru, S, keys, vals, nkeys,
                                                            (int32_t *)dptr(res)),
             who);
    release_if_temp(dB, ru_bait);
    release_if_temp(dO, ru_oe);
    UNPROTECT(3);
    return res;
}

/* .Call(chicdiff_hip_count_join_inner, ctx, ru_bait, ru_oe (device or host integer nru), tables (list of S key tables))
 * -> device integer nru x S.  mergedFiles <- Reduce(merge, tempForCounts) (an inner join over the replicates), then
 * merge(x, temp, all.x = TRUE); x[is.na(N), N := 0] per replicate (chicdiff.R:779-803) */
SEXP chicdiff_hip_count_join_inner(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP tables) {
    return join_over_tables(ctx, ru_bait, ru_oe, tables, 1, "chicdiff_hip_count_join_inner");
}

/* .Call(chicdiff_hip_count_join_multi, ctx, ru_bait, ru_oe (device or host integer nru), tables (list of S tables of
 *       chicdiff_hip_chinput_table)) -> device integer nru x S: the loop over the replicates of chicdiff.R:843-858 —
 * `merge(x, temp, all.x = TRUE); x[is.na(N), N := 0]` for every replicate — from ONE read of the RU rows; column s equals
 * chicdiff_hip_count_join with table s bit for bit */
SEXP chicdiff_hip_count_join_multi(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP tables) {
    return join_over_tables(ctx, ru_bait, ru_oe, tables, 0, "chicdiff_hip_count_join_multi");
}

/* .Call(chicdiff_hip_region_avdist, ctx, ru_bait, ru_oe (device or host integer, (regionID, otherEndID) order), region_ptr
 *       (double n + 1: 0-based offset of each region's first row), id_min, midsum (double nid: start + end of fragment
 *       id_min + k), chr (integer nid: chromosome code, -1 = ID not on the map; or NULL))
 * -> device double n: RU.recast[, list(avDist = mean(distSign)), by = "regionID"] (chicdiff.R:1965) without the long table */
SEXP chicdiff_hip_region_avdist(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP region_ptr, SEXP id_min, SEXP midsum, SEXP chr) {
    if (!Rf_isReal(region_ptr) || XLENGTH(region_ptr) < 2 || !Rf_isReal(midsum) || (!Rf_isNull(chr) && (!Rf_isInteger(chr) || XLENGTH(chr) != XLENGTH(midsum))))
        Rf_error("chicdiff_hip_region_avdist: bad arguments");
    chicdiff_hip_ctx *c = ctx_of(ctx);
    const R_xlen_t nru = TYPEOF(ru_bait) == EXTPTRSXP ? devbuf_of(ru_bait, INTSXP, -1, "ru_bait")->len : XLENGTH(ru_bait);
    const R_xlen_t n = XLENGTH(region_ptr) - 1, nid = XLENGTH(midsum);
    int64_t *hp = (int64_t *)R_alloc((size_t)(n + 1), 8), *hm = (int64_t *)R_alloc((size_t)nid, 8);
    for (R_xlen_t i = 0; i <= n; i++) {
        const double v = REAL(region_ptr)[i];
        if (!(v >= 0) || v > (double)nru || (i > 0 && v < REAL(region_ptr)[i - 1])) Rf_error("chicdiff_hip_region_avdist: region_ptr must ascend within 0..nru");
        hp[i] = (int64_t)v;
    }
    for (R_xlen_t i = 0; i < nid; i++) hm[i] = ISNAN(REAL(midsum)[i]) ? 0 : (int64_t)REAL(midsum)[i];
    SEXP dB = as_device(ctx, ru_bait, INTSXP, nru, "ru_bait"), dO = as_device(ctx, ru_oe, INTSXP, nru, "ru_oe");
    SEXP dP = devbuf_new(ctx, REALSXP, n + 1), dM = devbuf_new(ctx, REALSXP, nid);
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dP), hp, 8 * (uint64_t)(n + 1)), "chicdiff_hip_region_avdist");
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dM), hm, 8 * (uint64_t)nid), "chicdiff_hip_region_avdist");
    SEXP dC = R_NilValue;
    if (!Rf_isNull(chr)) dC = as_device(ctx, chr, INTSXP, nid, "chr"); else PROTECT(dC);
    SEXP res = devbuf_new(ctx, REALSXP, n);
    check_rc(ctx, chicdiff_hip_region_avdist_dev(c, (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)nru, (const int64_t *)dptr(dP), (int64_t)n,
                                                 Rf_asInteger(id_min), (int32_t)nid, (const int64_t *)dptr(dM),
                                                 Rf_isNull(chr) ? NULL : (const int32_t *)dptr(dC), (double *)dptr(res)),
             "chicdiff_hip_region_avdist");
    release_if_temp(dB, ru_bait);
    release_if_temp(dO, ru_oe);
    devbuf_finalizer(dP); devbuf_finalizer(dM);
    if (!Rf_isNull(chr)) release_if_temp(dC, chr);
    UNPROTECT(6);
    return res;
}

/* .Call(chicdiff_hip_count_join, ctx, ru_bait, ru_oe (integer nru, host or device), table (chicdiff_hip_chinput_table),
 *       out (device integer nru x S or NULL), column (0-based)) -> device integer: N of every RU row in that
 * replicate, 0 where the pair was not observed: `merge(x, temp, all.x = TRUE); x[is.na(N), N := 0]` (:850-853).
 * With `out` the column is written in place (one n x S fragment matrix for chicdiff_hip_window_sums). */
SEXP chicdiff_hip_count_join(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP table, SEXP out, SEXP column) {
    if (TYPEOF(table) != VECSXP || LENGTH(table) < 3) Rf_error("chicdiff_hip_count_join: table must come from chicdiff_hip_chinput_table");
    const R_xlen_t nru = TYPEOF(ru_bait) == EXTPTRSXP ? devbuf_of(ru_bait, INTSXP, -1, "ru_bait")->len : XLENGTH(ru_bait);
    const R_xlen_t nkeys = (R_xlen_t)Rf_asReal(VECTOR_ELT(table, 2));
    devbuf *k = devbuf_of(VECTOR_ELT(table, 0), REALSXP, -1, "table$keys"), *v = devbuf_of(VECTOR_ELT(table, 1), INTSXP, -1, "table$vals");
    if (nkeys < 0 || nkeys > k->len || nkeys > v->len) Rf_error("chicdiff_hip_count_join: table$nkeys does not fit the table");
    SEXP dB = as_device(ctx, ru_bait, INTSXP, nru, "ru_bait"), dO = as_device(ctx, ru_oe, INTSXP, nru, "ru_oe");
    SEXP res;
    int32_t *dst;
    if (Rf_isNull(out)) {
        res = devbuf_new(ctx, INTSXP, nru);
        dst = (int32_t *)dptr(res);
    } else {
        const R_xlen_t col = (R_xlen_t)Rf_asReal(column);
        devbuf *o = devbuf_of(out, INTSXP, -1, "out");
        if (col < 0 || (col + 1) * nru > o->len) Rf_error("chicdiff_hip_count_join: column outside `out`");
        res = PROTECT(out);
        dst = (int32_t *)o->d + col * nru;
    }
    check_rc(ctx, chicdiff_hip_count_join_dev(ctx_of(ctx), (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)nru, (const int64_t *)k->d,
                                              (const int32_t *)v->d, (int64_t)nkeys, dst),
             "chicdiff_hip_count_join");
    release_if_temp(dB, ru_bait);
    release_if_temp(dO, ru_oe);
    UNPROTECT(3);
    return res;
}

/* ---- a3: Bmean / Tmean / FullMean of every RU row and replicate, chicdiff.R:628-703, 894-896 ------------------------------ */
/* .Call(chicdiff_hip_fragment_background, ctx, ru_bait, ru_oe, id_min, midsum (double nid: start + end of fragment
 *       id_min + k), s_j, s_i (double nid x S, NA = absent), tblb, tlb (integer nid x S, NA = absent), Tmean (double
 *       ntlb x ntblb x S, NA = combination absent), distfun (double 10 x S: cubicFit[1:4], head.coef, tail.coef,
 *       obs.min, obs.max of .chicEstimateDistFun), S)
 * -> list(Bmean, Tmean, FullMean): device double nru x S.  Reading the Chicago objects, the by-bait / by-other-end
 * first values and the lm() refit of the distance function stay R (r/R/DESeq2Wrap_hip.R: .hipBackgroundTables). */
SEXP chicdiff_hip_fragment_background(SEXP ctx, SEXP ru_bait, SEXP ru_oe, SEXP id_min, SEXP midsum, SEXP sj, SEXP si, SEXP tblb, SEXP tlb,
                                      SEXP Tmean, SEXP distfun, SEXP nsamples) {
    const int S = Rf_asInteger(nsamples);
    if (S < 1 || !Rf_isReal(midsum) || !Rf_isReal(sj) || !Rf_isReal(si) || !Rf_isInteger(tblb) || !Rf_isInteger(tlb) || !Rf_isReal(Tmean) ||
        !Rf_isReal(distfun) || LENGTH(distfun) != 10 * S)
        Rf_error("chicdiff_hip_fragment_background: bad arguments");
    const R_xlen_t nid = XLENGTH(midsum);
    if (XLENGTH(sj) != nid * S || XLENGTH(si) != nid * S || XLENGTH(tblb) != nid * S || XLENGTH(tlb) != nid * S)
        Rf_error("chicdiff_hip_fragment_background: the per-fragment tables must be nid x S");
    SEXP dims = Rf_getAttrib(Tmean, R_DimSymbol);
    if (Rf_isNull(dims) || LENGTH(dims) != 3 || INTEGER(dims)[2] != S) Rf_error("chicdiff_hip_fragment_background: Tmean must be ntlb x ntblb x S");
    const int ntlb = INTEGER(dims)[0], ntblb = INTEGER(dims)[1];
    const R_xlen_t nru = TYPEOF(ru_bait) == EXTPTRSXP ? devbuf_of(ru_bait, INTSXP, -1, "ru_bait")->len : XLENGTH(ru_bait);
    chicdiff_hip_ctx *c = ctx_of(ctx);
    int np = 0;
    /* R's NA_integer_ bin codes become -1, the 1-based codes 0-based; start + end goes over as int64 */
    int32_t *hb = (int32_t *)R_alloc((size_t)(nid * S), 4), *hl = (int32_t *)R_alloc((size_t)(nid * S), 4);
    for (R_xlen_t i = 0; i < nid * S; i++) {
        const int b = INTEGER(tblb)[i], l = INTEGER(tlb)[i];
        if ((b != NA_INTEGER && (b < 1 || b > ntblb)) || (l != NA_INTEGER && (l < 1 || l > ntlb)))
            Rf_error("chicdiff_hip_fragment_background: bin code outside the Tmean table");
        hb[i] = b == NA_INTEGER ? -1 : b - 1;
        hl[i] = l == NA_INTEGER ? -1 : l - 1;
    }
    int64_t *hm = (int64_t *)R_alloc((size_t)nid, 8);
    for (R_xlen_t i = 0; i < nid; i++) hm[i] = ISNAN(REAL(midsum)[i]) ? 0 : (int64_t)REAL(midsum)[i];
    SEXP dB = as_device(ctx, ru_bait, INTSXP, nru, "ru_bait"); np++;
    SEXP dO = as_device(ctx, ru_oe, INTSXP, nru, "ru_oe"); np++;
    SEXP dM = devbuf_new(ctx, REALSXP, nid); np++;
    SEXP dTb = devbuf_new(ctx, INTSXP, nid * S); np++;
    SEXP dTl = devbuf_new(ctx, INTSXP, nid * S); np++;
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dM), hm, 8 * (uint64_t)nid), "chicdiff_hip_fragment_background");
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dTb), hb, 4 * (uint64_t)(nid * S)), "chicdiff_hip_fragment_background");
    check_rc(ctx, chicdiff_hip_memcpy_h2d(c, dptr(dTl), hl, 4 * (uint64_t)(nid * S)), "chicdiff_hip_fragment_background");
    SEXP dSj = as_device(ctx, sj, REALSXP, nid * S, "s_j"); np++;
    SEXP dSi = as_device(ctx, si, REALSXP, nid * S, "s_i"); np++;
    /* the library's table is [S][ntblb][ntlb] with tlb fastest: R's ntlb x ntblb x S array has exactly that layout */
    SEXP dT = as_device(ctx, Tmean, REALSXP, (R_xlen_t)ntlb * ntblb * S, "Tmean"); np++;
    static const char *names[] = {"Bmean", "Tmean", "FullMean"};
    SEXP out = PROTECT(named_list(3, names)); np++;
    SEXP res[3];
    for (int k = 0; k < 3; k++) {
        res[k] = devbuf_new(ctx, REALSXP, nru * S); np++;
        SET_VECTOR_ELT(out, k, res[k]);
    }
    /* distfun arrives 10 x S column-major = [S][10] row-major, the layout the library reads */
    check_rc(ctx, chicdiff_hip_fragment_background_dev(c, (const int32_t *)dptr(dB), (const int32_t *)dptr(dO), (int64_t)nru, Rf_asInteger(id_min),
                                                       (int32_t)nid, (const int64_t *)dptr(dM), S, (const double *)dptr(dSj), (const double *)dptr(dSi),
                                                       (const int32_t *)dptr(dTb), (const int32_t *)dptr(dTl), (const double *)dptr(dT), ntblb, ntlb,
                                                       REAL(distfun), (double *)dptr(res[0]), (double *)dptr(res[1]), (double *)dptr(res[2])),
                 "chicdiff_hip_fragment_background");
    release_if_temp(dB, ru_bait);
    release_if_temp(dO, ru_oe);
    devbuf_finalizer(dM); devbuf_finalizer(dTb); devbuf_finalizer(dTl);
    release_if_temp(dSj, sj);
    release_if_temp(dSi, si);
    release_if_temp(dT, Tmean);
    UNPROTECT(np);
    return out;
}

static const R_CallMethodDef call_methods[] = {{"chicdiff_hip_open", (DL_FUNC)&chicdiff_hip_open, 1},
                                               {"chicdiff_hip_close", (DL_FUNC)&chicdiff_hip_close, 1},
                                               {"chicdiff_hip_upload", (DL_FUNC)&chicdiff_hip_upload, 2},
                                               {"chicdiff_hip_alloc", (DL_FUNC)&chicdiff_hip_alloc, 3},
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
                                               {"chicdiff_hip_bait_flags", (DL_FUNC)&chicdiff_hip_bait_flags, 2},
                                               {"chicdiff_hip_chinput_table", (DL_FUNC)&chicdiff_hip_chinput_table, 3},
                                               {"chicdiff_hip_count_table", (DL_FUNC)&chicdiff_hip_count_table, 5},
                                               {"chicdiff_hip_count_join_inner", (DL_FUNC)&chicdiff_hip_count_join_inner, 4},
                                               {"chicdiff_hip_count_join_multi", (DL_FUNC)&chicdiff_hip_count_join_multi, 4},
                                               {"chicdiff_hip_region_avdist", (DL_FUNC)&chicdiff_hip_region_avdist, 7},
                                               {"chicdiff_hip_count_join", (DL_FUNC)&chicdiff_hip_count_join, 6},
                                               {"chicdiff_hip_fragment_background", (DL_FUNC)&chicdiff_hip_fragment_background, 12},
                                               {NULL, NULL, 0}};

void R_init_chicdiffhip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
