/*
 * chicdiff_hip_shim.c — the `.Call` shim a Chicdiff maintainer adds to reach the HIP library.
 *
 * NOT compiled or tested in this repository: R (R.h / Rinternals.h / libR) is absent from the
 * authoring image and from the GPU box.  It is kept deliberately thin:
 * every code path it reaches is the C ABI of include/chicdiff_hip.h, which IS tested (through
 * the ctypes binding in chicdiff_amd/hip.py).  Build where R exists:
 *     R CMD SHLIB chicdiff_hip_shim.c -I../../include -L../../chicdiff_amd/lib -lchicdiff_hip
 *
 * It replaces, inside DESeq2Wrap (chicdiff.R:1494), the DESeq2 calls at chicdiff.R:1557-1674:
 *   estimateSizeFactors            -> chicdiff_hip_size_factors      (host wrapper below)
 *   normalizationFactors<- / sc    -> chicdiff_hip_offsets           ( " )
 *   estimateDispersions + nbinomWaldTest -> chicdiff_hip_nbglm_fit
 * R matrices are column-major, so INTEGER(counts) / REAL(nf) are passed through untransposed.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "chicdiff_hip.h"

static chicdiff_hip_ctx *g_ctx = NULL;

static chicdiff_hip_ctx *ctx_or_error(void) {
    if (!g_ctx) {
        int rc = chicdiff_hip_create(&g_ctx, 0);
        if (rc) Rf_error("chicdiff_hip: %s", chicdiff_hip_last_error(NULL));
    }
    return g_ctx;
}

static SEXP named_list(int n, const char **names) {
    SEXP l = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(l, R_NamesSymbol, nm);
    UNPROTECT(2);
    return l;
}

/* .Call("chicdiff_hip_fit", counts (integer n x S), nf (double n x S), group (integer S, 0/1),
 *       dispPriorVar (double, NA = estimate))  ->  named list */
SEXP chicdiff_hip_fit(SEXP counts, SEXP nf, SEXP group, SEXP dispPriorVar) {
    if (!Rf_isInteger(counts) || !Rf_isReal(nf) || !Rf_isInteger(group)) Rf_error("chicdiff_hip_fit: bad argument types");
    SEXP dim = Rf_getAttrib(counts, R_DimSymbol);
    if (Rf_length(dim) != 2) Rf_error("chicdiff_hip_fit: counts must be a matrix");
    const R_xlen_t n = INTEGER(dim)[0];
    const int S = INTEGER(dim)[1];
    if (XLENGTH(nf) != n * S || LENGTH(group) != S) Rf_error("chicdiff_hip_fit: shapes do not match");
    chicdiff_hip_ctx *c = ctx_or_error();

    static const char *names[] = {"baseMean", "dispGeneEst", "dispFit", "dispersion", "log2FoldChange", "lfcSE", "stat",
                                  "pvalue", "deviance", "maxCooks", "betaConv", "cooksArgmax", "trendCoef",
                                  "dispPriorVar", "sumDeviance", "status"};
    SEXP out = PROTECT(named_list(16, names));
    chicdiff_nbglm_out o;
    memset(&o, 0, sizeof o);
    double **dst[] = {&o.baseMean, &o.dispGeneEst, &o.dispFit, &o.dispersion, &o.log2FoldChange, &o.lfcSE, &o.stat,
                      &o.pvalue, &o.deviance, &o.maxCooks};
    for (int k = 0; k < 10; k++) {
        SEXP v = Rf_allocVector(REALSXP, n);
        SET_VECTOR_ELT(out, k, v); /* protected by `out` */
        *dst[k] = REAL(v);
    }
    SEXP bc = Rf_allocVector(INTSXP, n);
    SET_VECTOR_ELT(out, 10, bc);
    o.betaConv = INTEGER(bc);
    SEXP am = Rf_allocVector(INTSXP, n);
    SET_VECTOR_ELT(out, 11, am);
    o.cooksArgmax = INTEGER(am);

    chicdiff_nbglm_opts opts;
    chicdiff_hip_default_opts(&opts);
    if (Rf_length(dispPriorVar) == 1 && !ISNA(Rf_asReal(dispPriorVar))) opts.dispPriorVar = Rf_asReal(dispPriorVar);
    chicdiff_nbglm_scalars sc;
    const int rc = chicdiff_hip_nbglm_fit(c, INTEGER(counts), REAL(nf), (int64_t)n, S, INTEGER(group), &opts, &o, &sc);
    if (rc) {
        /* copy the message before longjmp-ing out; `out` is released by UNPROTECT */
        char msg[512];
        strncpy(msg, chicdiff_hip_last_error(c), sizeof msg - 1);
        msg[sizeof msg - 1] = 0;
        UNPROTECT(1);
        Rf_error("chicdiff_hip_fit: %s", msg);
    }
    /* NaN -> NA_real_ for all-zero rows is left to the R wrapper (is.nan -> NA) */
    SEXP tc = Rf_allocVector(REALSXP, 2);
    SET_VECTOR_ELT(out, 12, tc);
    REAL(tc)[0] = sc.trendCoef[0];
    REAL(tc)[1] = sc.trendCoef[1];
    SET_VECTOR_ELT(out, 13, Rf_ScalarReal(sc.dispPriorVar));
    SET_VECTOR_ELT(out, 14, Rf_ScalarReal(sc.sumDeviance));
    SET_VECTOR_ELT(out, 15, Rf_ScalarInteger(sc.status));
    UNPROTECT(1);
    return out;
}

/* Device buffers come from the library itself (chicdiff_hip_malloc / memcpy_*): no HIP headers needed here. */
static void *to_device(chicdiff_hip_ctx *c, const void *h, size_t bytes) {
    void *d = NULL;
    if (chicdiff_hip_malloc(c, bytes, &d) || chicdiff_hip_memcpy_h2d(c, d, h, bytes)) {
        chicdiff_hip_free(c, d);
        Rf_error("chicdiff_hip: %s", chicdiff_hip_last_error(c));
    }
    return d;
}

/* .Call("chicdiff_hip_ihw_apply", avDist, pvalue, breaks, avWeights) -> list(group, weight, weighted_pvalue,
 * weighted_padj): chicdiff.R:2038-2049 */
SEXP chicdiff_hip_ihw_apply(SEXP avDist, SEXP pvalue, SEXP breaks, SEXP avWeights) {
    if (!Rf_isReal(avDist) || !Rf_isReal(pvalue) || !Rf_isReal(breaks) || !Rf_isReal(avWeights) ||
        XLENGTH(avDist) != XLENGTH(pvalue) || LENGTH(breaks) != LENGTH(avWeights) + 1)
        Rf_error("chicdiff_hip_ihw_apply: bad arguments");
    const R_xlen_t n = XLENGTH(avDist);
    chicdiff_hip_ctx *c = ctx_or_error();
    static const char *names[] = {"group", "weight", "weighted_pvalue", "weighted_padj"};
    SEXP out = PROTECT(named_list(4, names));
    SET_VECTOR_ELT(out, 0, Rf_allocVector(INTSXP, n));
    for (int k = 1; k < 4; k++) SET_VECTOR_ELT(out, k, Rf_allocVector(REALSXP, n));
    void *d_av = to_device(c, REAL(avDist), 8 * (size_t)n), *d_p = to_device(c, REAL(pvalue), 8 * (size_t)n);
    void *d_out[4] = {NULL, NULL, NULL, NULL};
    int rc = chicdiff_hip_malloc(c, 4 * (size_t)n, &d_out[0]);
    for (int k = 1; k < 4 && !rc; k++) rc = chicdiff_hip_malloc(c, 8 * (size_t)n, &d_out[k]);
    if (!rc)
        rc = chicdiff_hip_ihw_apply_dev(c, d_av, d_p, (int64_t)n, REAL(breaks), REAL(avWeights), LENGTH(avWeights), d_out[0], d_out[1],
                                        d_out[2], d_out[3]);
    if (!rc) rc = chicdiff_hip_memcpy_d2h(c, INTEGER(VECTOR_ELT(out, 0)), d_out[0], 4 * (size_t)n); /* INT32_MIN is NA_integer_ */
    for (int k = 1; k < 4 && !rc; k++) rc = chicdiff_hip_memcpy_d2h(c, REAL(VECTOR_ELT(out, k)), d_out[k], 8 * (size_t)n);
    char msg[512] = "";
    if (rc) strncpy(msg, chicdiff_hip_last_error(c), sizeof msg - 1);
    chicdiff_hip_free(c, d_av);
    chicdiff_hip_free(c, d_p);
    for (int k = 0; k < 4; k++) chicdiff_hip_free(c, d_out[k]);
    UNPROTECT(1);
    if (rc) Rf_error("chicdiff_hip_ihw_apply: %s", msg);
    return out;
}

/* .Call("chicdiff_hip_region_universe", baitID, oeID, RUexpand, chr_of) -> list(baitID, regionID, otherEndID)
 * in (regionID, otherEndID) order: chicdiff.R:376-401; chr_of[ID + 1] = chromosome code of rmap ID (0-based
 * vector of length maxfrag + 1, -1 = not on the map) */
SEXP chicdiff_hip_region_universe(SEXP baitID, SEXP oeID, SEXP RUexpand, SEXP chr_of) {
    if (!Rf_isInteger(baitID) || !Rf_isInteger(oeID) || !Rf_isInteger(chr_of) || XLENGTH(baitID) != XLENGTH(oeID))
        Rf_error("chicdiff_hip_region_universe: bad arguments");
    const R_xlen_t n = XLENGTH(baitID);
    const int maxfrag = LENGTH(chr_of) - 1, s = Rf_asInteger(RUexpand);
    chicdiff_hip_ctx *c = ctx_or_error();
    void *d_b = to_device(c, INTEGER(baitID), 4 * (size_t)n), *d_o = to_device(c, INTEGER(oeID), 4 * (size_t)n);
    void *d_chr = to_device(c, INTEGER(chr_of), 4 * (size_t)(maxfrag + 1)), *d_ptr = NULL, *d_rows[3] = {NULL, NULL, NULL};
    int64_t total = 0;
    int rc = chicdiff_hip_malloc(c, 8 * (size_t)(n + 1), &d_ptr);
    if (!rc) rc = chicdiff_hip_region_universe_count_dev(c, d_b, d_o, (int64_t)n, s, d_chr, maxfrag, d_ptr, NULL, NULL, &total);
    static const char *names[] = {"baitID", "regionID", "otherEndID"};
    SEXP out = PROTECT(named_list(3, names));
    for (int k = 0; k < 3 && !rc; k++) {
        SET_VECTOR_ELT(out, k, Rf_allocVector(INTSXP, total));
        rc = chicdiff_hip_malloc(c, 4 * (size_t)total, &d_rows[k]);
    }
    if (!rc && total > 0)
        rc = chicdiff_hip_region_universe_fill_dev(c, d_b, d_o, (int64_t)n, s, d_chr, maxfrag, d_ptr, d_rows[0], d_rows[1], d_rows[2]);
    for (int k = 0; k < 3 && !rc; k++) rc = chicdiff_hip_memcpy_d2h(c, INTEGER(VECTOR_ELT(out, k)), d_rows[k], 4 * (size_t)total);
    char msg[512] = "";
    if (rc) strncpy(msg, chicdiff_hip_last_error(c), sizeof msg - 1);
    chicdiff_hip_free(c, d_b);
    chicdiff_hip_free(c, d_o);
    chicdiff_hip_free(c, d_chr);
    chicdiff_hip_free(c, d_ptr);
    for (int k = 0; k < 3; k++) chicdiff_hip_free(c, d_rows[k]);
    UNPROTECT(1);
    if (rc) Rf_error("chicdiff_hip_region_universe: %s", msg);
    return out;
}

/* .Call("chicdiff_hip_padj", baseMean, pvalue, alpha) -> list(padj, filterThreshold, filterTheta, numRej):
 * DESeq2 pvalueAdjustment(independentFiltering = TRUE) as results() runs it at chicdiff.R:1721/1730/1739.
 * NA_real_ is a NaN, which is what the library treats as NA. */
SEXP chicdiff_hip_padj(SEXP baseMean, SEXP pvalue, SEXP alpha) {
    if (!Rf_isReal(baseMean) || !Rf_isReal(pvalue) || XLENGTH(baseMean) != XLENGTH(pvalue)) Rf_error("chicdiff_hip_padj: bad arguments");
    const R_xlen_t n = XLENGTH(pvalue);
    chicdiff_hip_ctx *c = ctx_or_error();
    static const char *names[] = {"padj", "filterThreshold", "filterTheta", "numRej"};
    SEXP out = PROTECT(named_list(4, names));
    SET_VECTOR_ELT(out, 0, Rf_allocVector(REALSXP, n));
    SET_VECTOR_ELT(out, 3, Rf_allocVector(REALSXP, 50));
    void *d_bm = to_device(c, REAL(baseMean), 8 * (size_t)n), *d_p = to_device(c, REAL(pvalue), 8 * (size_t)n), *d_q = NULL;
    chicdiff_results_info info;
    int rc = chicdiff_hip_malloc(c, 8 * (size_t)n, &d_q);
    if (!rc) rc = chicdiff_hip_independent_filtering_dev(c, d_bm, d_p, (int64_t)n, Rf_asReal(alpha), d_q, &info);
    if (!rc) rc = chicdiff_hip_memcpy_d2h(c, REAL(VECTOR_ELT(out, 0)), d_q, 8 * (size_t)n);
    char msg[512] = "";
    if (rc) strncpy(msg, chicdiff_hip_last_error(c), sizeof msg - 1);
    chicdiff_hip_free(c, d_bm);
    chicdiff_hip_free(c, d_p);
    chicdiff_hip_free(c, d_q);
    if (rc) {
        UNPROTECT(1);
        Rf_error("chicdiff_hip_padj: %s", msg);
    }
    SET_VECTOR_ELT(out, 1, Rf_ScalarReal(info.filterThreshold));
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal(info.filterTheta));
    memcpy(REAL(VECTOR_ELT(out, 3)), info.numRej, sizeof info.numRej);
    UNPROTECT(1);
    return out;
}

static const R_CallMethodDef call_methods[] = {{"chicdiff_hip_fit", (DL_FUNC)&chicdiff_hip_fit, 4},
                                               {"chicdiff_hip_padj", (DL_FUNC)&chicdiff_hip_padj, 3},
                                               {"chicdiff_hip_ihw_apply", (DL_FUNC)&chicdiff_hip_ihw_apply, 4},
                                               {"chicdiff_hip_region_universe", (DL_FUNC)&chicdiff_hip_region_universe, 4},
                                               {NULL, NULL, 0}};

void R_init_chicdiffhip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_chicdiffhip(DllInfo *dll) {
    (void)dll;
    if (g_ctx) {
        chicdiff_hip_destroy(g_ctx);
        g_ctx = NULL;
    }
}
