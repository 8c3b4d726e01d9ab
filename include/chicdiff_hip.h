/*
 * chicdiff_hip.h — C ABI of the MI355X-native differential-testing core of Chicdiff.
 *
 * The reference (pure R, /root/reference/Chicdiff/R/chicdiff.R) has no FFI: the seam it
 * offers is the exported R function DESeq2Wrap() (chicdiff.R:1494) and, below it, the DESeq2
 * calls at chicdiff.R:1557-1674.  Each entry point here replaces one of those call groups and
 * is what an R `.Call` shim (r/src/chicdiff_hip_shim.c) binds.  No R, no
 * torch, no C++ types cross this boundary: plain pointers, sizes and a status code.
 *
 * Conventions
 *   - matrices are n x S column-major (= sample-major; element (i, j) at [j*n + i]): exactly
 *     R's INTEGER(mat)/REAL(mat) for the matrices built at chicdiff.R:1551-1553 and :1583,
 *     and the coalesced layout for one-row-per-lane kernels.
 *   - `_dev` entry points take DEVICE pointers (HBM-resident inputs/outputs, the benchmarked
 *     form); the others take caller-owned HOST buffers and stage them (what R passes).
 *   - every function returns 0 on success, a CHICDIFF_E_* code otherwise;
 *     chicdiff_hip_last_error() gives the message.  Nothing throws, aborts or calls exit.
 *   - NA: counts must not be NA_integer_ (INT_MIN) -> CHICDIFF_E_INVALID; NaN/NA_real_ in
 *     FullMean is meaningful (row falls back to size factors, chicdiff.R:1588-1589).
 *   - threading: call from one host thread per context; work is enqueued on the context's
 *     stream (chicdiff_hip_set_stream) and the call returns after the results are complete.
 */
#ifndef CHICDIFF_HIP_H
#define CHICDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHICDIFF_OK 0
#define CHICDIFF_E_INVALID 1   /* bad argument (shape, NA count, unsupported design)          */
#define CHICDIFF_E_HIP 2       /* HIP runtime error (message has hipGetErrorString)           */
#define CHICDIFF_E_NOMEM 3
#define CHICDIFF_E_COMM 4      /* all-reduce callback failed                                  */
#define CHICDIFF_E_NUMERIC 5   /* e.g. every row has a zero (size factors undefined)          */

/* status bits reported in chicdiff_nbglm_scalars.status (fit completed, with caveats) */
#define CHICDIFF_ST_TREND_FAILED 1 /* no dispersion trend: the parametric fit failed and so did its substitute, the local regression
                                      (fewer than four usable rows), or the substitution was switched off (option
                                      local_trend_substitute = 0) — refit with opts.fitType = 1 ("mean") or supply trendCoef */
#define CHICDIFF_ST_PRIORVAR_MC 2  /* m-p<=3 and no dispPriorVar given: matched by simulation as DESeq2 does it (its set.seed(2) stream, hist(), loess()) */
#define CHICDIFF_ST_BETA_NONCONV 4 /* some rows hit betaMaxit (DESeq2 would call optim)        */
#define CHICDIFF_ST_ALLZERO_ROWS 8 /* some rows are all zero: their outputs are NaN (R: NA)    */
#define CHICDIFF_ST_TREND_LOCAL 16 /* dispFit is DESeq2's local-regression trend (localDispersionFit = locfit with its defaults): asked for
                                      (fitType 2), or substituted for a failed parametric fit as estimateDispersionsFit does;
                                      trendCoef is NaN */

typedef struct chicdiff_hip_ctx chicdiff_hip_ctx;

/* Sum-all-reduce `count` doubles at DEVICE pointer `dev_buf`, in place, ordered on the
 * context's stream, across the ranks sharing the fit.  Return 0 on success.  Replaces nothing
 * in the reference (single process); it is the hook for row sharding. */
typedef int (*chicdiff_allreduce_fn)(void *user, void *dev_buf, int64_t count);

/* Optional companion of the all-reduce hook: every rank contributes `count` doubles at DEVICE pointer `dev_send`; `dev_recv`
 * (world_size x count doubles, distinct from dev_send) receives rank r's block at r * count, ordered on the context's stream.
 * With it a sharded fit exchanges the rows of the dispersion trend by one all-gather (ncclAllGather); without it, by a
 * sum-all-reduce over zero-filled all-ranks arrays (twice the bytes).  Same results either way. */
typedef int (*chicdiff_allgather_fn)(void *user, const void *dev_send, void *dev_recv, int64_t count);

int chicdiff_hip_create(chicdiff_hip_ctx **ctx, int32_t device);
void chicdiff_hip_destroy(chicdiff_hip_ctx *ctx);
const char *chicdiff_hip_last_error(const chicdiff_hip_ctx *ctx); /* ctx may be NULL: last create() error */
/* hipStream_t to enqueue on (NULL = HIP's null stream).  Until this is called the context
 * uses a private non-blocking stream. */
int chicdiff_hip_set_stream(chicdiff_hip_ctx *ctx, void *hip_stream);
/* A non-NULL callback turns every global statistic (size-factor medians, nf column means, trend
 * sums, MAD medians, deviance sums) into local partials + one callback (also with world_size 1,
 * where the all-reduce is the identity); fn = NULL restores the single-process path. */
int chicdiff_hip_set_allreduce(chicdiff_hip_ctx *ctx, chicdiff_allreduce_fn fn, void *user,
                               int32_t world_size, int32_t rank);
/* After chicdiff_hip_set_allreduce (which clears it): the all-gather of the same transport; fn = NULL removes it. */
int chicdiff_hip_set_allgather(chicdiff_hip_ctx *ctx, chicdiff_allgather_fn fn, void *user);
/* Refits the last fit / size-factor / Wald-test call went through (a sharded select whose candidate list overflowed on some
 * rank, a grid-barrier timeout of the trend kernel on some rank, the local-regression substitute): every rank of a sharded
 * call reports the same number — each verdict is all-reduced before anybody acts on it. */
int32_t chicdiff_hip_last_refits(const chicdiff_hip_ctx *ctx);

/* Tuning / test options; results never depend on them, the defaults are what bench.py measures.
 *   "line_search_spread"        1 (default) | 0: evaluate straggler rows (line searches and IRLS) with their samples spread across lanes
 *   "line_search_min_waves"     2 (default) .. 4: waves per SIMD the line-search kernel variant is built for
 *   "line_search_schedule"      1 (default) | 0: the gene-wise line search visits the rows likely to need DESeq2's full 100
 *                               iterations first (score alpha_init * smaller group mean); 0 = natural row order
 *   "line_search_deal"          0 (default: chosen from the rows per wave), 1 .. 64: schedule entries per group of the static deal
 *   "theta_grid_concurrency"    5 (default), 1 .. 16: fits of the theta grid in flight at once (single rank only)
 *   "host_copy_threads"         12 (default), 1 .. 64: host threads staging caller buffers in chicdiff_hip_nbglm_fit
 *   "select_all_rounds"         0 (default) | 1: exact medians by histogram rounds only (no candidate-sort shortcut)
 *   "trend_one_launch_per_pass" 0 (default) | 1: trend fit as one launch per IRLS pass instead of one persistent kernel
 *   "sharded_trend_gather"      1 (default) | 0: sharded fits exchange the trend's rows once and fit them on every rank,
 *                               instead of one all-reduce per IRLS pass (same coefficients up to summation order)
 *   "trend_persistent_blocks"   0 (default: one workgroup per CU), 1 .. 256: cap on the workgroups of the single-launch trend
 *                               kernel, for fits that share one GPU (its grid barrier needs all of them resident at once); the
 *                               coefficients then differ in summation order only (1e-13)
 *   "trend_mad_in_kernel"       1 (default) | 0: the single-launch trend kernel goes on to the residuals, their exact median and MAD
 *                               and the closed-form prior variance; 0 = separate launches (residuals, two radix selects): same bits
 *   "fault_inject"              0 (default); test hook, one-shot bits consumed by the next call: 1 = this rank reports a select
 *                               overflow in its next fit, 2 = a grid-barrier timeout of its trend kernel, 4 = an overflow of its
 *                               next size-factor select — to prove that all ranks of a sharded fit refit together
 * and one that does change the outcome of a fit whose parametric trend fails (DESeq2 offers the same choice through fitType):
 *   "local_trend_substitute"    1 (default) | 0: report CHICDIFF_ST_TREND_FAILED instead of substituting the local regression */
int chicdiff_hip_set_option(chicdiff_hip_ctx *ctx, const char *name, int64_t value);

/* Direct RCCL (backend of choice on one node: RCCL over xGMI).  The library dlopen()s librccl (librccl_path, or
 * "librccl.so" when NULL/empty — pass the copy the host process already uses, e.g. torch's), creates its own
 * communicator and from then on calls ncclAllReduce(ncclFloat64, ncclSum) — and ncclAllGather for the rows of the
 * dispersion trend — itself, in place on its stream: no host callback per collective.  Rank 0 makes the 128-byte id with _unique_id and the host broadcasts it (any
 * transport); every rank then calls _init, which replaces a callback set with chicdiff_hip_set_allreduce. */
int chicdiff_hip_rccl_unique_id(chicdiff_hip_ctx *ctx, const char *librccl_path, void *id128);
int chicdiff_hip_rccl_init(chicdiff_hip_ctx *ctx, const char *librccl_path, const void *id128, int32_t world_size,
                           int32_t rank);

/* Device memory for hosts without a GPU array library of their own (the R shim): plain allocations on the
 * context's device, copies ordered on the context's stream and complete on return.  The context keeps a list of
 * them: chicdiff_hip_destroy() releases whatever is still outstanding (R runs the finalizers of one garbage
 * collection in no particular order, so a context can be finalized before its vectors — r/src/chicdiff_hip_shim.c:
 * devbuf_finalizer then finds the context gone and has nothing left to free); chicdiff_hip_free() of a pointer the
 * context does not own is CHICDIFF_E_INVALID.  _outstanding_allocations: how many are live (-1: NULL context). */
int chicdiff_hip_malloc(chicdiff_hip_ctx *ctx, uint64_t bytes, void **d_ptr);
int chicdiff_hip_free(chicdiff_hip_ctx *ctx, void *d_ptr);
int64_t chicdiff_hip_outstanding_allocations(chicdiff_hip_ctx *ctx);
int chicdiff_hip_memcpy_h2d(chicdiff_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int chicdiff_hip_memcpy_d2h(chicdiff_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);

/* DESeq2 defaults that Chicdiff never overrides (chicdiff.R:1573-1574 pass no arguments). */
typedef struct {
    double minDisp;      /* 1e-8 */
    double dispTol;      /* 1e-6 */
    double kappa0;       /* 1.0  */
    int32_t maxit;       /* 100  dispersion line search */
    int32_t betaMaxit;   /* 100  Wald IRLS              */
    double betaTol;      /* 1e-8 */
    double minmu;        /* 0.5  */
    double outlierSD;    /* 2.0  */
    double dispPriorVar; /* NaN = estimate; DESeq2's estimateDispersionsMAP(dispPriorVar=) */
    double trendCoef[2]; /* NaN = fit; else use alpha(mu) = c0 + c1/mu as given (DESeq2: dispersionFunction<-) */
    int32_t fitType;     /* 0 = "parametric" (DESeq2's and Chicdiff's default; when that fit fails the local regression is
                            substituted, as DESeq2 does, and CHICDIFF_ST_TREND_LOCAL is set);
                            1 = "mean": dispFit = mean(dispGeneEst[dispGeneEst > 10 minDisp], trim = 0.001) for every row,
                            DESeq2's estimateDispersions(fitType = "mean") (single process only);
                            2 = "local": locfit(log dispGeneEst ~ log baseMean, weights = baseMean) with locfit's defaults */
    int32_t _pad;
} chicdiff_nbglm_opts;
void chicdiff_hip_default_opts(chicdiff_nbglm_opts *opts);

/* Per-row outputs, length n each; any pointer may be NULL (not wanted).  Host or device
 * pointers according to the entry point used.  The device entry points write these columns IN PLACE while the fit
 * runs (its workspace points into them), so after a return other than CHICDIFF_OK — and while a fit that had to
 * start over is under way — their contents are undefined: partly written, not "untouched". */
typedef struct {
    double *baseMean;       /* mcols(dds)$baseMean                                   */
    double *baseVar;
    double *dispGeneEst;    /* mcols(dds)$dispGeneEst                                */
    double *dispFit;        /* mcols(dds)$dispFit                                    */
    double *dispMAP;
    double *dispersion;     /* dispersions(dds)                                      */
    double *log2FoldChange; /* results(dds)$log2FoldChange (NaN for design ~1)       */
    double *lfcSE;
    double *stat;
    double *pvalue;         /* before Cook's cutoff / independent filtering          */
    double *intercept;      /* log2 scale                                            */
    double *interceptSE;
    double *deviance;       /* mcols(dds)$deviance = -2 logLik                       */
    double *maxCooks;       /* NaN unless a group has >= 3 samples                   */
    int32_t *dispGeneIter, *dispIter, *dispOutlier, *betaConv, *betaIter, *allZero;
    int32_t *cooksArgmax;   /* 0-based sample with the largest Cook's distance (-1 if none)   */
} chicdiff_nbglm_out;

typedef struct {
    double trendCoef[2];    /* asymptDisp, extraPois: attr(dispersionFunction, "coefficients") */
    double varLogDispEsts;
    double dispPriorVar;
    double sumDeviance;     /* sum(mcols(dds)$deviance) as chicdiff.R:1647 (NaN if any all-zero row) */
    int64_t nAllZero;
    int32_t trendOuterIter;
    int32_t status;         /* CHICDIFF_ST_* bits */
} chicdiff_nbglm_scalars;

/* a5 — estimateSizeFactors (chicdiff.R:1561-1562): median-of-ratios, S doubles to HOST sf. */
int chicdiff_hip_size_factors_dev(chicdiff_hip_ctx *ctx, const int32_t *d_counts, int64_t n, int32_t S,
                                  double *sf_host);

/* a4 — offsets (chicdiff.R:1583-1589 M3; :1614-1615 nsf; :1635-1638 / :1666-1669 theta mix).
 * theta = NaN returns normFactorsM3 (norm="fullmean"); otherwise sc(theta).  d_fullMean = NULL returns the size
 * factors, one column per sample (norm="standard", chicdiff.R:1572-1575). */
int chicdiff_hip_offsets_dev(chicdiff_hip_ctx *ctx, const double *d_fullMean, const double *sf_host,
                             int64_t n, int32_t S, double theta, double *d_nf_out);

/* a2 — window sums (chicdiff.R:1540-1556).  Fragments of region i are rows
 * [region_ptr[i], region_ptr[i+1]) of the nfrag x S fragment matrices (ascending otherEndID,
 * the order setkey(fragData, otherEndID) at :1526 produces).  Either input may be NULL. */
int chicdiff_hip_window_sums_dev(chicdiff_hip_ctx *ctx, const int32_t *d_fragN, const double *d_fragFullMean,
                                 int64_t nfrag, int32_t S, const int64_t *d_region_ptr, int64_t n,
                                 int32_t *d_N, double *d_FullMean);

/* a1 — count join (chicdiff.R:843-858): out[r] = N of (bait[r], oe[r]) in the sample's sorted
 * key table (key = baitID<<32 | otherEndID, ascending, unique), 0 when absent. */
int chicdiff_hip_count_join_dev(chicdiff_hip_ctx *ctx, const int32_t *d_ru_bait, const int32_t *d_ru_oe,
                                int64_t nru, const int64_t *d_keys, const int32_t *d_vals, int64_t nkeys,
                                int32_t *d_out);

/* a1 for ALL replicates in one pass (chicdiff.R:843-858: the loop `for (i in 1:length(chicdiff.settings$countData))` around
 * merge(RU, temp, all.x = TRUE)): every replicate's column of N from one read of the RU rows.  d_keys / d_vals / nkeys are HOST
 * arrays of S entries (device pointers inside), one sorted key table per replicate; d_out is nru x S column-major (column s =
 * replicate s) and equals S calls of chicdiff_hip_count_join_dev bit for bit. */
int chicdiff_hip_count_join_multi_dev(chicdiff_hip_ctx *ctx, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                      int32_t S, const int64_t *const *d_keys, const int32_t *const *d_vals,
                                      const int64_t *nkeys, int32_t *d_out);

/* a1, branch without chinput files (chicdiff.R:774-807, = :1202-1260 in getFullRegionData2): N comes from the
 * replicates' Chicago objects.  tempForCounts[[i]] = x[, c("baitID", "otherEndID", "N")] per replicate;
 * mergedFiles <- Reduce(merge, tempForCounts) is merge()'s default INNER join on (baitID, otherEndID), so a pair keeps its
 * counts only when every replicate's table holds it; then merge(RU, ., all.x = TRUE) and N[is.na(N)] <- 0 per replicate.
 * d_keys / d_vals / nkeys are HOST arrays of S entries: one sorted key table per replicate as chicdiff_hip_count_table_dev
 * builds it (device pointers inside).  d_out is nru x S (column s = replicate s). */
int chicdiff_hip_count_join_inner_dev(chicdiff_hip_ctx *ctx, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                      int32_t S, const int64_t *const *d_keys, const int32_t *const *d_vals,
                                      const int64_t *nkeys, int32_t *d_out);

/* IHWcorrection's covariate (chicdiff.R:1965-1967 for the test set, :1980-1982 for the control set):
 *   RU.distances <- RU.recast[, list(avDist = mean(distSign)), by = "regionID"]
 * over the long table.  Every (region, fragment) row is repeated once per sample there with the same distSign, so this is
 * the mean over the region's RU rows of CountOut's distSign (chicdiff.R:868-882):
 *   midpoint <- round(0.5 * (start + end))  (per fragment; R's round(), half to even)
 *   distSign <- midpoint[otherEndID] - midpoint[baitID], NA when the fragments lie on different chromosomes.
 * RU rows [d_region_ptr[i], d_region_ptr[i+1]) belong to region i (the CSR chicdiff_hip_region_universe_count_dev
 * returns, or (regionID, otherEndID)-ordered RU rows of any origin).  d_midsum[nid] = start + end of fragment id_min + k;
 * d_chr[nid] = chromosome code (-1 = ID not on the map: such rows are dropped, as merge(x, rmap) drops them) or NULL =
 * all rows cis and on the map.  d_avDist[n]; NaN = NA (a trans row, or no row left).  Pinned by the reference's own
 * result table: its avDist column is reproduced exactly on all 24 863 regions (tests/test_results_postprocessing.py). */
int chicdiff_hip_region_avdist_dev(chicdiff_hip_ctx *ctx, const int32_t *d_ru_bait, const int32_t *d_ru_oe, int64_t nru,
                                   const int64_t *d_region_ptr, int64_t n, int32_t id_min, int32_t nid,
                                   const int64_t *d_midsum, const int32_t *d_chr, double *d_avDist);

/* a3 — per-fragment background, the offset ingredients (chicdiff.R:628-703, 894-896 and Chicago's
 * .estimateBMean/.distFun): for every RU row r = (bait, oe) and replicate s
 *   distSign = round(((start+end)[oe] - (start+end)[bait]) / 2)                         (:648)
 *   Bmean    = s_j[bait] * s_i[oe] * f_s(|distSign|);  s_i NA -> 1;  NA when s_j is NA  (:659-672, 701-702)
 *   Tmean    = T_s[tblb[bait]][tlb[oe]];  tlb NA and tblb known -> min over tlb;  else NA (:676-692)
 *   FullMean = Bmean + Tmean                                                             (:896)
 * Lookup tables are dense over fragment ids [id_min, id_min + nid): d_midsum[nid] (= start+end),
 * and per replicate d_sj, d_si [S][nid] (NaN = absent), d_tblb, d_tlb [S][nid] (-1 = NA),
 * d_T [S][ntblb][ntlb] (NaN = combination absent).  distfun_host[S][10] = cubicFit[0..3],
 * head.coef[0..1], tail.coef[0..1], obs.min, obs.max (chicdiff.R:553-569).  Outputs [S][nru], any
 * may be NULL.  Reading the Chicago objects and the lm() refit stay host R. */
int chicdiff_hip_fragment_background_dev(chicdiff_hip_ctx *ctx, const int32_t *d_bait, const int32_t *d_oe, int64_t nru,
                                         int32_t id_min, int32_t nid, const int64_t *d_midsum, int32_t S,
                                         const double *d_sj, const double *d_si, const int32_t *d_tblb,
                                         const int32_t *d_tlb, const double *d_T, int32_t ntblb, int32_t ntlb,
                                         const double *distfun_host, double *d_bmean, double *d_tmean,
                                         double *d_fullmean);

/* f2 (device part) — chinput columns -> the key table chicdiff_hip_count_join_dev searches:
 * setkey(x, baitID); x <- x[J(baits)] (chicdiff.R:828-831: only rows whose bait is an RU bait) and
 * setkey(temp, baitID, otherEndID) (:849).  d_bait_in_RU: one byte per ID 0..max_id (non-zero = keep) or NULL =
 * keep every row.  d_keys / d_vals hold nrows entries; the first *nkeys_host are the table, ascending in
 * (baitID << 32 | otherEndID).  Reading the chinput text stays host code. */
int chicdiff_hip_count_table_dev(chicdiff_hip_ctx *ctx, const int32_t *d_bait, const int32_t *d_oe, const int32_t *d_N,
                                 int64_t nrows, const uint8_t *d_bait_in_RU, int32_t max_id, int64_t *d_keys,
                                 int32_t *d_vals, int64_t *nkeys_host);

/* f2 (text part) — `x <- fread(chinput)` and the column pick `x[, c("baitID", "otherEndID", "N")]` (chicdiff.R:828, :849).
 * _read parses the file with host threads (optional '#' comment lines, a header naming the columns, tab / blank / comma
 * separated integer rows; nthreads <= 0 = the context's default) and keeps the three columns in the context;
 * *nrows_host = rows read.  _table_dev moves them to the device and builds the key table exactly as
 * chicdiff_hip_count_table_dev does (d_keys / d_vals must hold *nrows_host entries). */
int chicdiff_hip_chinput_read(chicdiff_hip_ctx *ctx, const char *path, int32_t nthreads, int64_t *nrows_host);
int chicdiff_hip_chinput_table_dev(chicdiff_hip_ctx *ctx, const uint8_t *d_bait_in_RU, int32_t max_id, int64_t *d_keys,
                                   int32_t *d_vals, int64_t *nkeys_host);

/* f1/f3 — p.adjust(p, method = "BH") (DESeq2 results() on the independent-filtering survivors; chicdiff.R:2049
 * on the weighted p-values).  NaN = NA: not counted, stays NaN.  n < 2^32. */
int chicdiff_hip_bh_adjust_dev(chicdiff_hip_ctx *ctx, const double *d_p, int64_t n, double *d_padj);

/* a9 — DESeq2 results() on device (chicdiff.R:1720-1741 call it with its defaults).
 *
 * Cook's cutoff: p <- NA where maxCooks > cutoff (host passes qf(.99, p, m - p)); for the two-group design the
 * p-value is kept when at least 3 counts of the row exceed the count of the sample with the largest Cook's
 * distance.  Only meaningful when some group has >= 3 samples (otherwise DESeq2 skips the step: do not call).
 * d_pvalue is modified in place; *n_outliers_host = rows set to NA. */
int chicdiff_hip_cooks_filter_dev(chicdiff_hip_ctx *ctx, const int32_t *d_counts, int64_t n, int32_t S, const int32_t *group,
                                  const double *d_maxCooks, const int32_t *d_cooksArgmax, double cutoff, double *d_pvalue,
                                  int64_t *n_outliers_host);

/* Independent filtering + BH (pvalueAdjustment, independentFiltering = TRUE): theta = seq(mean(baseMean == 0), 0.95,
 * length = 50), cutoffs = quantile(baseMean, theta), numRej[k] = #{BH-adjusted p < alpha among rows with baseMean >=
 * cutoff k}, lowess(numRej ~ theta, f = 1/5), first theta whose numRej exceeds max(fit) - RMSE; padj = BH over the rows
 * passing that cutoff, NaN elsewhere. */
typedef struct {
    double filterThreshold, filterTheta, alpha;
    int32_t index; /* 1-based position of the chosen theta */
    int32_t _pad;
    double theta[50], numRej[50], lowess[50];
} chicdiff_results_info;
int chicdiff_hip_independent_filtering_dev(chicdiff_hip_ctx *ctx, const double *d_baseMean, const double *d_pvalue, int64_t n,
                                           double alpha, double *d_padj, chicdiff_results_info *info);

/* f3 — application side of IHWcorrection (chicdiff.R:2038-2049), after ihw() has been trained in R:
 *   group <- as.integer(cut(log(abs(avDist)), breaks)); avWeights <- distLookup$avWeights[group];
 *   weight <- avWeights / mean(avWeights); weighted_pvalue <- pvalue / weight;
 *   weighted_padj <- p.adjust(weighted_pvalue, "BH").
 * breaks_host has ngroups + 1 ascending entries (chicdiff.R:2039), avWeights_host ngroups (<= 256).
 * d_group gets 1-based codes, INT32_MIN (NA_integer_) outside the breaks; any output may be NULL except
 * d_weighted_padj. */
int chicdiff_hip_ihw_apply_dev(chicdiff_hip_ctx *ctx, const double *d_avDist, const double *d_pvalue, int64_t n,
                               const double *breaks_host, const double *avWeights_host, int32_t ngroups,
                               int32_t *d_group, double *d_weight, double *d_weighted_pvalue,
                               double *d_weighted_padj);

/* f4 — getRegionUniverse, window mode (chicdiff.R:353-426).  For peak i (regionID i + 1): the otherEndIDs
 * .expandAvoidBait(baitID, oeID, RUexpand) (:353-367), kept when 1 <= ID <= maxfrag (:383-384) and on the bait's
 * chromosome (:386-401).  d_chr_of[0 .. maxfrag]: chromosome code of each restriction-map ID (-1 = ID not on
 * the map; entry 0 unused).  Two calls: _count fills d_region_ptr[n + 1] (CSR offsets), optional d_minOE /
 * d_maxOE (INT32_MIN for an empty region) and *total_host = number of RU rows; _fill writes the rows in
 * (regionID, otherEndID) order — RU.DT's own order is the stable sort of these rows by baitID after
 * otherEndID (setkey, :389/:393).  baitID == oeID is the reference's stop("Invalid parameters"): E_INVALID. */
int chicdiff_hip_region_universe_count_dev(chicdiff_hip_ctx *ctx, const int32_t *d_bait, const int32_t *d_oe, int64_t n,
                                           int32_t RUexpand, const int32_t *d_chr_of, int32_t maxfrag,
                                           int64_t *d_region_ptr, int32_t *d_minOE, int32_t *d_maxOE,
                                           int64_t *total_host);
int chicdiff_hip_region_universe_fill_dev(chicdiff_hip_ctx *ctx, const int32_t *d_bait, const int32_t *d_oe, int64_t n,
                                          int32_t RUexpand, const int32_t *d_chr_of, int32_t maxfrag,
                                          const int64_t *d_region_ptr, int32_t *d_ru_bait, int32_t *d_ru_region,
                                          int32_t *d_ru_oe);
/* ... both in ONE call: the caller gives room for the upper bound `capacity` >= n max(2 RUexpand + 1, 2) rows in the three row vectors
 * (the first *total_host of them are written; two rows per peak for RUexpand = 0: R's descending (bait + 2):(oe + 0) beside a bait,
 * chicdiff.R:359-363), so that nothing on the host stands between the scan and the fill. */
int chicdiff_hip_region_universe_dev(chicdiff_hip_ctx *ctx, const int32_t *d_baitID, const int32_t *d_oeID, int64_t n, int32_t RUexpand,
                                     const int32_t *d_chr_of, int32_t maxfrag, int64_t *d_region_ptr, int32_t *d_minOE,
                                     int32_t *d_maxOE, int32_t *d_ru_baitID, int32_t *d_ru_regionID, int32_t *d_ru_otherEndID,
                                     int64_t capacity, int64_t *total_host);

/* a6 + a7 — estimateDispersions + nbinomWaldTest (chicdiff.R:1573-1574, 1602-1603, 1643-1644,
 * 1673-1674) for design ~condition (group[j] in {0,1}, both present) or ~1 (all group[j]==0).
 * d_nf = normalizationFactors (n x S).  `group` is a HOST array of S ints. */
int chicdiff_hip_nbglm_fit_dev(chicdiff_hip_ctx *ctx, const int32_t *d_counts, const double *d_nf, int64_t n,
                               int32_t S, const int32_t *group, const chicdiff_nbglm_opts *opts,
                               const chicdiff_nbglm_out *d_out, chicdiff_nbglm_scalars *scalars);

/* Same, HOST buffers in and out (what the R .Call shim passes: INTEGER(counts), REAL(nf)). */
int chicdiff_hip_nbglm_fit(chicdiff_hip_ctx *ctx, const int32_t *counts, const double *nf, int64_t n, int32_t S,
                           const int32_t *group, const chicdiff_nbglm_opts *opts, const chicdiff_nbglm_out *out,
                           chicdiff_nbglm_scalars *scalars);

/* a5 + a4 + a6 + a7 in one call — size factors -> sc(theta) -> dispersions -> Wald test
 * (chicdiff.R:1561-1562, 1666-1674), with size factors and offsets kept in HBM.  theta = NaN uses
 * normFactorsM3 (norm = "fullmean"); d_fullMean = NULL uses the size factors alone (norm = "standard",
 * chicdiff.R:1572-1575).  sf_host (S doubles) may be NULL. */
int chicdiff_hip_wald_test_dev(chicdiff_hip_ctx *ctx, const int32_t *d_counts, const double *d_fullMean, int64_t n,
                               int32_t S, const int32_t *group, double theta, const chicdiff_nbglm_opts *opts,
                               const chicdiff_nbglm_out *d_out, chicdiff_nbglm_scalars *scalars, double *sf_host);

/* a8 — theta grid (chicdiff.R:1619-1662): for each theta, sc(theta) -> design ~1 fit ->
 * deviances[t] = sum(deviance).  d_fullMean n x S, sf_host the null size factors. */
int chicdiff_hip_theta_grid_dev(chicdiff_hip_ctx *ctx, const int32_t *d_counts, const double *d_fullMean,
                                const double *sf_host, int64_t n, int32_t S, const double *thetas,
                                int32_t ntheta, const chicdiff_nbglm_opts *opts, double *deviances_host);

/* Wald p-values alone: p[i] = 2*pnorm(-|stat[i]|) (Cody's algorithm, the one R's pnorm uses). */
int chicdiff_hip_wald_pvalues_dev(chicdiff_hip_ctx *ctx, const double *d_stat, int64_t n, double *d_p);

/* Device-math self test: out[i] = f(x[i]) with op 0 log (polynomial), 1 log (table), 2 reciprocal,
 * 3 lgamma, 4 digamma, 5 2*pnorm(-|x|), 8 exp (table) — the special functions the fit kernels are built on. */
int chicdiff_hip_selftest_math_dev(chicdiff_hip_ctx *ctx, int32_t op, const double *d_x, int64_t n, double *d_out);

/* Host-side self tests of the pieces behind CHICDIFF_ST_PRIORVAR_MC (no device, no context).
 * _r_random: set.seed(seed) followed by n draws of kind 0 runif(n), 1 rnorm(n), 2 rexp(n), 3 rgamma(n, shape = a,
 * scale = b) from R's default generators (Mersenne-Twister, inversion).
 * _prior_mc: DESeq2 estimateDispersionsPriorVar for residual d.f. df in 1..3: dens_out (200 x 40, row-major, may be
 * NULL) = the densities of its 200 simulated residual distributions; *prior_var_out (may be NULL) = the prior
 * variance matched to hist40, the counts of hist(residuals, breaks = -20:20/2). */
/* _chinput: the parser behind chicdiff_hip_chinput_read on its own: the three columns of up to `cap` rows, *nrows = rows in
 * the file; on a parse error the message goes to err[errcap]. */
int chicdiff_hip_selftest_chinput(const char *path, int32_t nthreads, int64_t cap, int32_t *bait, int32_t *oe, int32_t *N,
                                  int64_t *nrows, char *err, int32_t errcap);
int chicdiff_hip_selftest_r_random(int32_t kind, uint32_t seed, double a, double b, int64_t n, double *out);
int chicdiff_hip_selftest_prior_mc(int32_t df, const double *hist40, double *dens_out, double *prior_var_out);

/* Timing of the last *_dev call's kernels, measured with HIP events on the context's stream:
 * fills up to `cap` (name, milliseconds, launches) records; returns the number available. */
typedef struct {
    const char *name;
    double ms;
    int32_t launches;
    int32_t _pad;
    double bytes; /* "allreduce" / "allgather" (timing mode 1): payload this rank handed to the transport; 0 for kernels */
} chicdiff_kernel_time;
int32_t chicdiff_hip_kernel_times(chicdiff_hip_ctx *ctx, chicdiff_kernel_time *out, int32_t cap);
/* on: 0 = off, 1 = every stage of a call gets an event pair, 2 = only the three fit kernels (disp_gene, disp_map, wald_irls),
 * 3 = the gene-wise line search alone: an event pair is a packet pair on the stream — ~12 us of a 1.3 ms fit per bracketed
 * stage, 0.09 ms per fit with all ~20 stages bracketed */
int chicdiff_hip_enable_timing(chicdiff_hip_ctx *ctx, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
