// common.h — shared declarations of the HIP implementation behind include/chicdiff_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chicdiff_hip.h"
#include "fit_driver.h"
#include "fit_state.h"

namespace cd {

constexpr int kRedBlocks = 1024;   // fixed grid of the reduction passes (deterministic two-stage sums)

struct FitDims {
    int64_t n;
    int32_t S, p, nA, nB;
    uint64_t gmask;  // bit j set = sample j in group B
};

// workspace pointers of one fit (all device memory, length n unless noted)
struct FitWork {
    double *baseMean, *baseVar, *gm0, *gm1, *rough, *binit0, *binit1, *crow;
    double *dispGene, *dispFit, *dispMAP, *disp, *beta0, *beta1, *resid;
    int32_t *allZero, *geneIter, *mapIter, *outlier, *betaIter, *optimConv;
    int32_t *order;               // schedule of the gene-wise line search: row indices, likely-long rows first (disp_kernels.hip)
    uint8_t *cls;                 // ... and the class each row was put in (255 = all-zero row: not scheduled)
    int32_t *gridlist;            // rows whose dispersion line search did not converge (fitDispGrid runs in disp_grid_kernel); their number: queue[16] gene-wise, queue[24] MAP
    char *rowpack;                // row-major copy of the fit's inputs, row i at rowpack + i * row_stride(S): a 32-byte header (kRowHdr), nf[S]
                                  // doubles, counts[S] int32 (written by prep): the row-queue kernels visit rows out of order, and a row that
                                  // is 32 + 12 S contiguous bytes costs one or two cache lines instead of 2 S + 4
    double *start;                // 4 doubles per row, dense: what the kernel that visits the rows next needs besides the record — gene-wise
                                  // search: alpha_init, log alpha_init; MAP search: start value, prior mean; IRLS: alpha, row constant of the
                                  // deviance, the two start values.  (Rounds 3-4 kept them in the second half of the record's header: 16-byte
                                  // stores 128 bytes apart, which cost disp_init 50 of its 74 us at 2 M rows.)
    double *partials;             // kRedBlocks x 72 doubles
    double *hist;                 // kMaxS*2 x kSelBins doubles (f64 so it can ride the all-reduce)
    double *hist_local;           // same size: this rank's round-2 histogram, kept aside for the sharded shortcut
    double *selcnt;               // kSelMaxWorld x kMaxS*2 doubles: per-rank candidate counts
    unsigned long long *queue;    // work-queue heads (kQueueBytes): [0] gene-wise, [1] MAP, [8..15] spare, [16] / [24] lengths of the gene-wise / MAP grid lists, [32 + 8 h] the IRLS's eight heads, 64 bytes apart, [192 + 8 h] the MAP line search's
    unsigned int *barrier;        // 9 x 64 B: grid-barrier counters of the persistent trend kernel
    FitScalars *sc;
    const double *logfact;        // log(k!) for k < kLogFactN
};
constexpr int kLogFactN = 1024;
constexpr int kQueueBytes = 2048;  // FitWork::queue
// bytes between rows of FitWork::rowpack: 12 S rounded up so that a row never straddles more 128-byte lines than it must
constexpr int kRowHdr = 32;  // four doubles in front of every row: group mean A (its sign bit set: the row is all zero), group mean B, two
                             // spare (the per-stage start values moved to FitWork::start)
__host__ __device__ inline int64_t row_stride(int S) {
    const int64_t bytes = kRowHdr + (int64_t)S * 12;
    return bytes <= 64 ? 64 : (bytes + 127) / 128 * 128;
}
__device__ __forceinline__ double *row_hdr(char *rowpack, int64_t r, int S) { return reinterpret_cast<double *>(rowpack + r * row_stride(S)); }

struct Opts {
    double minDisp, dispTol, kappa0, betaTol, minmu, outlierSD, dispPriorVarIn, maxDisp, trendIn[2];
    int32_t maxit, betaMaxit;
    int32_t fit_type = 0;  // 0 parametric trend, 1 mean (chicdiff_nbglm_opts.fitType)
    // tuning (chicdiff_hip_set_option): not part of the algorithm, results do not depend on them
    int32_t spread = 1;     // line search: samples-across-lanes evaluation for straggler waves (0 = row per lane only)
    int32_t min_waves = 0;  // line search: waves per SIMD (2 .. 4; 0 = by launch_disp's rule)
    int32_t schedule = 1;   // gene-wise line search / IRLS: visit the rows likely-long first (0 = natural order; 2 = class order through the queue only)
    int32_t deal = 0;       // ... entries per group of its static deal (0 = chosen from the number of entries per wave)
    int32_t chunk = 0;      // line search: rows per dequeue (0 = chosen from the row count; 8 .. 64)
    int32_t classes_a = 0;  // gene-wise line search: score classes dealt out statically (0 = the default, 2; 1 .. 6)
    int32_t xim_here = 0;   // (set by the fit driver, not an option) single rank: disp_init forms xim from the column sums itself, no xim_kernel launch
    int32_t prio = 0;       // line search: issue priority by search age, one level per `prio` iterations (0 = off); option "line_search_prio"
    int32_t trend_blocks = 0;  // persistent trend kernel: at most this many workgroups (0 = one per CU); option "trend_persistent_blocks"
};

// ---- schedule of a row-queue kernel (disp_kernels.hip order_*): rows in class order; the class counts per tile of rows come from
// the kernel that writes the classes (disp_init for the gene-wise search; wald_prep, for fits of one row per thread, for the IRLS)
constexpr int kSchedClasses = 6, kSchedBlocks = 1024;
inline void order_tiles(int64_t n, int64_t &nblk, int64_t &tile) {
    nblk = (n + 255) / 256;
    if (nblk > kSchedBlocks) nblk = kSchedBlocks;
    tile = ((n + nblk - 1) / nblk + 255) / 256 * 256;
    nblk = (n + tile - 1) / tile;
}
#ifdef __HIPCC__
// per-thread class counts -> hist[class][block] (wave shuffles, then one LDS add per wave and class)
__device__ __forceinline__ void order_hist_store(const unsigned int (&mine)[kSchedClasses], unsigned int *hist) {
    __shared__ unsigned int s_cnt[kSchedClasses];
    if (threadIdx.x < kSchedClasses) s_cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSchedClasses; k++) {
        unsigned int v = mine[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[k], v);
    }
    __syncthreads();
    if (threadIdx.x < kSchedClasses) hist[threadIdx.x * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}
#endif

// ---- launchers (defined in the .hip files; all enqueue on `st` and never synchronise) -------
// fm != NULL (S <= 16): the offsets are formed here, from FullMean — sc(theta) of chicdiff.R:1635-1638 / M3 of :1583-1589, the very
// function offsets16_kernel runs — written to `nf` (the stages behind read them from there) and used at once: one read and one
// launch less than offsets + prep (round 5)
struct FusedOffsets { const double *fm = nullptr; const double *sf = nullptr; double theta = 0; int mix = 0; };
void launch_prep(const int32_t *counts, double *nf, FitDims d, FitWork w, Opts o, hipStream_t st, FusedOffsets fo = FusedOffsets());
void launch_prep_finish(FitDims d, FitWork w, double *slot, hipStream_t st);  // partials -> colsum, nnz (slot: as (hi, lo) pairs into this rank's slot instead)
void launch_xim(FitDims d, FitWork w, const double *slots, int world, hipStream_t st);  // (the ranks' slots ->) colsum -> xim
void launch_disp_gene(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st);
void launch_order_build(FitDims d, FitWork w, int classesA, bool have_hist, hipStream_t st);  // w.cls -> w.order (schedule of a row-queue kernel)
void launch_disp_map(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st);
// single rank (or the gathered rows of a sharded fit): whole trend fit, one launch; with_mad: the same launch goes on to the residuals,
// their median and MAD and the closed-form prior variance (w.resid, sc->med / nres / mad / varLogDispEsts / dispPriorVar)
void launch_trend_persistent(FitDims d, FitWork w, Opts o, hipStream_t st, bool with_mad);
int trend_persistent_blocks();  // workgroups that must be co-resident (grid barrier): needs that many CUs
void launch_trend_init(FitDims d, FitWork w, Opts o, hipStream_t st);
int trend_blocks();                                                       // grid of the trend pass = rows of 8 partial sums
void launch_trend_pass(FitDims d, FitWork w, Opts o, hipStream_t st, bool fused_step);  // pass (+ reduce + step when single rank)
void launch_trend_step(FitDims d, FitWork w, Opts o, hipStream_t st);   // fixed-order sum of the (all-reduced) partials + state machine step
constexpr int kLfMaxV = 100;  // locfit's maxk
struct LfVerts { int nv; int _pad; double x[kLfMaxV], f[kLfMaxV], d[kLfMaxV]; };  // vertices of the local trend, ascending x
void launch_lf_hist(FitDims d, FitWork w, Opts o, int use_dist, double xv, uint64_t prefix, int shift, double *hist, hipStream_t st);
void launch_lf_sums(FitDims d, FitWork w, Opts o, double xv, double h, double *partials, double *out8, hipStream_t st);
void launch_lf_eval(FitDims d, FitWork w, const LfVerts &v, hipStream_t st);
void launch_trend_gather(FitDims d, FitWork w, Opts o, double *xg, double *yg, hipStream_t st);
// all-gather transport of the sharded trend: rank r's block of `block` doubles holds x[maxn] | y[maxn] (its first off[r+1] - off[r]
// entries are rows); the copy below lays the ranks' rows out back to back
constexpr int kGatherMaxWorld = 64;
struct GatherLayout { int32_t world, _pad; int64_t block, maxn; int64_t off[kGatherMaxWorld + 1]; };
void launch_trend_compact(const GatherLayout &gl, const double *recv, double *xg, double *yg, hipStream_t st);
void launch_poke(int32_t *p, int32_t v, hipStream_t st);                       // *p = v on the stream (test hooks)
void launch_flag_to_double(const int32_t *flag, double *out, hipStream_t st);  // *out = *flag != 0 (a verdict on its way to a sum-all-reduce)
void launch_dispfit_resid(FitDims d, FitWork w, Opts o, hipStream_t st);
void launch_prior_var(FitDims d, FitWork w, Opts o, hipStream_t st);
void launch_resid_hist(FitDims d, FitWork w, double *out40, hipStream_t st);  // residual histogram for the d.f. <= 3 prior
void launch_prior_mc(FitDims d, FitWork w, const double *hist40, const void *table, hipStream_t st);  // simulation-matched prior variance
void launch_wald_prep(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st);
void launch_wald_irls(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st);
void launch_wald_final(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o,
                       const chicdiff_nbglm_out &out, hipStream_t st);
void launch_wald_intercept(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o,
                           const chicdiff_nbglm_out &out, hipStream_t st);
// the fit's last kernel: deviance / row-count sums and this rank's verdicts into sc->final_sums, the size factors into sc->final_sf;
// carry: overflow flag of the size-factor select the caller ran before the fit (may be NULL); sf_dev: its size factors (may be NULL)
void launch_dev_sum_finish(FitDims d, FitWork w, const int32_t *carry, const double *sf_dev, hipStream_t st);

// radix select over keys produced on the fly; `mode` (SelMode, fit_driver.h) picks the key generator
struct SelArgs {
    int mode;
    int ncol;               // columns selected simultaneously (1, or S for size factors)
    int64_t n;
    const double *resid;    // SEL_RESID / SEL_ABSDEV: residuals (NaN = excluded)
    const double *ratio;    // SEL_SIZEFACTOR: log(count) - row log geometric mean, S x n (NaN = excluded)
    int S;
    int shift;              // bit position of the current digit
    double *sf_out;         // SEL_SIZEFACTOR: where the size factors go as well (device, may be NULL)
    int32_t *overflow_out;  // sharded shortcut: set to 1 when a candidate list does not fit (NULL = sc->sel_overflow)
};
void launch_sel_hist(SelArgs a, FitWork w, hipStream_t st);     // digit histograms for the live prefixes
void launch_sel_step(SelArgs a, FitWork w, hipStream_t st);     // pick bins, extend prefixes
void launch_sel_shortcut(SelArgs a, FitWork w, hipStream_t st); // single rank: gather the candidates left after two rounds, finish by sorting
void launch_sel_finish(SelArgs a, FitWork w, hipStream_t st);   // prefixes -> values; median into sc
// sharded shortcut (fit_state.h): keep the local histogram, own count row, place candidates, sort + pick
void launch_sel_keep_local(SelArgs a, FitWork w, hipStream_t st);
void launch_sel_gather_counts(SelArgs a, FitWork w, int world, int rank, hipStream_t st);
void launch_sel_gather_place(SelArgs a, FitWork w, int world, int rank, hipStream_t st);
void launch_sel_gather_finish(SelArgs a, FitWork w, int world, int rank, hipStream_t st);

void launch_row_ratio(const int32_t *counts, int64_t n, int S, double *ratio, int32_t *clear_flag, hipStream_t st);  // keys of the size-factor medians (+ *clear_flag = 0)
void launch_offsets(const double *fullMean, const double *sf_dev, int64_t n, int S, double theta, int mix,
                    double *out, hipStream_t st);
void launch_window_sums(const int32_t *fragN, const double *fragFM, int64_t nfrag, int S, const int64_t *rptr,
                        int64_t n, int32_t *N, double *FM, hipStream_t st);
size_t count_join_scratch_bytes(int64_t nkeys);
void launch_count_join(const int32_t *bait, const int32_t *oe, int64_t nru, const int64_t *keys,
                       const int32_t *vals, int64_t nkeys, int32_t *out, void *scratch, hipStream_t st);
void launch_fragment_background(const int32_t *bait, const int32_t *oe, int64_t nru, int32_t id_min, int32_t nid,
                                const int64_t *midsum, int32_t S, const double *sj, const double *si, const int32_t *tblb,
                                const int32_t *tlb, const double *T, int32_t ntblb, int32_t ntlb, const double *distfun_dev,
                                double *bmean, double *tmean, double *fullmean, hipStream_t st);
// post_kernels.hip
size_t bh_workspace_bytes(int64_t n);
int launch_bh_adjust(const double *p, int64_t n, double *padj, char *ws, hipStream_t st);
size_t ihw_workspace_bytes();
void launch_ihw_apply(const double *avDist, const double *pvalue, int64_t n, const double *breaks, const double *weights,
                      int ng, int32_t *group, double *weight, double *wp, double *partials, hipStream_t st);
void launch_cooks_filter(const int32_t *counts, int64_t n, int S, int p, const double *maxCooks, const int32_t *argmax,
                         double cutoff, double *pvalue, unsigned long long *nout, hipStream_t st);
size_t if_workspace_bytes(int64_t n);
int run_independent_filtering(const double *d_bm, const double *d_p, int64_t n, double alpha, double *d_padj, char *ws, hipStream_t st,
                              hipStream_t st2, hipEvent_t fork, hipEvent_t join, chicdiff_results_info *info);  // synchronises (three small read-backs)
size_t ct_workspace_bytes(int64_t n);
int launch_count_table(const int32_t *bait, const int32_t *oe, const int32_t *N, int64_t n, const uint8_t *keep, int32_t max_id,
                       int64_t *keys_out, int32_t *vals_out, char *ws, hipStream_t st);
size_t ru_scan_bytes(int64_t n);
int launch_ru_count(const int32_t *bait, const int32_t *oe, int64_t n, int s, const int32_t *chr_of, int maxfrag,
                    int64_t *region_ptr, int32_t *minOE, int32_t *maxOE, int *bad, void *tmp, size_t tmp_bytes, hipStream_t st,
                    unsigned int *mask_out = nullptr);  // mask_out (n words, may be NULL): per region, which candidates of its window are kept
void launch_ru_fill(const int32_t *bait, const int32_t *oe, int64_t n, int s, const int32_t *chr_of, int maxfrag,
                    const int64_t *region_ptr, int32_t *ru_bait, int32_t *ru_region, int32_t *ru_oe, hipStream_t st,
                    const unsigned int *mask_in = nullptr);  // mask_in: launch_ru_count's masks (NULL: recomputed)
void launch_region_avdist(const int32_t *bait, const int32_t *oe, const int64_t *ptr, int64_t n, int32_t id_min, int32_t nid,
                          const int64_t *midsum, const int32_t *chr, double *avDist, hipStream_t st);
size_t count_join_multi_scratch_bytes(int S, const int64_t *nkeys);
void launch_count_join_multi(const int32_t *bait, const int32_t *oe, int64_t nru, int S, const int64_t *const *keys, const int32_t *const *vals,
                             const int64_t *nkeys, int32_t *out, void *scratch, hipStream_t st);
void launch_count_join_inner(const int32_t *bait, const int32_t *oe, int64_t nru, int S, const int64_t *const *keys,
                             const int32_t *const *vals, const int64_t *nkeys, int32_t *out, hipStream_t st);
void launch_math_selftest(int op, const double *x, int64_t n, double *out, hipStream_t st);
void launch_pvalues(const double *stat, int64_t n, double *p, hipStream_t st);

}  // namespace cd
