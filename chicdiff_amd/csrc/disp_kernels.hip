// disp_kernels.hip — gene-wise and MAP dispersion estimation on gfx950.
//
// Replaces DESeq2's estimateDispersionsGeneEst / estimateDispersionsMAP (R) and their C++
// core fitDisp / fitDispGrid, which Chicdiff reaches through estimateDispersions() at
// chicdiff.R:1573, :1602, :1643, :1673.
//
// Mapping (not MFMA: millions of independent 1-D line searches):
//   * one interaction (row) per LANE; the row's counts and offsets are staged in LDS
//     ([sample][lane] so a wave's ds_read is conflict-free), its line-search state lives in
//     registers;
//   * every loop trip ("tick") each lane evaluates the Cox-Reid adjusted profile
//     log-likelihood AND its derivative at one point of its own search — the proposal of the
//     Armijo step or the start point — so lanes in different phases still execute the same
//     instruction stream;
//   * iteration counts are heavy-tailed (median ~7, 1.7 % of rows run all 100 steps), so a
//     finished lane immediately pulls the next row from a global queue (wave-private chunks
//     of 64 rows, one atomic per chunk) instead of idling until its 63 neighbours finish;
//   * a row whose search does not converge goes on a list; the 2 x 20 points of its fitDispGrid
//     fallback are evaluated across lanes by disp_grid_kernel, three rows per wave (round 6);
//   * at the end of the launch, waves left with a few stragglers evaluate them with the samples
//     spread across lanes (eval_point_spread), bit-identical to the row-per-lane evaluation.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "common.h"
#include "devmath.h"

namespace cd {

// ------------------------------------------------------------------------------------------
// prep: per-row moments of the normalised counts (getBaseMeansAndVariances, roughDispEstimate,
// linearModelMu group means).  One thread per row, sample-major loads are coalesced.
// per-row statistics from the normalised counts q_j = k_j / nf_j (shared by both variants)
// double-double accumulation (error-free TwoSum): the column sums of nf decide xim, xim enters every row's start value, and a
// start value one ulp away moves a noise-decided row elsewhere (and through the trend every row's 8th digit).  Summed this way the
// result is the correctly rounded exact sum whatever the order, so a permuted matrix gives the permuted results bit for bit
// (test_full_size_2Mx8_...: trend to 1e-12) although the sums now ride on prep's tiles instead of a pass of their own.
struct DD { double hi, lo; };
__device__ __forceinline__ void dd_add(DD &a, double x) {
    const double s = a.hi + x, bb = s - a.hi;
    a.lo += (a.hi - (s - bb)) + (x - bb);
    a.hi = s;
}
__device__ __forceinline__ void dd_add(DD &a, const DD &b) {
    dd_add(a, b.hi);
    a.lo += b.lo;
    const double s = a.hi + a.lo;  // renormalise
    a.lo = a.lo - (s - a.hi);
    a.hi = s;
}

// The row's count profile: c_i = #{j : y_j > i} for i = 0..9, one byte each (S <= 64), in three ints.  It depends on the counts alone,
// so it is formed once per fit, by prep, and kept in the spare half of the row record's header (bytes 16..27; first version of round 6:
// formed when a lane stages its row — ~100 integer instructions that, like the whole refill, every lane of a wave executes on
// nearly every tick of the bulk: gene-wise 1.34 -> 1.41 ms); the sum over the samples of the harmonic sums H_n = sum_{i<n} 1/(r+i), n = min(y_j, nr),
// that the derivative needs is then sum_{i<nr} c_i / (r+i) — a ROW-level sum of ten terms per tick instead of a tabulated H_n per
// sample (round 6: the per-tick prefix table shrinks from 22 to 10 LDS slots per lane, which is what lets a third wave per SIMD in).
struct CountProfile { unsigned int w0 = 0, w1 = 0, w2 = 0; };
__device__ __forceinline__ void profile_add(CountProfile &c, int y) {
    const unsigned int m = (unsigned int)(y < 10 ? y : 10);      // bytes [0, m) get + 1
    const unsigned int a = m < 4u ? m : 4u, b = m < 4u ? 0u : (m < 8u ? m - 4u : 4u), d = m < 8u ? 0u : m - 8u;
    c.w0 += 0x01010101u & (a == 4u ? 0xffffffffu : ((1u << (8u * a)) - 1u));
    c.w1 += 0x01010101u & (b == 4u ? 0xffffffffu : ((1u << (8u * b)) - 1u));
    c.w2 += 0x00000101u & ((1u << (8u * d)) - 1u);
}
__device__ __forceinline__ void prep_store(FitDims d, FitWork w, int64_t i, double s, double g0, double g1, double v,
                                           double est, int64_t tot) {
    w.baseMean[i] = s / d.S;
    w.baseVar[i] = v / (d.S - 1);
    w.gm0[i] = g0;
    w.gm1[i] = g1;
    w.rough[i] = fmax(est / (d.S - d.p), 0.0);
    w.allZero[i] = (tot == 0);
}

// S <= 16: the row's q_j live in registers — one read of counts and offsets, one division per sample.  The same pass writes
// FitWork::rowpack, the row-major copy the row-queue kernels read (a row = 12 S contiguous bytes): the block's 256 rows go
// through an LDS tile (row pitch + 1 dword: conflict-free both ways) and leave as one contiguous run of 16-byte stores.
// (Measured, 2 M x 8: with every thread storing its own row straight to global memory — 16 partial-line stores per row — the
// pass took 0.47 ms instead of 0.12; through the tile it is 0.19 ms, and the three row-queue kernels gain 0.46 ms.)
template <bool FUSED>
__global__ __launch_bounds__(256) void prep16_kernel(const int32_t *__restrict__ counts,
                                                     double *__restrict__ nf, FitDims d, FitWork w, FusedOffsets fo) {
    __shared__ LogEntry s_lt[FUSED ? 64 : 1];
    if (FUSED) log_table_to_lds(s_lt);
    extern __shared__ uint32_t s_tile[];  // T rows x (stride / 4 + 1) dwords, T = blockDim.x (256, or 128 when a row is longer than 128 bytes)
    __shared__ DD s_part[256];            // column sums of the tile by row group: [group][column]
    __shared__ unsigned char s_live[256]; // row of the tile is not all zero
    const int T = blockDim.x;
    const int64_t n = d.n;
    const int S = d.S;
    const int64_t stride = row_stride(S);
    const int ldw = (int)(stride / 4) + 1, qpr = (int)(stride / 16);  // tile pitch in dwords; 16-byte chunks per row
    const int tid = threadIdx.x;
    // column sums of nf over the non-all-zero rows + their number (momentsDispEstimate's xim): the tile holds the values anyway.
    // Thread (group g, column c) adds the rows g, g + G, ... of column c; thread c then adds the G group sums in order: fixed order
    const int ncol = S + 1, G = T / ncol, cg = tid / ncol, cc = tid % ncol;
    DD colacc{0.0, 0.0};  // threads 0..S: this block's sum of column tid (column S counts the rows)
    for (int64_t base = (int64_t)blockIdx.x * T; base < n; base += (int64_t)gridDim.x * T) {
        const int64_t i = base + tid;
        const int nrows = n - base < T ? (int)(n - base) : T;
        if (i < n) {
            double q[16];
            double s = 0, g0 = 0, g1 = 0;
            int64_t tot = 0;
            int32_t sign = 0;
            CountProfile cprof;
            uint32_t *row = s_tile + tid * ldw;
            row[7] = 0;                                                        // second half of the header: the count profile (three words, below) + one spare
            for (int k = 8 + 3 * S; k < (int)(stride / 4); k++) row[k] = 0;  // the pad behind the row
            double fv[16];
            if (FUSED) {  // the row's normalisation factors from FullMean (offsets_row16: the function offsets16_kernel runs)
#pragma unroll
                for (int j = 0; j < 16; j++) fv[j] = j < S ? fo.fm[(int64_t)j * n + i] : 1.0;
                offsets_row16(fv, S, fo.sf, fo.theta, fo.mix, s_lt);
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (j < S) nf[(int64_t)j * n + i] = fv[j];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                q[j] = 0;
                if (j < S) {
                    const int32_t k = counts[(int64_t)j * n + i];
                    const double f = FUSED ? fv[j] : nf[(int64_t)j * n + i];
                    row[8 + 2 * j] = (uint32_t)__double2loint(f);
                    row[8 + 2 * j + 1] = (uint32_t)__double2hiint(f);
                    row[8 + 2 * S + j] = (uint32_t)k;
                    profile_add(cprof, k);
                    sign |= k;
                    q[j] = (double)k / f;
                    tot += k;
                    s += q[j];
                    if ((d.gmask >> j) & 1) g1 += q[j]; else g0 += q[j];
                }
            }
            const double bm = s / S;
            g0 /= d.nA;
            if (d.p == 2) g1 /= d.nB;
            row[0] = (uint32_t)__double2loint(g0);  // first half of the header: the group means
            row[1] = (uint32_t)__double2hiint(g0) | (tot == 0 ? 0x80000000u : 0u);  // sign bit of the (never negative) mean: the row is all zero — the row-queue kernels read the flag with the record
            row[2] = (uint32_t)__double2loint(g1);
            row[3] = (uint32_t)__double2hiint(g1);
            row[4] = cprof.w0;
            row[5] = cprof.w1;
            row[6] = cprof.w2;
            s_live[tid] = tot != 0;
            const double m0 = fmax(1.0, g0), m1 = fmax(1.0, g1);
            const double i0 = 1.0 / (m0 * m0), i1 = 1.0 / (m1 * m1);  // two divisions per row instead of one per sample
            double v = 0, est = 0;
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (j < S) {
                    v += (q[j] - bm) * (q[j] - bm);
                    const bool g = (d.gmask >> j) & 1;
                    const double mj = g ? m1 : m0;
                    est += ((q[j] - mj) * (q[j] - mj) - mj) * (g ? i1 : i0);
                }
            prep_store(d, w, i, s, g0, g1, v, est, tot);
            if (sign < 0) w.sc->neg_counts = 1;  // NA_integer_ / negative count: the fit is refused (include/chicdiff_hip.h)
        }
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(w.rowpack + base * stride);
        for (int c = tid; c < nrows * qpr; c += T) {
            const uint32_t *src = s_tile + (c / qpr) * ldw + (c % qpr) * 4;
            dst[c] = make_uint4(src[0], src[1], src[2], src[3]);
        }
        if (cg < G) {
            DD a{0.0, 0.0};
            for (int r = cg; r < nrows; r += G)
                if (s_live[r]) {
                    const uint32_t *src = s_tile + r * ldw + 8 + 2 * cc;
                    dd_add(a, cc < S ? __hiloint2double((int)src[1], (int)src[0]) : 1.0);
                }
            s_part[cg * ncol + cc] = a;
        }
        __syncthreads();
        if (tid < ncol)
            for (int g = 0; g < G; g++) dd_add(colacc, s_part[g * ncol + tid]);
        __syncthreads();
    }
    if (tid < ncol) {
        w.partials[((int64_t)tid * gridDim.x + blockIdx.x) * 2] = colacc.hi;
        w.partials[((int64_t)tid * gridDim.x + blockIdx.x) * 2 + 1] = colacc.lo;
    }
}

__global__ __launch_bounds__(256) void prep_kernel(const int32_t *__restrict__ counts,
                                                   const double *__restrict__ nf, FitDims d, FitWork w) {
    const int64_t n = d.n;
    const int S = d.S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0, g0 = 0, g1 = 0;
        int64_t tot = 0;
        double *rp_nf = reinterpret_cast<double *>(w.rowpack + i * row_stride(S) + kRowHdr);
        int32_t *rp_y = reinterpret_cast<int32_t *>(rp_nf + S);
        CountProfile cprof;
        for (int j = 0; j < S; j++) {
            const int32_t k = counts[(int64_t)j * n + i];
            if (k < 0) w.sc->neg_counts = 1;
            profile_add(cprof, k);
            rp_nf[j] = nf[(int64_t)j * n + i];
            rp_y[j] = k;
            const double q = (double)k / nf[(int64_t)j * n + i];
            tot += k;
            s += q;
            if ((d.gmask >> j) & 1) g1 += q; else g0 += q;
        }
        const double bm = s / S;
        g0 /= d.nA;
        if (d.p == 2) g1 /= d.nB;
        reinterpret_cast<double2 *>(row_hdr(w.rowpack, i, S))[0] = make_double2(tot == 0 ? -0.0 : g0, g1);  // first half of the row's header; sign bit = all-zero row
        reinterpret_cast<uint4 *>(row_hdr(w.rowpack, i, S))[1] = make_uint4(cprof.w0, cprof.w1, cprof.w2, 0u);  // second half: the count profile
        const double m0 = fmax(1.0, g0), m1 = fmax(1.0, g1);
        double v = 0, est = 0;
        for (int j = 0; j < S; j++) {
            const double q = (double)counts[(int64_t)j * n + i] / nf[(int64_t)j * n + i];
            v += (q - bm) * (q - bm);
            const double mj = ((d.gmask >> j) & 1) ? m1 : m0;
            est += ((q - mj) * (q - mj) - mj) / (mj * mj);
        }
        prep_store(d, w, i, s, g0, g1, v, est, tot);
    }
}

// column sums of nf over non-all-zero rows + their count: grid (kRedBlocks/8... , S+1)
__global__ __launch_bounds__(256) void colsum_kernel(const double *__restrict__ nf, FitDims d, FitWork w) {
    __shared__ double red[256];
    const int j = blockIdx.y;  // column; j == S counts rows
    const int64_t n = d.n;
    double acc = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!w.allZero[i]) acc += (j < d.S) ? nf[(int64_t)j * n + i] : 1.0;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) w.partials[(int64_t)j * gridDim.x + blockIdx.x] = red[0];
}

// one wave per column: strided fixed-order sum of the block partials, then a shuffle tree; dd: the partials are double-double pairs.
// slot != NULL (sharded fit): the column's sum leaves as a (hi, lo) pair in this rank's slot of the all-ranks buffer instead of
// rounded into the scalars — the ranks' pairs are exchanged (a sum-all-reduce over zero-filled slots) and added in rank order by
// xim_kernel, in double-double: the correctly rounded exact sum again, so a sharded fit starts from the very xim of the one-rank fit
__global__ void colsum_finish_kernel(FitDims d, FitWork w, int nblk, int dd, double *slot) {
    const int j = blockIdx.x, lane = threadIdx.x;
    DD a{0.0, 0.0};
    if (dd) {
        for (int b = lane; b < nblk; b += 64) dd_add(a, DD{w.partials[((int64_t)j * nblk + b) * 2], w.partials[((int64_t)j * nblk + b) * 2 + 1]});
        for (int off = 32; off > 0; off >>= 1) dd_add(a, DD{__shfl_down(a.hi, off), __shfl_down(a.lo, off)});
    } else {
        double s = 0;
        for (int b = lane; b < nblk; b += 64) s += w.partials[(int64_t)j * nblk + b];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        a.hi = s;
    }
    if (lane == 0) {
        if (slot) {
            slot[2 * j] = a.hi;
            slot[2 * j + 1] = a.lo;
        } else {
            const double s = a.hi + a.lo;
            if (j < d.S) w.sc->colsum[j] = s; else w.sc->nnz = s;
        }
    }
}

// slots != NULL: first the ranks' (hi, lo) column sums, in rank order (see colsum_finish_kernel)
__global__ void xim_kernel(FitDims d, FitWork w, const double *slots, int world) {
    if (threadIdx.x || blockIdx.x) return;
    if (slots)
        for (int j = 0; j <= d.S; j++) {
            DD a{0.0, 0.0};
            for (int r = 0; r < world; r++) dd_add(a, DD{slots[((size_t)r * (d.S + 1) + j) * 2], slots[((size_t)r * (d.S + 1) + j) * 2 + 1]});
            const double s = a.hi + a.lo;
            if (j < d.S) w.sc->colsum[j] = s; else w.sc->nnz = s;
        }
    double x = 0;
    for (int j = 0; j < d.S; j++) x += 1.0 / (w.sc->colsum[j] / w.sc->nnz);
    w.sc->xim = x / d.S;
}

constexpr int kColsumBlocks = 512;  // per column; (S+1) x 512 partials fit the 1024 x 72 partials buffer for S <= 64
static int prep16_blocks(int) { return 1024; }  // four workgroups per CU — what their LDS tiles (34-38 KB) allow at once: one full round, no tail (768: 0.148 -> 0.127 ms at 2 M x 8, 1280: 0.163; S = 16: 1536 -> 1024: 0.40 -> 0.32, 2048: 0.33; round 4); (S + 1) x 1024 partials
void launch_prep(const int32_t *counts, double *nf, FitDims d, FitWork w, Opts, hipStream_t st, FusedOffsets fo) {
    if (d.S <= 16) {  // one resident round: 3 workgroups of 256 per CU; the LDS tile stays under 34 KB; the column sums ride along
        const int T = row_stride(d.S) > 128 ? 128 : 256;
        if (fo.fm) prep16_kernel<true><<<prep16_blocks(d.S), T, (size_t)T * (row_stride(d.S) + 4), st>>>(counts, nf, d, w, fo);
        else prep16_kernel<false><<<prep16_blocks(d.S), T, (size_t)T * (row_stride(d.S) + 4), st>>>(counts, nf, d, w, fo);
    } else {
        prep_kernel<<<kRedBlocks, 256, 0, st>>>(counts, nf, d, w);
        colsum_kernel<<<dim3(kColsumBlocks, d.S + 1), 256, 0, st>>>(nf, d, w);
    }
}
void launch_prep_finish(FitDims d, FitWork w, double *slot, hipStream_t st) {
    colsum_finish_kernel<<<d.S + 1, 64, 0, st>>>(d, w, d.S <= 16 ? prep16_blocks(d.S) : kColsumBlocks, d.S <= 16, slot);
}
void launch_xim(FitDims d, FitWork w, const double *slots, int world, hipStream_t st) { xim_kernel<<<1, 64, 0, st>>>(d, w, slots, world); }

// ------------------------------------------------------------------------------------------
// Start values of the two line searches, one thread per row, so that the queue refill inside the
// search kernels is pure data movement.
//   gene-wise (A2.3-2.4): alpha_init = clamp(min(roughDisp, momentsDisp)); rough[] <- alpha_init, resid[] <- log
//   MAP (A4): dispFit, prior mean log(dispFit), start log(dispGeneEst > 0.1 dispFit ? dispGeneEst : dispFit),
//             outlier flag log(dispGeneEst) > log(dispFit) + outlierSD * sqrt(varLogDispEsts)
// ---- schedule of the gene-wise line search --------------------------------------------------------------------------
// DESeq2's line search moves by kappa * dlp with kappa <= 1 shrinking by 0.8 every five accepted steps, so a row whose
// likelihood is flat in log(alpha) — alpha * mu << 1, the near-Poisson corner — creeps for all 100 iterations and then
// walks the 40 grid points: 2.7 % of the rows of the benchmark matrix take >= 50 iterations and hold 27 % of all ticks.
// Which rows those are can be told beforehand: alpha_init * (smaller group mean) ranks them (400 k synthetic rows: the
// lowest 8 % of the scores hold 82 % of the rows with >= 50 iterations, the lowest 23 % hold 99.2 %, the lowest 54 %
// 99.95 %; rows that start at minDisp are never long).  A row dequeued late that needs 100 + 40 serial ticks IS the tail
// of the launch (0.5 ms of 2.0 ms at 2 M rows, 0.55 of 0.72 ms at 250 k), so the rows are visited in score order:
//   classes 0, 1 (score < 0.1, 0.32: 8 % of the rows, 82 % of the long ones) -> "A": dealt out statically, in groups of
//       eight consecutive schedule entries round-robin over the waves, so every wave starts its share of the likely-long
//       rows at once and — when rows are few (a rank's share of a sharded fit) — is left with a handful of them, which is
//       what the samples-across-lanes evaluation wants (250 k x 8: 0.74 ms against 0.87 ms with the queue alone);
//   classes 2..5 (score < 1, 3.2, 10; the rest and the minDisp starts) -> "B": the dynamic queue, in class order, after A.
// Measured at 2 M x 8 (disp_gene incl. the two order_* launches): natural order 2.08-2.11 ms, class order through the
// queue alone 1.92, A = classes 0-1 1.88-1.90, A = 0-2 1.90-1.95, A = 0-3 1.94-2.05, everything dealt statically 2.33-2.39:
// the drain shrinks from 0.48 to 0.14 ms, but its ticks move into the bulk (143 -> 167 per wave), which is bound by fp64
// issue — the launch cannot beat (all ticks) / (waves x lanes) ~ 1.7 ms.  At 250 k rows the launch stays at the latency of one
// long row (~ 105 serial ticks at 3.6-4 us in the samples-across-lanes layout): 0.79 -> 0.74 ms.
// Within a class rows keep their natural order (neighbouring lanes read neighbouring rows).  Results never depend on
// the schedule (tests/test_gpu_parity.py::test_line_search_layouts_agree_bit_for_bit runs both).
constexpr int kSchedClassesA = 2, kSchedDeal = 8;  // (kSchedClasses, kSchedBlocks, order_tiles(), order_hist_store(): common.h — wald_prep builds the IRLS's class counts too)
__device__ __forceinline__ int sched_class(double a0, double gmin, double minDisp) {
    if (!(a0 > 1.5 * minDisp)) return 5;
    const double s = a0 * gmin;
    return s < 0.1 ? 0 : (s < 0.316 ? 1 : (s < 1.0 ? 2 : (s < 3.16 ? 3 : (s < 10.0 ? 4 : 5))));
}
// per-block class counts over contiguous tiles of rows: hist[class][block]
__global__ __launch_bounds__(256) void order_hist_kernel(const uint8_t *__restrict__ cls, int64_t n, int64_t tile, unsigned int *hist) {
    const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
    unsigned int mine[kSchedClasses] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int c = cls[i];
#pragma unroll
        for (int k = 0; k < kSchedClasses; k++) mine[k] += (c == k);
    }
    order_hist_store(mine, hist);
}
// class-major exclusive scan of hist (every block recomputes the offsets it needs: 6 x 1024 entries from L2), then a
// stable scatter of the block's rows: order[] = class 0 rows in row order, class 1 rows, ...
__global__ __launch_bounds__(256) void order_scatter_kernel(const uint8_t *__restrict__ cls, int64_t n, int64_t tile,
                                                            const unsigned int *__restrict__ hist, int32_t *__restrict__ order,
                                                            FitScalars *sc, int classesA) {
    __shared__ unsigned long long s_base[kSchedClasses + 1];
    __shared__ unsigned int s_wcnt[4][kSchedClasses];
    const int nblk = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // s_base[c] = number of schedule entries before this block's class-c rows = the sum of hist[0 .. c nblk + blockIdx.x); block 0
    // also leaves |A| = sum of hist[0 .. classesA nblk) and the total.  ONE pass over the 6 nblk counts for all eight sums (round 4;
    // before: a pass, a shuffle tree and two workgroup barriers per sum — half of the kernel's 20 us at 1024 tiles).
    {
        unsigned long long part[kSchedClasses + 2] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int total = kSchedClasses * nblk, cutA = classesA * nblk;
        for (int e = threadIdx.x; e < total; e += 256) {
            const unsigned long long hv = hist[e];
#pragma unroll
            for (int c = 0; c < kSchedClasses; c++) part[c] += e < c * nblk + (int)blockIdx.x ? hv : 0ull;
            part[kSchedClasses] += e < cutA ? hv : 0ull;
            part[kSchedClasses + 1] += hv;
        }
        __shared__ unsigned long long s_part[4][kSchedClasses + 2];
#pragma unroll
        for (int c = 0; c < kSchedClasses + 2; c++) {
            unsigned long long v = part[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) s_part[wave][c] = v;
        }
        __syncthreads();
        if (threadIdx.x < kSchedClasses + 2) {
            const unsigned long long tot = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
            if (threadIdx.x < kSchedClasses) s_base[threadIdx.x] = tot;
            else if (blockIdx.x == 0) {
                if (threadIdx.x == kSchedClasses) sc->ord_na = (int64_t)tot;
                else sc->ord_n = (int64_t)tot;
            }
        }
        __syncthreads();
    }
    const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
    for (int64_t i0 = lo; i0 < hi; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        const int c = i < hi ? cls[i] : 255;
        unsigned int rank = 0;
#pragma unroll
        for (int k = 0; k < kSchedClasses; k++) {
            const unsigned long long m = __ballot(c == k);
            if (c == k) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_wcnt[wave][k] = __popcll(m);
        }
        __syncthreads();
        if (c < kSchedClasses) {
            unsigned int before = 0;
            for (int w2 = 0; w2 < wave; w2++) before += s_wcnt[w2][c];
            order[s_base[c] + before + rank] = (int32_t)i;
        }
        __syncthreads();
        if (threadIdx.x < kSchedClasses) s_base[threadIdx.x] += s_wcnt[0][threadIdx.x] + s_wcnt[1][threadIdx.x] + s_wcnt[2][threadIdx.x] + s_wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// tile > 0 (gene-wise launch with the schedule on): block b owns rows [b tile, (b + 1) tile) and leaves its class counts in
// hist (the first half of order_*); tile == 0: rows grid-strided
template <bool MAP>
__global__ __launch_bounds__(256) void disp_init_kernel(FitDims d, FitWork w, Opts o, int64_t tile, unsigned int *hist, int xim_here) {
    const FitScalars *sc = w.sc;
    const double c0 = sc->coefs[0], c1 = sc->coefs[1];
    // xim = mean_j 1 / (colsum_j / nnz) (momentsDispEstimate).  xim_here (single rank, round 6): formed here from the column sums, by every
    // thread with xim_kernel's own operations (same bits), instead of by a one-thread launch of its own in front of this one (~4 us)
    double xim = 0.0;
    if (!MAP) {
        if (xim_here) {
            double x = 0;
            for (int j = 0; j < d.S; j++) x += 1.0 / (sc->colsum[j] / sc->nnz);
            xim = x / d.S;
            if (blockIdx.x == 0 && threadIdx.x == 0) w.sc->xim = xim;
        } else {
            xim = sc->xim;
        }
    }
    const double out_thr = MAP ? o.outlierSD * sqrt(sc->varLogDispEsts) : 0.0;
    unsigned int mine[kSchedClasses] = {0, 0, 0, 0, 0, 0};
    const int64_t lo = tile > 0 ? (int64_t)blockIdx.x * tile : (int64_t)blockIdx.x * 256;
    const int64_t hi = tile > 0 ? (lo + tile < d.n ? lo + tile : d.n) : d.n;
    const int64_t step = tile > 0 ? 256 : (int64_t)gridDim.x * 256;
    // (the next row's inputs are loaded before the current row is worked on: a thread walks ~8 rows, and with the loads issued
    // only when their row's turn came every row cost a full memory round trip)
    struct In { int az; double bm, a, b, g0, g1; };
    auto load_in = [&](int64_t i) {
        In r;
        r.az = w.allZero[i];
        r.bm = w.baseMean[i];
        if (!MAP) {
            r.a = w.baseVar[i];
            r.b = w.rough[i];
            r.g0 = w.gm0[i];
            r.g1 = w.gm1[i];
        } else {
            r.a = w.dispGene[i];
            r.b = sc->trend_local ? w.dispFit[i] : 0.0;
            r.g0 = r.g1 = 0.0;
        }
        return r;
    };
    int64_t i = lo + threadIdx.x;
    In nxt{};
    if (i < hi) nxt = load_in(i);
    for (; i < hi; i += step) {
        const In cur = nxt;
        if (i + step < hi) nxt = load_in(i + step);
        if (cur.az) {
            if (!MAP) {  // not scheduled at all (order_*): the line search never sees the row
                w.cls[i] = 255;
                w.dispGene[i] = NAN;
                w.geneIter[i] = 0;
            }
            continue;
        }
        const double bm = cur.bm;
        if (!MAP) {
            const double moments = (cur.a - xim * bm) / (bm * bm);
            const double a0 = fmin(fmax(o.minDisp, fmin(cur.b, moments)), o.maxDisp);
            const double g0 = cur.g0, g1 = cur.g1;
            reinterpret_cast<double2 *>(w.start)[2 * i] = make_double2(a0, log(a0));  // what the search reads with the row
            const int c = sched_class(a0, d.p == 2 ? fmin(g0, g1) : g0, o.minDisp);
            w.cls[i] = (uint8_t)c;
#pragma unroll
            for (int k = 0; k < kSchedClasses; k++) mine[k] += (c == k);
        } else {
            const double dg = cur.a, df = sc->trend_local ? cur.b : c0 + c1 / bm;
            const double ldf = log(df);
            w.dispFit[i] = df;
            const int is_out = log(dg) > ldf + out_thr;
            reinterpret_cast<double2 *>(w.start)[2 * i] = make_double2(dg > 0.1 * df ? log(dg) : ldf, ldf);  // start value, prior mean
            reinterpret_cast<double2 *>(w.start)[2 * i + 1] = make_double2(dg, (double)is_out);  // ... and what the search hands through: the gene-wise estimate, the outlier flag
            w.outlier[i] = is_out;
            if (is_out) w.disp[i] = dg;  // an outlier keeps its gene-wise estimate (A4): written HERE, so that the search carries one bit for it, not the value
        }
    }
    if (!MAP && tile > 0) order_hist_store(mine, hist);
}

constexpr int kChunk = 64;   // rows a wave takes from the global queue per atomic (at most: DispArgs::chunk)
constexpr int kTabSlots = 10;  // LDS slots (64 doubles each) of a wave's prefix table = its samples-across-lanes exchange area (128 entries of 36 bytes)
static inline size_t disp_lds_per_wave(int S) { return (size_t)kTabSlots * 64 * 8 + (size_t)S * 64 * 12 + 3 * 64 * 4; }
enum Phase : int { PH_NEED = 0, PH_INIT = 1, PH_SEARCH = 2, PH_DONE = 5 };

// One row of FitWork::rowpack -> the lane's LDS column, with mu_j = max(nf_j * groupmean_g, minmu) formed on the way (what the
// line search needs of nf_j).  All of the record's 16-byte loads are in flight before the first is used (S a multiple of four up
// to 16: one round trip instead of one per four samples), and the all-zero flag comes with the record (sign bit of the first header
// word, set by prep) instead of from a load of its own in front of it.  Returns false for an all-zero row.
__device__ __forceinline__ double max_num(double x, double m) {  // fmax() without the canonicalising copies of its operands
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
    return r;
}
template <int Q>  // S = 4 Q
__device__ __forceinline__ bool load_row_mu_fixed(const char *row, double *s_nf, int *s_y, int lane, uint64_t gmask, double minmu) {
    const double2 *p = reinterpret_cast<const double2 *>(row);
    const int4 *py = reinterpret_cast<const int4 *>(row + kRowHdr + 32 * Q);
    const double2 h0 = p[0];  // the two group means; sign bit of the first: all-zero row
    const int4 prof = *reinterpret_cast<const int4 *>(row + 16);  // the count profile (prep)
    double2 f[2 * Q];
    int4 y[Q];
#pragma unroll
    for (int k = 0; k < 2 * Q; k++) f[k] = p[2 + k];
#pragma unroll
    for (int k = 0; k < Q; k++) y[k] = py[k];
    // (no branch on the flag here: an all-zero row's values go to the lane's LDS column like any other's and are never used —
    // with a branch the compiler moves the loads behind it, i.e. behind the wait for the header.)  The sample's group is a bit
    // of a wave-uniform word; it is taken from a copy the compiler cannot see through, or it builds all S lane masks outside
    // the launch's main loop and spills them.
    uint32_t gbits = (uint32_t)gmask;
    asm volatile("" : "+s"(gbits));
#pragma unroll
    for (int k = 0; k < 2 * Q; k++) {
        s_nf[(2 * k) * 64 + lane] = max_num(f[k].x * (((gbits >> (2 * k)) & 1u) ? h0.y : h0.x), minmu);
        s_nf[(2 * k + 1) * 64 + lane] = max_num(f[k].y * (((gbits >> (2 * k + 1)) & 1u) ? h0.y : h0.x), minmu);
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
        s_y[(4 * k) * 64 + lane] = y[k].x; s_y[(4 * k + 1) * 64 + lane] = y[k].y;
        s_y[(4 * k + 2) * 64 + lane] = y[k].z; s_y[(4 * k + 3) * 64 + lane] = y[k].w;
    }
    s_y[(4 * Q) * 64 + lane] = prof.x;
    s_y[(4 * Q + 1) * 64 + lane] = prof.y;
    s_y[(4 * Q + 2) * 64 + lane] = prof.z;
    return __double2hiint(h0.x) >= 0;
}
__device__ __forceinline__ bool load_row_mu(const char *row, int S, double *s_nf, int *s_y, int lane, uint64_t gmask, double minmu) {
    if (S == 8) return load_row_mu_fixed<2>(row, s_nf, s_y, lane, gmask, minmu);
    if (S == 4) return load_row_mu_fixed<1>(row, s_nf, s_y, lane, gmask, minmu);
    if (S == 16) return load_row_mu_fixed<4>(row, s_nf, s_y, lane, gmask, minmu);
    if (S == 12) return load_row_mu_fixed<3>(row, s_nf, s_y, lane, gmask, minmu);
    double hdr[2];
    {
        const double2 h0 = reinterpret_cast<const double2 *>(row)[0];
        hdr[0] = h0.x; hdr[1] = h0.y;
    }
    if (__double2hiint(hdr[0]) < 0) return false;
    const double *pf = reinterpret_cast<const double *>(row + kRowHdr);
    const int *py = reinterpret_cast<const int *>(row + kRowHdr + 8 * S);
    const int4 prof = *reinterpret_cast<const int4 *>(row + 16);  // the count profile (prep)
    for (int j = 0; j < S; j++) {
        s_nf[j * 64 + lane] = max_num(pf[j] * (((gmask >> j) & 1) ? hdr[1] : hdr[0]), minmu);
        s_y[j * 64 + lane] = py[j];
    }
    s_y[S * 64 + lane] = prof.x;
    s_y[(S + 1) * 64 + lane] = prof.y;
    s_y[(S + 2) * 64 + lane] = prof.z;
    return true;
}

struct DispArgs {
    const int32_t *counts;
    const double *nf;
    FitDims d;
    FitWork w;
    Opts o;
    unsigned long long *stamps;  // CHICDIFF_DIAG builds only (make DIAG=1): per wave timestamps and tick counts
    int spread;                  // 0 = row-per-lane evaluation only (option "line_search_spread", for the bit-identity test)
    const int32_t *order;        // gene-wise launch: the schedule (order_*); NULL = rows 0..n-1 through the queue (MAP, option "line_search_schedule" 0)
    int deal;                    // entries per group of the static deal (0 = by the number of entries per wave)
    int prefetch;                // 1 = warm the cache lines of the rows handed out next
    int chunk;                   // rows per dequeue (<= kChunk)
    int32_t *gridlist;           // rows whose line search did not converge: fitDispGrid's two stages run in disp_grid_kernel (round 6)
    unsigned int *gridcount;     // ... and their number
    int prio;                    // > 0: a wave raises its issue priority (s_setprio) by one level per `prio` iterations of its oldest search
};
// make ISA_MARK=1 (tools/isa_account.py): comment lines in the generated assembly that delimit the parts of a tick; a volatile asm
// statement also keeps the compiler from moving code across it, so the marked build is for counting, not for running
#ifdef CHICDIFF_ISA_MARK
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name)
#endif
#ifdef CHICDIFF_DIAG
#define DIAG(...) __VA_ARGS__
constexpr int kStampSlots = 34;  // start, queue-empty, exit (s_memrealtime), live rows at queue-empty, ticks after queue-empty: row-per-lane / spread / burst, all ticks, s_memtime cycles after queue-empty in row / spread / burst ticks, spare
#else
#define DIAG(...)
#endif

// log posterior of a = log(alpha) and its derivative for one row held in LDS (A2.6).
//
// With r = 1/alpha, the per-sample terms of DESeq2's log_posterior / dlog_posterior are
//   lgamma(y+r) - lgamma(r) - y log(mu+r) - r log(1+mu alpha)            (value)
//   digamma(r) - digamma(y+r) + log(1+mu alpha) - mu alpha/(1+mu alpha) + y/(mu+r)   (derivative)
// and are evaluated here as
//   * log(mu+r) = log(1+mu alpha) - a, so one log L covers both logs;
//   * y is an integer count: lgamma(y+r)-lgamma(r) = log prod_{i<n}(r+i) + [lgS(y+r)-lgS(r+n)]
//     with n = min(y, nr), nr = the number of unit steps that lift r to >= 10 (per row and tick),
//     lgS = Stirling's series (valid as both arguments are >= 10); likewise for digamma with the
//     derivative of the product.  Samples with y <= nr need no Stirling term, samples on rows
//     with alpha <= 0.1 need no product;
//     The nr prefix products P_1 .. P_nr are tabulated once per row and tick in LDS (P_0 = 1); the harmonic sums are not: summed
//     over the samples they are sum_{i<nr} c_i / (r+i) with the row's count profile c_i = #{j : y_j > i} (CountProfile above),
//     ten terms at row level (round 6; before: H_n tabulated beside P_n, 22 LDS slots per lane instead of 10);
//   * the products of all samples are multiplied up (mantissa/exponent) and logged ONCE per row.
// mu_j = max(nf_j * groupmean_g, minmu) sits in LDS (formed when the row is staged).
struct RowConsts {  // what depends only on the evaluation point a = log(alpha)
    double a, alpha, r, lgS0, dgS0;
    int nr;
};
__device__ __forceinline__ RowConsts row_consts(double a, const LogEntry *lt, const ExpEntry *et) {
    RowConsts c;
    c.a = a;
    c.alpha = texp(a, et);
    c.r = rcp(c.alpha);
    c.nr = c.r < 10.0 ? (int)ceil(10.0 - c.r) : 0;  // unit steps lifting r to r0 = r + nr >= 10
    const double r0 = c.r + (double)c.nr;
    stirling(r0, tlog(r0, lt), rcp(r0), c.lgS0, c.dgS0);
    return c;
}
struct Acc {  // sums over samples
    double ll = 0, sd = 0, wA = 0, wB = 0, dA = 0, dB = 0;
    double pm = 1.0;  // product of the samples' shift products (mantissas) ...
    int pe = 0;       // ... and of their binary exponents
};
// One sample's contribution as five finished values, and the step that folds them into the row sums.
// The library is compiled with -ffp-contract=off and every fused operation is written out, so the two
// evaluation layouts below (row per lane / samples across lanes) produce the same bits: both call
// sample_values() on the same inputs and both fold the S results in sample order with accumulate().
struct SampleVals {
    double wj, pm, tll, tsd;
    int pe;
};
// P = prod_{i<n}(r+i) for n = min(y, nr) (the harmonic sums H_n of the derivative are added at row level: harmonic_row)
// mu = max(nf_j * groupmean_g, minmu) does not change during a row's search: it is formed once, when the row is staged, and kept in
// the LDS column in place of nf_j (round 3: five instructions per sample and tick less, two shuffled operands less per
// samples-across-lanes tick; same product, same bits)
__device__ __forceinline__ SampleVals sample_values(const RowConsts &c, double mu, int yi, double P, const LogEntry *lt) {
    SampleVals v;
    const double y = (double)yi;
    const double ma = mu * c.alpha;
    const double t = 1.0 + ma;
    const double rt = rcp(t);
    const double L = tlog1p_from(ma, t, rt, lt);
    v.wj = mu * rt;  // 1 / (1/mu + alpha)
    double dlg = 0.0, ddg = 0.0;
    v.pe = __builtin_amdgcn_frexp_exp(P);
    v.pm = __builtin_amdgcn_frexp_mant(P);
    // (Measured, round 4: the two halves of a sample — log1p(mu alpha) with its reciprocal, the Stirling difference at z = y + r
    // with its logarithm and reciprocal — written side by side and branch-free, so that the compiler interleaves the two
    // dependency chains: bit-identical, 207 VGPRs instead of 196, and no faster — gene-wise 1.560 -> 1.565 ms at 2 M x 8,
    // 0.603 -> 0.614 at 250 k.  profiles/r04_ab_line_search_trims.txt)
    if (yi > c.nr) {
        const double z = y + c.r;
        double lgz, dgz;
        stirling(z, tlog(z, lt), rcp(z), lgz, dgz);
        dlg = lgz - c.lgS0;
        ddg = dgz - c.dgS0;
    }
    v.tll = fma(-c.r, L, fma(-y, L - c.a, dlg));          // dlg - y (L - a) - r L
    v.tsd = fma(y * c.alpha, rt, fma(-ma, rt, L - ddg));  // L - ddg - ma/t + y alpha/t
    return v;
}
__device__ __forceinline__ void accumulate(Acc &acc, const SampleVals &v, bool g) {
    // the sample's group is wave-uniform: multiply by an exact 1.0 / 0.0 (scalar operands) instead of selecting
    // registers — x*1 + s and x*0 + s round exactly like s + x and s, at 6 instructions instead of 14
    const double gB = g ? 1.0 : 0.0, gA = g ? 0.0 : 1.0;
    const double tA = v.wj * gA, tB = v.wj * gB;
    acc.wA += tA;
    acc.wB += tB;
    acc.dA = fma(-tA, v.wj, acc.dA);
    acc.dB = fma(-tB, v.wj, acc.dB);
    acc.pe += v.pe;
    acc.pm *= v.pm;
    acc.ll += v.tll;
    acc.sd += v.tsd;
}
// sum over the samples of H_{min(y_j, nr)} = sum_{i < nr} c_i / (r + i), i ascending, from the row's count profile (three ints, ten
// bytes).  TABLE: the same ten steps also leave the prefix products P_1 .. P_10 in the lane's LDS column (entries beyond the lane's nr
// are never read; all ten steps in every lane, no trip count: round 4).  Both evaluation layouts call this with the same operands —
// same bits.  r < 6e30 keeps r^10 finite.
template <bool TABLE>
__device__ __forceinline__ double harmonic_row(const RowConsts &c, unsigned int w0, unsigned int w1, unsigned int w2, double *s_tab, int lane) {
    // bytes i >= nr of the profile do not count
    const unsigned int nr = (unsigned int)c.nr;
    const unsigned int a = nr < 4u ? nr : 4u, b = nr < 4u ? 0u : (nr < 8u ? nr - 4u : 4u), d = nr < 8u ? 0u : nr - 8u;
    w0 &= a == 4u ? 0xffffffffu : ((1u << (8u * a)) - 1u);
    w1 &= b == 4u ? 0xffffffffu : ((1u << (8u * b)) - 1u);
    w2 &= (1u << (8u * d)) - 1u;
    double P = 1.0, Hr = 0.0, zz = c.r;
#ifdef HR_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int i = 0; i < 10; i++) {
        const unsigned int w = i < 4 ? w0 : (i < 8 ? w1 : w2);
        const double ci = (double)((w >> (8 * (i & 3))) & 0xffu);
        Hr = fma(ci, rcp(zz), Hr);
        if (TABLE) {
            P *= zz;
            s_tab[i * 64 + lane] = P;
        }
        zz += 1.0;
#ifdef HR_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return Hr;
}
__device__ __forceinline__ void finish_point(const Acc &acc, const RowConsts &c, double Hrow, bool p2, bool use_prior,
                                             double prior_mean, double prior_isig, double &lp, double &dlp,
                                             const LogEntry *lt) {
    const double ll = acc.ll + fma((double)acc.pe, 0.69314718055994530942, tlog(acc.pm, lt));
    double cr, dcr;
    if (p2) {
        cr = -0.5 * tlog(acc.wA * acc.wB, lt);
        dcr = -0.5 * (acc.dA * rcp(acc.wA) + acc.dB * rcp(acc.wB));
    } else {
        cr = -0.5 * tlog(acc.wA, lt);
        dcr = -0.5 * (acc.dA * rcp(acc.wA));
    }
    double pr = 0, dpr = 0;
    if (use_prior) {
        const double dd = c.a - prior_mean;
        pr = -0.5 * dd * dd * prior_isig;
        dpr = -dd * prior_isig;
    }
    lp = ll + pr + cr;
    dlp = (c.r * c.r * (acc.sd - Hrow) + dcr) * c.alpha + dpr;
}

// Row-per-lane evaluation: all S samples of the row in LDS column `slot` (the lane's own row; in disp_grid_kernel the column of
// the row whose grid point the lane evaluates), the prefix table in the lane's own column.
__device__ __forceinline__ void eval_point(const double *s_nf, const int *s_y, double *s_tab, int lane, int slot, int S, uint64_t gmask,
                                           bool p2, double a,
                                           bool use_prior, double prior_mean, double prior_isig,
                                           double &lp, double &dlp, double &alpha_out, const LogEntry *lt, const ExpEntry *et
                                           DIAG(, unsigned long long *tm)) {
    MARK("row:begin");
    DIAG(tm[0] = __builtin_amdgcn_s_memtime();)
    const RowConsts c = row_consts(a, lt, et);
    alpha_out = c.alpha;
    MARK("row:row_consts_end");
    DIAG(tm[1] = __builtin_amdgcn_s_memtime();)
    // per-tick table (LDS, [entry][lane]): P_n for n = 1..10, and the row-level harmonic sum from the count profile of the row in
    // column `slot` (harmonic_row: ten unconditional steps — in-kernel timers, round 4: a loop to the lane's own nr ran, in SIMD, to the
    // wave's largest, and unrolled by eight plus a remainder loop)
    double Hrow = 0.0;
    if (__ballot(c.nr > 0) != 0ull)
        Hrow = harmonic_row<true>(c, (unsigned int)s_y[S * 64 + slot], (unsigned int)s_y[(S + 1) * 64 + slot], (unsigned int)s_y[(S + 2) * 64 + slot], s_tab, lane);
    MARK("row:table_end");
    DIAG(tm[2] = __builtin_amdgcn_s_memtime();)
    Acc acc;
    for (int j = 0; j < S; j++) {
        const int yi = s_y[j * 64 + slot];
        const int n = yi < c.nr ? yi : c.nr;
        const bool g = (gmask >> j) & 1;
        const double Pt = s_tab[((n > 0 ? n : 1) - 1) * 64 + lane];  // (n = 0: P_0 = 1; the entry read instead is never used)
        accumulate(acc, sample_values(c, s_nf[j * 64 + slot], yi, n > 0 ? Pt : 1.0, lt), g);
    }
    MARK("row:samples_end");
    DIAG(tm[3] = __builtin_amdgcn_s_memtime();)
    finish_point(acc, c, Hrow, p2, use_prior, prior_mean, prior_isig, lp, dlp, lt);
    MARK("row:finish_end");
    DIAG(tm[4] = __builtin_amdgcn_s_memtime();)
}

// Samples-across-lanes evaluation for the end of the launch.  Once the queue is empty every wave is left
// with a dozen rows that still need up to ~130 serial evaluations (flat likelihoods: DESeq2's step size
// decays faster than the search converges, then the grid takes over), and a row-per-lane tick costs the same
// ~1600 instructions whether 64 lanes or one are busy.  When at most 64/G rows are live (G = 2^lg lanes per row), the
// g-th live row is evaluated by lanes G*g .. G*g+G-1 — one sample each when G >= S, else samples jj, jj + G, ... (S = 8:
// four lanes per row for 9-16 live rows, two for 17-32; round 3) —: every lane rebuilds the row
// constants (same instructions, no extra cost in SIMD), walks its own prefix product, computes its sample's
// five values, then every lane of the group folds the S results in sample order (so each holds the row's
// sums) and the owning lane picks the result up.  Same functions, same operand values, same order of the
// floating-point operations as eval_point(): the bits do not depend on which layout evaluated a tick,
// hence not on the schedule (tests/test_gpu_parity.py::test_line_search_layouts_agree_bit_for_bit).
// Gain at 2 M rows: 1 % of the gene-wise launch at S = 8, 4 % at S = 4, 10 % at S = 16.
// (Measured and dropped: handing the stragglers to a second, densely packed launch — a wave's tick takes
// ~5 us alone or with a neighbour on its SIMD, the tail is bound by the ~130 serial ticks, not by issue.)
// Which lanes evaluate which live row in a samples-across-lanes tick.  It depends only on the set of live lanes (and, for the MAP
// search, on the rows' prior means), so the launch's end — tick after tick with the same few rows — builds it once per change of
// that set (round 5) instead of once per tick.
struct SpreadMap {
    unsigned long long mask;  // the live lanes it was built for
    int lg;                   // log2(lanes per row); -1: the rows do not fit (row-per-lane tick)
    int owner;                // the lane whose LDS column holds this lane's row
    int src;                  // first lane of the group that evaluates this lane's own row (live lanes)
    bool has;                 // this lane's group has a row
    double pm_o;              // the row's prior mean (MAP)
};
__device__ __forceinline__ SpreadMap spread_map(double *s_x, int lane, int lg, unsigned long long actmask, bool active, bool use_prior, double prior_mean) {
    SpreadMap m;
    m.mask = actmask;
    m.lg = lg;
    const int grp = lane >> lg;
    // owner of group g = the g-th live lane: every live lane leaves its number at its rank among the live lanes, group g reads entry
    // g (one LDS round trip instead of a walk over the set bits: ~50 instructions per tick of a launch's latency-bound end)
    const int nact = __popcll(actmask);
    const int myrank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(actmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)actmask, 0u));
    int *s_own = reinterpret_cast<int *>(s_x);  // in the exchange area itself: read here, before an evaluation writes the area again (a wave's LDS operations execute in order)
    __builtin_amdgcn_wave_barrier();            // (every lane has read the last evaluation's values)
    if (active) s_own[myrank] = lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    m.has = grp < nact;
    m.owner = m.has ? s_own[grp] : 0;
    __builtin_amdgcn_wave_barrier();            // (... and the look-up is over before the area is written again)
    m.src = (active ? myrank : 0) << lg;  // an active lane's group is its rank among the active lanes
    m.pm_o = use_prior ? __shfl(prior_mean, m.owner) : 0.0;
    return m;
}
__device__ __forceinline__ void eval_point_spread(const double *s_nf, const int *s_y, double *s_x, int lane, int S, const SpreadMap &map, uint64_t gmask,
                                                  bool p2, double a_eval,
                                                  bool use_prior, double prior_isig,
                                                  double &lp, double &dlp, double &alpha_out, const LogEntry *lt, const ExpEntry *et
                                                  DIAG(, unsigned long long *tm)) {
    DIAG(tm[0] = __builtin_amdgcn_s_memtime();)
    // lanes per row L = 2^lg: one sample per lane when L >= S (at most 64 / L rows), else samples jj, jj + L, ... per lane — the
    // layout also serves 9 .. 32 live rows (S = 8: four or two lanes per row), where a row-per-lane tick would still walk all S
    // samples in every lane
    const int lg = map.lg;
    const int L = 1 << lg, grp = lane >> lg, jj = lane & (L - 1), R = 64 >> lg;
    const bool has = map.has;
    const int owner = map.owner;
    MARK("spread:owner_walk_end");
    DIAG(tm[1] = __builtin_amdgcn_s_memtime();)
    const double a_o = __shfl(a_eval, owner);
    const double pm_o = map.pm_o;
    const RowConsts c = row_consts(a_o, lt, et);
    MARK("spread:row_consts_end");
    DIAG(tm[2] = __builtin_amdgcn_s_memtime();)
    // the five values of every sample pass through the wave's prefix-table area (idle in this layout), [value][sample R + group]:
    // each lane then reads its group's samples — four samples' loads in flight at a time, same address within a group (a
    // broadcast), neighbouring banks across groups — and folds them in sample order.  (Round 2 fetched them with nine
    // ds_bpermute per sample inside the fold loop: one LDS round trip per sample on the critical path of a tick that is all
    // latency.)  The area holds 128 entries (4 x 128 doubles + 128 ints of the 10 x 64 doubles: round 6, when the table lost its
    // harmonic half); a layout of S R <= 256 entries goes through it in two rounds of ceil(S / 2) samples each, folded in sample
    // order as before: same bits.
    // the row-level harmonic sum from the owner's count profile (every lane of the group: same instructions, no extra cost in SIMD);
    // skipped — profile reads included — when no row of the wave has r < 10 (the flat-likelihood rows of a launch's end never have)
    double Hrow = 0.0;
    if (__ballot(c.nr > 0) != 0ull) {
        const unsigned int w0 = has ? (unsigned int)s_y[S * 64 + owner] : 0u, w1 = has ? (unsigned int)s_y[(S + 1) * 64 + owner] : 0u;
        const unsigned int w2 = has ? (unsigned int)s_y[(S + 2) * 64 + owner] : 0u;
        Hrow = harmonic_row<false>(c, w0, w1, w2, nullptr, lane);
    }
    int *s_xe = reinterpret_cast<int *>(s_x + 4 * 128);
    Acc acc;
    // one round of the exchange: the samples [j0, j1) of every group's row
    auto round = [&](const int j0, const int j1, const int js) {
        for (int j = js; j < j1; j += L) {  // (the same trip count in every lane of a group up to the guard)
            const int yi = has ? s_y[j * 64 + owner] : 0;
            const double nfj = has ? s_nf[j * 64 + owner] : 1.0;
            double P = 1.0;
            {
                const int n = yi < c.nr ? yi : c.nr;
                double zz = c.r;
                for (int i = 0; i < n; i++) {  // the same recurrence as the table of eval_point(), stopped at entry n
                    P *= zz;
                    zz += 1.0;
                }
            }
            MARK("spread:prefix_walk_end");
            const SampleVals v = sample_values(c, nfj, yi, P, lt);
            MARK("spread:sample_end");
            const int e = (j - j0) * R + grp;
            s_x[e] = v.wj;
            s_x[128 + e] = v.pm;
            s_x[256 + e] = v.tll;
            s_x[384 + e] = v.tsd;
            s_xe[e] = v.pe;
        }
        MARK("spread:exchange_store_end");
        DIAG(if (j0 == 0) tm[3] = __builtin_amdgcn_s_memtime();)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q0 = j0; q0 < j1; q0 += 4) {
            SampleVals u[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int src = ((q0 + t < j1 ? q0 + t : q0) - j0) * R + grp;
                u[t].wj = s_x[src];
                u[t].pm = s_x[128 + src];
                u[t].tll = s_x[256 + src];
                u[t].tsd = s_x[384 + src];
                u[t].pe = s_xe[src];
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (q0 + t < j1) accumulate(acc, u[t], (gmask >> (q0 + t)) & 1);
        }
        __builtin_amdgcn_wave_barrier();  // (the area is written again only after every lane has read it)
    };
    if (S * R <= 128) {  // (the usual case at a launch's end: one round, as before round 6)
        round(0, S, jj);
    } else {
        const int Sh = (S + 1) >> 1;
        round(0, Sh, jj);
        round(Sh, S, jj >= Sh ? jj : jj + (((Sh - jj + L - 1) >> lg) << lg));  // (this lane's first sample of the second round)
    }
    MARK("spread:fold_end");
    DIAG(tm[4] = __builtin_amdgcn_s_memtime();)
    double lp_g, dlp_g;
    finish_point(acc, c, Hrow, p2, use_prior, pm_o, prior_isig, lp_g, dlp_g, lt);
    MARK("spread:finish_end");
    DIAG(tm[5] = __builtin_amdgcn_s_memtime();)
    const int src = map.src;
    lp = __shfl(lp_g, src);
    dlp = __shfl(dlp_g, src);
    alpha_out = __shfl(c.alpha, src);
    MARK("spread:pickup_end");
    DIAG(tm[6] = __builtin_amdgcn_s_memtime();)
}

template <bool MAP, int MINW>
__global__ __launch_bounds__(256, MINW) void disp_fit_kernel(DispArgs A) {
    extern __shared__ double smem[];
    __shared__ LogEntry s_logtab[64];
    __shared__ ExpEntry s_exptab[64];
    exp_table_to_lds(s_exptab);
    log_table_to_lds(s_logtab);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (tells the compiler it is wave-uniform: what derives from it — LDS bases, the wave's place in the deal — stays in scalar registers)
    const int S = A.d.S;
    const int64_t n = A.d.n;
    // per wave: 10*64 doubles (prefix products P_1..P_10; the samples-across-lanes exchange) + S*64 doubles (mu) + (S + 3)*64 ints (counts, count profile)
    double *s_tab = smem + (size_t)wave * (kTabSlots * 64 + S * 96 + 96);
    double *s_nf = s_tab + kTabSlots * 64;
    int *s_y = reinterpret_cast<int *>(s_nf + S * 64);
    const uint64_t gmask = A.d.gmask;
    const bool p2 = A.d.p == 2;
    const Opts o = A.o;
    const double min_log_alpha = log(o.minDisp / 10.0);
#ifdef EXP_NO_SPREAD
    int spread_lg = -1;
#else
    int spread_lg = A.spread ? 1 : -1;  // log2(lanes per row) of the samples-across-lanes layout; -1 = never
#endif
    while (spread_lg >= 0 && (1 << spread_lg) < S) spread_lg++;
    FitScalars *sc = A.w.sc;
    // fit-wide scalars (uniform)
    const double prior_isig = MAP ? 1.0 / sc->dispPriorVar : 0.0;
    // eight queue heads, 64 bytes apart (chunk c = 8 k + h is the k-th chunk of head h; a wave starts at its XCD's head and moves on
    // when a head runs dry): one head is one hot word, and every dequeue stalls its wave for the round trip
    // (MAP launch only — natural row order, every row through the queue: 1.23 -> 1.22 ms at 2 M x 8, 0.233 -> 0.219 at 250 k; the
    // gene-wise launch, whose queue carries the classes in order behind the static deal, lost 1.6 % with it and keeps one head)
    unsigned long long *heads = A.w.queue + (MAP ? 192 : 0);
    int cur_head = MAP ? (blockIdx.x & 7) : 0;
    unsigned int heads_left = MAP ? 0xffu : 0x1u;
    constexpr unsigned long long kHeads = MAP ? 8ull : 1ull;

    int phase = PH_NEED, iter = 0, iacc = 0;
    int row = -1;
    double a = 0, lp = 0, dlp = 0, kappa = 0, init_lp = 0, a0 = 0, prior_mean = 0;
    double a_new = 0, alpha_cur = 0;
    bool queue_empty = false;
    // the open chunk (wave-uniform): entries [chunk_pos, chunk_len) of it are still to be handed out; lane l holds the row of entry l
    uint32_t chunk_base = 0, chunk_pos = 0, chunk_len = 0;
    const uint32_t chunk_rows = (uint32_t)A.chunk;
    int ord_reg = 0;
    bool touch = false;
    const int64_t rstride = row_stride(S);
    // schedule (gene-wise launch): positions [0, nA) of order[] are dealt out statically — group g of kSchedDeal entries
    // belongs to wave g mod W — the positions [nA, nTot) go through the queue; without a schedule the queue covers rows 0..n-1
    const int32_t *__restrict__ order = MAP ? nullptr : A.order;
    // (32-bit positions: n < 2^31 — api.hip refuses more —, and a head's counter overshoots the end by a few chunks per wave at most;
    // the scalar unit has no ordered 64-bit compare, with 64-bit positions the whole bookkeeping moves to the vector unit)
    const uint32_t nA = order ? (uint32_t)sc->ord_na : 0u;
    const uint32_t nTot = order ? (uint32_t)sc->ord_n : (uint32_t)n;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t mywave = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave;
    uint32_t a_k = 0;
    bool a_done = nA == 0;
    unsigned int pf_val = 0, pf_acc = 0;
    // entries per group of the deal: eight while every wave gets several groups (neighbouring rows, neighbouring lanes), down to
    // one when rows are few, so that the likely-long rows spread evenly over the waves
    uint32_t deal = A.deal > 0 ? (uint32_t)A.deal : nA / (4u * nwaves);
    deal = deal < 1 ? 1 : (deal > (uint32_t)kSchedDeal ? (uint32_t)kSchedDeal : deal);
    DIAG(const int gwave = blockIdx.x * (blockDim.x >> 6) + wave;)
    DIAG(unsigned long long tk_lean = 0; unsigned long long rej_spread = 0, srch_spread = 0, rej_bulk = 0, srch_bulk = 0; unsigned long long cy_sp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sp_t6 = 0; bool sp_on = false; unsigned long long cy_ev[4] = {0, 0, 0, 0}; unsigned long long cy_sec[5] = {0, 0, 0, 0, 0}; bool stamped = false; unsigned long long tk_row = 0, tk_spread = 0, tk_burst = 0, tk_all = 0, cy_row = 0, cy_spread = 0, cy_burst = 0, cy_last = 0; int tk_kind = -1;
         if (A.stamps && lane == 0) A.stamps[gwave * kStampSlots + 0] = __builtin_amdgcn_s_memrealtime();)

    // the line search of this lane's row is over: its result, or the grid fallback (A2.7 / A4)
    auto search_over = [&](double &result, bool &have_result) {
        bool grid;
        if (!MAP) {
            double dd = fmin(alpha_cur, o.maxDisp);
            if (lp < init_lp + fabs(init_lp) / 1e6) dd = a0;  // noIncrease: keep alpha_init
            const bool conv = (iter < o.maxit) && (iter != 1);
            grid = !conv && dd > o.minDisp * 10;
            result = dd;
        } else {
            grid = !(iter < o.maxit);
            result = alpha_cur;
        }
        if (grid) {
            // fitDispGrid (A2.7 / A4) is not this kernel's business any more (round 6): the row goes on a list, and a small launch behind
            // this one evaluates the 20 points of each of its two stages ACROSS LANES (disp_grid_kernel).  The stages' state (point
            // counter, best value and index, centre of the second stage) and their code were carried by every wave of this kernel
            // for the 1.7 % of the rows that use them — registers that the three-waves-per-SIMD build had to spill —, and a row
            // inside a stage held its lane for 40 more serial evaluations (the end of a small fit's launch: ~105 + 40 ticks).
            const unsigned int at = atomicAdd(A.gridcount, 1u);
            A.gridlist[at] = MAP ? (row & 0x7fffffff) : row;
            if (!MAP) A.w.geneIter[row] = iter; else A.w.mapIter[row & 0x7fffffff] = iter;
            phase = queue_empty ? (int)PH_DONE : (int)PH_NEED;
        } else {
            have_result = true;
        }
    };
    auto store_result = [&](double result) {
        const double dd = fmin(fmax(result, o.minDisp), o.maxDisp);
        if (!MAP) {
            A.w.dispGene[row] = dd;
            A.w.geneIter[row] = iter;
        } else {
            // (the outlier flag rides in the sign bit of `row`, and an outlier's final dispersion — its gene-wise estimate — was written
            // by disp_init: rounds 4-5 carried estimate and flag through the search in three registers, the three the three-waves-per-
            // SIMD build was short of; a first version of round 6 read them back here, a dependent global load on nearly every tick of
            // the bulk: + 230 cycles per tick)
            const int r_ = row & 0x7fffffff;
            A.w.dispMAP[r_] = dd;
            if (row >= 0) A.w.disp[r_] = dd;
            A.w.mapIter[r_] = iter;
        }
    };
    const int lg_depth = A.spread == 3 ? 1 : 2;  // (option line_search_spread = 3: no layout of more than 128 exchange entries — an experiment of round 6)
    const bool lean_on = A.spread == 1 || A.spread == 3;  // (option line_search_spread = 2: samples across lanes, but every tick through the general path)
    SpreadMap lmap;
    lmap.mask = 0ull;
    lmap.lg = -1;
    lmap.owner = lmap.src = 0;
    lmap.has = false;
    lmap.pm_o = 0.0;

    for (;;) {
        // ---- lean tick of the launch's end (round 5) -----------------------------------------------------------------------
        // Once the queue is empty a wave is left with a few rows that creep along a flat likelihood for ~100 evaluations, one per
        // tick, in the samples-across-lanes layout: the launch's last 0.3-0.4 ms ARE that chain (250 k x 8: queue empty at 0.13 ms,
        // last wave out at 0.53), and a wave alone on its SIMD issues one instruction per ~8 cycles whatever it is — the tick costs
        // what it has instructions.  Of the ~870 of a general tick more than 400 were not evaluation but the tick's frame: the
        // refill's ballots and branches, the choice of the point and the state machine for all six phases behind divergent
        // branches, the grid-burst test, the owner look-up.  When every live lane of the wave is inside its line search — nine
        // ticks in ten — none of that is needed: the step, the evaluation, and the Armijo / stop update written with selects; the
        // lane map is rebuilt only when the set of live lanes changes; whatever is rare (a clamped step, a search that ends) sits
        // behind a wave-uniform branch.  Same operations on the same operands in the same order as the general tick: same bits
        // (test_line_search_layouts_agree_bit_for_bit runs both).
        if (lean_on && queue_empty && chunk_pos >= chunk_len && spread_lg >= 0) {
            const unsigned long long srch = __ballot(phase == PH_SEARCH);
            if (srch != 0ull && __ballot(phase != PH_SEARCH && phase != PH_DONE) == 0ull) {
                const bool active = phase == PH_SEARCH;
                if (srch != lmap.mask) {
                    const int nact = __popcll(srch), lg_min = spread_lg - lg_depth > 1 ? spread_lg - lg_depth : 1;
                    int lg_t = spread_lg;
                    while (lg_t >= lg_min && (nact << lg_t) > 64) lg_t--;
                    if (lg_t < lg_min) {
                        lmap.mask = srch;
                        lmap.lg = -1;
                    } else {
                        lmap = spread_map(s_tab, lane, lg_t, srch, active, MAP, prior_mean);
                    }
                }
                if (lmap.lg >= 0) {
                    DIAG({
                        const unsigned long long now = __builtin_amdgcn_s_memtime();
                        if (tk_kind == 0) cy_row += now - cy_last; else if (tk_kind == 1) cy_spread += now - cy_last; else if (tk_kind == 2) cy_burst += now - cy_last;
                        cy_last = now;
                        tk_all++;
                        tk_spread++;
                        tk_lean++;
                        tk_kind = 1;
                    })
                    MARK("lean:begin");
                    DIAG(const unsigned long long lean_t0 = __builtin_amdgcn_s_memtime();)
                    iter += active ? 1 : 0;
                    const double a_prop = a + kappa * dlp;
                    if (__ballot(active && (a_prop < -30.0 || a_prop > 10.0)) != 0ull) {  // (rare: the step leaves [-30, 10])
                        if (a_prop < -30.0) kappa = (-30.0 - a) / dlp;
                        if (a_prop > 10.0) kappa = (10.0 - a) / dlp;
                    }
                    a_new = a + kappa * dlp;
                    double l_new = 0, dl_new = 0, alpha_new = 0;
                    DIAG(unsigned long long tms[7];)
                    MARK("lean:eval_begin");
                    eval_point_spread(s_nf, s_y, s_tab, lane, S, lmap, gmask, p2, a_new, MAP, prior_isig, l_new, dl_new, alpha_new, s_logtab, s_exptab DIAG(, tms));
                    // Armijo test and stop rules of the general tick's PH_SEARCH branch, as selects
                    MARK("lean:eval_end");
                    const double theta_kappa = -l_new;
                    const double theta_hat_kappa = -lp - kappa * 1.0e-4 * dlp * dlp;
                    const bool acc = active && theta_kappa <= theta_hat_kappa;
                    const double change = l_new - lp;
                    const bool fin_tol = acc && change < o.dispTol;
                    const bool fin_low = acc && !fin_tol && a_new < min_log_alpha;
                    const bool go_on = acc && !fin_tol && !fin_low;
                    iacc += acc ? 1 : 0;
                    a = acc ? a_new : a;
                    alpha_cur = acc ? alpha_new : alpha_cur;
                    lp = (acc && !fin_low) ? l_new : lp;
                    dlp = go_on ? dl_new : dlp;
                    DIAG(if (active && !acc) rej_spread++; if (active) srch_spread++;)
                    {
                        double k_acc = fmin(kappa * 1.1, o.kappa0);
                        if (iacc % 5 == 0) k_acc *= 0.5;
                        kappa = go_on ? k_acc : ((active && !acc) ? kappa * 0.5 : kappa);
                    }
                    const bool finished = active && (fin_tol || fin_low || iter >= o.maxit);
                    MARK("lean:update_end");
                    if (__ballot(finished) != 0ull) {
                        if (finished) {
                            double result = 0;
                            bool have_result = false;
                            search_over(result, have_result);
                            if (have_result) {
                                store_result(result);
                                phase = PH_DONE;  // (the queue is empty: what the refill makes of PH_NEED)
                            }
                        }
                    }
                    MARK("lean:end");
                    DIAG(for (int q = 0; q < 6; q++) cy_sp[q] += tms[q + 1] - tms[q]; cy_sp[6] += tms[0] - lean_t0; cy_sp[7]++; cy_sp[8] += __builtin_amdgcn_s_memtime() - tms[6];)
                    continue;
                }
            }
        }
        // ---- issue priority by the age of the wave's oldest search (round 6) --------------------------------------------------------
        // The launch ends with its longest rows: ~105 dependent evaluations each, at the pace of a wave that shares its SIMD — with
        // three waves per SIMD a bulk tick takes ~12 us instead of ~9, and 105 of them are the whole launch.  A wave that holds an old
        // search asks the instruction arbiter for precedence over its neighbours (which lose what it gains: throughput is unchanged,
        // the critical path is shorter).
        if (A.prio > 0) {
            const int t1 = A.prio;
            const bool in_s = phase == PH_SEARCH;
            if (__ballot(in_s && iter >= 3 * t1) != 0ull) __builtin_amdgcn_s_setprio(3);
            else if (__ballot(in_s && iter >= 2 * t1) != 0ull) __builtin_amdgcn_s_setprio(2);
            else if (__ballot(in_s && iter >= t1) != 0ull) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        MARK("tick:begin");
        DIAG(const bool sec_on = !queue_empty; const unsigned long long sec_t0 = __builtin_amdgcn_s_memtime();)
        // ---- refill: lanes without a row pull the next ones from the queue -----------------
        // (Round 4: this part was a quarter of a tick's time in the bulk of the launch — in-kernel section timers, tools/stamps.py —,
        // most of it instruction issue: the chunk bookkeeping ran in vector registers behind divergent-looking branches, every row cost
        // three dependent loads (schedule entry, all-zero flag, record, the record in two trips) and the offsets were turned into
        // means by a read-modify-write loop over LDS.  Now: chunk state is wave-uniform (scalar registers, scalar branches); a chunk's
        // schedule entries are read once, when it opens, one per lane, and handed out with a lane permute; the all-zero flag comes
        // with the record; the record's loads are all in flight at once and the means are formed on the way into LDS.)
        unsigned long long needmask = __ballot(phase == PH_NEED);
#pragma unroll 1
        for (int attempt = 0; needmask != 0ull && attempt < (MAP ? 4 : 10); attempt++) {
            if (chunk_pos >= chunk_len) {
                if (queue_empty) {
                    if (phase == PH_NEED) phase = PH_DONE;
                    break;
                }
                uint32_t b, e;
                if (!a_done) {
                    // this wave's next 64 / deal groups of the static deal in one go: lane l reads entry l % deal of group a_k + l / deal
                    // (positions grow with the lane, so the entries inside [0, nA) are the first lanes).  One group per chunk meant a
                    // dependent load of 1 .. 8 schedule entries per attempt: at 250 k rows a refill of the launch's first 0.1 ms took
                    // 7 900 cycles of a 20 800-cycle tick (in-kernel timers, round 4).
                    const uint32_t per = 64u / deal, gq = (uint32_t)lane / deal, go = (uint32_t)lane - gq * deal;
                    const uint32_t pos = ((a_k + gq) * nwaves + mywave) * deal + go;
                    const bool valid = gq < per && pos < nA;
                    const unsigned long long vm = __ballot(valid);
                    a_k += per;
                    if (vm == 0ull) {
                        a_done = true;
                        continue;
                    }
                    chunk_base = 0;
                    chunk_len = (uint32_t)__popcll(vm);
                    chunk_pos = 0;
                    ord_reg = valid ? order[pos] : 0;
                    touch = true;
                } else {
                    if (heads_left == 0u) {
                        queue_empty = true;
                        continue;
                    }
                    // one atomic on the global head per kChunk rows (a single hot word saturates near 90 dequeues/us, which one
                    // atomic per tick per wave hit)
                    unsigned int kq = 0;
                    if (lane == 0) kq = (unsigned int)atomicAdd(heads + 8 * cur_head, 1ull);
                    kq = __builtin_amdgcn_readfirstlane(kq);
                    b = nA + (kq * (uint32_t)kHeads + (uint32_t)cur_head) * chunk_rows;
                    if (b >= nTot) {  // this head is dry: on to the next one that is not known to be (uses up one attempt)
                        heads_left &= ~(1u << cur_head);
                        if (MAP)
                            for (int t = 1; t <= 8; t++) {
                                const int hn = (cur_head + t) & 7;
                                if (heads_left & (1u << hn)) { cur_head = hn; break; }
                            }
                        continue;
                    }
                    e = b + chunk_rows < nTot ? b + chunk_rows : nTot;
                    chunk_base = b;
                    chunk_len = e - b;
                    chunk_pos = 0;
                    ord_reg = (int)chunk_base + lane;
                    if (order && (uint32_t)lane < chunk_len) ord_reg = order[chunk_base + lane];
                    touch = true;
                }
            }
            const uint32_t cnt = (uint32_t)__popcll(needmask), avail = chunk_len - chunk_pos;
            const uint32_t take = cnt < avail ? cnt : avail;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(needmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needmask, 0u));
            const int r = MAP ? (int)(chunk_base + chunk_pos + rank) : __shfl(ord_reg, (int)((chunk_pos + rank) & 63u));
            chunk_pos += take;
            if (phase == PH_NEED && rank < take) {
                const double2 st = reinterpret_cast<const double2 *>(A.w.start)[2 * (int64_t)r];  // start values (disp_init_kernel)
                double ol = 0.0;
                if (MAP) ol = A.w.start[4 * (int64_t)r + 3];  // outlier flag
                if (!load_row_mu(A.w.rowpack + (int64_t)r * rstride, S, s_nf, s_y, lane, gmask, o.minmu)) {
                    if (!MAP) {
                        A.w.dispGene[r] = NAN;
                        A.w.geneIter[r] = 0;
                    } else {
                        A.w.dispFit[r] = NAN;
                        A.w.dispMAP[r] = NAN;
                        A.w.disp[r] = NAN;
                        A.w.mapIter[r] = 0;
                        A.w.outlier[r] = 0;
                    }
                } else {
                    row = (MAP && ol != 0.0) ? (r | (int)0x80000000) : r;
                    if (!MAP) {
                        a0 = st.x;
                        a = st.y;
                    } else {
                        a = st.x;
                        prior_mean = st.y;
                    }
                    phase = PH_INIT;
                }
            }
            needmask = __ballot(phase == PH_NEED);
        }
        if (__ballot(phase != PH_DONE) == 0ull) break;
        MARK("tick:refill_end");
        DIAG(const unsigned long long sec_t1 = __builtin_amdgcn_s_memtime();)
        // warm the lines of the rows this wave hands out next (the rest of the chunk it has just opened): the load is consumed when
        // the next chunk opens, long after it has landed, so the coming refills find their rows in the cache instead of in HBM
        if (touch) {
            touch = false;
            pf_acc ^= pf_val;
            pf_val = 0;
            if (A.prefetch && (uint32_t)lane >= chunk_pos && (uint32_t)lane < chunk_len)
                pf_val = *reinterpret_cast<const unsigned int *>(A.w.rowpack + (int64_t)ord_reg * rstride) ^
                         *reinterpret_cast<const unsigned int *>(A.w.start + 4 * (int64_t)ord_reg);  // (the record's line and the start values')
        }
        DIAG(if (A.stamps && queue_empty && !stamped) {
            stamped = true;
            const int live = __popcll(__ballot(phase != PH_DONE && phase != PH_NEED));
            if (lane == 0) {
                A.stamps[gwave * kStampSlots + 1] = __builtin_amdgcn_s_memrealtime();
                A.stamps[gwave * kStampSlots + 3] = live;
            }
        })

        // ---- choose this tick's evaluation point -------------------------------------------
        // (Round 5: written with selects, the rare cases behind wave-uniform tests.  A branch around a lane-dependent block makes the
        // scalar unit wait for the vector compare — ~90 cycles for a lone wave, ~190 with the second wave of the bulk on the SIMD,
        // profiles/r05_lone_wave_latency.txt — and the if / else-if chains over the six phases, here and in the state machine below,
        // were a dozen of them per tick.  Same operations on the same operands: same bits.)
        const bool in_search = phase == PH_SEARCH, in_init = phase == PH_INIT;
        iter += in_search ? 1 : 0;
        {
            const double a_prop = a + kappa * dlp;
            if (__ballot(in_search && (a_prop < -30.0 || a_prop > 10.0)) != 0ull) {  // (rare: the step leaves [-30, 10])
                if (in_search && a_prop < -30.0) kappa = (-30.0 - a) / dlp;
                if (in_search && a_prop > 10.0) kappa = (10.0 - a) / dlp;
            }
        }
        a_new = in_search ? a + kappa * dlp : a_new;
        const double a_eval = in_search ? a_new : a;

        // ---- evaluate -------------------------------------------------------------------------
        MARK("tick:choose_point_end");
        double l_new = 0, dl_new = 0, alpha_new = 0;
        const bool active = phase != PH_DONE && phase != PH_NEED;
        const unsigned long long actmask = __ballot(active);
        // samples-across-lanes layout for this tick: the most lanes per row that still hold every live row (S = 8: eight lanes
        // for up to 8 rows, four for up to 16, two for up to 32); -1 = row per lane
        int lg_t = -1;
        if (queue_empty && spread_lg >= 0) {
            const int nact = __popcll(actmask), lg_min = spread_lg - lg_depth > 1 ? spread_lg - lg_depth : 1;
            lg_t = spread_lg;
            while (lg_t >= lg_min && (nact << lg_t) > 64) lg_t--;
            if (lg_t < lg_min) lg_t = -1;
        }
        DIAG({
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (tk_kind == 0) cy_row += now - cy_last; else if (tk_kind == 1) cy_spread += now - cy_last; else if (tk_kind == 2) cy_burst += now - cy_last;
            cy_last = now;
            tk_all++;
            tk_kind = -1;
            if (queue_empty) {
                if (lg_t >= 0) { tk_spread++; tk_kind = 1; }
                else { tk_row++; tk_kind = 0; }
            }
        })
        DIAG(const unsigned long long sec_t2 = __builtin_amdgcn_s_memtime();)
        if (lg_t >= 0) {
            DIAG(unsigned long long tms[7];)
            const SpreadMap map = spread_map(s_tab, lane, lg_t, actmask, active, MAP, prior_mean);
            eval_point_spread(s_nf, s_y, s_tab, lane, S, map, gmask, p2, a_eval, MAP, prior_isig, l_new, dl_new, alpha_new, s_logtab, s_exptab DIAG(, tms));
            DIAG(for (int q = 0; q < 6; q++) cy_sp[q] += tms[q + 1] - tms[q]; cy_sp[6] += tms[0] - sec_t0; cy_sp[7]++; sp_t6 = tms[6]; sp_on = true;)
        } else if (active) {
            DIAG(unsigned long long tm[5];)
            eval_point(s_nf, s_y, s_tab, lane, lane, S, gmask, p2, a_eval, MAP, prior_mean, prior_isig, l_new,
                       dl_new, alpha_new, s_logtab, s_exptab DIAG(, tm));
            DIAG(if (sec_on) for (int q = 0; q < 4; q++) cy_ev[q] += tm[q + 1] - tm[q];)
        }
        MARK("tick:evaluate_end");
        DIAG(const unsigned long long sec_t3 = __builtin_amdgcn_s_memtime();)
        // ---- advance the per-lane state machine ---------------------------------------------
        MARK("tick:burst_reduce_end");
        double result = 0;
        bool have_result = false;
        // a lane that entered this tick at its start point (in_init) or inside its line search (in_search): the update as selects
        const double theta_kappa = -l_new;
        const double theta_hat_kappa = -lp - kappa * 1.0e-4 * dlp * dlp;
        const bool acc = in_search && theta_kappa <= theta_hat_kappa;
        const double change = l_new - lp;
        const bool fin_tol = acc && change < o.dispTol;
        const bool fin_low = acc && !fin_tol && a_new < min_log_alpha;
        const bool go_on = acc && !fin_tol && !fin_low;
        iacc += acc ? 1 : 0;
        a = acc ? a_new : a;
        alpha_cur = (acc || in_init) ? alpha_new : alpha_cur;  // (start point: exp(a), kept so that finishing needs no further exp())
        lp = ((acc && !fin_low) || in_init) ? l_new : lp;
        dlp = (go_on || in_init) ? dl_new : dlp;
        init_lp = in_init ? l_new : init_lp;
        {
            double k_acc = fmin(kappa * 1.1, o.kappa0);
            if (iacc % 5 == 0) k_acc *= 0.5;
            kappa = in_init ? o.kappa0 : (go_on ? k_acc : ((in_search && !acc) ? kappa * 0.5 : kappa));
        }
        DIAG(if (in_search) { if (lg_t >= 0) { srch_spread++; if (!acc) rej_spread++; } else if (!queue_empty) { srch_bulk++; if (!acc) rej_bulk++; } })
        const bool finished = in_search && (fin_tol || fin_low || iter >= o.maxit);  // line search over: result, or the grid fallback
        iter = in_init ? 0 : iter;
        iacc = in_init ? 0 : iacc;
        phase = in_init ? (int)PH_SEARCH : phase;
        if (__ballot(finished) != 0ull) {
            if (finished) search_over(result, have_result);
        }
        if (__ballot(have_result) != 0ull) {
            if (have_result) {
                store_result(result);
                phase = PH_NEED;
            }
        }
        MARK("tick:state_machine_end");
        DIAG(if (sp_on) { cy_sp[8] += __builtin_amdgcn_s_memtime() - sp_t6; sp_on = false; })
        DIAG(if (sec_on) {
            const unsigned long long sec_t4 = __builtin_amdgcn_s_memtime();
            cy_sec[0] += sec_t1 - sec_t0; cy_sec[1] += sec_t2 - sec_t1; cy_sec[2] += sec_t3 - sec_t2; cy_sec[3] += sec_t4 - sec_t3; cy_sec[4]++;
        })
    }
    if (A.prefetch == 0x7fffffff) A.w.queue[8 + (pf_acc & 7)] = pf_acc;  // (never: keeps the warming loads alive)
    DIAG(if (A.stamps && lane == 0) {
        A.stamps[gwave * kStampSlots + 2] = __builtin_amdgcn_s_memrealtime();
        A.stamps[gwave * kStampSlots + 4] = tk_row;
        A.stamps[gwave * kStampSlots + 5] = tk_spread;
        A.stamps[gwave * kStampSlots + 6] = tk_burst;
        A.stamps[gwave * kStampSlots + 7] = tk_all;
        A.stamps[gwave * kStampSlots + 8] = cy_row;
        A.stamps[gwave * kStampSlots + 9] = cy_spread;
        A.stamps[gwave * kStampSlots + 10] = cy_burst;
        for (int q = 0; q < 5; q++) A.stamps[gwave * kStampSlots + 11 + q] = cy_sec[q];
        for (int q = 0; q < 4; q++) A.stamps[gwave * kStampSlots + 16 + q] = cy_ev[q];
        for (int q = 0; q < 9; q++) A.stamps[gwave * kStampSlots + 20 + q] = cy_sp[q];
        A.stamps[gwave * kStampSlots + 33] = tk_lean;
    })
    DIAG({  // rejected / all line-search steps, summed over the wave's lanes: in samples-across-lanes ticks, in bulk ticks
        unsigned long long v4[4] = {rej_spread, srch_spread, rej_bulk, srch_bulk};
        for (int q = 0; q < 4; q++) {
            unsigned long long x = v4[q];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (A.stamps && lane == 0) A.stamps[gwave * kStampSlots + 29 + q] = x;
        }
    })
}

// ---- fitDispGrid (A2.7 gene-wise, A4 MAP) for the rows whose line search did not converge --------------------------------------
// DESeq2's fallback: the log posterior at 20 points log(1e-8) .. log(maxDisp), then at 20 points around the first maximum, +- one
// coarse step; the first maximum of the second stage is the estimate.  The 2 x 20 evaluations of a row are independent, so a
// wave takes THREE listed rows at a time — lanes 20 k .. 20 k + 19 evaluate the points of row k, whose counts and means sit in LDS
// column k — and a row is done in two evaluations instead of 40 serial ones.  eval_point() is the function the line search uses
// (the prefix table in the lane's own column, the row's operands read from column k: a broadcast within the 20 lanes): the value at
// a point has the bits it had when the line-search kernel walked the grid itself (rounds 1-5, row per lane or as a "grid burst").
// Rows per launch: ~1.7 % of a gene-wise fit's, fewer for MAP; the launch reads their number from the device (no host stop), an
// empty list costs one launch of idle waves.
template <bool MAP>
__global__ __launch_bounds__(128) void disp_grid_kernel(DispArgs A) {
    extern __shared__ double smem[];
    __shared__ LogEntry s_logtab[64];
    __shared__ ExpEntry s_exptab[64];
    exp_table_to_lds(s_exptab);
    log_table_to_lds(s_logtab);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = A.d.S;
    double *s_tab = smem + (size_t)wave * (kTabSlots * 64 + S * 96 + 96);
    double *s_nf = s_tab + kTabSlots * 64;
    int *s_y = reinterpret_cast<int *>(s_nf + S * 64);
    const uint64_t gmask = A.d.gmask;
    const bool p2 = A.d.p == 2;
    const Opts o = A.o;
    const double glo = log(1e-8), ghi = log(o.maxDisp), gstep = (ghi - glo) / 19.0;
    const double prior_isig = MAP ? 1.0 / A.w.sc->dispPriorVar : 0.0;
    const unsigned int count = *A.gridcount;
    const int64_t rstride = row_stride(S);
    const unsigned int nwaves = gridDim.x * (blockDim.x >> 6), mywave = blockIdx.x * (blockDim.x >> 6) + (unsigned int)wave;
    const int k = lane / 20 < 3 ? lane / 20 : 2, pt = lane - 20 * (lane / 20);  // row of the triple, point (lanes 60-63: idle copies of row 2's first points, never looked at)
    for (unsigned int g0 = 3u * mywave; g0 < count; g0 += 3u * nwaves) {
        const int nrow = count - g0 < 3u ? (int)(count - g0) : 3;
        __builtin_amdgcn_wave_barrier();
        int r_mine = 0;
        if (lane < nrow) {
            r_mine = A.gridlist[g0 + (unsigned int)lane];
            load_row_mu(A.w.rowpack + (int64_t)r_mine * rstride, S, s_nf, s_y, lane, gmask, o.minmu);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool live = lane < 60 && k < nrow;
        const int r_k = __shfl(r_mine, k);
        double prior_mean = 0.0;
        if (MAP) prior_mean = reinterpret_cast<const double2 *>(A.w.start)[2 * (int64_t)(live ? r_k : __shfl(r_mine, 0))].y;
        double ghat = 0.0;
        int gbi = 0;
        for (int stage = 0; stage < 2; stage++) {
            const double a_eval = stage == 0 ? ((pt == 19) ? ghi : glo + pt * gstep)
                                             : ((pt == 19) ? ghat + gstep : (ghat - gstep) + pt * (2.0 * gstep / 19.0));
            double l_new = 0, dl_new = 0, alpha_new = 0;
            DIAG(unsigned long long tm[5];)
            eval_point(s_nf, s_y, s_tab, lane, live ? k : 0, S, gmask, p2, a_eval, MAP, prior_mean, prior_isig, l_new, dl_new, alpha_new, s_logtab, s_exptab DIAG(, tm));
            // first maximum over the row's 20 points, in point order, NaN never beats anything: what `if (l_new > gbest)` did one by one
            __builtin_amdgcn_wave_barrier();
            s_tab[lane] = l_new;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double gbest = -INFINITY;
            gbi = 0;
            for (int t = 0; t < 20; t++) {
                const double v = s_tab[20 * k + t];
                if (v > gbest) {
                    gbest = v;
                    gbi = t;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (stage == 0) ghat = (gbi == 19) ? ghi : glo + gbi * gstep;
        }
        if (live && pt == 0) {
            const double fa = (gbi == 19) ? ghat + gstep : (ghat - gstep) + gbi * (2.0 * gstep / 19.0);
            const double dd = fmin(fmax(exp(fa), o.minDisp), o.maxDisp);
            if (!MAP) {
                A.w.dispGene[r_k] = dd;
            } else {
                const double2 st1 = reinterpret_cast<const double2 *>(A.w.start)[2 * (int64_t)r_k + 1];  // gene-wise estimate, outlier flag
                A.w.dispMAP[r_k] = dd;
                A.w.disp[r_k] = st1.y != 0.0 ? st1.x : dd;
            }
        }
    }
}

// w.cls[] -> w.order[], sc->ord_na (entries of classes < classesA), sc->ord_n; w.hist is idle between the selects
// have_hist: the class counts per tile are in w.hist already (left there by the kernel that wrote w.cls)
void launch_order_build(FitDims d, FitWork w, int classesA, bool have_hist, hipStream_t st) {
    int64_t nblk, tile;
    order_tiles(d.n, nblk, tile);
    unsigned int *hist = reinterpret_cast<unsigned int *>(w.hist);
    if (!have_hist) order_hist_kernel<<<(unsigned)nblk, 256, 0, st>>>(w.cls, d.n, tile, hist);
    order_scatter_kernel<<<(unsigned)nblk, 256, 0, st>>>(w.cls, d.n, tile, hist, w.order, w.sc, classesA);
}

static void launch_disp(bool map, const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o,
                        hipStream_t st) {
    // (no schedule for a fit every row of which starts at once — n <= 65 536 is half the lanes of two waves per SIMD —: the order then
    // decides nothing, and building it is a launch and a pass: 30 k x 4 gene-wise stage 0.309 -> 0.301 ms, round 6)
    const bool sched = !map && o.schedule && (d.n > 65536 || o.schedule == 2);
    if (map) disp_init_kernel<true><<<kRedBlocks, 256, 0, st>>>(d, w, o, 0, nullptr, 0);
    else if (!sched) disp_init_kernel<false><<<kRedBlocks, 256, 0, st>>>(d, w, o, 0, nullptr, o.xim_here);
    else {
        int64_t nblk, tile;
        order_tiles(d.n, nblk, tile);
        disp_init_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(d, w, o, tile, reinterpret_cast<unsigned int *>(w.hist), o.xim_here);
    }
    const size_t lds_per_wave = disp_lds_per_wave(d.S);
    // 128-thread blocks while two waves' rows fit comfortably in LDS, else 64-thread blocks
    int threads = 128;
    while (threads > 64 && lds_per_wave * (threads / 64) > 40 * 1024) threads >>= 1;
    const size_t lds = lds_per_wave * (threads / 64);
    // persistent grid: enough waves to fill 256 CUs; rows are pulled from the queue
    int64_t waves_needed = (d.n + 63) / 64;
    int64_t blocks = (waves_needed + threads / 64 - 1) / (threads / 64);
    // never more waves than are resident at once — by LDS, and by registers (the kernel is built for `min_waves` per SIMD): a
    // wave that starts only when another has left finds the queue empty at best, and at worst owns dealt-out rows that then
    // start late (S = 4: LDS would allow 10 waves per CU, the registers allow 8; 200 k x 4 went from 0.57 to 0.72 ms)
    const int waves_per_block = threads / 64;
    int64_t blocks_per_cu = (int64_t)(160 * 1024 / (lds + 2048));  // (+ the workgroup's log and exp tables)
    // Waves per SIMD (option "line_search_min_waves"; 0 = this rule).  Round 6: the kernel fits 168 registers without scratch (fitDispGrid
    // and its state left for disp_grid_kernel, the MAP search's outlier value for disp_init) and its LDS lets twelve waves into a CU at
    // S <= 8 (the harmonic half of the prefix table became a row-level sum).  Measured on one box (profiles/r06_ab_three_waves.txt):
    // the BULK of a launch gains ~10 % (queue empty at 1.04 instead of 1.17 ms, gene-wise, 2 M x 8), and the MAP launch keeps it
    // (1.09 -> 0.97 ms at 2 M x 8, 0.60 -> 0.53 at 1 M); the gene-wise launch does not: its end is the ~105 dependent evaluations of the
    // flat-likelihood rows, each at the pace of a wave that now shares its SIMD with two others (bulk tick 10.7 instead of 8.0 us,
    // drain 160 / 430 us median / longest instead of 130 / 250) — 1.34 -> 1.36 ms, and worse for fits of <= 500 k rows, which are
    // all drain (250 k x 8: 0.50 -> 0.63).  So: three waves for the MAP search of >= 750 k rows; for the gene-wise search only where the
    // launch is long against that chain (105 ticks of ~2 + 1.1 S us): from 3 M rows (3 M x 8 1.85 -> 1.82 ms, 4 M 2.39 -> 2.25, 6 M 3.45 ->
    // 3.20), at S <= 4 from 1.5 M (2 M x 4 1.16 -> 1.13, 4 M x 4 1.97 -> 1.91); two otherwise.
    const bool lds3 = (int64_t)(160 * 1024 / (lds + 2048)) * waves_per_block >= 12;
    const bool three = lds3 && (map ? d.n >= 750000 : (d.n >= 3000000 || (d.S <= 4 && d.n >= 1500000)));
    const int min_waves = (o.min_waves >= 2 && o.min_waves <= 4) ? o.min_waves : (three ? 3 : 2);
    const int64_t by_regs = (int64_t)(4 * min_waves) / waves_per_block;
    if (blocks_per_cu > by_regs) blocks_per_cu = by_regs;
    if (blocks_per_cu > 8) blocks_per_cu = 8;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    const int64_t max_blocks = 256 * blocks_per_cu;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    // the gene-wise launch visits the rows likely-long first; schedule 2 (a fit that shares the GPU with other fits: the theta
    // grid's lanes) keeps the class order but deals nothing out statically — its waves are not all resident at once, and a wave
    // that starts late must not be the owner of likely-long rows
    // classes dealt out statically: 0-1 (the likely-long 8 % of the rows) — or, when the launch has 1.8 .. 8 rows per lane, all but
    // the last (everything except the minDisp starts and the highest scores), in groups of four.  The queue hands a class out in
    // chunks of 64 consecutive schedule entries: a fifth of the waves each received a whole chunk of class-2 rows (20-50 steps)
    // and, with so few rows per lane, nothing afterwards to even that out — they were the waves that left last.  Measured, two or
    // three repeats each (profiles/r05_ab_static_deal_classes.txt): gene-wise 0.525 -> 0.495 ms at 250 k x 8 (1.9 rows per lane),
    // 0.702 -> 0.658 at 500 k x 8, 0.934 -> 0.915 at 1 M x 8 (7.6), 0.562 -> 0.531 at 500 k x 4, 0.733 -> 0.716 at 1 M x 4,
    // 0.666 -> 0.644 at 250 k x 16 (2.5); NOT at 2 M x 8 (15 rows per lane: the queue balances better than any deal: 1.356 -> 1.44),
    // 1 M x 16 (10: + 1.5 %), 200 k x 4 (1.5: + 1.5 %), 150 k x 8 (1.1: + 2 %).  Groups of 8 (the automatic size for so many entries)
    // gave nothing: 2 or 4.
    const double rows_per_lane = (double)d.n / (double)(blocks * threads);
    const bool deal_most = !map && o.schedule == 1 && o.classes_a == 0 && rows_per_lane >= 1.8 && rows_per_lane <= 8.0;
    const int classes_a = o.schedule == 2 ? 0 : (o.classes_a > 0 ? o.classes_a : (deal_most ? 5 : kSchedClassesA));
    if (sched) launch_order_build(d, w, classes_a, true, st);
    DispArgs A{counts, nf, d, w, o, nullptr, o.spread, sched ? w.order : nullptr, (deal_most && o.deal == 0) ? 4 : o.deal, 1, kChunk,
               w.gridlist, reinterpret_cast<unsigned int *>(w.queue + (map ? 24 : 16)), o.prio};
#ifdef CHICDIFF_DIAG
    const char *stamp_file = getenv("CHICDIFF_DISP_STAMPS");  // diagnostic build only: blocking, never timed
    const size_t stamp_words = (size_t)blocks * (threads / 64) * kStampSlots;
    if (stamp_file) {
        (void)hipMalloc((void **)&A.stamps, stamp_words * 8);
        (void)hipMemsetAsync(A.stamps, 0, stamp_words * 8, st);
    }
#endif
    // rows per dequeue: 64 — unless a wave's share of the rows is about ONE such chunk (the reference's own data set: 30 k rows on
    // 470 waves): the queue then balances nothing, and which wave gets its chunk last decides which leaves last.  Then 16 at a time:
    // gene-wise 0.313 -> 0.300 ms, MAP 0.084 -> 0.076 at 30 k x 4.  Not beyond: at 200 k rows 16-row chunks cost 0.424 -> 0.525 ms (the
    // one queue head saturates near 90 dequeues per us, and every dequeue is a refill on cold lines): profiles/r05_ab_queue_chunk.txt
    // (option "line_search_chunk" overrides; the rule is on the row count, not on rows per wave: the 3-waves-per-SIMD MAP build at
    // 200 k x 4 has 65 rows per wave and lost 0.157 -> 0.204 ms to 16-row chunks)
    A.chunk = o.chunk > 0 ? o.chunk : (d.n <= 65536 ? 16 : kChunk);
    const int variant = min_waves;
#define LAUNCH(M, W) disp_fit_kernel<M, W><<<(unsigned)blocks, threads, lds, st>>>(A)
    if (map) {
        if (variant >= 4) LAUNCH(true, 4); else if (variant == 3) LAUNCH(true, 3); else LAUNCH(true, 2);
    } else {
        if (variant >= 4) LAUNCH(false, 4); else if (variant == 3) LAUNCH(false, 3); else LAUNCH(false, 2);
    }
#undef LAUNCH
    {   // fitDispGrid for the rows the searches have listed; the list's length stays on the device.  Waves: what ~2 % of the rows need at
        // three rows per wave and pass, at most one resident round
        int64_t gblocks = (d.n / 50 / 3 + 1 + 1) / 2;
        if (gblocks > 1536) gblocks = 1536;  // (six workgroups per CU — 150 VGPRs, 26 KB of LDS each: one resident round; 1024 / 1536 / 2048: 75 / 71 / 77 us at 2 M x 8)
        if (gblocks < 1) gblocks = 1;
        if (map) disp_grid_kernel<true><<<(unsigned)gblocks, 128, 2 * lds_per_wave, st>>>(A);
        else disp_grid_kernel<false><<<(unsigned)gblocks, 128, 2 * lds_per_wave, st>>>(A);
    }
#ifdef CHICDIFF_DIAG
    if (stamp_file) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(stamp_words);
        (void)hipMemcpy(h.data(), A.stamps, h.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(stamp_file, map ? "ab" : "wb");
        if (f) {
            unsigned long long hdr[2] = {map ? 1ull : 0ull, h.size() / kStampSlots};
            fwrite(hdr, 8, 2, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
        (void)hipFree(A.stamps);
    }
#endif
}

void launch_disp_gene(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    launch_disp(false, counts, nf, d, w, o, st);
}
void launch_disp_map(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    launch_disp(true, counts, nf, d, w, o, st);
}

}  // namespace cd
