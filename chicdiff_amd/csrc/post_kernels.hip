// post_kernels.hip — the steps either side of the test itself:
//   f1/f3  p.adjust(method = "BH") on device (DESeq2 results() and chicdiff.R:2049), and the application side of
//          IHWcorrection (chicdiff.R:2038-2049): distance -> group cut, weight lookup, renormalisation,
//          weighted p-values, BH;
//   f4     getRegionUniverse, window mode (chicdiff.R:353-426): .expandAvoidBait ranges, clipped to the
//          restriction map and to the bait's chromosome, as a CSR over regions.
// HBM-bound integer/byte work plus one device sort; rocPRIM provides the sort and the scans (plain library
// primitives), the rest is hand-written.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace cd {

// ---- Benjamini-Hochberg ----------------------------------------------------------------------------------
// R: n <- sum(!is.na(p)); o <- order(p, decreasing = TRUE); pmin(1, cummin(n / (n:1) * p[o]))[order(o)].
// Here: ascending sort of (key(p), index) with NA last, q_k = n/k * p_(k), suffix minimum, scatter.
__global__ __launch_bounds__(256) void bh_keys_kernel(const double *__restrict__ p, int64_t n, uint64_t *keys,
                                                      uint32_t *idx, unsigned long long *count) {
    __shared__ unsigned int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = p[i];
        const bool na = x != x;
        keys[i] = na ? ~0ull : key_of(x);
        idx[i] = (uint32_t)i;
        mine += na ? 0u : 1u;
    }
    atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(count, (unsigned long long)s_cnt);
}
__global__ __launch_bounds__(256) void bh_q_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                   const unsigned long long *count, double *q) {
    const double m = (double)*count;
    for (int64_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        q[k] = (unsigned long long)k < *count ? m / (double)(k + 1) * value_of(keys[k]) : INFINITY;
}
__global__ __launch_bounds__(256) void bh_scatter_kernel(const double *__restrict__ smin, const uint32_t *__restrict__ idx,
                                                         int64_t n, const unsigned long long *count, double *padj) {
    for (int64_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        padj[idx[k]] = (unsigned long long)k < *count ? fmin(1.0, smin[k]) : NAN;
}

size_t bh_workspace_bytes(int64_t n) {
    size_t sort_tmp = 0, scan_tmp = 0;
    uint64_t *k = nullptr;
    uint32_t *v = nullptr;
    double *d = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_tmp, rocprim::make_reverse_iterator(d), rocprim::make_reverse_iterator(d),
                                  (size_t)n, rocprim::minimum<double>(), (hipStream_t)0);
    const size_t tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    return (256 + ((size_t)n * (8 + 8 + 4 + 4 + 8 + 8) + 6 * 256) + tmp + 511) / 256 * 256;  // a multiple of 256: callers append to it
}

// `ws` holds bh_workspace_bytes(n) bytes; everything is enqueued on `st`
int launch_bh_adjust(const double *p, int64_t n, double *padj, char *ws, hipStream_t st) {
    auto take = [&](size_t bytes) { char *r = ws; ws += (bytes + 255) / 256 * 256; return r; };
    unsigned long long *count = (unsigned long long *)take(256);
    uint64_t *k0 = (uint64_t *)take(8 * (size_t)n), *k1 = (uint64_t *)take(8 * (size_t)n);
    uint32_t *i0 = (uint32_t *)take(4 * (size_t)n), *i1 = (uint32_t *)take(4 * (size_t)n);
    double *q = (double *)take(8 * (size_t)n), *s = (double *)take(8 * (size_t)n);
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, k0, k1, i0, i1, (size_t)n, 0, 64, st);
    (void)rocprim::inclusive_scan(nullptr, scan_tmp, rocprim::make_reverse_iterator(q + n), rocprim::make_reverse_iterator(s + n),
                                  (size_t)n, rocprim::minimum<double>(), st);
    void *tmp = ws;
    if (hipMemsetAsync(count, 0, 8, st) != hipSuccess) return 1;
    int blocks = (int)((n + 2047) / 2048);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    bh_keys_kernel<<<blocks, 256, 0, st>>>(p, n, k0, i0, count);
    if (rocprim::radix_sort_pairs(tmp, sort_tmp, k0, k1, i0, i1, (size_t)n, 0, 64, st) != hipSuccess) return 1;  // stable
    bh_q_kernel<<<blocks, 256, 0, st>>>(k1, n, count, q);
    if (rocprim::inclusive_scan(tmp, scan_tmp, rocprim::make_reverse_iterator(q + n), rocprim::make_reverse_iterator(s + n),
                                (size_t)n, rocprim::minimum<double>(), st) != hipSuccess)
        return 1;
    bh_scatter_kernel<<<blocks, 256, 0, st>>>(s, i1, n, count, padj);
    return 0;
}

// ---- IHW application (chicdiff.R:2038-2046) ---------------------------------------------------------------
// group <- as.integer(cut(log(abs(avDist)), breaks))   : (b_k, b_k+1], NA outside or for NA
// avWeights <- distLookup$avWeights[group]             : NA for NA group
// partial sums of avWeights for mean(out$avWeights) (no na.rm: one NA makes every weight NA)
constexpr int kIhwMaxGroups = 256;
struct IhwTables {
    double breaks[kIhwMaxGroups + 1];
    double w[kIhwMaxGroups];
    int ng;
};
__global__ __launch_bounds__(256) void ihw_group_kernel(const double *__restrict__ avDist, int64_t n, IhwTables t,
                                                        int32_t *group, double *avw, double *partials) {
    __shared__ double red[256];
    double s = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = log(fabs(avDist[i]));
        int g = -1;
        if (x == x && x > t.breaks[0] && x <= t.breaks[t.ng]) {
            int lo = 0, hi = t.ng;  // largest k with breaks[k] < x
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (t.breaks[mid] < x) lo = mid; else hi = mid;
            }
            g = lo;
        }
        const double wv = g >= 0 ? t.w[g] : NAN;
        if (group) group[i] = g >= 0 ? g + 1 : INT32_MIN;  // R's NA_integer_
        avw[i] = wv;
        s += wv;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void ihw_mean_kernel(double *partials, int nblocks, int64_t n) {
    __shared__ double red[256];
    double s = 0;
    for (int k = threadIdx.x; k < nblocks; k += 256) s += partials[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[nblocks] = red[0] / (double)n;
}
__global__ __launch_bounds__(256) void ihw_weight_kernel(const double *__restrict__ pvalue, int64_t n, const double *mean,
                                                         double *avw_to_weight, double *wp) {
    const double m = *mean;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double w = avw_to_weight[i] / m;
        avw_to_weight[i] = w;
        wp[i] = pvalue[i] / w;
    }
}
constexpr int kIhwBlocks = 1024;
size_t ihw_workspace_bytes() { return sizeof(double) * (kIhwBlocks + 1 + 31); }
void launch_ihw_apply(const double *avDist, const double *pvalue, int64_t n, const double *breaks, const double *weights,
                      int ng, int32_t *group, double *weight, double *wp, double *partials, hipStream_t st) {
    IhwTables t;
    for (int k = 0; k <= ng; k++) t.breaks[k] = breaks[k];
    for (int k = 0; k < ng; k++) t.w[k] = weights[k];
    t.ng = ng;
    ihw_group_kernel<<<kIhwBlocks, 256, 0, st>>>(avDist, n, t, group, weight, partials);
    ihw_mean_kernel<<<1, 256, 0, st>>>(partials, kIhwBlocks, n);
    ihw_weight_kernel<<<kIhwBlocks, 256, 0, st>>>(pvalue, n, partials + kIhwBlocks, weight, wp);
}

// ---- f2: chinput columns -> sorted key table of the count join (chicdiff.R:826-831, :849) -----------------------
__global__ __launch_bounds__(256) void ct_keys_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t n,
                                                      const uint8_t *__restrict__ keep, int32_t max_id, uint64_t *keys,
                                                      unsigned long long *count) {
    __shared__ unsigned int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t b = bait[i];
        const bool k = b >= 0 && (!keep || (b <= max_id && keep[b]));
        keys[i] = k ? (((uint64_t)(uint32_t)b << 32) | (uint32_t)oe[i]) : ~0ull;  // dropped rows sort to the end
        mine += k ? 1u : 0u;
    }
    atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(count, (unsigned long long)s_cnt);
}
size_t ct_workspace_bytes(int64_t n) {
    size_t tmp = 0;
    uint64_t *k = nullptr;
    int32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
    return (256 + ((size_t)n * 8 + 255) / 256 * 256 + tmp + 511) / 256 * 256;
}
int launch_count_table(const int32_t *bait, const int32_t *oe, const int32_t *N, int64_t n, const uint8_t *keep, int32_t max_id,
                       int64_t *keys_out, int32_t *vals_out, char *ws, hipStream_t st) {
    unsigned long long *count = (unsigned long long *)ws;
    uint64_t *k0 = (uint64_t *)(ws + 256);
    void *tmp = ws + 256 + ((size_t)n * 8 + 255) / 256 * 256;
    size_t tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, (uint64_t *)keys_out, N, vals_out, (size_t)n, 0, 64, st);
    if (hipMemsetAsync(count, 0, 8, st) != hipSuccess) return 1;
    int blocks = (int)((n + 2047) / 2048);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    ct_keys_kernel<<<blocks, 256, 0, st>>>(bait, oe, n, keep, max_id, k0, count);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, (uint64_t *)keys_out, N, vals_out, (size_t)n, 0, 64, st) == hipSuccess ? 0 : 1;
}

// ---- region universe (chicdiff.R:353-426) ----------------------------------------------------------------
// .expandAvoidBait(bait, oe, s): (oe-s):(oe+s) unless the bait is within s+1 fragments, then the range stops two
// fragments short of the bait.  R's a:b counts down when a > b, so the set is [min(a,b), max(a,b)].
__device__ __forceinline__ bool expand_range(int bait, int oe, int s, int &lo, int &hi) {
    const int dist = bait > oe ? bait - oe : oe - bait;
    int a, b;
    if (dist > s + 1) { a = oe - s; b = oe + s; }
    else if (oe > bait) { a = bait + 2; b = oe + s; }
    else if (oe < bait) { a = oe - s; b = bait - 2; }
    else return false;  // stop("Invalid parameters ...")
    lo = a < b ? a : b;
    hi = a < b ? b : a;
    return true;
}
// kept: 1 <= id <= maxfrag (the rmap join drops ids the map does not hold; `otherEndID <= maxfrag`, :384),
// same chromosome as the bait (:399), bait itself on the map
__device__ __forceinline__ bool ru_keep(int id, int bait_chr, const int32_t *chr_of, int maxfrag) {
    return id >= 1 && id <= maxfrag && chr_of[id] >= 0 && chr_of[id] == bait_chr;
}
__global__ __launch_bounds__(256) void ru_count_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t n,
                                                       int s, const int32_t *__restrict__ chr_of, int maxfrag, int64_t *len,
                                                       int32_t *minOE, int32_t *maxOE, int *bad, unsigned int *mask_out) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int lo, hi, cnt = 0, mn = INT32_MAX, mx = INT32_MIN;
        unsigned int mask = 0;  // bit (id - lo): candidate id is kept (windows of <= 32 candidates; what the fill kernel reads instead of walking)
        const int b = bait[i];
        if (!expand_range(b, oe[i], s, lo, hi)) {
            atomicExch(bad, 1);
        } else {
            const int bc = (b >= 1 && b <= maxfrag) ? chr_of[b] : -1;
            // (four candidates' chromosome look-ups in flight at a time: the window is a run of consecutive IDs, the look-ups are independent,
            // and as `ru_keep(id, ...)` inside the loop's condition each was a load the next iteration waited for)
            for (int id0 = lo; id0 <= hi; id0 += 4) {
                int cc[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int id = id0 + q;
                    cc[q] = (id <= hi && id >= 1 && id <= maxfrag) ? chr_of[id] : -1;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int id = id0 + q;
                    if (bc >= 0 && cc[q] >= 0 && cc[q] == bc) {  // ru_keep()
                        cnt++;
                        mn = id < mn ? id : mn;
                        mx = id > mx ? id : mx;
                        if (id - lo < 32) mask |= 1u << (id - lo);
                    }
                }
            }
        }
        if (mask_out) mask_out[i] = mask;
        len[i] = cnt;
        if (minOE) minOE[i] = cnt ? mn : INT32_MIN;
        if (maxOE) maxOE[i] = cnt ? mx : INT32_MIN;
    }
}
// (Measured, round 2: one thread per region writing its ~11 consecutive rows — a sixth of the look-ups, but stores strided by
// 44 bytes across the wave: 0.61 ms against 0.25 ms at 2 M peaks.  Coalesced stores win.)
// rows are written by consecutive threads (coalesced): a block owns 256 regions, finds each of its rows' region by
// binary search over the block's CSR offsets in LDS, and walks to the row's candidate (at most 2s+1 steps)
__global__ __launch_bounds__(256) void ru_fill_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t n,
                                                      int s, const int32_t *__restrict__ chr_of, int maxfrag,
                                                      const int64_t *__restrict__ ptr, int32_t *ru_bait, int32_t *ru_region,
                                                      int32_t *ru_oe, const unsigned int *__restrict__ mask_in) {
    // (round 6: which of a region's <= 2 s + 1 candidates are kept is worked out ONCE per region, by the region's thread — a bit mask
    // in LDS —, and a row picks the k-th set bit; before, every ROW walked its region's candidates, a dependent gather from chr_of per
    // step: ~6 gathers per output row on average, the kernel at a tenth of the HBM roof.  Windows wider than 32 candidates
    // (RUexpand > 15) keep the walk.)
    __shared__ int64_t s_ptr[257];
    __shared__ int s_lo[256], s_hi[256], s_bait[256], s_chr[256];
    __shared__ unsigned int s_mask[256];
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int nreg = n - i0 < 256 ? (int)(n - i0) : 256;
        __syncthreads();
        if ((int)threadIdx.x <= nreg) s_ptr[threadIdx.x] = ptr[i0 + threadIdx.x];
        if (threadIdx.x == 0) s_ptr[nreg] = ptr[i0 + nreg];
        if ((int)threadIdx.x < nreg) {
            int lo = 0, hi = -1;
            const int b = bait[i0 + threadIdx.x];
            (void)expand_range(b, oe[i0 + threadIdx.x], s, lo, hi);
            const int bc = (b >= 1 && b <= maxfrag) ? chr_of[b] : -1;
            unsigned int mask = 0;
            if (mask_in) mask = mask_in[i0 + threadIdx.x];  // (left by ru_count_kernel in the one-call entry point: no gathers here at all)
            else if (hi - lo < 32)
                for (int id = lo; id <= hi; id++)
                    if (ru_keep(id, bc, chr_of, maxfrag)) mask |= 1u << (id - lo);
            s_lo[threadIdx.x] = lo;
            s_hi[threadIdx.x] = hi;
            s_bait[threadIdx.x] = b;
            s_chr[threadIdx.x] = bc;
            s_mask[threadIdx.x] = mask;
        }
        __syncthreads();
        const int64_t r0 = s_ptr[0], r1 = s_ptr[nreg];
        for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
            int a = 0, e = nreg;  // last region with s_ptr[a] <= r
            while (e - a > 1) {
                const int mid = (a + e) >> 1;
                if (s_ptr[mid] <= r) a = mid; else e = mid;
            }
            int k = (int)(r - s_ptr[a]);  // the k-th kept candidate of region a
            int id = s_lo[a];
            if (s_hi[a] - id < 32) {
                unsigned int m = s_mask[a];
                for (int q = 0; q < k; q++) m &= m - 1u;  // drop the k lowest set bits
                id += __ffs((int)m) - 1;
            } else {
                const int bc = s_chr[a];
                for (; id <= s_hi[a]; id++)
                    if (ru_keep(id, bc, chr_of, maxfrag) && k-- == 0) break;
            }
            ru_bait[r] = s_bait[a];
            ru_region[r] = (int32_t)(i0 + a + 1);  // regionID <- 1:nrow
            ru_oe[r] = id;
        }
    }
}
size_t ru_scan_bytes(int64_t n) {
    size_t tmp = 0;
    int64_t *d = nullptr;
    (void)rocprim::exclusive_scan(nullptr, tmp, d, d, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), (hipStream_t)0);
    return tmp + 256;
}
// region_ptr (n+1): first used as the lengths (entry n = 0), then scanned in place
int launch_ru_count(const int32_t *bait, const int32_t *oe, int64_t n, int s, const int32_t *chr_of, int maxfrag,
                    int64_t *region_ptr, int32_t *minOE, int32_t *maxOE, int *bad, void *tmp, size_t tmp_bytes, hipStream_t st, unsigned int *mask_out) {
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess) return 1;
    if (hipMemsetAsync(region_ptr + n, 0, sizeof(int64_t), st) != hipSuccess) return 1;
    ru_count_kernel<<<blocks, 256, 0, st>>>(bait, oe, n, s, chr_of, maxfrag, region_ptr, minOE, maxOE, bad, mask_out);
    size_t need = 0;
    (void)rocprim::exclusive_scan(nullptr, need, region_ptr, region_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st);
    if (need > tmp_bytes) return 2;
    return rocprim::exclusive_scan(tmp, need, region_ptr, region_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st) == hipSuccess ? 0 : 1;
}
void launch_ru_fill(const int32_t *bait, const int32_t *oe, int64_t n, int s, const int32_t *chr_of, int maxfrag,
                    const int64_t *region_ptr, int32_t *ru_bait, int32_t *ru_region, int32_t *ru_oe, hipStream_t st, const unsigned int *mask_in) {
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    ru_fill_kernel<<<blocks, 256, 0, st>>>(bait, oe, n, s, chr_of, maxfrag, region_ptr, ru_bait, ru_region, ru_oe, mask_in);
}


// ---- IHWcorrection's covariate: avDist = mean(distSign) by regionID (chicdiff.R:1965-1967, :1980-1982) ----------------
// The reference takes it from the long "recast" table (one row per region, fragment and sample — 176 M rows at 2 M
// regions x 8 samples, 3.5 G at 20 M x 16): RU.recast[, list(avDist = mean(distSign)), by = "regionID"].  Every
// (region, fragment) pair appears once per sample there with the same distSign, so the mean over the table is the mean
// over the region's fragments.  distSign is the one of CountOut (chicdiff.R:868-882): midpoint := round(0.5 * (start +
// end)) per fragment (R's round(): half to even), distSign := midpoint[otherEndID] - midpoint[baitID], NA when the two
// fragments lie on different chromosomes.  Sums of integers below 2^53 are exact, so one division gives R's mean().
// One thread per region over the CSR of (regionID, otherEndID)-ordered RU rows; the midpoint table is L2-resident.
__global__ __launch_bounds__(256) void region_avdist_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe,
                                                            const int64_t *__restrict__ ptr, int64_t n, int32_t id_min, int32_t nid,
                                                            const int64_t *__restrict__ midsum, const int32_t *__restrict__ chr,
                                                            double *__restrict__ avDist) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t lo = ptr[i], hi = ptr[i + 1];
        double sum = 0.0;
        int64_t cnt = 0;
        bool na = false;
        for (int64_t r = lo; r < hi; r++) {
            const int32_t b = bait[r] - id_min, o = oe[r] - id_min;
            if (b < 0 || b >= nid || o < 0 || o >= nid || (chr && (chr[b] < 0 || chr[o] < 0)))
                continue;  // a fragment the map does not hold: merge(x, rmap) drops the row (chicdiff.R:873-874)
            cnt++;
            if (chr && chr[b] != chr[o]) na = true;  // ifelse(chr.x == chr.y, ., NA), :877-880; mean() has no na.rm
            sum += rint(0.5 * (double)midsum[o]) - rint(0.5 * (double)midsum[b]);
        }
        avDist[i] = (na || cnt == 0) ? NAN : sum / (double)cnt;
    }
}
void launch_region_avdist(const int32_t *bait, const int32_t *oe, const int64_t *ptr, int64_t n, int32_t id_min, int32_t nid,
                          const int64_t *midsum, const int32_t *chr, double *avDist, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    region_avdist_kernel<<<blocks, 256, 0, st>>>(bait, oe, ptr, n, id_min, nid, midsum, chr, avDist);
}

// ---- a1, no-chinput branch (chicdiff.R:774-807, = :1202-1260 in getFullRegionData2) ------------------------------------
// Without chinput files N comes from the Chicago objects: tempForCounts[[i]] = x[, c(baitID, otherEndID, N)] keyed by
// (baitID, otherEndID); mergedFiles <- Reduce(merge, tempForCounts) is an INNER join over the replicates (merge()'s
// default), so a pair survives only when every replicate's object holds it; then per replicate merge(RU, ., all.x = TRUE)
// and N[is.na(N)] <- 0.  Net effect per RU row: its N in every replicate when all S tables hold the pair, 0 in every
// replicate otherwise.  One thread per RU row, S lower-bound searches in global memory: this is the fallback path (the
// chinput path is the tuned count_join_kernel), 22 M rows x 8 tables x ~24 probes is a few milliseconds.
struct JoinInnerArgs {
    const int64_t *keys[64];
    const int32_t *vals[64];
    int64_t nkeys[64];
};
__global__ __launch_bounds__(256) void count_join_inner_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe,
                                                               int64_t nru, int S, JoinInnerArgs a, int32_t *__restrict__ out) {
    for (int64_t r = blockIdx.x * 256 + threadIdx.x; r < nru; r += (int64_t)gridDim.x * 256) {
        const int64_t key = ((int64_t)bait[r] << 32) | (uint32_t)oe[r];
        bool all = true;
        for (int s = 0; s < S; s++) {
            const int64_t *k = a.keys[s];
            int64_t lo = 0, hi = a.nkeys[s];
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (k[mid] < key) lo = mid + 1; else hi = mid;
            }
            const bool hit = lo < a.nkeys[s] && k[lo] == key;
            out[(int64_t)s * nru + r] = hit ? a.vals[s][lo] : 0;
            all = all && hit;
        }
        if (!all)
            for (int s = 0; s < S; s++) out[(int64_t)s * nru + r] = 0;
    }
}
void launch_count_join_inner(const int32_t *bait, const int32_t *oe, int64_t nru, int S, const int64_t *const *keys,
                             const int32_t *const *vals, const int64_t *nkeys, int32_t *out, hipStream_t st) {
    JoinInnerArgs a;
    for (int s = 0; s < 64; s++) {
        a.keys[s] = s < S ? keys[s] : nullptr;
        a.vals[s] = s < S ? vals[s] : nullptr;
        a.nkeys[s] = s < S ? nkeys[s] : 0;
    }
    int blocks = (int)((nru + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    count_join_inner_kernel<<<blocks, 256, 0, st>>>(bait, oe, nru, S, a, out);
}

}  // namespace cd

// ================================================================================================================
// a9 — DESeq2 results(): Cook's cutoff and independent filtering on device (chicdiff.R:1720-1741 call results()
// with its defaults).  Host numpy needs ~5 s for 2 M rows (50 filtered BH passes); here the
// 50 rejection counts come from ONE sort by p-value and a 50-wide prefix count.
namespace cd {

// p <- NA for Cook's outliers: maxCooks > cutoff, unless (two-group design) at least 3 counts of the row exceed the
// count of the sample with the largest Cook's distance (then the outlier is a low count and the p-value stays)
__global__ __launch_bounds__(256) void cooks_filter_kernel(const int32_t *__restrict__ counts, int64_t n, int S, int p,
                                                           const double *__restrict__ maxCooks, const int32_t *__restrict__ argmax,
                                                           double cutoff, double *pvalue, unsigned long long *nout) {
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double mc = maxCooks[i];
        if (!(mc > cutoff)) continue;  // NaN: not an outlier
        bool outlier = true;
        if (p == 2) {
            const int a = argmax[i];
            if (a >= 0 && a < S) {
                const int32_t oc = counts[(int64_t)a * n + i];
                int larger = 0;
                for (int j = 0; j < S; j++) larger += counts[(int64_t)j * n + i] > oc ? 1 : 0;
                if (larger >= 3) outlier = false;
            }
        }
        if (outlier) {
            pvalue[i] = NAN;
            mine++;
        }
    }
    if (mine) atomicAdd(nout, (unsigned long long)mine);
}
void launch_cooks_filter(const int32_t *counts, int64_t n, int S, int p, const double *maxCooks, const int32_t *argmax,
                         double cutoff, double *pvalue, unsigned long long *nout, hipStream_t st) {
    (void)hipMemsetAsync(nout, 0, 8, st);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    cooks_filter_kernel<<<blocks, 256, 0, st>>>(counts, n, S, p, maxCooks, argmax, cutoff, pvalue, nout);
}

constexpr int kIfN = 50;      // quantile cutoffs (DESeq2: theta <- seq(lower, upper, length = 50))
constexpr int kIfBlock = 1024;

// Independent filtering (DESeq2 pvalueAdjustment): what the kernels hand to each other.  One sort of baseMean (the
// 50 quantile cutoffs), one sort of the p-values; the 50 filtered BH rejection counts and the final adjustment are
// ranks inside that one p-order.  The host looks once in between, for the 50-point lowess that picks the filter.
struct IfState {
    unsigned long long nzero, count;  // rows with baseMean == 0; rows with a p-value
    unsigned int total[kIfN];         // rows passing filter f (and holding a p-value)
    unsigned int numRej[kIfN];        // BH rejections at level alpha among them
    double cut[kIfN];                 // quantile(baseMean, theta[f])
    chicdiff_results_info info;       // theta[] only (the rest is filled in on the host)
};

__global__ __launch_bounds__(256) void if_keys_kernel(const double *__restrict__ bm, int64_t n, uint64_t *keys, IfState *st) {
    unsigned int mine = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = bm[i];
        keys[i] = x != x ? ~0ull : key_of(x);
        mine += x == 0.0 ? 1u : 0u;
    }
    if (mine) atomicAdd(&st->nzero, (unsigned long long)mine);
}
// lower = mean(baseMean == 0); theta = seq(lower, upper, length = 50); cutoffs = quantile(baseMean, theta), type 7
__global__ void if_cuts_kernel(const uint64_t *__restrict__ sorted, int64_t n, double alpha, IfState *st) {
    const int k = threadIdx.x;
    if (k >= kIfN) return;
    const double lower = (double)st->nzero / (double)n, upper = lower < 0.95 ? 0.95 : 1.0;
    const double theta = lower + (double)k * ((upper - lower) / (double)(kIfN - 1));
    const double h = (double)(n - 1) * theta;
    const int64_t lo = (int64_t)floor(h), hi = lo + 1 < n ? lo + 1 : n - 1;
    const double frac = h - (double)lo, xlo = value_of(sorted[lo]), xhi = value_of(sorted[hi]);
    st->cut[k] = (frac > 0.0 && xhi != xlo) ? (1.0 - frac) * xlo + frac * xhi : xlo;  // quantile.default(type = 7): qs[i] <- (1 - h) * qs[i] + h * x[hi[i]]
    st->info.theta[k] = k == kIfN - 1 ? upper : theta;
}
// T = number of cutoffs <= baseMean (cutoffs ascending): the row passes the filters 0 .. T-1
__device__ __forceinline__ int if_T(double bm, const double *cut) {
    int lo = 0, hi = kIfN;  // first cutoff > bm
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cut[mid] <= bm) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// pass 1 over the p-sorted rows: per block and filter, how many rows pass (and have a p-value)
__global__ __launch_bounds__(kIfBlock) void if_count_kernel(const uint32_t *__restrict__ idx, const double *__restrict__ bm, int64_t n,
                                                             const IfState *st, uint8_t *T, uint32_t *blockcnt, int nblocks) {
    __shared__ unsigned int h[kIfN + 1];
    __shared__ double s_cut[kIfN];
    if (threadIdx.x <= kIfN) h[threadIdx.x] = 0;
    if (threadIdx.x < kIfN) s_cut[threadIdx.x] = st->cut[threadIdx.x];
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * kIfBlock + threadIdx.x;
    int t = 0;
    if (k < n && (unsigned long long)k < st->count) t = if_T(bm[idx[k]], s_cut);
    if (k < n) T[k] = (uint8_t)t;
    atomicAdd(&h[t], 1u);
    __syncthreads();
    if (threadIdx.x < kIfN) {  // rows with T > f pass filter f
        unsigned int s = 0;
        for (int b = threadIdx.x + 1; b <= kIfN; b++) s += h[b];
        blockcnt[(size_t)threadIdx.x * nblocks + blockIdx.x] = s;
    }
}
// exclusive scan over the blocks, one workgroup per filter; total[f] = rows passing filter f
__global__ __launch_bounds__(1024) void if_scan_kernel(uint32_t *blockcnt, int nblocks, IfState *st) {
    __shared__ unsigned int part[1024];
    uint32_t *c = blockcnt + (size_t)blockIdx.x * nblocks;
    const int per = (nblocks + 1023) / 1024, b0 = threadIdx.x * per;
    unsigned int s = 0;
    for (int b = b0; b < b0 + per && b < nblocks; b++) s += c[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned int add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - s;  // exclusive
    for (int b = b0; b < b0 + per && b < nblocks; b++) {
        const unsigned int v = c[b];
        c[b] = run;
        run += v;
    }
    if (threadIdx.x == 1023) st->total[blockIdx.x] = part[1023];
}
// pass 2: BH at level alpha rejects the hypotheses up to the largest filtered rank r with m/r * p_(r) < alpha
// (DESeq2 counts padj < alpha): numRej[f] = that r.  m/r >= 1, so only the rows with p < alpha can qualify: the
// workgroups behind them in the p-order leave at once; the others first tabulate, per wave and filter, how many of
// their rows pass, then every wave ranks its own rows (ballot + the waves before it) and reports its last qualifying
// rank — no workgroup barrier inside the loop over the 50 filters.
__global__ __launch_bounds__(kIfBlock) void if_rej_kernel(const uint64_t *__restrict__ pkeys, const uint8_t *__restrict__ T, int64_t n,
                                                           const uint32_t *__restrict__ blockoff, int nblocks, double alpha, IfState *st) {
    __shared__ unsigned int wcnt[kIfBlock / 64][kIfN];
    __shared__ unsigned int best[kIfN];
    const int64_t k0 = (int64_t)blockIdx.x * kIfBlock;
    if (pkeys[k0] >= key_of(alpha)) return;  // also true for the NA keys (all ones) at the end
    if (threadIdx.x < kIfN) best[threadIdx.x] = 0;
    const int64_t k = k0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = k < n ? T[k] : 0;
    const double p = t > 0 ? value_of(pkeys[k]) : 0.0;
    for (int f = 0; f < kIfN; f++) {
        const unsigned long long bal = __ballot(t > f);
        if (lane == 0) wcnt[wave][f] = (unsigned int)__popcll(bal);
    }
    __syncthreads();
    for (int f = 0; f < kIfN; f++) {
        const unsigned long long bal = __ballot(t > f);
        if (!bal) continue;
        unsigned int before = blockoff[(size_t)f * nblocks + blockIdx.x];
        for (int w2 = 0; w2 < wave; w2++) before += wcnt[w2][f];
        const unsigned int rank = before + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull)) + 1u;
        const bool ok = t > f && (double)st->total[f] / (double)rank * p < alpha;
        const unsigned long long okb = __ballot(ok);
        if (okb && lane == 63 - __clzll((long long)okb)) atomicMax(&best[f], rank);  // ranks ascend with the lane
    }
    __syncthreads();
    if (threadIdx.x < kIfN && best[threadIdx.x]) atomicMax(&st->numRej[threadIdx.x], best[threadIdx.x]);
}

// Cleveland's LOWESS as in R stats::lowess / clowess.c (x ascending), host side: 50 points
static void lowess_host(const double *x, const double *y, int n, double f, int nsteps, double delta, double *ys) {
    std::vector<double> rw((size_t)n, 1.0), res((size_t)n, 0.0), w((size_t)n, 0.0);
    if (n < 2) {
        for (int i = 0; i < n; i++) ys[i] = y[i];
        return;
    }
    int ns = (int)(f * n + 1e-7);
    if (ns > n) ns = n;
    if (ns < 2) ns = 2;
    auto lowest = [&](double xs, int nleft, int nright, bool userw, double &out) -> bool {
        const double range = x[n - 1] - x[0];
        const double h = std::max(xs - x[nleft], x[nright] - xs), h9 = 0.999 * h, h1 = 0.001 * h;
        double a = 0.0;
        int j = nleft;
        while (j < n) {
            w[j] = 0.0;
            const double r = fabs(x[j] - xs);
            if (r <= h9) {
                if (r <= h1) w[j] = 1.0;
                else { const double q = r / h, c = 1.0 - q * q * q; w[j] = c * c * c; }
                if (userw) w[j] *= rw[j];
                a += w[j];
            } else if (x[j] > xs) break;
            j++;
        }
        const int nrt = j - 1;
        if (a <= 0.0) return false;
        for (j = nleft; j <= nrt; j++) w[j] /= a;
        if (h > 0.0) {
            a = 0.0;
            for (j = nleft; j <= nrt; j++) a += w[j] * x[j];
            double b = xs - a, c = 0.0;
            for (j = nleft; j <= nrt; j++) c += w[j] * (x[j] - a) * (x[j] - a);
            if (sqrt(c) > 0.001 * range) {
                b /= c;
                for (j = nleft; j <= nrt; j++) w[j] *= (b * (x[j] - a) + 1.0);
            }
        }
        out = 0.0;
        for (j = nleft; j <= nrt; j++) out += w[j] * y[j];
        return true;
    };
    for (int it = 0; it <= nsteps; it++) {
        int nleft = 0, nright = ns - 1, last = -1, i = 0;
        for (;;) {
            if (nright < n - 1) {
                const double d1 = x[i] - x[nleft], d2 = x[nright + 1] - x[i];
                if (d1 > d2) { nleft++; nright++; continue; }
            }
            double v;
            ys[i] = lowest(x[i], nleft, nright, it > 0, v) ? v : y[i];
            if (last < i - 1) {
                const double denom = x[i] - x[last];
                for (int j = last + 1; j < i; j++) {
                    const double al = (x[j] - x[last]) / denom;
                    ys[j] = al * ys[i] + (1.0 - al) * ys[last];
                }
            }
            last = i;
            const double cut = x[last] + delta;
            for (i = last + 1; i < n; i++) {
                if (x[i] > cut) break;
                if (x[i] == x[last]) { ys[i] = ys[last]; last = i; }
            }
            i = std::max(last + 1, i - 1);
            if (last >= n - 1) break;
        }
        for (int k = 0; k < n; k++) res[k] = y[k] - ys[k];
        if (it >= nsteps) break;
        double sc = 0.0;
        for (int k = 0; k < n; k++) { rw[k] = fabs(res[k]); sc += rw[k]; }
        sc /= n;
        std::vector<double> srt(rw);
        std::sort(srt.begin(), srt.end());
        const int m1 = n / 2;
        const double cmad = 3.0 * (srt[m1] + srt[n - m1 - 1]);
        if (cmad < 1e-7 * sc) break;
        const double c9 = 0.999 * cmad, c1 = 0.001 * cmad;
        for (int k = 0; k < n; k++) {
            const double r = fabs(res[k]);
            if (r <= c1) rw[k] = 1.0;
            else if (r <= c9) { const double q = r / cmad, c = 1.0 - q * q; rw[k] = c * c; }
            else rw[k] = 0.0;
        }
    }
}


// padj = BH over the rows that pass the chosen filter, in the p-order already at hand: q = m / rank * p for a passing
// row (rank among the passing rows), +Inf otherwise; a suffix minimum; scatter back (NaN for the filtered rows)
__global__ __launch_bounds__(kIfBlock) void if_final_q_kernel(const uint64_t *__restrict__ pkeys, const uint8_t *__restrict__ T, int64_t n,
                                                               const uint32_t *__restrict__ blockoff, int nblocks, const IfState *st, int j,
                                                               double *q) {
    __shared__ unsigned int wcnt[kIfBlock / 64];
    const int64_t k = (int64_t)blockIdx.x * kIfBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool pass = k < n && T[k] > j;
    const unsigned long long bal = __ballot(pass);
    if (lane == 0) wcnt[wave] = (unsigned int)__popcll(bal);
    __syncthreads();
    unsigned int before = blockoff[(size_t)j * nblocks + blockIdx.x];
    for (int w2 = 0; w2 < wave; w2++) before += wcnt[w2];
    const unsigned int rank = before + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull)) + 1u;
    if (k < n) q[k] = pass ? (double)st->total[j] / (double)rank * value_of(pkeys[k]) : INFINITY;
}
__global__ __launch_bounds__(256) void if_final_scatter_kernel(const double *__restrict__ smin, const uint32_t *__restrict__ idx,
                                                               const uint8_t *__restrict__ T, int64_t n, int j, double *padj) {
    for (int64_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) padj[idx[k]] = T[k] > j ? fmin(1.0, smin[k]) : NAN;
}

size_t if_workspace_bytes(int64_t n) {
    const int nblocks = (int)((n + kIfBlock - 1) / kIfBlock);
    size_t sort_tmp = 0;
    uint64_t *k = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, sort_tmp, k, k, (size_t)n, 0, 64, (hipStream_t)0);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    return bh_workspace_bytes(n) + 2 * al(8 * (size_t)n) + al((size_t)n) + al(4 * (size_t)kIfN * nblocks) + al(sort_tmp) + al(sizeof(IfState)) + 256;
}

// returns 0 ok.  info: filterThreshold, filterTheta, index (1-based), theta[50], numRej[50], lowess[50]
// st2 / fork / join: a second stream and two events of the caller's, so that the two independent sorts (baseMean keys
// for the cutoffs, p-values for everything else — each a chain of ~20 small launches that does not fill the GPU at
// 2 M rows) run side by side
int run_independent_filtering(const double *d_bm, const double *d_p, int64_t n, double alpha, double *d_padj, char *ws, hipStream_t st,
                              hipStream_t st2, hipEvent_t fork, hipEvent_t join, chicdiff_results_info *info) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const int nblocks = (int)((n + kIfBlock - 1) / kIfBlock);
    char *bh_ws = ws;
    char *p = ws + bh_workspace_bytes(n);
    uint64_t *k0 = (uint64_t *)p; p += al(8 * (size_t)n);
    uint64_t *k1 = (uint64_t *)p; p += al(8 * (size_t)n);
    uint8_t *T = (uint8_t *)p; p += al((size_t)n);
    uint32_t *blockcnt = (uint32_t *)p; p += al(4 * (size_t)kIfN * nblocks);
    IfState *state = (IfState *)p; p += al(sizeof(IfState));
    void *sort_tmp = p;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, sort_bytes, k0, k1, (size_t)n, 0, 64, st);
    int g = (int)((n + 2047) / 2048);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    if (hipMemsetAsync(state, 0, sizeof(IfState), st) != hipSuccess) return 1;
    // 1. lower = mean(baseMean == 0); cutoffs = quantile(baseMean, theta) (type 7) — on the second stream
    if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(st2, fork, 0) != hipSuccess) return 1;
    if_keys_kernel<<<g, 256, 0, st2>>>(d_bm, n, k0, state);
    if (rocprim::radix_sort_keys(sort_tmp, sort_bytes, k0, k1, (size_t)n, 0, 64, st2) != hipSuccess) return 1;
    if_cuts_kernel<<<1, 64, 0, st2>>>(k1, n, alpha, state);
    if (hipEventRecord(join, st2) != hipSuccess) return 1;
    // 2. one sort by p-value (NA last) in the BH machinery's buffers, then the 50 filtered rejection counts
    //    (bh_ws layout: count | k0 | k1 | i0 | i1 | q | s — see launch_bh_adjust)
    uint64_t *pk0 = (uint64_t *)(bh_ws + 256), *pk1 = (uint64_t *)((char *)pk0 + al(8 * (size_t)n));
    uint32_t *pi0 = (uint32_t *)((char *)pk1 + al(8 * (size_t)n)), *pi1 = (uint32_t *)((char *)pi0 + al(4 * (size_t)n));
    double *q = (double *)((char *)pi1 + al(4 * (size_t)n)), *s = (double *)((char *)q + al(8 * (size_t)n));
    void *ptmp = (char *)s + al(8 * (size_t)n);
    size_t psort = 0, pscan = 0;
    (void)rocprim::radix_sort_pairs(nullptr, psort, pk0, pk1, pi0, pi1, (size_t)n, 0, 64, st);
    (void)rocprim::inclusive_scan(nullptr, pscan, rocprim::make_reverse_iterator(q + n), rocprim::make_reverse_iterator(s + n), (size_t)n,
                                  rocprim::minimum<double>(), st);
    bh_keys_kernel<<<g, 256, 0, st>>>(d_p, n, pk0, pi0, &state->count);
    if (rocprim::radix_sort_pairs(ptmp, psort, pk0, pk1, pi0, pi1, (size_t)n, 0, 64, st) != hipSuccess) return 1;
    if (hipStreamWaitEvent(st, join, 0) != hipSuccess) return 1;  // the cutoffs are there
    if_count_kernel<<<nblocks, kIfBlock, 0, st>>>(pi1, d_bm, n, state, T, blockcnt, nblocks);
    if_scan_kernel<<<kIfN, 1024, 0, st>>>(blockcnt, nblocks, state);
    if_rej_kernel<<<nblocks, kIfBlock, 0, st>>>(pk1, T, n, blockcnt, nblocks, alpha, state);
    // 3. lowess(numRej ~ theta, f = 1/5), threshold choice (DESeq2 pvalueAdjustment): 50 points, on the host — the
    //    one place the host looks at the device's numbers before the end
    IfState h;
    if (hipMemcpyAsync(&h, state, sizeof(IfState), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    double maxRej = 0;
    for (int k = 0; k < kIfN; k++) {
        info->theta[k] = h.info.theta[k];
        info->numRej[k] = (double)h.numRej[k];
        maxRej = std::max(maxRej, info->numRej[k]);
    }
    lowess_host(info->theta, info->numRej, kIfN, 1.0 / 5.0, 3, 0.01 * (info->theta[kIfN - 1] - info->theta[0]), info->lowess);
    int j = 0;
    if (maxRej > 10) {
        double ss = 0, fitmax = info->lowess[0];
        int cnt = 0;
        for (int k = 0; k < kIfN; k++) {
            fitmax = std::max(fitmax, info->lowess[k]);
            if (info->numRej[k] > 0) {
                const double r = info->numRej[k] - info->lowess[k];
                ss += r * r;
                cnt++;
            }
        }
        const double thresh = fitmax - sqrt(cnt ? ss / cnt : 0.0);
        for (int k = 0; k < kIfN; k++)
            if (info->numRej[k] > thresh) { j = k; break; }
    }
    info->index = j + 1;
    info->filterThreshold = h.cut[j];
    info->filterTheta = info->theta[j];
    info->alpha = alpha;
    // 4. padj = BH over the rows that pass the chosen filter
    if_final_q_kernel<<<nblocks, kIfBlock, 0, st>>>(pk1, T, n, blockcnt, nblocks, state, j, q);
    if (rocprim::inclusive_scan(ptmp, pscan, rocprim::make_reverse_iterator(q + n), rocprim::make_reverse_iterator(s + n), (size_t)n,
                                rocprim::minimum<double>(), st) != hipSuccess)
        return 1;
    if_final_scatter_kernel<<<g, 256, 0, st>>>(s, pi1, T, n, j, d_padj);
    return 0;
}

}  // namespace cd
