// post_kernels.hip — the steps either side of the test itself:
//   f1/f3  p.adjust(method = "BH") on device (DESeq2 results() and chicdiff.R:2049), and the application side of
//          IHWcorrection (chicdiff.R:2038-2049): distance -> group cut, weight lookup, renormalisation,
//          weighted p-values, BH;
//   f4     getRegionUniverse, window mode (chicdiff.R:353-426): .expandAvoidBait ranges, clipped to the
//          restriction map and to the bait's chromosome, as a CSR over regions.
// HBM-bound integer/byte work plus one device sort; rocPRIM provides the sort and the scans (plain library
// primitives), the rest is hand-written.
#include <string.h>

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
    return 256 + ((size_t)n * (8 + 8 + 4 + 4 + 8 + 8) + 6 * 256) + tmp + 256;
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
    return 256 + ((size_t)n * 8 + 255) / 256 * 256 + tmp + 256;
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
                                                       int32_t *minOE, int32_t *maxOE, int *bad) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int lo, hi, cnt = 0, mn = INT32_MAX, mx = INT32_MIN;
        const int b = bait[i];
        if (!expand_range(b, oe[i], s, lo, hi)) {
            atomicExch(bad, 1);
        } else {
            const int bc = (b >= 1 && b <= maxfrag) ? chr_of[b] : -1;
            for (int id = lo; id <= hi; id++)
                if (bc >= 0 && ru_keep(id, bc, chr_of, maxfrag)) {
                    cnt++;
                    mn = id < mn ? id : mn;
                    mx = id > mx ? id : mx;
                }
        }
        len[i] = cnt;
        if (minOE) minOE[i] = cnt ? mn : INT32_MIN;
        if (maxOE) maxOE[i] = cnt ? mx : INT32_MIN;
    }
}
// rows are written by consecutive threads (coalesced): a block owns 256 regions, finds each of its rows' region by
// binary search over the block's CSR offsets in LDS, and walks to the row's candidate (at most 2s+1 steps)
__global__ __launch_bounds__(256) void ru_fill_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t n,
                                                      int s, const int32_t *__restrict__ chr_of, int maxfrag,
                                                      const int64_t *__restrict__ ptr, int32_t *ru_bait, int32_t *ru_region,
                                                      int32_t *ru_oe) {
    __shared__ int64_t s_ptr[257];
    __shared__ int s_lo[256], s_hi[256], s_bait[256], s_chr[256];
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int nreg = n - i0 < 256 ? (int)(n - i0) : 256;
        __syncthreads();
        if ((int)threadIdx.x <= nreg) s_ptr[threadIdx.x] = ptr[i0 + threadIdx.x];
        if (threadIdx.x == 0) s_ptr[nreg] = ptr[i0 + nreg];
        if ((int)threadIdx.x < nreg) {
            int lo = 0, hi = -1;
            const int b = bait[i0 + threadIdx.x];
            (void)expand_range(b, oe[i0 + threadIdx.x], s, lo, hi);
            s_lo[threadIdx.x] = lo;
            s_hi[threadIdx.x] = hi;
            s_bait[threadIdx.x] = b;
            s_chr[threadIdx.x] = (b >= 1 && b <= maxfrag) ? chr_of[b] : -1;
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
            const int bc = s_chr[a];
            int id = s_lo[a];
            for (; id <= s_hi[a]; id++)
                if (ru_keep(id, bc, chr_of, maxfrag) && k-- == 0) break;
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
                    int64_t *region_ptr, int32_t *minOE, int32_t *maxOE, int *bad, void *tmp, size_t tmp_bytes, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess) return 1;
    if (hipMemsetAsync(region_ptr + n, 0, sizeof(int64_t), st) != hipSuccess) return 1;
    ru_count_kernel<<<blocks, 256, 0, st>>>(bait, oe, n, s, chr_of, maxfrag, region_ptr, minOE, maxOE, bad);
    size_t need = 0;
    (void)rocprim::exclusive_scan(nullptr, need, region_ptr, region_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st);
    if (need > tmp_bytes) return 2;
    return rocprim::exclusive_scan(tmp, need, region_ptr, region_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st) == hipSuccess ? 0 : 1;
}
void launch_ru_fill(const int32_t *bait, const int32_t *oe, int64_t n, int s, const int32_t *chr_of, int maxfrag,
                    const int64_t *region_ptr, int32_t *ru_bait, int32_t *ru_region, int32_t *ru_oe, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    ru_fill_kernel<<<blocks, 256, 0, st>>>(bait, oe, n, s, chr_of, maxfrag, region_ptr, ru_bait, ru_region, ru_oe);
}

}  // namespace cd
