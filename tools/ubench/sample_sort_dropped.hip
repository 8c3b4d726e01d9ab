// NOT PART OF THE LIBRARY: the sample sort of round 3, kept as the record of a measured-and-dropped variant (DESIGN.md §5:
// 480 us against rocPRIM's 300 us at 2 M pairs).  It compiled into post_kernels.hip's BH / independent-filtering paths through
// launch_sample_sort_pairs(); nothing builds or loads this file.
// sort_kernels.hip — the sort of results() / p.adjust("BH") / the IHW application side:
// (u64 key, u32 row) pairs, ascending by (key, row) — the order a stable sort by key leaves when the rows come in as
// 0..n-1 (DESeq2 pvalueAdjustment's order(), chicdiff.R:2049's p.adjust).
//
// A sample sort written for the sizes this path sees (0.5 M .. 3 M rows per call; outside that range, and if a bucket ever
// overflows, the callers fall back to the library radix sort — post_kernels.hip):
//   1. one workgroup sorts 16 384 evenly spaced samples (16 per thread in registers, bitonic network; partners in other
//      threads are reached through LDS, eight elements at a time) and keeps every 16th as a splitter: 1023 splitters,
//      compared as (key, row) so that no two elements are equal and any number of tied keys spreads over buckets;
//   2. count: every workgroup takes a tile of 8192 elements, finds each element's bucket (binary search over the splitters
//      in LDS), writes the bucket ids (u16) and its tile's bucket histogram;
//   3. one workgroup turns the histograms into bucket offsets (and flags a bucket over the LDS capacity of step 5);
//   4. scatter: a workgroup reserves room in every bucket for its tile (one atomic per non-empty bucket) and moves its
//      elements there;
//   5. one workgroup per bucket sorts it in registers + LDS (the network of step 1, 8 or 16 elements per thread) and
//      writes it to its final place.
// HBM traffic: 12 B in, 2 B + 2 B of ids, 12 B out and in again, 12 B out per element = 52 B against the radix sort's
// 8 passes x 24 B; 6 launches against 8 passes + 17 buffer fills.  Oversampling by 16 makes a bucket's size a Gamma(16)
// multiple of n/1024: the capacity is 2.6 x the mean at the largest n taken, i.e. < 1e-3 chance of one overflow per call.
#include "common.h"

namespace cd {

constexpr int kSsBuckets = 1024, kSsOver = 16, kSsSample = kSsBuckets * kSsOver;
constexpr int kSsTile = 8192;  // elements per workgroup in the count and scatter passes
constexpr int kSsCap = 8192;   // elements a bucket may hold (step 5: 512 threads x 16)
constexpr int64_t kSsMinN = 400000, kSsMaxN = (int64_t)kSsBuckets * kSsCap * 10 / 26;  // mean bucket <= capacity / 2.6

__device__ __forceinline__ bool ss_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}
__device__ __forceinline__ void ss_cswap(uint64_t &ka, uint32_t &ia, uint64_t &kb, uint32_t &ib, bool up) {
    // ascending (up): the smaller of the two ends in a
    if (ss_less(kb, ib, ka, ia) == up) {
        const uint64_t tk = ka; ka = kb; kb = tk;
        const uint32_t ti = ia; ia = ib; ib = ti;
    }
}

// Bitonic sort of M = E x T elements held E per thread (thread t: positions t E .. t E + E - 1), ascending by (key, row).
// Compare-exchanges with a partner inside the thread run on registers; with a partner in another thread (distance >= E)
// the two threads swap their elements through LDS, eight at a time ([element][thread]: conflict-free), and each keeps the
// smaller or the larger.  s_key: 8 T uint64, s_idx: 8 T uint32.
template <int E, int T>
__device__ __forceinline__ void ss_block_sort(uint64_t (&key)[E], uint32_t (&idx)[E], uint64_t *s_key, uint32_t *s_idx) {
    const int t = threadIdx.x;
    constexpr int M = E * T;
#pragma unroll
    for (int k = 2; k <= E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int l = e ^ j;
                if (l > e) ss_cswap(key[e], idx[e], key[l], idx[l], ((t * E + e) & k) == 0);
            }
        }
    }
    for (int k = 2 * E; k <= M; k <<= 1) {
        const bool up = ((t * E) & k) == 0;  // (uniform over the thread's elements: k >= 2 E)
        for (int j = k >> 1; j >= E; j >>= 1) {
            const int pt = t ^ (j / E);
            const bool keep_small = (pt > t) == up;
#pragma unroll
            for (int h = 0; h < E; h += 8) {
                __syncthreads();  // (the previous round's reads are done)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    s_key[e * T + t] = key[h + e];
                    s_idx[e * T + t] = idx[h + e];
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint64_t ok = s_key[e * T + pt];
                    const uint32_t oi = s_idx[e * T + pt];
                    const bool other_less = ss_less(ok, oi, key[h + e], idx[h + e]);
                    if (other_less == keep_small) {
                        key[h + e] = ok;
                        idx[h + e] = oi;
                    }
                }
            }
        }
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int l = e ^ j;
                if (l > e) ss_cswap(key[e], idx[e], key[l], idx[l], up);
            }
        }
    }
}

// rows: the row number that travels with keys[i]; NULL = i itself
__global__ __launch_bounds__(1024) void ss_sample_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ rows, int64_t n,
                                                         uint64_t *spl_key, uint32_t *spl_idx) {
    extern __shared__ uint64_t ss_lds[];
    uint64_t *s_key = ss_lds;
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(s_key + 8 * 1024);
    uint64_t key[kSsOver];
    uint32_t idx[kSsOver];
    const int t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < kSsOver; e++) {
        const int64_t pos = (int64_t)(((unsigned long long)(e * 1024 + t) * (unsigned long long)n) / (unsigned long long)kSsSample);
        key[e] = keys[pos];
        idx[e] = rows ? rows[pos] : (uint32_t)pos;
    }
    ss_block_sort<kSsOver, 1024>(key, idx, s_key, s_idx);
    // sorted sample s sits in thread s / 16, element s % 16: splitter b = sample 16 b
    if (t > 0) {
        spl_key[t - 1] = key[0];
        spl_idx[t - 1] = idx[0];
    }
}

// bucket of x = number of splitters <= x
__device__ __forceinline__ int ss_bucket(const uint64_t *s_sk, const uint32_t *s_si, uint64_t k, uint32_t i) {
    int lo = 0, hi = kSsBuckets - 1;  // answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ss_less(k, i, s_sk[mid], s_si[mid])) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void ss_count_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ rows, int64_t n,
                                                       const uint64_t *__restrict__ spl_key, const uint32_t *__restrict__ spl_idx,
                                                       uint16_t *bid, uint32_t *cnt) {
    __shared__ uint64_t s_sk[kSsBuckets];
    __shared__ uint32_t s_si[kSsBuckets];
    __shared__ uint32_t s_h[kSsBuckets];
    for (int k = threadIdx.x; k < kSsBuckets; k += 256) {
        s_sk[k] = k < kSsBuckets - 1 ? spl_key[k] : ~0ull;
        s_si[k] = k < kSsBuckets - 1 ? spl_idx[k] : ~0u;
        s_h[k] = 0;
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSsTile;
    for (int q = threadIdx.x; q < kSsTile; q += 256) {
        const int64_t i = base + q;
        if (i >= n) break;
        const int b = ss_bucket(s_sk, s_si, keys[i], rows ? rows[i] : (uint32_t)i);
        bid[i] = (uint16_t)b;
        atomicAdd(&s_h[b], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kSsBuckets; k += 256) cnt[(size_t)blockIdx.x * kSsBuckets + k] = s_h[k];
}

// bucket totals -> offsets (off[0..1024]) and the cursors of the scatter pass; overflow: a bucket over the capacity
__global__ __launch_bounds__(1024) void ss_scan_kernel(const uint32_t *__restrict__ cnt, int nblk, uint32_t *off, uint32_t *cursor,
                                                       int *overflow) {
    __shared__ uint32_t s[kSsBuckets];
    const int b = threadIdx.x;
    uint32_t tot = 0;
    for (int g = 0; g < nblk; g++) tot += cnt[(size_t)g * kSsBuckets + b];  // (coalesced across the workgroup)
    if (tot > (uint32_t)kSsCap) *overflow = 1;
    s[b] = tot;
    __syncthreads();
    for (int d = 1; d < kSsBuckets; d <<= 1) {
        const uint32_t add = b >= d ? s[b - d] : 0u;
        __syncthreads();
        s[b] += add;
        __syncthreads();
    }
    const uint32_t excl = s[b] - tot;
    off[b] = excl;
    cursor[b] = excl;
    if (b == kSsBuckets - 1) off[kSsBuckets] = s[b];
}

__global__ __launch_bounds__(256) void ss_scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ rows, int64_t n,
                                                         const uint16_t *__restrict__ bid, const uint32_t *__restrict__ cnt,
                                                         uint32_t *cursor, const int *overflow, uint64_t *tkey, uint32_t *tidx) {
    __shared__ uint32_t s_pos[kSsBuckets];
    if (*overflow) return;
    for (int k = threadIdx.x; k < kSsBuckets; k += 256) {
        const uint32_t c = cnt[(size_t)blockIdx.x * kSsBuckets + k];
        s_pos[k] = c ? atomicAdd(&cursor[k], c) : 0u;  // room for this tile's share of bucket k
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSsTile;
    for (int q = threadIdx.x; q < kSsTile; q += 256) {
        const int64_t i = base + q;
        if (i >= n) break;
        const uint32_t p = atomicAdd(&s_pos[bid[i]], 1u);
        tkey[p] = keys[i];
        tidx[p] = rows ? rows[i] : (uint32_t)i;
    }
}

template <int E>
__device__ __forceinline__ void ss_sort_bucket(const uint64_t *__restrict__ tkey, const uint32_t *__restrict__ tidx, uint32_t lo, uint32_t m,
                                               uint64_t *okey, uint32_t *oidx, uint64_t *s_key, uint32_t *s_idx) {
    uint64_t key[E];
    uint32_t idx[E];
    const int t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < E; e++) {  // any placement will do: coalesced loads, the tail padded with the largest pair
        const uint32_t q = (uint32_t)(e * 512 + t);
        key[e] = q < m ? tkey[lo + q] : ~0ull;
        idx[e] = q < m ? tidx[lo + q] : ~0u;
    }
    ss_block_sort<E, 512>(key, idx, s_key, s_idx);
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t q = (uint32_t)(t * E + e);
        if (q < m) {
            okey[lo + q] = key[e];
            oidx[lo + q] = idx[e];
        }
    }
}
__global__ __launch_bounds__(512) void ss_bucket_kernel(const uint64_t *__restrict__ tkey, const uint32_t *__restrict__ tidx,
                                                        const uint32_t *__restrict__ off, const int *overflow, uint64_t *okey, uint32_t *oidx) {
    __shared__ uint64_t s_key[8 * 512];
    __shared__ uint32_t s_idx[8 * 512];
    if (*overflow) return;
    const uint32_t lo = off[blockIdx.x], m = off[blockIdx.x + 1] - lo;
    if (m == 0) return;
    if (m <= 8 * 512) ss_sort_bucket<8>(tkey, tidx, lo, m, okey, oidx, s_key, s_idx);
    else ss_sort_bucket<16>(tkey, tidx, lo, m, okey, oidx, s_key, s_idx);
}

bool sample_sort_takes(int64_t n) { return n >= kSsMinN && n <= kSsMaxN; }
size_t sample_sort_workspace_bytes(int64_t n) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t nblk = (size_t)((n + kSsTile - 1) / kSsTile);
    return al(8 * (size_t)n) + al(4 * (size_t)n) + al(2 * (size_t)n) + al(4 * nblk * kSsBuckets) + al(12 * kSsBuckets) + al(4 * (kSsBuckets + 1)) +
           al(4 * kSsBuckets) + 256;
}
// (keys, rows or NULL = 0..n-1) -> (okey, oidx) ascending by (key, row).  *overflow (device, zeroed by the caller before the
// first sort that shares it) is set when a bucket did not fit: the outputs are then unspecified and the caller sorts again with
// the library.  Everything is enqueued on st; returns non-zero when a launch failed.
int launch_sample_sort_pairs(const uint64_t *keys, const uint32_t *rows, int64_t n, uint64_t *okey, uint32_t *oidx, char *ws, int *overflow,
                             hipStream_t st) {
    auto take = [&](size_t bytes) { char *r = ws; ws += (bytes + 255) / 256 * 256; return r; };
    const int nblk = (int)((n + kSsTile - 1) / kSsTile);
    uint64_t *tkey = (uint64_t *)take(8 * (size_t)n);
    uint32_t *tidx = (uint32_t *)take(4 * (size_t)n);
    uint16_t *bid = (uint16_t *)take(2 * (size_t)n);
    uint32_t *cnt = (uint32_t *)take(4 * (size_t)nblk * kSsBuckets);
    uint64_t *spl_key = (uint64_t *)take(8 * kSsBuckets);
    uint32_t *spl_idx = (uint32_t *)take(4 * kSsBuckets);
    uint32_t *off = (uint32_t *)take(4 * (kSsBuckets + 1));
    uint32_t *cursor = (uint32_t *)take(4 * kSsBuckets);
    static bool attr_set = false;
    const size_t sample_lds = 8 * 1024 * 12;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ss_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sample_lds) != hipSuccess) return 1;
        attr_set = true;
    }
    ss_sample_kernel<<<1, 1024, sample_lds, st>>>(keys, rows, n, spl_key, spl_idx);
    ss_count_kernel<<<nblk, 256, 0, st>>>(keys, rows, n, spl_key, spl_idx, bid, cnt);
    ss_scan_kernel<<<1, 1024, 0, st>>>(cnt, nblk, off, cursor, overflow);
    ss_scatter_kernel<<<nblk, 256, 0, st>>>(keys, rows, n, bid, cnt, cursor, overflow, tkey, tidx);
    ss_bucket_kernel<<<kSsBuckets, 512, 0, st>>>(tkey, tidx, off, overflow, okey, oidx);
    return hipGetLastError() != hipSuccess;
}

}  // namespace cd
