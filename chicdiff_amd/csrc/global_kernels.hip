// global_kernels.hip — the fit's global statistics and the HBM-bound Chicdiff-side kernels.
//
//  * dispersion trend  alpha(mu) = a0 + a1/mu  — DESeq2 parametricDispersionFit + R glm.fit for
//    Gamma(link="identity"): one fused pass per IRLS step (deviance of
//    the current iterate + weighted-LS sums for the next), fixed-order reductions, and a state
//    machine that runs on the device (fit_state.h).  Single rank: ONE persistent launch with the
//    rows resident in LDS and a grid barrier per pass; sharded: one launch + one all-reduce of the
//    block partials + one reduce-and-step launch per pass;
//  * exact medians (size factors, chicdiff.R:1561-1562 / A1; MAD of log residuals, A3) by a
//    12-bit radix select over order-preserving 64-bit keys: two histogram rounds, then the few
//    hundred candidates left are gathered and sorted (single rank: compaction; sharded: count rows
//    + a sum-all-reduce of disjoint entries) — every step is a SUM, so it shards;
//  * offsets (chicdiff.R:1583-1589, 1635-1638), window sums (:1540-1547), count join (:843-858).
#include "common.h"
#include "devmath.h"
#include "prior_mc.h"

namespace cd {

// ------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double *out) {
    __shared__ double red[K][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < K) out[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// sums[k] = sum_b partials[b*K + k], fixed order (deterministic)
__global__ void reduce_partials_kernel(const double *partials, int nblk, int K, double *sums) {
    __shared__ double red[256];
    for (int k = 0; k < K; k++) {
        double acc = 0;
        for (int b = threadIdx.x; b < nblk; b += 256) acc += partials[(int64_t)b * K + k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[k] = red[0];
        __syncthreads();
    }
}

static inline double *sums_of(const FitWork &w) { return w.partials + (size_t)kRedBlocks * 72; }

// ------------------------------------------------------------------------------------------
// trend (state machine: fit_state.h)
__global__ void trend_init_kernel(FitWork w) { trend_init(w.sc); }

// One row's contribution to a trend pass (fit_state.h trend_row, which the CPU harness runs as written) with the device's
// lean arithmetic: x = 1/baseMean comes in ready, the two quotients become reciprocals (rcp: <= 1 ulp) and the
// logarithm the table-driven tlog — about 55 instructions instead of about 200 (five IEEE divisions and a library log),
// which was half of a pass of the persistent kernel at 2 M rows.
__device__ __forceinline__ void trend_row_dev(const FitScalars *sc, double x, double disp, double *v, const LogEntry *lt) {
    const double r = disp * rcp(fma(sc->coefs[1], x, sc->coefs[0]));
    if (!(r > 1e-4 && r < 15)) return;
    const double mu = fma(sc->b[1], x, sc->b[0]);
    if (!(mu > 0) || !isfinite(mu)) {
        v[7] += 1;
        return;
    }
    const double rmu = rcp(mu), q = disp * rmu;
    v[0] += -2.0 * (tlog(q, lt) - fma(-mu, rmu, q));  // Gamma deviance residual: log(y/mu) - (y - mu)/mu
    const double wt = rmu * rmu;                        // glm.fit weight for Gamma/identity
    v[1] += wt;
    v[2] += wt * x;
    v[3] += wt * x * x;
    v[4] += wt * disp;
    v[5] += wt * x * disp;
    v[6] += 1;
}

__global__ __launch_bounds__(256) void trend_pass_kernel(FitDims d, FitWork w, double minDisp) {
    __shared__ LogEntry s_lt[64];
    log_table_to_lds(s_lt);
    const FitScalars *sc = w.sc;
    if (sc->finished) return;
    double v[kTrendSums] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        if (w.allZero[i]) continue;
        const double y = w.dispGene[i];
        if (!(y > 100 * minDisp)) continue;  // useForFit <- dispGeneEst > 100*minDisp
        trend_row_dev(sc, 1.0 / w.baseMean[i], y, v, s_lt);
    }
    block_reduce_store<kTrendSums>(v, w.partials + (size_t)blockIdx.x * kTrendSums);
}

// one block: fixed-order sum of the block partials; with a single rank the same block also
// advances the state machine (with several ranks the sums are all-reduced first)
__global__ __launch_bounds__(256) void trend_reduce_kernel(FitWork w, int nblk, int do_step) {
    __shared__ double red[kTrendSums][4];
    double *sums = w.partials + (size_t)kRedBlocks * 72;
    if (w.sc->finished) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < kTrendSums; k++) {
        double acc = 0;
        for (int b = threadIdx.x; b < nblk; b += 256) acc += w.partials[(size_t)b * kTrendSums + k];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) red[k][wave] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < kTrendSums; k++) sums[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        if (do_step) trend_step(w.sc, sums);
    }
}
__global__ void trend_step_kernel(FitWork w) { trend_step(w.sc, w.partials + (size_t)kRedBlocks * 72); }

// ---- single-rank fast path: the whole trend fit in ONE persistent launch ------------------------------
// One 1024-thread workgroup per CU; each owns a contiguous block of rows and keeps their
// (baseMean, dispGeneEst) pairs in LDS (2 M rows = 125 KB per CU of the 160 KB), so the ~20 IRLS passes
// of glm.fit read HBM once instead of 20 times and need no kernel boundary: per pass every workgroup
// publishes 8 partial sums, a grid barrier (monotonic counter, agent-scope release/acquire, cdna guide
// G16) follows, and every workgroup adds the partials in the same fixed order and advances its own copy
// of the state machine (fit_state.h) — identical in all workgroups, no broadcast needed.
constexpr int kTpBlocks = 256, kTpThreads = 1024, kTpCap = 8000, kTpMinRows = 2048;  // rows cached per workgroup (2 x 64 000 B of LDS)

// Two-level grid barrier: workgroups arrive at one of 8 group counters (blockIdx % 8, i.e. one per XCD under
// round-robin dispatch — different cache lines, so the arrivals of different groups do not serialise behind one
// another), the last arrival of a group bumps the top counter, everybody polls the top counter.  256 atomics on
// one word cost ~10 us per barrier; this way the longest chain is 32 + 8.  `pass` counts barriers from 0.
__device__ __forceinline__ bool grid_sync(unsigned int *top, unsigned int *groups, unsigned int pass) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int nblk = gridDim.x, g = blockIdx.x & 7u;
        const unsigned int ngroups = nblk < 8u ? nblk : 8u, gsize = (nblk - g + 7u) >> 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int old = __hip_atomic_fetch_add(groups + g * 16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 64-byte stride
        if (old + 1u == (pass + 1u) * gsize) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned int target = (pass + 1u) * ngroups;
        int spins = 0;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < (1 << 24)) {
            __builtin_amdgcn_s_sleep(2);
            spins++;
        }
        ok = spins < (1 << 24);  // a bounded spin: a lost workgroup must not hang the device
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok != 0;
}
// (Measured and dropped, round 4: a barrier by flags — workgroup b publishes `pass + 1` in flags[b], thread t of every workgroup
// polls flags[t]; no read-modify-write, no chain of arrivals.  Slower: 4.6-5.4 us per barrier instead of 3.3-3.8 at 122 workgroups,
// 12.8 instead of 5-6 at 256 — 256 x 256 polling loads per round trip swamp the path to the flags' home.  And the counters
// without the top level — arrivals by an atomic add nobody waits for, lanes 0..7 of every workgroup polling the eight group
// counters, so that no arrival depends on a returned count: workgroup 0 passes the barriers of the trend passes in 2.2 instead of
// 3.6 us (250 k rows) and 3.7 instead of 5-6 (2 M), but the barriers of the MAD's histogram rounds, where every workgroup arrives
// in a burst behind its global atomics, get slower by as much: launch 0.190 -> 0.188 ms at 250 k, 0.365 -> 0.380 at 2 M.)

// Sum over the 64 lanes of a wave, result in lane 63, by DPP row shifts and row broadcasts (an inclusive scan: fixed order, no
// LDS round trips).  The eight sums of a trend pass through ds_bpermute shuffles were 48 dependent LDS trips per wave — with 16
// waves per workgroup 4-6 us of a 12-20 us pass (same stamps).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return x + __hiloint2double(hi, lo);  // lanes without a source (or masked out) add +0.0
}
__device__ __forceinline__ double wave_sum_to_lane63(double x) {
    x = dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = dpp_add<0x118, 0xf>(x);  // row_shr:8   -> lane 15 of every row of 16: the row's sum
    x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3 -> lane 63: everything
    return x;
}

// ---- MAD of the log residuals inside the persistent trend kernel ------------------------------------------------------
// Round 3 followed the trend kernel with resid_kernel and two radix selects — 14 dependent launches and 4 fills that move 16 MB
// each and take 0.17 ms at 2 M rows, 0.11 ms at 250 k: pure launch latency.  The trend kernel has every row in LDS and a grid
// barrier at hand, so it goes on: residuals (written to w.resid as before), then for the median and for the median of the
// absolute deviations the same exact radix select — 12-bit digit histograms (LDS, then one global atomic per non-empty bin), a
// grid barrier, every workgroup picks the bins that hold the two middle ranks, next digit ... until at most kSelCap keys share
// the chosen prefix (after two rounds unless the data are massively tied), then the candidates are appended to a global list,
// one more barrier, and every workgroup sorts the same short list in LDS and reads the order statistics off it.  Exact, so the
// medians — and everything downstream — are the bits of the launches it replaces.
// Global scratch (w.hist, as 32-bit words, zeroed by the kernel): per select s in {median, absdev} and slot q in {lower, upper}
//   hist[s][round 0..5][q][4096], cnt[s][q], then 64-bit keys cand[s][q][kSelCap].
constexpr int kMadSortMax = 512;  // candidates per slot that are sorted rather than narrowed down by another round (<= kSelCap)
constexpr int kMadHistWords = 2 * 6 * 2 * kSelBins;
constexpr int kMadCntWords = 8;                                  // 2 x 2 counters (+ pad)
constexpr int kMadScratchWords = kMadHistWords + kMadCntWords;   // what must be zero; the candidate lists follow (8-byte aligned)
struct MadArgs {
    int enabled;       // 0: the kernel stops after the trend (the host launches the separate MAD step)
    int S, p;
    double prior_in;   // NaN = estimate; the closed-form prior variance is finished here unless by_simulation
    int by_simulation; // residual d.f. <= 3: the host follows up with the residual histogram and prior_mc
};

#ifdef CHICDIFF_MAD_STAMPS
#define MSTAMP(label) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_mad_n < 64) { g_mad_lab[g_mad_n] = (label); g_mad_t[g_mad_n++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
__device__ unsigned long long g_mad_t0, g_mad_t[64];
__device__ int g_mad_lab[64], g_mad_n;
#else
#define MSTAMP(label)
#endif
#ifdef CHICDIFF_TREND_STAMPS  // diagnostic build only: where a pass of the persistent trend kernel spends its time (workgroup 0)
#define TSTAMP(label) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_tr_n < 160) { g_tr_lab[g_tr_n] = (label); g_tr_t[g_tr_n++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
__device__ unsigned long long g_tr_t[160];
__device__ int g_tr_lab[160], g_tr_n;
#else
#define TSTAMP(label)
#endif
// wave-aggregated histogram update: lanes that hold the same digit add once (the first digit — sign and exponent — is shared by
// almost every key of a column, and 64 lanes adding 1 to one LDS word serialise)
__device__ __forceinline__ void hist_add(unsigned int *h, bool valid, unsigned int digit) {
    unsigned long long todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < 4 && todo; it++) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned int dl = (unsigned int)__shfl((int)digit, leader);
        const unsigned long long same = __ballot(valid && digit == dl) & todo;
        if (lane == leader) atomicAdd(&h[dl], (unsigned int)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&h[digit], 1u);  // (digits of the later rounds differ from lane to lane)
}

// One exact select over the keys key_fn(k) of this workgroup's rows; returns the two middle order statistics as keys.
// `sel` picks the scratch; s_a / s_b: two LDS histograms of kSelBins words; sortbuf: kSelCap 64-bit words (may alias them).
// Every workgroup executes the same barriers and reaches the same result.  Returns false if a grid barrier timed out.
template <class KeyFn>
__device__ bool mad_select(KeyFn key_fn, int64_t nrows_block, int sel, unsigned int *gscratch, unsigned int *s_a, unsigned int *s_b,
                           uint64_t *sortbuf, unsigned int *ctr, unsigned int *grp, unsigned int &pass, uint64_t (&result)[2], double &population) {
    __shared__ uint64_t sh_pre[2], sh_res[2];
    __shared__ double sh_rank[2], sh_pop;
    __shared__ unsigned int sh_cnt[2];
    __shared__ double sh_scan[kTpThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned int *ghist = gscratch + (size_t)sel * 6 * 2 * kSelBins;
    unsigned int *gcnt = gscratch + kMadHistWords + sel * 2;
    uint64_t *gcand = reinterpret_cast<uint64_t *>(gscratch + kMadScratchWords) + (size_t)sel * 2 * kSelCap;
    uint64_t p0 = 0, p1 = 0;
    double rank0 = 0, rank1 = 0;
    int fixed_hi = 64;  // bits [fixed_hi, 64) of the two prefixes are decided
    for (int r = 0; r < 6; r++) {
        const int shift = kSelShifts[r], bits = sel_bits(shift), nb = 1 << bits;
        const bool same = p0 == p1;
        for (int k = tid; k < kSelBins; k += kTpThreads) { s_a[k] = 0; s_b[k] = 0; }
        __syncthreads();
        const int64_t trips = (nrows_block + kTpThreads - 1) / kTpThreads;  // (every thread the same number: hist_add ballots)
        for (int64_t t = 0; t < trips; t++) {
            const int64_t k = t * kTpThreads + tid;
            uint64_t key = 0;
            const bool valid = k < nrows_block && key_fn(k, key);
            const unsigned int dig = (unsigned int)((key >> shift) & (uint64_t)(nb - 1));
            const bool m0 = valid && sel_match(key, p0, fixed_hi), m1 = valid && !same && !m0 && sel_match(key, p1, fixed_hi);
            hist_add(s_a, m0, dig);
            if (!same) hist_add(s_b, m1, dig);
        }
        __syncthreads();
        unsigned int *g0 = ghist + ((size_t)r * 2 + 0) * kSelBins, *g1 = g0 + kSelBins;
        for (int k = tid; k < nb; k += kTpThreads) {
            if (s_a[k]) atomicAdd(&g0[k], s_a[k]);
            if (!same && s_b[k]) atomicAdd(&g1[k], s_b[k]);
        }
        MSTAMP(1);
        if (!grid_sync(ctr, grp, pass++)) return false;
        MSTAMP(2);
        // every workgroup: pick, for each slot, the bin that holds its rank (fit_state.h sel_pick, parallel: per-thread partial
        // sums over 4 consecutive bins, a scan over the 1024 threads, the owning thread refines)
        for (int hsel = 0; hsel < (same ? 1 : 2); hsel++) {  // one scan per distinct histogram; both slots read theirs off it
            const unsigned int *g = hsel ? g1 : g0;
            const int per = (nb + kTpThreads - 1) / kTpThreads;  // 4 (1 in the last, 16-bin round)
            double mine[4] = {0, 0, 0, 0}, acc = 0;
            for (int q = 0; q < per; q++) {
                const int b = tid * per + q;
                mine[q] = b < nb ? (double)g[b] : 0.0;
                acc += mine[q];
            }
            double incl = acc;  // inclusive scan over the workgroup (counts: exact in fp64)
            for (int off = 1; off < 64; off <<= 1) {
                const double o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            if (lane == 63) sh_scan[wave] = incl;
            __syncthreads();
            double before_wave = 0, total = 0;
            for (int q = 0; q < kTpThreads / 64; q++) {
                if (q < wave) before_wave += sh_scan[q];
                total += sh_scan[q];
            }
            incl += before_wave;
            if (r == 0 && tid == 0) sh_pop = total;
            const double before = incl - acc;
            const bool last_thread = tid == (nb - 1) / per;
            for (int slot = 0; slot < 2; slot++) {
                if ((slot == 1 && !same) != (hsel == 1)) continue;  // slot 1 reads the second histogram when the prefixes differ
                double rank = slot ? rank1 : rank0;
                if (r == 0) {  // the first histogram's total is the population: ranks of the two middles (R median())
                    const int64_t mi = (int64_t)total;
                    rank = slot ? (double)(mi / 2) : (double)((mi - 1) / 2);
                }
                if ((before <= rank && rank < incl) || (last_thread && rank >= incl && incl == total)) {
                    double cum = before;
                    int q = 0;
                    for (; q < per - 1 && tid * per + q < nb - 1; q++) {
                        if (cum + mine[q] > rank) break;
                        cum += mine[q];
                    }
                    const int b = tid * per + q;
                    sh_pre[slot] = (slot ? p1 : p0) | ((uint64_t)b << shift);
                    sh_rank[slot] = rank - cum;
                    sh_cnt[slot] = g[b];
                }
            }
            __syncthreads();
        }
        MSTAMP(3);
        p0 = sh_pre[0]; p1 = sh_pre[1];
        rank0 = sh_rank[0]; rank1 = sh_rank[1];
        fixed_hi = shift;
        population = sh_pop;
        const unsigned int c0 = sh_cnt[0], c1 = sh_cnt[1];
        __syncthreads();
        if (population <= 0) { result[0] = result[1] = 0; return true; }  // empty: the caller sees population 0
        if (shift == 0) { result[0] = p0; result[1] = p1; return true; }  // every bit decided
        // few enough keys share the prefixes: gather and sort.  (A further round costs ~14 us — two passes over LDS and a grid barrier
        // — and a bitonic sort of 4096 keys ~55 us, of 512 ~10 us: measured, profiles/r04_mad_in_kernel_stamps.txt)
        if (c0 <= (unsigned int)kMadSortMax && c1 <= (unsigned int)kMadSortMax) break;
    }
    // candidates -> global lists (order irrelevant: they are sorted), then every workgroup sorts the same lists
    const bool same = p0 == p1;
    for (int64_t k = tid; k < nrows_block; k += kTpThreads) {
        uint64_t key;
        if (!key_fn(k, key)) continue;
        int slot = -1;
        if (sel_match(key, p0, fixed_hi)) slot = 0;
        else if (!same && sel_match(key, p1, fixed_hi)) slot = 1;
        if (slot < 0) continue;
        const unsigned int pos = atomicAdd(&gcnt[slot], 1u);
        if (pos < (unsigned int)kSelCap) gcand[(size_t)slot * kSelCap + pos] = key;
    }
    MSTAMP(4);
    if (!grid_sync(ctr, grp, pass++)) return false;
    MSTAMP(5);
    // The order statistics wanted, read off each list by COUNTING instead of sorting it (round 4: a bitonic sort of ~100 keys cost
    // 4 us — 28 rounds of compare-exchange with a workgroup barrier each — and there are up to four lists per MAD): the list goes
    // to LDS, P = 1024 / (its length rounded up to a power of two) threads share one candidate, each counts the keys that come
    // before it (smaller, or equal with a lower index) in its slice of the list, the P counts are added by shuffles, and the
    // candidate whose count is the wanted rank is the answer.  Lists longer than 512 keep the sort.
    for (int slot = 0; slot < 2; slot++) {
        const int hslot = (slot == 1 && !same) ? 1 : 0;
        const int m = (int)gcnt[hslot];
        MSTAMP(100 + m);
        if (slot == 1 && same) break;  // (both middles were read off the one list below)
        const int rkA = (int)(slot ? rank1 : rank0), rkB = same ? (int)rank1 : -1;
        const int wantA = rkA < m ? rkA : m - 1, wantB = rkB < 0 ? -1 : (rkB < m ? rkB : m - 1);
        __syncthreads();
        if (m <= kMadSortMax) {
            int mp = 64;
            while (mp < m) mp <<= 1;
            const int P = kTpThreads / mp;  // 2 .. 16 threads per candidate, neighbouring lanes
            for (int e = tid; e < m; e += kTpThreads) sortbuf[e] = gcand[(size_t)hslot * kSelCap + e];
            __syncthreads();
            const int e = tid / P, part = tid - e * P;
            int before = 0;
            const uint64_t mine = e < m ? sortbuf[e] : 0ull;
            if (e < m)
                for (int j = part; j < m; j += P) {
                    const uint64_t o = sortbuf[j];
                    before += (o < mine || (o == mine && j < e)) ? 1 : 0;
                }
            for (int off = 1; off < P; off <<= 1) before += __shfl_xor(before, off);
            if (e < m && part == 0) {  // (ranks are distinct: exactly one candidate has each)
                if (before == wantA) sh_res[0] = mine;
                if (before == wantB) sh_res[1] = mine;
            }
            __syncthreads();
            result[slot] = sh_res[0];
            if (wantB >= 0) result[1] = sh_res[1];
            continue;
        }
        int len = 64;
        while (len < m) len <<= 1;
        for (int e = tid; e < len; e += kTpThreads) sortbuf[e] = e < m ? gcand[(size_t)hslot * kSelCap + e] : ~0ull;
        __syncthreads();
        for (int k2 = 2; k2 <= len; k2 <<= 1)  // bitonic sort, ascending
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int e = tid; e < len; e += kTpThreads) {
                    const int q = e ^ j;
                    if (q > e) {
                        const uint64_t x = sortbuf[e], y = sortbuf[q];
                        const bool up = (e & k2) == 0;
                        if ((x > y) == up) { sortbuf[e] = y; sortbuf[q] = x; }
                    }
                }
                __syncthreads();
            }
        result[slot] = sortbuf[wantA];
        if (wantB >= 0) result[1] = sortbuf[wantB];
        __syncthreads();
    }
    MSTAMP(6);
    return true;
}

__global__ __launch_bounds__(kTpThreads) void trend_persistent_kernel(FitDims d, FitWork w, double minDisp, MadArgs madargs) {
    const bool mad = madargs.enabled != 0;
    __shared__ double s_bm[kTpCap], s_y[kTpCap];  // 1 / baseMean and dispGeneEst (NaN = not used for the fit)
    __shared__ double red[kTrendSums][16];
    __shared__ FitScalars st;  // only the trend fields are used
    __shared__ LogEntry s_lt[64];
    if (threadIdx.x < 64) s_lt[threadIdx.x] = kLogTable[threadIdx.x];  // (the barrier below covers it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (d.n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = (r0 + per < d.n) ? r0 + per : d.n;
    const int64_t nrows = r1 > r0 ? r1 - r0 : 0;
    const int ncache = nrows < kTpCap ? (int)nrows : kTpCap;
    for (int k = tid; k < ncache; k += kTpThreads) {
        const int64_t i = r0 + k;
        const double y = w.dispGene[i];
        s_bm[k] = 1.0 / w.baseMean[i];
        s_y[k] = w.allZero[i] ? NAN : y;  // (useForFit, y > 100 minDisp, is tested per pass: the MAD step below wants y >= 100 minDisp)
    }
    if (tid == 0) trend_init(&st);
    __syncthreads();
    unsigned int *ctr = reinterpret_cast<unsigned int *>(w.barrier), *grp = ctr + 16;  // top counter, then 8 group counters 64 B apart
    double *slots = w.partials;  // [2][gridDim.x][kTrendSums], double-buffered by pass parity
    const double thr = 100 * minDisp;
    // scratch of the MAD step (below): zeroed here, by one workgroup; the barriers of the trend passes publish it
    unsigned int *gscratch = reinterpret_cast<unsigned int *>(w.hist);
    if (mad && blockIdx.x == 0)
        for (int k = tid; k < kMadScratchWords; k += kTpThreads) gscratch[k] = 0;
    bool alive = true;
    unsigned int pass = 0;
#ifdef CHICDIFF_TREND_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tr_n = 0;
#endif
    TSTAMP(9);
    for (; pass < 11 * 27 + 16; pass++) {
        if (st.finished) break;
        TSTAMP(10);
        double v[kTrendSums] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = tid; k < ncache; k += kTpThreads) {
            const double y = s_y[k];
            if (y > thr) trend_row_dev(&st, s_bm[k], y, v, s_lt);  // (NaN fails)
        }
        for (int64_t i = r0 + kTpCap + tid; i < r1; i += kTpThreads) {  // rows beyond the LDS cache stream from HBM
            const double y = w.dispGene[i];
            if (!w.allZero[i] && (y > 100 * minDisp)) trend_row_dev(&st, 1.0 / w.baseMean[i], y, v, s_lt);
        }
        TSTAMP(11);
#pragma unroll
        for (int k = 0; k < kTrendSums; k++) {
            const double x = wave_sum_to_lane63(v[k]);
            if (lane == 63) red[k][wave] = x;
        }
        __syncthreads();
        double *mine = slots + ((size_t)(pass & 1) * gridDim.x + blockIdx.x) * kTrendSums;
        if (tid < kTrendSums) {
            double acc = 0;
            for (int q = 0; q < kTpThreads / 64; q++) acc += red[tid][q];
            mine[tid] = acc;
        }
        TSTAMP(12);
        alive = grid_sync(ctr, grp, pass);
        TSTAMP(13);
        if (!alive) break;
        // every workgroup: fixed-order sum of all partials (wave k sums quantity k: 64 lanes x strided
        // partials, then a shuffle tree — the same order in every workgroup), then the same state-machine step
        const double *all = slots + (size_t)(pass & 1) * gridDim.x * kTrendSums;
        if (wave < kTrendSums) {
            double acc = 0;
            for (unsigned int b = lane; b < gridDim.x; b += 64) acc += all[(size_t)b * kTrendSums + wave];
            acc = wave_sum_to_lane63(acc);
            if (lane == 63) red[wave][0] = acc;
        }
        __syncthreads();
        TSTAMP(14);
        if (tid == 0) {
            double sums[kTrendSums];
            for (int k = 0; k < kTrendSums; k++) sums[k] = red[k][0];
            trend_step(&st, sums);
        }
        __syncthreads();
        TSTAMP(15);
    }
    TSTAMP(16);
    // ---- MAD of the log residuals around the trend (estimateDispersionsFit's varLogDispEsts), see mad_select above --------
    double med = NAN, madv = NAN, nres = 0;
    if (mad && alive) {
#ifdef CHICDIFF_MAD_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_mad_t0 = __builtin_amdgcn_s_memrealtime(); g_mad_n = 0; }
#endif
        const double c0 = st.coefs[0], c1 = st.coefs[1];
        // residuals as resid_kernel forms them (same expressions: same bits), to w.resid and — cached rows — over y in LDS
        for (int64_t k = tid; k < nrows; k += kTpThreads) {
            const int64_t i = r0 + k;
            const double y = k < ncache ? s_y[k] : (w.allZero[i] ? NAN : w.dispGene[i]);
            double r = NAN;
            if (y >= thr) r = log(y) - log(c0 + c1 / w.baseMean[i]);
            w.resid[i] = r;
            if (k < ncache) s_y[k] = r;
        }
        __syncthreads();
        MSTAMP(0);
        // the freed 1 / baseMean cache holds the select's LDS histograms and, afterwards, the candidates being sorted
        unsigned int *s_a = reinterpret_cast<unsigned int *>(s_bm), *s_b = s_a + kSelBins;
        uint64_t *sortbuf = reinterpret_cast<uint64_t *>(s_bm);
        const double *resid_g = w.resid + r0;
        uint64_t res[2];
        double pop = 0;
        auto key_resid = [&](int64_t k, uint64_t &key) {
            const double x = k < ncache ? s_y[k] : resid_g[k];
            if (x != x) return false;
            key = key_of(x);
            return true;
        };
        alive = mad_select(key_resid, nrows, 0, gscratch, s_a, s_b, sortbuf, ctr, grp, pass, res, pop);
        if (alive) {
            nres = pop;
            med = pop > 0 ? (value_of(res[0]) + value_of(res[1])) / 2.0 : NAN;  // R median(): mean of the two middles
            const double m0 = med;
            auto key_absdev = [&](int64_t k, uint64_t &key) {
                double x = k < ncache ? s_y[k] : resid_g[k];
                if (x != x) return false;
                x = fabs(x - m0);
                if (x != x) return false;
                key = key_of(x);
                return true;
            };
            alive = mad_select(key_absdev, nrows, 1, gscratch, s_a, s_b, sortbuf, ctr, grp, pass, res, pop);
            if (alive) madv = 1.4826 * (pop > 0 ? (value_of(res[0]) + value_of(res[1])) / 2.0 : NAN);  // R mad(): constant 1.4826
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        FitScalars *sc = w.sc;
        sc->coefs[0] = st.coefs[0]; sc->coefs[1] = st.coefs[1];
        sc->b[0] = st.b[0]; sc->b[1] = st.b[1];
        sc->devold = st.devold;
        sc->inner_it = st.inner_it; sc->outer_it = st.outer_it; sc->phase = st.phase;
        sc->conv = st.conv;
        sc->failed = alive ? (st.finished ? st.failed : 2) : 3;  // 3: grid barrier timed out
        sc->finished = 1;
#ifdef CHICDIFF_TREND_STAMPS
        for (int q = 0; q < g_tr_n; q++) printf("  trend stamp %3d %8.2f us\n", g_tr_lab[q], (double)(g_tr_t[q] - g_tr_t[0]) / 100.0);
#endif
#ifdef CHICDIFF_MAD_STAMPS
        for (int q = 0; q < g_mad_n; q++) printf("  mad stamp %3d %8.2f us\n", g_mad_lab[q], (double)(g_mad_t[q] - g_mad_t0) / 100.0);
#endif
        if (mad && alive) {
            sc->med = med;
            sc->nres = nres;
            sc->mad = madv;
            if (!madargs.by_simulation) prior_var(sc, madargs.S, madargs.p, madargs.prior_in);  // (else prior_mc_kernel follows)
        }
    }
}
// sharded fits: this rank's slice of the rows the trend is fitted to, written into the all-ranks arrays (zero elsewhere; the
// sum-all-reduce that follows is then an all-gather).  y = NaN marks an all-zero row.
__global__ __launch_bounds__(256) void trend_gather_kernel(FitDims d, FitWork w, double minDisp, double *xg, double *yg) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        const double y = w.dispGene[i];
        xg[i] = w.baseMean[i];
        yg[i] = w.allZero[i] ? NAN : y;  // (the trend kernel applies useForFit itself; the MAD of the residuals wants y >= 100 minDisp)
    }
}
void launch_trend_gather(FitDims d, FitWork w, Opts o, double *xg, double *yg, hipStream_t st) {
    trend_gather_kernel<<<kRedBlocks, 256, 0, st>>>(d, w, o.minDisp, xg, yg);
}
__global__ __launch_bounds__(256) void trend_compact_kernel(GatherLayout gl, const double *__restrict__ recv, double *__restrict__ xg,
                                                            double *__restrict__ yg) {
    const int r = blockIdx.y;
    const int64_t cnt = gl.off[r + 1] - gl.off[r];
    const double *src = recv + (int64_t)r * gl.block;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
        xg[gl.off[r] + i] = src[i];
        yg[gl.off[r] + i] = src[gl.maxn + i];
    }
}
void launch_trend_compact(const GatherLayout &gl, const double *recv, double *xg, double *yg, hipStream_t st) {
    int64_t bx = (gl.maxn + 255) / 256;
    if (bx > 2048 / gl.world) bx = 2048 / gl.world;
    if (bx < 1) bx = 1;
    trend_compact_kernel<<<dim3((unsigned)bx, (unsigned)gl.world), 256, 0, st>>>(gl, recv, xg, yg);
}
__global__ void poke_kernel(int32_t *p, int32_t v) { *p = v; }
void launch_poke(int32_t *p, int32_t v, hipStream_t st) { poke_kernel<<<1, 1, 0, st>>>(p, v); }
__global__ void flag_to_double_kernel(const int32_t *flag, double *out) { *out = *flag ? 1.0 : 0.0; }
void launch_flag_to_double(const int32_t *flag, double *out, hipStream_t st) { flag_to_double_kernel<<<1, 1, 0, st>>>(flag, out); }
int trend_persistent_blocks() { return kTpBlocks; }
void launch_trend_persistent(FitDims d, FitWork w, Opts o, hipStream_t st, bool with_mad) {
    // (the barrier counters were zeroed with the fit's scalars; the trend runs once per fit)
    // as few workgroups as keep every row LDS-resident: the pass time is the grid barrier plus the all-partials sum,
    // both of which grow with the number of workgroups (small fits are latency-bound by these ~20 passes)
    // rows per workgroup: at most kTpCap (LDS), at least ~2 per thread — with 8 per thread the row loop (4 us) was as long
    // as the grid barrier of a small fit, with 2 it is 1 us and the barrier of the 4x larger grid costs less than that saved
    int64_t blocks = (d.n + kTpMinRows - 1) / kTpMinRows;
    if (blocks < 1) blocks = 1;
    if (blocks > kTpBlocks) blocks = kTpBlocks;
    // fewer on request: several fits sharing one GPU (the one-GPU rehearsal of a sharded fit) must keep ALL their trend kernels'
    // workgroups resident at once, or the grid barriers time out; rows beyond the LDS cache stream from HBM / L2
    if (o.trend_blocks > 0 && blocks > o.trend_blocks) blocks = o.trend_blocks;
    MadArgs m{};
    m.enabled = with_mad ? 1 : 0;
    m.S = d.S;
    m.p = d.p;
    m.prior_in = o.dispPriorVarIn;
    m.by_simulation = (!(o.dispPriorVarIn == o.dispPriorVarIn) && d.S - d.p <= 3 && d.S > d.p) ? 1 : 0;
    trend_persistent_kernel<<<(unsigned)blocks, kTpThreads, 0, st>>>(d, w, o.minDisp, m);
}

void launch_trend_init(FitDims, FitWork w, Opts, hipStream_t st) { trend_init_kernel<<<1, 1, 0, st>>>(w); }
constexpr int kTrendBlocks = 512;
int trend_blocks() { return kTrendBlocks; }
void launch_trend_pass(FitDims d, FitWork w, Opts o, hipStream_t st, bool fused_step) {
    trend_pass_kernel<<<kTrendBlocks, 256, 0, st>>>(d, w, o.minDisp);
    if (fused_step) trend_reduce_kernel<<<1, 256, 0, st>>>(w, kTrendBlocks, 1);
}
void launch_trend_step(FitDims, FitWork w, Opts, hipStream_t st) { trend_reduce_kernel<<<1, 256, 0, st>>>(w, kTrendBlocks, 1); }

// residuals of log gene-wise estimates around the trend (rows with dispGeneEst >= 100*minDisp)
__global__ __launch_bounds__(256) void resid_kernel(FitDims d, FitWork w, double minDisp) {
    const double c0 = w.sc->coefs[0], c1 = w.sc->coefs[1];
    const bool local = w.sc->trend_local != 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        double r = NAN;
        if (!w.allZero[i]) {
            const double y = w.dispGene[i];
            if (y >= 100 * minDisp) r = log(y) - log(local ? w.dispFit[i] : c0 + c1 / w.baseMean[i]);
        }
        w.resid[i] = r;
    }
}
void launch_dispfit_resid(FitDims d, FitWork w, Opts o, hipStream_t st) {
    resid_kernel<<<kRedBlocks, 256, 0, st>>>(d, w, o.minDisp);
}

// ---- local-regression trend: DESeq2 localDispersionFit = locfit(log disp ~ log mean, weights = mean) with locfit's
// defaults (alpha = 0.7 nearest-neighbour bandwidth, local quadratic, tricube kernel, rbox(cut = 0.8) tree, Hermite
// interpolation) — DESeq2's own substitute when the parametric trend fails, or fitType = "local" on request.  A handful
// of vertices (about ten), each needing one order statistic (the k-th smallest |x_i - x_v|: a byte-wise radix select,
// eight histogram rounds) and eight weighted sums over the rows; the tree itself grows on the host (api.hip,
// local_trend_fit).  Every statistic is a sum over rows, so the sharded path all-reduces it like the others.  Rare
// path: clarity over speed.  Rows in the fit: dispGeneEst > 100 minDisp (useForFit), x = log baseMean.
__device__ __forceinline__ bool lf_row(const FitWork &w, int64_t i, double minDisp, double &x) {
    if (w.allZero[i] || !(w.dispGene[i] > 100 * minDisp)) return false;
    x = log(w.baseMean[i]);
    return true;
}
// hist[b] += rows whose key (of x, or of |x - xv|) agrees with `prefix` above bit shift + 8 and has byte b at `shift`
__global__ __launch_bounds__(256) void lf_hist_kernel(FitDims d, FitWork w, double minDisp, int use_dist, double xv, uint64_t prefix, int shift,
                                                      double *hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        double x;
        if (!lf_row(w, i, minDisp, x)) continue;
        const uint64_t key = key_of(use_dist ? fabs(x - xv) : x);
        if (shift < 56 && (key >> (shift + 8)) != (prefix >> (shift + 8))) continue;
        atomicAdd(&h[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (double)h[threadIdx.x]);  // integer counts: exact in any order
}
// partial sums of w K(u) dx^q (q = 0..4) and w K(u) y dx^q (q = 0..2) per block; lf_sums_finish adds them in a fixed order
__global__ __launch_bounds__(256) void lf_sums_kernel(FitDims d, FitWork w, double minDisp, double xv, double h, double *partials) {
    __shared__ double red[256];
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        double x;
        if (!lf_row(w, i, minDisp, x)) continue;
        const double dx = x - xv, u = fabs(dx) / h;
        if (u >= 1.0) continue;
        const double c = 1.0 - u * u * u, y = log(w.dispGene[i]);
        double p = w.baseMean[i] * (c * c * c);
        for (int q = 0; q < 5; q++) {
            v[q] += p;
            if (q < 3) v[5 + q] += p * y;
            p *= dx;
        }
    }
    for (int k = 0; k < 8; k++) {
        red[threadIdx.x] = v[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * 8 + k] = red[0];
        __syncthreads();
    }
}
__global__ void lf_sums_finish_kernel(const double *partials, int nblocks, double *out) {
    const int k = threadIdx.x;
    if (k >= 8) return;
    double s = 0;
    for (int b = 0; b < nblocks; b++) s += partials[(size_t)b * 8 + k];
    out[k] = s;
}
// dispFit = exp(predict(fit, log baseMean)) for every row with counts; marks the trend as local
__global__ __launch_bounds__(256) void lf_eval_kernel(FitDims d, FitWork w, LfVerts v) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.sc->trend_local = 1;
        w.sc->coefs[0] = w.sc->coefs[1] = NAN;
        w.sc->failed = 0;
        w.sc->finished = 1;
    }
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        if (w.allZero[i]) continue;
        const double x = log(w.baseMean[i]);
        int j = 0;
        while (j + 2 < v.nv && x >= v.x[j + 1]) j++;  // the cell a descent with "left when x < midpoint" ends in
        const double z = v.x[j + 1] - v.x[j], t = (x - v.x[j]) / z;
        double p0, p1, p2, p3;  // locfit hermite2: cubic inside the cell, the end vertex's line outside the data range
        if (t < 0) { p0 = 1; p1 = 0; p2 = t; p3 = 0; }
        else if (t > 1) { p0 = 0; p1 = 1; p2 = 0; p3 = t - 1; }
        else { p1 = t * t * (3 - 2 * t); p0 = 1 - p1; p2 = t * (1 - t) * (1 - t); p3 = t * t * (t - 1); }
        w.dispFit[i] = exp(p0 * v.f[j] + p1 * v.f[j + 1] + (p2 * v.d[j] + p3 * v.d[j + 1]) * z);
    }
}
void launch_lf_hist(FitDims d, FitWork w, Opts o, int use_dist, double xv, uint64_t prefix, int shift, double *hist, hipStream_t st) {
    (void)hipMemsetAsync(hist, 0, sizeof(double) * 256, st);
    lf_hist_kernel<<<kRedBlocks, 256, 0, st>>>(d, w, o.minDisp, use_dist, xv, prefix, shift, hist);
}
void launch_lf_sums(FitDims d, FitWork w, Opts o, double xv, double h, double *partials, double *out8, hipStream_t st) {
    lf_sums_kernel<<<kRedBlocks, 256, 0, st>>>(d, w, o.minDisp, xv, h, partials);
    lf_sums_finish_kernel<<<1, 64, 0, st>>>(partials, kRedBlocks, out8);
}
void launch_lf_eval(FitDims d, FitWork w, const LfVerts &v, hipStream_t st) { lf_eval_kernel<<<kRedBlocks, 256, 0, st>>>(d, w, v); }

// 40-bin histogram of the log dispersion residuals inside (-10, 10) (hist(breaks = -20:20/2) as hist.default counts
// it, pmc_bin): input of the simulation-matched prior variance for residual d.f. <= 3 (prior_mc.h).  Counts as
// doubles: a sum-all-reducible statistic like the others.
__global__ __launch_bounds__(256) void resid_hist_kernel(FitDims d, FitWork w, double *out) {
    __shared__ unsigned int h[kPmcBins];
    if (threadIdx.x < kPmcBins) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        const int b = pmc_bin(w.resid[i]);  // NaN = not part of the fit
        if (b >= 0) atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kPmcBins && h[threadIdx.x]) atomicAdd(&out[threadIdx.x], (double)h[threadIdx.x]);
}
void launch_resid_hist(FitDims d, FitWork w, double *out40, hipStream_t st) {
    (void)hipMemsetAsync(out40, 0, sizeof(double) * 40, st);
    resid_hist_kernel<<<256, 256, 0, st>>>(d, w, out40);
}

// estimateDispersionsPriorVar for residual d.f. <= 3 (prior_mc.h): KL of the observed residual density against the 200
// simulated ones, loess as R evaluates it (local fits at the k-d tree vertices, cubic Hermite in between) on the fine
// grid, first minimum — the pieces of pmc_prior_var(), one workgroup, no host round trip.  t: the table of this
// d.f. (built once per process on the host).
__global__ __launch_bounds__(256) void prior_mc_kernel(FitDims d, FitWork w, const double *hist, const PmcTable *t) {
    __shared__ double obs[kPmcBins], kl[kPmcGrid], s_nobs, bestv[256], vert[kPmcMaxVert], val[kPmcMaxVert], slope[kPmcMaxVert];
    __shared__ int besti[256];
    if (threadIdx.x == 0) {
        double nobs = 0;
        for (int b = 0; b < kPmcBins; b++) nobs += hist[b];
        s_nobs = nobs;
    }
    __syncthreads();
    if (!(s_nobs > 0)) {  // no residuals at all: the closed form (its 0.25 floor)
        if (threadIdx.x == 0) prior_var(w.sc, d.S, d.p, NAN);
        return;
    }
    if (threadIdx.x < kPmcBins) obs[threadIdx.x] = hist[threadIdx.x] / (s_nobs * 0.5);
    __syncthreads();
    for (int g = threadIdx.x; g < kPmcGrid; g += 256) kl[g] = pmc_kl(obs, t->dens[g]);
    __syncthreads();
    const int nvert = t->nvert;
    if ((int)threadIdx.x < nvert) {
        vert[threadIdx.x] = t->vert[threadIdx.x];
        pmc_vertex(*t, threadIdx.x, kl, val[threadIdx.x], slope[threadIdx.x]);
    }
    __syncthreads();
    double best = INFINITY;
    int bi = kPmcFine;
    for (int f = threadIdx.x; f < kPmcFine; f += 256) {
        const double fit = pmc_loess_eval(vert, nvert, val, slope, pmc_fine_x(f));
        if (fit < best) { best = fit; bi = f; }
    }
    bestv[threadIdx.x] = best;
    besti[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {  // which.min: the first minimum
        if ((int)threadIdx.x < off) {
            const double ov = bestv[threadIdx.x + off];
            const int oi = besti[threadIdx.x + off];
            if (ov < bestv[threadIdx.x] || (ov == bestv[threadIdx.x] && oi < besti[threadIdx.x])) {
                bestv[threadIdx.x] = ov;
                besti[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double arg = besti[0] < kPmcFine ? pmc_fine_x(besti[0]) : 0.0;
        prior_var(w.sc, d.S, d.p, arg > 0.25 ? arg : 0.25);
    }
}
void launch_prior_mc(FitDims d, FitWork w, const double *hist40, const void *table, hipStream_t st) {
    prior_mc_kernel<<<1, 256, 0, st>>>(d, w, hist40, (const PmcTable *)table);
}

// estimateDispersionsPriorVar, closed-form branch (A4): fit_state.h
__global__ void prior_var_kernel(FitDims d, FitWork w, double prior_in) { prior_var(w.sc, d.S, d.p, prior_in); }
void launch_prior_var(FitDims d, FitWork w, Opts o, hipStream_t st) {
    prior_var_kernel<<<1, 1, 0, st>>>(d, w, o.dispPriorVarIn);
}

// ------------------------------------------------------------------------------------------
// radix select
__device__ __forceinline__ bool sel_key(const SelArgs &a, const FitScalars *sc, int col, int64_t i, uint64_t &key) {
    double x;
    if (a.mode == SEL_SIZEFACTOR) {
        x = a.ratio[(int64_t)col * a.n + i];  // log(k) - loggeomean, NaN = excluded (row_ratio_kernel)
        if (x != x) return false;
    } else {
        x = a.resid[i];
        if (x != x) return false;
        if (a.mode == SEL_ABSDEV) x = fabs(x - sc->med);
    }
    key = key_of(x);
    return true;
}

// the element behind sel_key() as it sits in memory (NaN = excluded), and the key of a loaded element: the row loops below load
// four elements before they work on the first, so that a thread has four loads in flight instead of one
__device__ __forceinline__ double sel_raw(const SelArgs &a, int col, int64_t i) {
    return a.mode == SEL_SIZEFACTOR ? a.ratio[(int64_t)col * a.n + i] : a.resid[i];
}
__device__ __forceinline__ bool sel_key_of_raw(const SelArgs &a, const FitScalars *sc, double x, uint64_t &key) {
    if (x != x) return false;
    if (a.mode == SEL_ABSDEV) x = fabs(x - sc->med);
    key = key_of(x);
    return true;
}

// histograms of the current digit for the two live prefixes of column blockIdx.y
__global__ __launch_bounds__(256) void sel_hist_kernel(SelArgs a, FitWork w, int bits) {
    __shared__ unsigned int h[2][kSelBins];
    const int col = blockIdx.y;
    const FitScalars *sc = w.sc;
    if (!sel_first_round(a.shift) && sc->sel_fast_done) return;  // the shortcut already has the answers
    for (int k = threadIdx.x; k < 2 * kSelBins; k += 256) (&h[0][0])[k] = 0;
    __syncthreads();
    const bool first = sel_first_round(a.shift);  // first round: every key, whatever an earlier select left behind
    const uint64_t p0 = first ? 0 : sc->sel_prefix[2 * col], p1 = first ? 0 : sc->sel_prefix[2 * col + 1];
    const int hi = a.shift + bits;  // bits above `hi` are fixed by the prefix
    const bool same = (p0 == p1);
    const uint64_t mask = (1ull << bits) - 1ull;
    uint64_t key;
    // Run-length accumulation per thread: the first digit is the sign and exponent of the key, which a whole column
    // shares but for a handful of values — 64 lanes adding 1 to the same LDS word serialise, so a thread adds a run of
    // equal digits at once (order-free sums: the histogram is the same).
    int cur = -1;
    unsigned cnt = 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < a.n; i0 += 4 * step) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = i0 + u * step < a.n ? sel_raw(a, col, i0 + u * step) : NAN;
#pragma unroll
        for (int u = 0; u < 4; u++) {  // (a thread's elements in the order of the one-by-one loop: the same runs)
            if (!sel_key_of_raw(a, sc, x[u], key)) continue;
            const unsigned dig = (unsigned)((key >> a.shift) & mask);
            int bin;
            if (sel_match(key, p0, hi)) bin = (int)dig;
            else if (!same && sel_match(key, p1, hi)) bin = (int)(kSelBins + dig);
            else continue;
            if (bin == cur) cnt++;
            else {
                if (cnt) atomicAdd(&(&h[0][0])[cur], cnt);
                cur = bin;
                cnt = 1;
            }
        }
    }
    if (cnt) atomicAdd(&(&h[0][0])[cur], cnt);
    __syncthreads();
    double *g = w.hist + (size_t)col * 2 * kSelBins;
    for (int k = threadIdx.x; k < 2 * kSelBins; k += 256) {
        const unsigned c = (&h[0][0])[k];
        if (c) atomicAdd(&g[k], (double)c);
    }
}

// pick the bin holding the wanted rank (parallel form of fit_state.h sel_pick); one block per
// column, both rank slots in turn: per-thread partial sums, block scan, the owning thread refines
__global__ __launch_bounds__(256) void sel_step_kernel(SelArgs a, FitWork w, int bits) {
    __shared__ double scan[256];
    const int col = blockIdx.x;
    FitScalars *sc = w.sc;
    const bool first = sel_first_round(a.shift);
    if (!first && sc->sel_fast_done) return;
    if (first && col == 0 && threadIdx.x == 0) sc->sel_fast_done = 0;  // (read again only after later launches)
    if (a.shift == 40 && threadIdx.x < 2) sc->sel_cnt[2 * col + threadIdx.x] = 0;  // write positions of the compaction that follows
    const uint64_t p0 = first ? 0 : sc->sel_prefix[2 * col], p1 = first ? 0 : sc->sel_prefix[2 * col + 1];
    double rank0 = sc->sel_rank[2 * col], rank1 = sc->sel_rank[2 * col + 1];
    const int nb = 1 << bits, per = (nb + 255) / 256;
    for (int slot = 0; slot < 2; slot++) {
        const int hslot = (slot == 1 && p0 != p1) ? 1 : 0;
        const double *g = w.hist + ((size_t)col * 2 + hslot) * kSelBins;
        double mine[16];
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int b = threadIdx.x * per + k;
            mine[k] = (k < per && b < nb) ? g[b] : 0.0;
            acc += mine[k];
        }
        __syncthreads();
        scan[threadIdx.x] = acc;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // inclusive Hillis-Steele scan (counts: exact in fp64)
            const double add = (int)threadIdx.x >= off ? scan[threadIdx.x - off] : 0.0;
            __syncthreads();
            scan[threadIdx.x] += add;
            __syncthreads();
        }
        if (first && slot == 0) {  // the first histogram's total is the population: ranks of the two middles
            const int64_t mi = (int64_t)scan[255];
            rank0 = (double)((mi - 1) / 2);
            rank1 = (double)(mi / 2);
            if (threadIdx.x == 0) sc->sel_count[col] = (double)mi;
        }
        const double rank = slot ? rank1 : rank0;
        const double incl = scan[threadIdx.x], before = incl - acc;
        const bool last_thread = (int)threadIdx.x == (nb - 1) / per;
        if ((before <= rank && rank < incl) || (last_thread && rank >= incl && incl == scan[255])) {
            double cum = before;
            int k = 0;
            for (; k < per - 1 && threadIdx.x * per + k < nb - 1; k++) {
                if (cum + mine[k] > rank) break;
                cum += mine[k];
            }
            const int b = threadIdx.x * per + k;
            const uint64_t pre = (slot == 0) ? p0 : p1;
            sc->sel_prefix[2 * col + slot] = pre | ((uint64_t)b << a.shift);
            sc->sel_rank[2 * col + slot] = rank - cum;
        }
    }
}

// ---- single-rank shortcut: after two rounds (24 of 64 key bits fixed) only a handful of keys share
// the live prefixes; gather them and finish in one block per column instead of four more passes over
// all the data.  Exact and order-independent (a sort), so results equal the six-round path bit for bit.
__global__ __launch_bounds__(256) void sel_compact_kernel(SelArgs a, FitWork w) {
    const int col = blockIdx.y;
    FitScalars *sc = w.sc;
    const uint64_t p0 = sc->sel_prefix[2 * col], p1 = sc->sel_prefix[2 * col + 1];
    const bool same = p0 == p1;
    uint64_t *cand = reinterpret_cast<uint64_t *>(w.hist) + (size_t)col * 2 * kSelCap;
    uint64_t key;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < a.n; i0 += 4 * step) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = i0 + u * step < a.n ? sel_raw(a, col, i0 + u * step) : NAN;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!sel_key_of_raw(a, sc, x[u], key)) continue;
            int slot = -1;
            if (sel_match(key, p0, 40)) slot = 0;
            else if (!same && sel_match(key, p1, 40)) slot = 1;
            if (slot < 0) continue;
            const unsigned pos = atomicAdd(&sc->sel_cnt[2 * col + slot], 1u);
            if (pos < (unsigned)kSelCap) cand[(size_t)slot * kSelCap + pos] = key;
        }
    }
}
__device__ __forceinline__ void sel_tail_rounds(const SelArgs &a, FitScalars *sc, int col, unsigned int *h);
__device__ __forceinline__ void sel_finish_col(const SelArgs &a, FitScalars *sc, int c);
__global__ __launch_bounds__(256) void sel_small_kernel(SelArgs a, FitWork w) {
    __shared__ uint64_t s_k[kSelCap];
    FitScalars *sc = w.sc;
    bool fits = true;
    for (int q = 0; q < 2 * a.ncol; q++) fits &= sc->sel_cnt[q] <= (unsigned)kSelCap;  // same verdict in every block
    const int col = blockIdx.x;
    if (!fits) {  // massive ties: this column's remaining radix rounds, here and now
        sel_tail_rounds(a, sc, col, reinterpret_cast<unsigned int *>(s_k));
        if (threadIdx.x == 0) sel_finish_col(a, sc, col);
        return;
    }
    const uint64_t p0 = sc->sel_prefix[2 * col], p1 = sc->sel_prefix[2 * col + 1];
    const uint64_t *cand = reinterpret_cast<const uint64_t *>(w.hist) + (size_t)col * 2 * kSelCap;
    uint64_t result[2] = {p0, p1};
    for (int slot = 0; slot < 2; slot++) {
        const int hslot = (slot == 1 && p0 != p1) ? 1 : 0;
        const int m = (int)sc->sel_cnt[2 * col + hslot];
        if (m == 0) continue;  // empty population: prefixes stay as they are (median NaN via sel_count)
        int len = 64;
        while (len < m) len <<= 1;
        __syncthreads();
        for (int e = threadIdx.x; e < len; e += 256) s_k[e] = e < m ? cand[(size_t)hslot * kSelCap + e] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= len; k <<= 1)  // bitonic sort, ascending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int e = threadIdx.x; e < len; e += 256) {
                    const int p = e ^ j;
                    if (p > e) {
                        const uint64_t x = s_k[e], y = s_k[p];
                        const bool up = (e & k) == 0;
                        if ((x > y) == up) { s_k[e] = y; s_k[p] = x; }
                    }
                }
                __syncthreads();
            }
        const int r = (int)sc->sel_rank[2 * col + slot];  // rank inside the prefix (set by the second round)
        result[slot] = s_k[r < m ? r : m - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc->sel_prefix[2 * col] = result[0];
        sc->sel_prefix[2 * col + 1] = result[1];
        if (col == 0) sc->sel_fast_done = 1;
        sel_finish_col(a, sc, col);  // (a column's order statistics and what follows from them are its own)
    }
}

// The shortcut's own fallback: more than kSelCap candidates share 24 key bits with a median (massive ties).
// The column's workgroup runs the four remaining radix rounds by itself — slow (one workgroup scans all n), but it
// keeps the host from launching those rounds every time just in case.  `h` : 2 x kSelBins counters in LDS.
__device__ __forceinline__ void sel_tail_rounds(const SelArgs &a, FitScalars *sc, int col, unsigned int *h) {
    __shared__ uint64_t s_pre[2];
    __shared__ double s_rank[2];
    const int nthr = blockDim.x;
    if (threadIdx.x < 2) {
        s_pre[threadIdx.x] = sc->sel_prefix[2 * col + threadIdx.x];
        s_rank[threadIdx.x] = sc->sel_rank[2 * col + threadIdx.x];
    }
    for (int r = 2; r < 6; r++) {
        const int shift = kSelShifts[r], bits = sel_bits(shift), hi = shift + bits;
        __syncthreads();
        for (int k = threadIdx.x; k < 2 * kSelBins; k += nthr) h[k] = 0;
        __syncthreads();
        const uint64_t p0 = s_pre[0], p1 = s_pre[1], mask = (1ull << bits) - 1ull;
        const bool same = p0 == p1;
        uint64_t key;
        for (int64_t i = threadIdx.x; i < a.n; i += nthr) {
            if (!sel_key(a, sc, col, i, key)) continue;
            const unsigned dig = (unsigned)((key >> shift) & mask);
            if (sel_match(key, p0, hi)) atomicAdd(&h[dig], 1u);
            else if (!same && sel_match(key, p1, hi)) atomicAdd(&h[kSelBins + dig], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 2) {  // fit_state.h sel_pick, on the workgroup's own histogram
            const int slot = threadIdx.x, hslot = (slot == 1 && !same) ? 1 : 0;
            const double rank = s_rank[slot];
            const int nb = 1 << bits;
            double cum = 0;
            int b = 0;
            for (; b < nb - 1; b++) {
                if (cum + (double)h[hslot * kSelBins + b] > rank) break;
                cum += (double)h[hslot * kSelBins + b];
            }
            s_pre[slot] = (slot ? p1 : p0) | ((uint64_t)b << shift);
            s_rank[slot] = rank - cum;
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) sc->sel_prefix[2 * col + threadIdx.x] = s_pre[threadIdx.x];
    __syncthreads();
}
// prefixes -> order statistics -> what the select was for (one thread per column)
__device__ __forceinline__ void sel_finish_col(const SelArgs &a, FitScalars *sc, int c) {
    const double med = sel_median(sc, c);
    sc->sel_value[2 * c] = value_of(sc->sel_prefix[2 * c]);
    sc->sel_value[2 * c + 1] = value_of(sc->sel_prefix[2 * c + 1]);
    if (a.mode == SEL_RESID) {
        sc->med = med;
        sc->nres = sc->sel_count[c];
    } else if (a.mode == SEL_ABSDEV) {
        sc->mad = 1.4826 * med;  // R mad(): constant 1.4826
    } else {
        sc->sel_value[2 * c] = exp(med);  // size factor of column c
        if (a.sf_out) a.sf_out[c] = sc->sel_value[2 * c];
    }
}

// ---- sharded shortcut (protocol and layout: fit_state.h) -------------------------------------------------------
__global__ void sel_gcount_kernel(SelArgs a, FitWork w, int world, int rank) {
    const int nq = 2 * a.ncol;
    for (int k = threadIdx.x; k < world * nq; k += blockDim.x) {
        const int r = k / nq, q = k - r * nq;
        w.selcnt[k] = r == rank ? sel_local_count(w.sc, w.hist_local, q >> 1, q & 1) : 0.0;
    }
    if (threadIdx.x < (unsigned)nq) w.sc->sel_cnt[threadIdx.x] = 0;  // write positions of the place pass
    if (threadIdx.x == 0) w.sc->sel_fast_done = 0;
}
__device__ __forceinline__ bool sel_gather_fits(const double *cnt, int world, int nq) {
    for (int q = 0; q < nq; q++) {
        double t = 0;
        for (int r = 0; r < world; r++) t += cnt[(size_t)r * nq + q];
        if (t > (double)kSelCap) return false;
    }
    return true;
}
// this rank's candidates, as values, at its offset of the (zeroed) hist buffer; the sum-all-reduce then gathers
__global__ __launch_bounds__(256) void sel_gplace_kernel(SelArgs a, FitWork w, int world, int rank) {
    const int col = blockIdx.y, nq = 2 * a.ncol;
    FitScalars *sc = w.sc;
    if (!sel_gather_fits(w.selcnt, world, nq)) return;  // same verdict on every rank: the histogram rounds go on
    const uint64_t p0 = sc->sel_prefix[2 * col], p1 = sc->sel_prefix[2 * col + 1];
    const bool same = p0 == p1;
    double base[2], tot;
    sel_gather_layout(w.selcnt, world, rank, nq, 2 * col, &base[0], &tot);
    sel_gather_layout(w.selcnt, world, rank, nq, 2 * col + 1, &base[1], &tot);
    uint64_t key;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        if (!sel_key(a, sc, col, i, key)) continue;
        int slot = -1;
        if (sel_match(key, p0, 40)) slot = 0;
        else if (!same && sel_match(key, p1, 40)) slot = 1;
        if (slot < 0) continue;
        const unsigned pos = atomicAdd(&sc->sel_cnt[2 * col + slot], 1u);
        w.hist[((size_t)2 * col + slot) * kSelCap + (size_t)base[slot] + pos] = value_of(key);
    }
}
__global__ __launch_bounds__(256) void sel_gfinish_kernel(SelArgs a, FitWork w, int world, int rank) {
    __shared__ uint64_t s_k[kSelCap];
    FitScalars *sc = w.sc;
    const int nq = 2 * a.ncol;
    if (!sel_gather_fits(w.selcnt, world, nq)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *(a.overflow_out ? a.overflow_out : &sc->sel_overflow) = 1;  // the host refits with every histogram round (api.hip)
        return;
    }
    const int col = blockIdx.x;
    const uint64_t p0 = sc->sel_prefix[2 * col], p1 = sc->sel_prefix[2 * col + 1];
    uint64_t result[2] = {p0, p1};
    for (int slot = 0; slot < 2; slot++) {
        const int hslot = (slot == 1 && p0 != p1) ? 1 : 0;
        double b, t;
        sel_gather_layout(w.selcnt, world, rank, nq, 2 * col + hslot, &b, &t);
        const int m = (int)t;
        if (m == 0) continue;
        int len = 64;
        while (len < m) len <<= 1;
        __syncthreads();
        for (int e = threadIdx.x; e < len; e += 256) s_k[e] = e < m ? key_of(w.hist[((size_t)2 * col + hslot) * kSelCap + e]) : ~0ull;
        __syncthreads();
        for (int k = 2; k <= len; k <<= 1)  // bitonic sort, ascending
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int e = threadIdx.x; e < len; e += 256) {
                    const int p = e ^ j;
                    if (p > e) {
                        const uint64_t x = s_k[e], y = s_k[p];
                        const bool up = (e & k) == 0;
                        if ((x > y) == up) { s_k[e] = y; s_k[p] = x; }
                    }
                }
                __syncthreads();
            }
        const int r = (int)sc->sel_rank[2 * col + slot];
        result[slot] = s_k[r < m ? r : m - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc->sel_prefix[2 * col] = result[0];
        sc->sel_prefix[2 * col + 1] = result[1];
        if (col == 0) sc->sel_fast_done = 1;
    }
}

__global__ void sel_finish_kernel(SelArgs a, FitWork w) {
    if ((int)threadIdx.x < a.ncol) sel_finish_col(a, w.sc, threadIdx.x);
}

// workgroups per column: every workgroup ends with up to 2 x 4096 global atomics into the column's histogram, so
// more of them is not better — measured at 2 M rows: 512 for one column (MAD), 128 per column for eight (size factors)
static int sel_blocks(int64_t n, int ncol) {
    int64_t b = (n + 256 * 8 - 1) / (256 * 8);
    const int64_t cap = ncol >= 4 ? (1024 / ncol > 64 ? 1024 / ncol : 64) : 512;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
void launch_sel_hist(SelArgs a, FitWork w, hipStream_t st) {
    const int bits = sel_bits(a.shift);
    (void)hipMemsetAsync(w.hist, 0, sizeof(double) * (size_t)a.ncol * 2 * kSelBins, st);
    sel_hist_kernel<<<dim3(sel_blocks(a.n, a.ncol), a.ncol), 256, 0, st>>>(a, w, bits);
}
void launch_sel_step(SelArgs a, FitWork w, hipStream_t st) {
    const int bits = sel_bits(a.shift);
    sel_step_kernel<<<a.ncol, 256, 0, st>>>(a, w, bits);
}
void launch_sel_shortcut(SelArgs a, FitWork w, hipStream_t st) {
    sel_compact_kernel<<<dim3(sel_blocks(a.n, a.ncol), a.ncol), 256, 0, st>>>(a, w);  // (sel_cnt was zeroed by the round-2 step)
    sel_small_kernel<<<a.ncol, 256, 0, st>>>(a, w);  // sort + pick, or the fallback rounds; then the column's finish
}
void launch_sel_finish(SelArgs a, FitWork w, hipStream_t st) { sel_finish_kernel<<<1, 64, 0, st>>>(a, w); }
void launch_sel_keep_local(SelArgs a, FitWork w, hipStream_t st) {
    (void)hipMemcpyAsync(w.hist_local, w.hist, sizeof(double) * (size_t)a.ncol * 2 * kSelBins, hipMemcpyDeviceToDevice, st);
}
void launch_sel_gather_counts(SelArgs a, FitWork w, int world, int rank, hipStream_t st) {
    sel_gcount_kernel<<<1, 256, 0, st>>>(a, w, world, rank);
}
void launch_sel_gather_place(SelArgs a, FitWork w, int world, int rank, hipStream_t st) {
    (void)hipMemsetAsync(w.hist, 0, sizeof(double) * (size_t)a.ncol * 2 * kSelCap, st);
    sel_gplace_kernel<<<dim3(sel_blocks(a.n, a.ncol), a.ncol), 256, 0, st>>>(a, w, world, rank);
}
void launch_sel_gather_finish(SelArgs a, FitWork w, int world, int rank, hipStream_t st) {
    sel_gfinish_kernel<<<a.ncol, 256, 0, st>>>(a, w, world, rank);
}

// ------------------------------------------------------------------------------------------

// a5 helper: the keys of the size-factor medians, computed once: ratio[j][i] = log(counts[j][i]) -
// rowMeans(log(counts))[i] for rows without a zero count (estimateSizeFactorsForMatrix uses only rows with
// a finite log geometric mean, and only positive counts), NaN otherwise.  The select passes then stream
// 8 B per element instead of recomputing a log.
__global__ __launch_bounds__(256) void row_ratio_kernel(const int32_t *__restrict__ counts, int64_t n, int S,
                                                        double *__restrict__ ratio, int32_t *clear_flag) {
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double s = 0;
        for (int j = 0; j < S; j++) {
            const double l = log((double)counts[(int64_t)j * n + i]);  // log(0) = -inf
            ratio[(int64_t)j * n + i] = l;
            s += l;
        }
        const double lg = s / S;
        const bool use = isfinite(lg);
        for (int j = 0; j < S; j++) ratio[(int64_t)j * n + i] = use ? ratio[(int64_t)j * n + i] - lg : NAN;
    }
}
__global__ __launch_bounds__(256) void row_ratio16_kernel(const int32_t *__restrict__ counts, int64_t n, int S,
                                                          double *__restrict__ ratio, int32_t *clear_flag) {
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;  // (the select that follows may set it)
    __shared__ LogEntry s_lt[64];
    log_table_to_lds(s_lt);
    const int64_t step = (int64_t)gridDim.x * 256;
    int32_t kn[16];  // the next row's counts are loaded before the current row is worked on
    int64_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; j++) kn[j] = (j < S && i < n) ? counts[(int64_t)j * n + i] : 1;
    for (; i < n; i += step) {
        double l[16];
        int32_t kc[16];
#pragma unroll
        for (int j = 0; j < 16; j++) kc[j] = kn[j];
        if (i + step < n) {
#pragma unroll
            for (int j = 0; j < 16; j++) kn[j] = j < S ? counts[(int64_t)j * n + i + step] : 1;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int32_t k = kc[j];
            l[j] = k > 0 ? tlog((double)k, s_lt) : (k == 0 ? -INFINITY : NAN);  // log(0) = -Inf drops the row, as in R
        }
        double s = 0;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < S) s += l[j];  // sample order, as the loop above
        const double lg = s / S;
        const bool use = isfinite(lg);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < S) ratio[(int64_t)j * n + i] = use ? l[j] - lg : NAN;
    }
}
void launch_row_ratio(const int32_t *counts, int64_t n, int S, double *ratio, int32_t *clear_flag, hipStream_t st) {
    if (S <= 16) row_ratio16_kernel<<<kRedBlocks * 2, 256, 0, st>>>(counts, n, S, ratio, clear_flag);
    else row_ratio_kernel<<<kRedBlocks, 256, 0, st>>>(counts, n, S, ratio, clear_flag);
}

// a4: offsets.  One thread per row.  For S <= 16 the row lives in registers (one HBM read, one write);
// larger S re-reads the row from L1/L2.  log() keeps R's semantics for NA/0/negative inputs, the
// common positive-finite case takes the cheaper flog().
__device__ __forceinline__ double rlog(double x) { return (x > 0.0 && x < 1.7e308) ? flog(x) : log(x); }

// 2 S logarithms, 2 exponentials and 2 S quotients per row sit beside 16 S bytes of traffic: the kernel is only
// HBM-bound if that arithmetic is lean — table-driven log (devmath.h tlog, 1 KB table in LDS) and one reciprocal per
// geometric mean instead of S divisions (x * (1/g) against x / g: <= 1.5 ulp, far inside the 1e-13 the parity test asks).
__global__ __launch_bounds__(256) void offsets16_kernel(const double *__restrict__ fm, const double *__restrict__ sf,
                                                        int64_t n, int S, double theta, int mix,
                                                        double *__restrict__ out) {
    __shared__ LogEntry s_lt[64];
    log_table_to_lds(s_lt);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = j < S ? fm[(int64_t)j * n + i] : 1.0;
        offsets_row16(v, S, sf, theta, mix, s_lt);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < S) out[(int64_t)j * n + i] = v[j];
    }
}

__global__ __launch_bounds__(256) void offsets_kernel(const double *__restrict__ fm, const double *__restrict__ sf,
                                                      int64_t n, int S, double theta, int mix,
                                                      double *__restrict__ out) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double sl = 0;
        for (int j = 0; j < S; j++) sl += rlog(fm[(int64_t)j * n + i]);
        const double gmean = exp(sl / S);
        bool anyna = false;
        for (int j = 0; j < S; j++) {
            const double m3 = fm[(int64_t)j * n + i] / gmean;
            anyna |= (m3 != m3);
        }
        double g2 = 1.0;
        if (mix) {
            double sl2 = 0;
            for (int j = 0; j < S; j++) {
                const double m3 = anyna ? sf[j] : fm[(int64_t)j * n + i] / gmean;
                sl2 += rlog(m3 * (1 - theta) + sf[j] * theta);
            }
            g2 = exp(sl2 / S);
        }
        for (int j = 0; j < S; j++) {
            double m3 = anyna ? sf[j] : fm[(int64_t)j * n + i] / gmean;
            if (mix) m3 = (m3 * (1 - theta) + sf[j] * theta) / g2;
            out[(int64_t)j * n + i] = m3;
        }
    }
}
// norm = "standard" (chicdiff.R:1572-1575): the size factors themselves, one column per sample
__global__ __launch_bounds__(256) void offsets_sf_kernel(const double *__restrict__ sf, int64_t n, int S, double *__restrict__ out) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        for (int j = 0; j < S; j++) out[(int64_t)j * n + i] = sf[j];
}
void launch_offsets(const double *fm, const double *sf_dev, int64_t n, int S, double theta, int mix, double *out,
                    hipStream_t st) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (!fm) {
        offsets_sf_kernel<<<(unsigned)blocks, 256, 0, st>>>(sf_dev, n, S, out);
        return;
    }
    if (S <= 16)
        offsets16_kernel<<<(unsigned)blocks, 256, 0, st>>>(fm, sf_dev, n, S, theta, mix, out);
    else
        offsets_kernel<<<(unsigned)blocks, 256, 0, st>>>(fm, sf_dev, n, S, theta, mix, out);
}

// a2: window sums.  A block owns 256 consecutive regions of one sample; their fragments are one
// contiguous range (regions are contiguous fragment windows), which the block streams into LDS
// with coalesced loads; each thread then adds up its own window from LDS, in fragment order
// (sequential fp64 sum = the order sum() sees after setkey(otherEndID)).  Windows whose span does
// not fit the staging buffer (never for RUexpand <= 5) take the direct path.
constexpr int kWinCap = 256 * 12;  // staged fragments per block (F <= 11 for the default RUexpand = 5)
__global__ __launch_bounds__(256) void window_sums_kernel(const int32_t *__restrict__ fragN,
                                                          const double *__restrict__ fragFM, int64_t nfrag, int S,
                                                          const int64_t *__restrict__ rptr, int64_t n,
                                                          int32_t *__restrict__ N, double *__restrict__ FM) {
    __shared__ int32_t s_n[kWinCap];
    __shared__ double s_f[kWinCap];
    const int j = blockIdx.y;
    const int64_t nblk = (n + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t i0 = blk * 256, i1 = (i0 + 256 < n) ? i0 + 256 : n;
        const int64_t f0 = rptr[i0], f1 = rptr[i1];
        const int64_t i = i0 + threadIdx.x;
        const bool staged = (f1 - f0) <= kWinCap;
        int64_t lo = 0, hi = 0;
        if (i < i1) { lo = rptr[i]; hi = rptr[i + 1]; }
        if (staged) {
            __syncthreads();  // previous iteration's readers are done with the buffers
            // all 12 loads of a thread are issued before the first LDS store (memory-level parallelism)
            const int span = (int)(f1 - f0);
            if (fragN) {
                int32_t v[12];
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int e = threadIdx.x + k * 256;
                    v[k] = e < span ? fragN[(int64_t)j * nfrag + f0 + e] : 0;
                }
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int e = threadIdx.x + k * 256;
                    if (e < span) s_n[e] = v[k];
                }
            }
            if (fragFM) {
                double v[12];
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int e = threadIdx.x + k * 256;
                    v[k] = e < span ? fragFM[(int64_t)j * nfrag + f0 + e] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int e = threadIdx.x + k * 256;
                    if (e < span) s_f[e] = v[k];
                }
            }
            __syncthreads();
            if (i < i1) {
                if (fragN) {
                    int32_t s = 0;
                    for (int64_t f = lo; f < hi; f++) s += s_n[f - f0];
                    N[(int64_t)j * n + i] = s;
                }
                if (fragFM) {
                    double s = 0;
                    for (int64_t f = lo; f < hi; f++) s += s_f[f - f0];
                    FM[(int64_t)j * n + i] = s;
                }
            }
        } else if (i < i1) {
            if (fragN) {
                int32_t s = 0;
                for (int64_t f = lo; f < hi; f++) s += fragN[(int64_t)j * nfrag + f];
                N[(int64_t)j * n + i] = s;
            }
            if (fragFM) {
                double s = 0;
                for (int64_t f = lo; f < hi; f++) s += fragFM[(int64_t)j * nfrag + f];
                FM[(int64_t)j * n + i] = s;
            }
        }
    }
}
void launch_window_sums(const int32_t *fragN, const double *fragFM, int64_t nfrag, int S, const int64_t *rptr,
                        int64_t n, int32_t *N, double *FM, hipStream_t st) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;  // x S samples in grid.y: >> 256 CUs
    window_sums_kernel<<<dim3((unsigned)blocks, S), 256, 0, st>>>(fragN, fragFM, nfrag, S, rptr, n, N, FM);
}

// a1: count join.  RU is keyed by baitID (chicdiff.R:425) and a region's fragments are consecutive IDs, so the 512
// queries of a wave's tile (eight consecutive rows per lane, moved as 4 x int32) fall in a narrow key range.  Each
// wave works alone — no block barrier — and keeps the chain of DEPENDENT global loads per tile short, which is what
// bounds the kernel (measured at 22 M queries x 10 M keys: a block-wide LDS window 0.31 ms, one global binary search
// per query 0.24 ms, a per-lane register run of 16 keys 0.17 ms):
//   1. lo = lower bound of the tile's smallest query: a 64-ary search by the whole wave over the table's coarse
//      level (every 64th key, copied out once per call: 1/64 of the table, L2-resident), then one dense load of
//      the 64 keys it points at — probing the table itself every 64th key costs a 128-byte line per 8 bytes used
//      and two to three HBM round trips per tile;
//   2. one dense load of the next 64 coarse entries bounds the tile's largest query: hi
//      (a tile that spans more than 4096 keys gets a proper search instead);
//   3. a range of at most kJoinCap keys goes to the wave's LDS slice with its values (coalesced, read once) and
//      every lane resolves its eight queries there by a branch-free binary search, eight independent LDS reads per
//      step; a wider range (a far denser table, an unsorted caller) is searched in global memory the same way;
//   4. key and value at the lower bound decide match / no match (N <- 0, chicdiff.R:851-853).
// Results do not depend on the tiling or on the path taken.
// kJoinCap = 768 (round 5; 512 before): a table nearly as dense as RU (the end-to-end leg: 19 M keys under 21.5 M rows) sent every
// other tile through the global search — 0.20 ms at 22 M x 19.8 M against 0.135 with the wider window, four blocks per CU instead of
// six; a sparser table (10 M keys) is unchanged at 0.117.  Tried on the way and dropped (no gain: the tile's chain of dependent loads
// is NOT what bounds the kernel any more): runs of consecutive tiles per wave with the last tile's answer as a hint for the next.
constexpr int kJoinPerLane = 8, kJoinTile = 64 * kJoinPerLane, kJoinCap = 768;
// lower_bound over keys[lo, hi) by the 64 lanes of a wave; every lane returns the same value
__device__ __forceinline__ int64_t wave_lower_bound(const int64_t *__restrict__ keys, int64_t lo, int64_t hi, int64_t target, int lane) {
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) / 64;
        const int64_t pos = lo + (int64_t)lane * step;
        const int c = __popcll(__ballot(pos < hi && keys[pos] < target));  // monotone in the lane
        const int64_t nlo = c > 0 ? lo + (int64_t)(c - 1) * step + 1 : lo;
        const int64_t pc = lo + (int64_t)c * step;
        hi = pc < hi ? pc : hi;  // the answer is in [nlo, hi] (it may be `hi` itself)
        lo = nlo;
    }
    const int64_t pos = lo + lane;
    return lo + __popcll(__ballot(pos < hi && keys[pos] < target));
}

// One tile's 512 queries (eight per lane, q[]; kmin / kmax = the tile's smallest / largest, wave-uniform) against ONE sorted key table:
// steps 1-4 above.  sk / sv = the wave's LDS slice.  Shared by the single-table and the all-replicates kernels.
__device__ __forceinline__ void join_tile_table(const int64_t (&q)[kJoinPerLane], int64_t kmin, int64_t kmax, const int64_t *__restrict__ keys,
                                                const int32_t *__restrict__ vals, int64_t nkeys, const int64_t *__restrict__ index, int64_t nidx,
                                                int64_t *sk, int32_t *sv, int lane, int32_t (&res)[kJoinPerLane]) {
    // lo: the coarse level (every 64th key, L2-resident) says which 64 keys hold the lower bound of the tile's
    // smallest query, one dense load of those finishes; hi: the first coarse entry >= the largest query, from the
    // 64 entries after lo (one dense load), or a proper search for a tile that spans more than 4096 keys
    const int64_t jlo = wave_lower_bound(index, 0, nidx, kmin, lane);  // first j with keys[64 j] >= kmin
    int64_t lo = 0;
    if (jlo > 0) {
        const int64_t e = 64 * jlo < nkeys ? 64 * jlo : nkeys;
        lo = wave_lower_bound(keys, 64 * (jlo - 1) + 1, e, kmin, lane);
    }
    int64_t hi;
    {
        const int64_t j0 = lo / 64 + 1, pos = j0 + lane;
        const int c = __popcll(__ballot(pos < nidx && index[pos] < kmax));
        int64_t jhi = j0 + c;  // keys[64 jhi] >= kmax, or jhi >= nidx
        if (c == 64) jhi = wave_lower_bound(index, j0 + 64, nidx, kmax, lane);
        hi = jhi < nidx ? 64 * jhi + 1 : nkeys;
    }
    if (hi - lo <= kJoinCap) {
        const int w = (int)(hi - lo);
        {
            int64_t tk[kJoinCap / 64];
            int32_t tv[kJoinCap / 64];
#pragma unroll
            for (int e = 0; e < kJoinCap / 64; e++) {
                const int at = e * 64 + lane;
                tk[e] = at < w ? keys[lo + at] : 0;
                tv[e] = at < w ? vals[lo + at] : 0;
            }
#pragma unroll
            for (int e = 0; e < kJoinCap / 64; e++) {
                sk[e * 64 + lane] = tk[e];
                sv[e * 64 + lane] = tv[e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int b[kJoinPerLane];  // the answer of query k stays in [b[k], b[k] + len]
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) b[k] = 0;
        int len = w;
        while (len > 1) {
            const int half = len >> 1;
#pragma unroll
            for (int k = 0; k < kJoinPerLane; k++) b[k] += sk[b[k] + half - 1] < q[k] ? half : 0;
            len -= half;
        }
        if (len == 1) {
#pragma unroll
            for (int k = 0; k < kJoinPerLane; k++) b[k] += sk[b[k]] < q[k] ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) {
            const bool in = b[k] < w;
            const int at = in ? b[k] : 0;
            res[k] = (in && sk[at] == q[k]) ? sv[at] : 0;
        }
        __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next tile
    } else {
        int64_t b[kJoinPerLane];
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) b[k] = lo;
        int64_t len = hi - lo;
        while (len > 1) {
            const int64_t half = len >> 1;
#pragma unroll
            for (int k = 0; k < kJoinPerLane; k++) b[k] += keys[b[k] + half - 1] < q[k] ? half : 0;
            len -= half;
        }
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) b[k] += keys[b[k]] < q[k] ? 1 : 0;  // len == 1: the range has > kJoinCap keys
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) {
            const bool in = b[k] < hi;
            const int64_t at = in ? b[k] : lo;
            res[k] = (in && keys[at] == q[k]) ? vals[at] : 0;
        }
    }
}
// a tile's queries: eight consecutive RU rows per lane -> q[], and the tile's smallest / largest key (wave-uniform)
template <bool VEC>
__device__ __forceinline__ void join_tile_queries(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t nru, int64_t r0,
                                                  int64_t (&q)[kJoinPerLane], int64_t &kmin, int64_t &kmax) {
    int32_t qb[kJoinPerLane], qo[kJoinPerLane];
    if (VEC) {
        const int4 b0 = *(const int4 *)(bait + r0), b1 = *(const int4 *)(bait + r0 + 4);
        const int4 o0 = *(const int4 *)(oe + r0), o1 = *(const int4 *)(oe + r0 + 4);
        qb[0] = b0.x; qb[1] = b0.y; qb[2] = b0.z; qb[3] = b0.w; qb[4] = b1.x; qb[5] = b1.y; qb[6] = b1.z; qb[7] = b1.w;
        qo[0] = o0.x; qo[1] = o0.y; qo[2] = o0.z; qo[3] = o0.w; qo[4] = o1.x; qo[5] = o1.y; qo[6] = o1.z; qo[7] = o1.w;
    } else {
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++) {
            const bool v = r0 + k < nru;
            qb[k] = v ? bait[r0 + k] : 0;
            qo[k] = v ? oe[r0 + k] : 0;
        }
    }
    kmin = INT64_MAX;
    kmax = INT64_MIN;
#pragma unroll
    for (int k = 0; k < kJoinPerLane; k++) {
        q[k] = ((int64_t)qb[k] << 32) | (uint32_t)qo[k];
        if (VEC || r0 + k < nru) {
            kmin = q[k] < kmin ? q[k] : kmin;
            kmax = q[k] > kmax ? q[k] : kmax;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t a = __shfl_xor(kmin, off), c = __shfl_xor(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
}
template <bool VEC>
__device__ __forceinline__ void join_tile_store(int32_t *__restrict__ out, int64_t nru, int64_t r0, const int32_t (&res)[kJoinPerLane]) {
    if (VEC) {
        *(int4 *)(out + r0) = make_int4(res[0], res[1], res[2], res[3]);
        *(int4 *)(out + r0 + 4) = make_int4(res[4], res[5], res[6], res[7]);
    } else {
#pragma unroll
        for (int k = 0; k < kJoinPerLane; k++)
            if (r0 + k < nru) out[r0 + k] = res[k];
    }
}

// A full tile's 512 results to a column that is only 4-byte aligned (column t of an S x nru matrix starts at t * nru: 16-byte aligned
// for one nru in four).  The tile is one contiguous run of 2 KB; with k = the number of leading elements up to the next 16-byte
// boundary (wave-uniform: r0 is a multiple of 8), lane L stores the ALIGNED eight elements [8 L + k, 8 L + k + 8) — its own res[k..7]
// and the next lane's res[0..k-1] — as 2 x int4; lane 0 adds the tile's first k elements, lane 63 keeps its last 8 - k, as dwords.
__device__ __forceinline__ void join_tile_store_shifted(int32_t *__restrict__ col, int64_t r0, int lane, const int32_t (&res)[kJoinPerLane]) {
    const int k = (int)((4u - (unsigned)(((uintptr_t)(col + r0)) >> 2)) & 3u);  // same in every lane
    if (k == 0) {
        *(int4 *)(col + r0) = make_int4(res[0], res[1], res[2], res[3]);
        *(int4 *)(col + r0 + 4) = make_int4(res[4], res[5], res[6], res[7]);
        return;
    }
    const int n0 = __shfl_down(res[0], 1), n1 = __shfl_down(res[1], 1), n2 = __shfl_down(res[2], 1);
    int32_t v[kJoinPerLane];
    if (k == 1) { v[0] = res[1]; v[1] = res[2]; v[2] = res[3]; v[3] = res[4]; v[4] = res[5]; v[5] = res[6]; v[6] = res[7]; v[7] = n0; }
    else if (k == 2) { v[0] = res[2]; v[1] = res[3]; v[2] = res[4]; v[3] = res[5]; v[4] = res[6]; v[5] = res[7]; v[6] = n0; v[7] = n1; }
    else { v[0] = res[3]; v[1] = res[4]; v[2] = res[5]; v[3] = res[6]; v[4] = res[7]; v[5] = n0; v[6] = n1; v[7] = n2; }
    int32_t *dst = col + r0 + k;
    if (lane < 63) {
        *(int4 *)dst = make_int4(v[0], v[1], v[2], v[3]);
        *(int4 *)(dst + 4) = make_int4(v[4], v[5], v[6], v[7]);
    } else {
        for (int i = 0; i < kJoinPerLane - k; i++) dst[i] = v[i];
    }
    if (lane == 0)
        for (int i = 0; i < k; i++) col[r0 + i] = res[i];
}

// (gfx950: 160 KB of LDS per CU — the four waves' slices take 4 x 768 x 12 = 36 864 bytes per workgroup, so FOUR workgroups = 16 waves
// are resident per CU, which is what the launch bound asks the register allocator for; with the 512-key window it was six)
template <bool VEC>  // VEC: bait / oe / out are 16-byte aligned, full tiles move as 4 x int32
__global__ __launch_bounds__(256, 4) void count_join_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe,
                                                         int64_t nru, const int64_t *__restrict__ keys,
                                                         const int32_t *__restrict__ vals, int64_t nkeys,
                                                         const int64_t *__restrict__ index, int64_t nidx,
                                                         int32_t *__restrict__ out) {
    __shared__ int64_t s_keys[4][kJoinCap];
    __shared__ int32_t s_vals[4][kJoinCap];
    const int lane = threadIdx.x & 63;
    int64_t *const sk = s_keys[threadIdx.x >> 6];
    int32_t *const sv = s_vals[threadIdx.x >> 6];
    // VEC launches cover the full tiles only; the ragged end (and unaligned callers) go through the scalar variant
    const int64_t ntile = VEC ? nru / kJoinTile : (nru + kJoinTile - 1) / kJoinTile;
    // blocks b and b + 8 share an XCD (round-robin dispatch; speed only): each of the eight L2s serves one contiguous
    // eighth of the tiles, so the sparse probes of neighbouring tiles (every 64th key: 128-byte lines of which 8
    // bytes are used) hit lines a neighbour has just brought in instead of going to HBM again
    const int64_t per_xcd = (ntile + 7) / 8, t_begin = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t t_end = t_begin + per_xcd < ntile ? t_begin + per_xcd : ntile;
    const int64_t wave0 = (int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), nwave = (int64_t)(gridDim.x >> 3) * 4;
    for (int64_t tile = t_begin + wave0; tile < t_end; tile += nwave) {
        const int64_t r0 = tile * kJoinTile + (int64_t)lane * kJoinPerLane;
        int64_t q[kJoinPerLane], kmin, kmax;
        join_tile_queries<VEC>(bait, oe, nru, r0, q, kmin, kmax);
        int32_t res[kJoinPerLane];
        join_tile_table(q, kmin, kmax, keys, vals, nkeys, index, nidx, sk, sv, lane, res);
        join_tile_store<VEC>(out, nru, r0, res);
    }
}

// a1 for ALL replicates in one pass (round 6): chicdiff.R:843-858 is a loop over the replicates, each merge() reading the same RU
// rows; here a tile's (baitID, otherEndID) pairs are read ONCE and resolved against every replicate's key table in turn — per set of
// S joins 8 nru + S (4 nru + 12 nkeys) bytes instead of S (12 nru + 12 nkeys) (S = 8, 21.5 M rows, 19 M keys: 2.7 instead of 3.9 GB),
// one launch instead of S, and the coarse levels of all tables built by one launch.  Same per-table steps (join_tile_table): same bits.
constexpr int kJoinMaxTables = 16;  // per launch (the tables' pointers travel as kernel arguments); more replicates: several launches
struct JoinTables {
    const int64_t *keys[kJoinMaxTables];
    const int32_t *vals[kJoinMaxTables];
    const int64_t *index[kJoinMaxTables];
    int64_t nkeys[kJoinMaxTables];
    int32_t T;
};
template <bool VEC>
__global__ __launch_bounds__(256, 4) void count_join_multi_kernel(const int32_t *__restrict__ bait, const int32_t *__restrict__ oe, int64_t nru,
                                                                  JoinTables tb, int32_t *__restrict__ out, int64_t out_stride) {
    __shared__ int64_t s_keys[4][kJoinCap];
    __shared__ int32_t s_vals[4][kJoinCap];
    const int lane = threadIdx.x & 63;
    int64_t *const sk = s_keys[threadIdx.x >> 6];
    int32_t *const sv = s_vals[threadIdx.x >> 6];
    const int64_t ntile = VEC ? nru / kJoinTile : (nru + kJoinTile - 1) / kJoinTile;
    const int64_t per_xcd = (ntile + 7) / 8, t_begin = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t t_end = t_begin + per_xcd < ntile ? t_begin + per_xcd : ntile;
    const int64_t wave0 = (int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), nwave = (int64_t)(gridDim.x >> 3) * 4;
    for (int64_t tile = t_begin + wave0; tile < t_end; tile += nwave) {
        const int64_t r0 = tile * kJoinTile + (int64_t)lane * kJoinPerLane;
        int64_t q[kJoinPerLane], kmin, kmax;
        join_tile_queries<VEC>(bait, oe, nru, r0, q, kmin, kmax);
#pragma unroll 1
        for (int t = 0; t < tb.T; t++) {
            int32_t res[kJoinPerLane];
            const int64_t nk = tb.nkeys[t];
            join_tile_table(q, kmin, kmax, tb.keys[t], tb.vals[t], nk, tb.index[t], (nk + 63) / 64, sk, sv, lane, res);
            int32_t *col = out + (int64_t)t * out_stride;
            if (VEC) join_tile_store_shifted(col, r0, lane, res);
            else join_tile_store<false>(col, nru, r0, res);
        }
    }
}
__global__ __launch_bounds__(256) void join_index_kernel(const int64_t *__restrict__ keys, int64_t nidx, int64_t *__restrict__ index) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < nidx) index[j] = keys[64 * j];
}
size_t count_join_scratch_bytes(int64_t nkeys) { return sizeof(int64_t) * (size_t)((nkeys + 63) / 64 + 1); }
// scratch: count_join_scratch_bytes(nkeys) bytes
void launch_count_join(const int32_t *bait, const int32_t *oe, int64_t nru, const int64_t *keys, const int32_t *vals,
                       int64_t nkeys, int32_t *out, void *scratch, hipStream_t st) {
    auto grid = [](int64_t rows) {  // a multiple of 8: one share of the tiles per XCD
        int64_t blocks = (((rows + kJoinTile - 1) / kJoinTile + 3) / 4 + 7) / 8 * 8;
        return (unsigned)(blocks > 8192 ? 8192 : (blocks < 8 ? 8 : blocks));
    };
    int64_t *index = (int64_t *)scratch;
    const int64_t nidx = (nkeys + 63) / 64;
    if (nidx > 0) join_index_kernel<<<(unsigned)((nidx + 255) / 256), 256, 0, st>>>(keys, nidx, index);
    const bool vec = (((uintptr_t)bait | (uintptr_t)oe | (uintptr_t)out) & 15) == 0;
    const int64_t body = vec ? nru / kJoinTile * kJoinTile : 0;
    if (body > 0) count_join_kernel<true><<<grid(body), 256, 0, st>>>(bait, oe, body, keys, vals, nkeys, index, nidx, out);
    if (nru > body)
        count_join_kernel<false><<<grid(nru - body), 256, 0, st>>>(bait + body, oe + body, nru - body, keys, vals, nkeys, index, nidx, out + body);
}

// all tables' coarse levels in one launch: grid.y = table
__global__ __launch_bounds__(256) void join_index_multi_kernel(JoinTables tb) {
    const int t = blockIdx.y;
    const int64_t nidx = (tb.nkeys[t] + 63) / 64;
    int64_t *index = const_cast<int64_t *>(tb.index[t]);
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nidx; j += (int64_t)gridDim.x * 256) index[j] = tb.keys[t][64 * j];
}
size_t count_join_multi_scratch_bytes(int S, const int64_t *nkeys) {
    size_t b = 0;
    for (int s = 0; s < S; s++) b += count_join_scratch_bytes(nkeys[s]);
    return b;
}
// out: S x nru (replicate-major); keys / vals / nkeys: host arrays of S entries (device pointers inside); scratch: count_join_multi_scratch_bytes
void launch_count_join_multi(const int32_t *bait, const int32_t *oe, int64_t nru, int S, const int64_t *const *keys, const int32_t *const *vals,
                             const int64_t *nkeys, int32_t *out, void *scratch, hipStream_t st) {
    auto grid = [](int64_t rows) {
        int64_t blocks = (((rows + kJoinTile - 1) / kJoinTile + 3) / 4 + 7) / 8 * 8;
        return (unsigned)(blocks > 8192 ? 8192 : (blocks < 8 ? 8 : blocks));
    };
    int64_t *index = (int64_t *)scratch;
    for (int s0 = 0; s0 < S; s0 += kJoinMaxTables) {
        JoinTables tb{};
        tb.T = S - s0 < kJoinMaxTables ? S - s0 : kJoinMaxTables;
        int64_t maxidx = 0;
        for (int t = 0; t < tb.T; t++) {
            tb.keys[t] = keys[s0 + t];
            tb.vals[t] = vals[s0 + t];
            tb.nkeys[t] = nkeys[s0 + t];
            tb.index[t] = index;
            const int64_t nidx = (nkeys[s0 + t] + 63) / 64;
            index += nidx + 1;
            maxidx = nidx > maxidx ? nidx : maxidx;
        }
        if (maxidx > 0) {
            int64_t bx = (maxidx + 255) / 256;
            join_index_multi_kernel<<<dim3((unsigned)(bx > 1024 ? 1024 : bx), tb.T), 256, 0, st>>>(tb);
        }
        int32_t *o = out + (int64_t)s0 * nru;
        const bool vec = (((uintptr_t)bait | (uintptr_t)oe) & 15) == 0;  // (the columns of `out` need not be: join_tile_store_shifted)
        const int64_t body = vec ? nru / kJoinTile * kJoinTile : 0;
        if (body > 0) count_join_multi_kernel<true><<<grid(body), 256, 0, st>>>(bait, oe, body, tb, o, nru);
        if (nru > body) count_join_multi_kernel<false><<<grid(nru - body), 256, 0, st>>>(bait + body, oe + body, nru - body, tb, o + body, nru);
    }
}

__global__ __launch_bounds__(256) void pvalue_kernel(const double *__restrict__ stat, int64_t n, double *__restrict__ p) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = pnorm_two_sided(stat[i]);
}
void launch_pvalues(const double *stat, int64_t n, double *p, hipStream_t st) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    pvalue_kernel<<<(unsigned)blocks, 256, 0, st>>>(stat, n, p);
}

// a3: per-fragment background (chicdiff.R:628-703 + Chicago .estimateBMean/.distFun): a gather from
// dense per-fragment tables plus one log and one exp per (RU row, replicate).  HBM/L2-gather bound.
struct BgArgs {
    const int32_t *bait, *oe;
    int64_t nru;
    int32_t id_min, nid, S, ntblb, ntlb;
    const int64_t *midsum;
    const double *sj, *si;
    const int32_t *tblb, *tlb;
    const double *T;
    const double *distfun;  // device, S x 10
    double *bmean, *tmean, *fullmean;
};
// What bounds it (round 5, 22 M rows x 8 replicates, FullMean only: 0.95 ms = 1.7 TB/s of algorithmic bytes; profiles/r05_fragment_background_ablation.txt):
// instruction issue, not memory — 830 vector + 420 scalar instructions per 64 rows (SQ_INSTS_VALU / SQ_INSTS_SALU), HBM traffic 0.53 GB
// read + 1.375 GB written (as computed: every XCD reads the tables once).  Without the gathers 0.76 ms, without the stores 0.74, without
// either and without the midpoints 0.58.  Measured and NOT taken, each within +-2 % of the kernel below: the gathers of four replicates
// issued together, table-driven exp / log (devmath.h) in place of the device library's, the Tmean tables in LDS.
// One thread per RU row, the replicates in a loop (round 5; before: one thread per (row, replicate), grid.y = S): the row's two
// fragment IDs, the two midpoints and log|distance| are the same for every replicate — S - 1 of every S logarithms, ID loads and
// midpoint gathers were repeats — and a caller that wants FullMean alone (chicdiff.R:896: the one column DESeq2Wrap reads) passes
// NULL for the other two and saves two thirds of the stores.  Same expressions in the same order: same bits.
__global__ __launch_bounds__(256) void fragment_background_kernel(BgArgs a) {
    extern __shared__ double s_df[];  // S x 10: the replicates' distance functions
    for (int k = threadIdx.x; k < 10 * a.S; k += 256) s_df[k] = a.distfun[k];
    __syncthreads();
    for (int64_t r = blockIdx.x * 256 + threadIdx.x; r < a.nru; r += (int64_t)gridDim.x * 256) {
        const int32_t b = a.bait[r] - a.id_min, o = a.oe[r] - a.id_min;
        const bool on_map = b >= 0 && b < a.nid && o >= 0 && o < a.nid;
        double ld = 0.0;
        if (on_map) {
            const double dist = rint((double)(a.midsum[o] - a.midsum[b]) / 2.0);  // R round(): half to even
            ld = log(fabs(dist));
        }
        for (int s = 0; s < a.S; s++) {
            double B = NAN, Tm = NAN;
            if (on_map) {
                const double *p = s_df + 10 * s;
                const double s_j = a.sj[(int64_t)s * a.nid + b];
                double s_i = a.si[(int64_t)s * a.nid + o];
                if (s_i != s_i) s_i = 1.0;
                double e;
                if (ld > p[9]) e = p[6] + ld * p[7];
                else if (ld < p[8]) e = p[4] + ld * p[5];
                else e = p[0] + p[1] * ld + p[2] * (ld * ld) + p[3] * (ld * ld * ld);
                B = s_j * s_i * exp(e);
                const int32_t tb = a.tblb[(int64_t)s * a.nid + b], tl = a.tlb[(int64_t)s * a.nid + o];
                if (tb >= 0 && tl >= 0) {
                    Tm = a.T[((int64_t)s * a.ntblb + tb) * a.ntlb + tl];
                } else if (tb >= 0) {
                    double m = INFINITY;
                    for (int k = 0; k < a.ntlb; k++) {
                        const double v = a.T[((int64_t)s * a.ntblb + tb) * a.ntlb + k];
                        if (v == v && v < m) m = v;
                    }
                    Tm = isfinite(m) ? m : NAN;
                }
            }
            if (a.bmean) a.bmean[(int64_t)s * a.nru + r] = B;
            if (a.tmean) a.tmean[(int64_t)s * a.nru + r] = Tm;
            if (a.fullmean) a.fullmean[(int64_t)s * a.nru + r] = B + Tm;
        }
    }
}
void launch_fragment_background(const int32_t *bait, const int32_t *oe, int64_t nru, int32_t id_min, int32_t nid,
                                const int64_t *midsum, int32_t S, const double *sj, const double *si, const int32_t *tblb,
                                const int32_t *tlb, const double *T, int32_t ntblb, int32_t ntlb, const double *distfun_dev,
                                double *bmean, double *tmean, double *fullmean, hipStream_t st) {
    BgArgs a{bait, oe, nru, id_min, nid, S, ntblb, ntlb, midsum, sj, si, tblb, tlb, T, distfun_dev, bmean, tmean, fullmean};
    int64_t blocks = (nru + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    fragment_background_kernel<<<(unsigned)blocks, 256, sizeof(double) * 10 * S, st>>>(a);
}

// device-math self test (tests/test_gpu_parity.py::test_device_math): out[i] = f_op(x[i])
__global__ __launch_bounds__(256) void math_selftest_kernel(int op, const double *__restrict__ x, int64_t n,
                                                            double *__restrict__ out) {
    __shared__ LogEntry s_logtab[64];
    __shared__ ExpEntry s_exptab[64];
    exp_table_to_lds(s_exptab);
    log_table_to_lds(s_logtab);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i];
        double r = NAN, t;
        switch (op) {
            case 0: r = flog(v); break;
            case 1: r = tlog(v, s_logtab); break;
            case 2: r = rcp(v); break;
            case 3: r = lgamma_pos(v); break;
            case 4: lgamma_digamma(v, t, r); break;
            case 5: r = pnorm_two_sided(v); break;
            case 6: r = __builtin_amdgcn_rcp(v); break;  // raw v_rcp_f64 (~25 bits)
            case 7: { double q = __builtin_amdgcn_rcp(v); r = fma(q, fma(-v, q, 1.0), q); } break;  // + 1 Newton step
            case 8: r = texp(v, s_exptab); break;
        }
        out[i] = r;
    }
}
void launch_math_selftest(int op, const double *x, int64_t n, double *out, hipStream_t st) {
    math_selftest_kernel<<<256, 256, 0, st>>>(op, x, n, out);
}

}  // namespace cd
