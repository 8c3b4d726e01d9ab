// wald_kernels.hip — NB-GLM ridge IRLS, Wald statistics, deviance and Cook's distances.
//
// Replaces DESeq2's nbinomWaldTest -> fitNbinomGLMs (R) -> fitBeta (C++), reached from
// chicdiff.R:1574, :1603, :1644, :1674.  For the two designs Chicdiff
// builds (~condition with two levels; ~1) the (S+2)x2 QR solve of fitBeta's ridge-augmented
// system collapses to a closed-form 2x2 solve (det(X'WX) = sum_A w * sum_B w), so there is no
// dense contraction and no MFMA: three kernels,
//   wald_prep   (row per thread) : beta start = LS of log(q+0.1); the mu-independent parts of the
//                                   NB log-likelihood (log y! from a table), so that an IRLS tick
//                                   needs one log1p and one reciprocal per sample, no log(mu), no 1/mu;
//   wald_irls   (row per lane + queue refill, as disp_fit_kernel) : IRLS ticks;
//                 the few rows it leaves unconverged go through the optim fallback on the spot (optim_row);
//   wald_final  (row per thread) : sandwich SE, stat, p (Cody), deviance, hat diagonals -> max
//                                   Cook's distance (trimmed cell variances by rank counting in LDS).
#include "common.h"
#include "devmath.h"

namespace cd {

constexpr double kLog2e = 1.4426950408889634074;

// tile > 0 (a fit of at most one row per thread, schedule on; round 6): workgroup b owns rows [b tile, (b + 1) tile) and leaves its class
// counts in hist — what order_hist_kernel did in a launch of its own behind this one (~5 us of a small fit's step); tile == 0: grid-strided
__global__ __launch_bounds__(256) void wald_prep_kernel(const int32_t *__restrict__ counts,
                                                        const double *__restrict__ nf, FitDims d, FitWork w, int sched, int64_t tile, unsigned int *hist) {
    __shared__ double s_logfact[kLogFactN];  // log(y!) for ordinary counts: a look-up instead of a Stirling difference
    __shared__ LogEntry s_lt[64];
    for (int k = threadIdx.x; k < kLogFactN; k += 256) s_logfact[k] = w.logfact[k];
    log_table_to_lds(s_lt);  // (ends with the barrier)
    const int64_t n = d.n;
    const int S = d.S;
    unsigned int mine[kSchedClasses] = {0, 0, 0, 0, 0, 0};
    const int64_t lo = tile > 0 ? (int64_t)blockIdx.x * tile : (int64_t)blockIdx.x * 256;
    const int64_t hi = tile > 0 ? (lo + tile < n ? lo + tile : n) : n;
    const int64_t step = tile > 0 ? 256 : (int64_t)gridDim.x * 256;
    for (int64_t i = lo + threadIdx.x; i < hi; i += step) {
        if (w.allZero[i]) {
            reinterpret_cast<double2 *>(w.start)[2 * i] = make_double2(NAN, 0.0);  // alpha = NaN: how the IRLS learns that the row is all zero
            if (sched) {  // not scheduled (order_*): the IRLS kernel never sees the row
                w.cls[i] = 255;
                w.beta0[i] = NAN;
                w.beta1[i] = NAN;
                w.betaIter[i] = 0;
            }
            continue;
        }
        const double alpha = w.disp[i], la = flog(alpha);
        if (sched) {
            // Schedule of the IRLS: rows converge in 3-5 steps unless a group's mean is tiny or the dispersion huge (400 k
            // synthetic rows: `smaller group mean < 2` is 10 % of the rows and holds every row with >= 30 steps and 98.8 % of
            // those with >= 20; the rest have alpha > 1.5).  Those rows go first, so that a 20-100 step row never starts late.
            const double gmin = fmin(w.gm0[i], w.gm1[i]);
            const int cl = gmin < 0.5 ? 0 : (gmin < 1.0 ? 1 : ((gmin < 2.0 || alpha > 1.5) ? 2 : 3));
            w.cls[i] = (uint8_t)cl;
#pragma unroll
            for (int k = 0; k < kSchedClasses; k++) mine[k] += (cl == k);
        }
        const LgrCtx cs = lgr_make_t(rcp(alpha), s_lt), c1 = lgr_one();
        double lA = 0, lB = 0, c = 0, cst = 0;
        int yi_next = counts[i];
        double nf_next = nf[i];
        for (int j = 0; j < S; j++) {
            const int yi = yi_next;
            const double nfj = nf_next;
            if (j + 1 < S) {  // the next sample's loads are in flight while this one is worked on
                yi_next = counts[(int64_t)(j + 1) * n + i];
                nf_next = nf[(int64_t)(j + 1) * n + i];
            }
            const double l = tlog((double)yi / nfj + 0.1, s_lt);
            if ((d.gmask >> j) & 1) lB += l; else lA += l;
            if (yi > 0) {
                // lgamma(y+size) - lgamma(size) - lgamma(y+1): the mu-independent part of log dnbinom
                c += lgr_eval_t(cs, yi, s_lt) - (yi < kLogFactN ? s_logfact[yi] : lgr_eval(c1, yi));
                // sum_j y_j (log alpha + log nf_j): with mu = nf e^eta the rest of sum_j y_j log(alpha mu_j) is
                // eta_A sum_A y + eta_B sum_B y, so the IRLS ticks need no log(mu)
                cst = fma((double)yi, la + tlog(nfj, s_lt), cst);
            }
        }
        lA /= d.nA;
        lB /= d.nB;
        w.binit0[i] = lA;
        w.binit1[i] = lB - lA;
        w.crow[i] = c;
        double2 *h = reinterpret_cast<double2 *>(w.start) + 2 * i;  // what the IRLS reads with the row
        h[0] = make_double2(alpha, c + cst);
        h[1] = make_double2(lA, lB - lA);
    }
    if (tile > 0) order_hist_store(mine, hist);
}


// Fallback for rows whose IRLS diverged or ran out of iterations.  DESeq2 hands them to
// optim(L-BFGS-B, bounds +-30) on the log2-scale negative log posterior (fitNbinomGLMsOptim); what is
// reproduced here is that optimiser's target — the posterior mode inside the box — by damped Fisher
// scoring with backtracking on the same objective, started from the least-squares start values.
// A handful of rows per million (single extreme count outliers): the lane that holds the row in the IRLS kernel
// does it on the spot (the row is in its LDS column; the wave's other lanes wait the few tens of microseconds).
// counts / offsets of the row are read as y[j * stride], f[j * stride] (global memory, or the LDS copy below)
__device__ __forceinline__ double optim_objective(const int32_t *y_, const double *f_, int64_t stride, int S, uint64_t gmask,
                                                  double alpha, double size, double la, double crow, double lam, double b0,
                                                  double b1) {
    double f = 0.5 * lam * (b0 * b0 + b1 * b1) - crow;
    const double E0 = exp(b0), E1 = exp(b0 + b1);
    for (int j = 0; j < S; j++) {
        const double y = (double)y_[j * stride];
        const double mu = f_[j * stride] * (((gmask >> j) & 1) ? E1 : E0);
        const double ma = alpha * mu, t = 1.0 + ma;
        f += (size + y) * flog1p_from(ma, t, rcp(t));
        if (y > 0) f -= y * (la + flog(mu));
    }
    return f;
}
// returns true when the mode was reached (DESeq2: optim's convergence code 0)
__device__ __noinline__ bool optim_row(const int32_t *y_, const double *f_, int64_t stride, int S, uint64_t gmask, double alpha, double crow,
                                       double &b0, double &b1) {
    const double lam = 1e-6 / (0.69314718055994530942 * 0.69314718055994530942);
    const double bound = 30.0 * 0.69314718055994530942;
    const double size = rcp(alpha), la = flog(alpha);
    double f = optim_objective(y_, f_, stride, S, gmask, alpha, size, la, crow, lam, b0, b1);
    bool converged = false;
    for (int it = 0; it < 200 && !converged; it++) {
        double g0 = lam * b0, g1 = lam * b1, wA = 0, wB = 0;
        const double E0 = exp(b0), E1 = exp(b0 + b1);
        for (int j = 0; j < S; j++) {
            const bool g = (gmask >> j) & 1;
            const double y = (double)y_[j * stride];
            const double mu = f_[j * stride] * (g ? E1 : E0);
            const double rt = rcp(fma(alpha, mu, 1.0));
            const double sc = (y - mu) * rt, wj = mu * rt;
            g0 -= sc;
            if (g) { g1 -= sc; wB += wj; } else wA += wj;
        }
        const double m00 = wA + wB + lam, m01 = wB, m11 = wB + lam, det = m00 * m11 - m01 * m01;
        const double d0 = -(m11 * g0 - m01 * g1) / det, d1 = -(m00 * g1 - m01 * g0) / det;
        double t = 1.0;
        bool moved = false;
        for (int h = 0; h < 40; h++, t *= 0.5) {
            const double n0 = fmin(fmax(b0 + t * d0, -bound), bound), n1 = fmin(fmax(b1 + t * d1, -bound), bound);
            const double fn = optim_objective(y_, f_, stride, S, gmask, alpha, size, la, crow, lam, n0, n1);
            if (fn < f) {
                if (f - fn < 1e-13 * (fabs(f) + 1.0)) converged = true;
                b0 = n0; b1 = n1; f = fn; moved = true;
                break;
            }
        }
        if (!moved) converged = true;
    }
    return converged;
}

// One row of FitWork::rowpack -> the lane's LDS column for the IRLS (offsets as they are), with the two groups' count sums formed
// on the way.  As load_row_mu() of disp_kernels.hip: every 16-byte load of the record in flight before the first is used; whether the
// row is all zero comes with the start values (alpha = NaN, wald_prep_kernel).
template <int Q>  // S = 4 Q
__device__ __forceinline__ void load_row_sums_fixed(const char *row, double *s_nf, int *s_y, int lane, uint64_t gmask, int &iyA, int &iyB) {
    const double2 *p = reinterpret_cast<const double2 *>(row);
    const int4 *py = reinterpret_cast<const int4 *>(row + kRowHdr + 32 * Q);
    double2 f[2 * Q];
    int4 y[Q];
#pragma unroll
    for (int k = 0; k < 2 * Q; k++) f[k] = p[2 + k];
#pragma unroll
    for (int k = 0; k < Q; k++) y[k] = py[k];
    uint32_t gbits = (uint32_t)gmask;
    asm volatile("" : "+s"(gbits));  // (a copy the compiler cannot see through: it would build all S lane masks outside the main loop)
#pragma unroll
    for (int k = 0; k < 2 * Q; k++) {
        s_nf[(2 * k) * 64 + lane] = f[k].x;
        s_nf[(2 * k + 1) * 64 + lane] = f[k].y;
    }
    int all = 0, b = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int v[4] = {y[k].x, y[k].y, y[k].z, y[k].w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            s_y[(4 * k + t) * 64 + lane] = v[t];
            all += v[t];
            b += ((gbits >> (4 * k + t)) & 1u) ? v[t] : 0;
        }
    }
    iyB = b;
    iyA = all - b;
}
__device__ __forceinline__ void load_row_sums(const char *row, int S, double *s_nf, int *s_y, int lane, uint64_t gmask, int &iyA, int &iyB) {
    if (S == 8) return load_row_sums_fixed<2>(row, s_nf, s_y, lane, gmask, iyA, iyB);
    if (S == 4) return load_row_sums_fixed<1>(row, s_nf, s_y, lane, gmask, iyA, iyB);
    if (S == 16) return load_row_sums_fixed<4>(row, s_nf, s_y, lane, gmask, iyA, iyB);
    if (S == 12) return load_row_sums_fixed<3>(row, s_nf, s_y, lane, gmask, iyA, iyB);
    const double *pf = reinterpret_cast<const double *>(row + kRowHdr);
    const int *py = reinterpret_cast<const int *>(row + kRowHdr + 8 * S);
    int a = 0, b = 0;
    for (int j = 0; j < S; j++) {
        const int yi = py[j];
        s_nf[j * 64 + lane] = pf[j];
        s_y[j * 64 + lane] = yi;
        if ((gmask >> j) & 1) b += yi; else a += yi;
    }
    iyA = a;
    iyB = b;
}

#ifdef CHICDIFF_DIAG
#define WDIAG(...) __VA_ARGS__
__device__ unsigned long long g_irls_cy[8], g_irls_open[2];  // s_memtime cycles summed over all waves: refill / evaluate / rest / ticks, while the queue has rows and after
#else
#define WDIAG(...)
#endif
struct WaldArgs {
    const int32_t *counts;
    const double *nf;
    FitDims d;
    FitWork w;
    Opts o;
    int chunk;  // rows a wave takes from the global queue per atomic
    const int32_t *order;  // schedule (slow rows first, wald_prep_kernel + order_*), NULL = rows 0..n-1
    int spread;            // 0 = row-per-lane ticks only (option "line_search_spread", for the bit-identity test)
};

// IRLS.  Tick k evaluates at beta_k: deviance(beta_k) for the convergence test and the
// weighted sums that give beta_{k+1}.  DESeq2's loop index t equals k-1.
// Three waves per SIMD: at four (128 VGPRs) the kernel spills 44 VGPRs to scratch and, once the row header is read with the row, is
// slower (0.50 ms at 2 M x 8 against 0.43 at three and 0.44 at two; 250 k x 8: 0.25 / 0.25 / 0.23).
#ifndef WALD_MINW
#define WALD_MINW 3
#endif
__global__ __launch_bounds__(256, WALD_MINW) void wald_irls_kernel(WaldArgs A) {
    extern __shared__ double smem[];
    __shared__ LogEntry s_logtab[64];
    __shared__ ExpEntry s_exptab[64];
    exp_table_to_lds(s_exptab);
    log_table_to_lds(s_logtab);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform, and the compiler knows)
    const int S = A.d.S;
    const int64_t n = A.d.n;
    double *s_nf = smem + (size_t)wave * S * 96;
    int *s_y = reinterpret_cast<int *>(s_nf + S * 64);
    const uint64_t gmask = A.d.gmask;
    const Opts o = A.o;
    const double lambda = 1e-6 / (0.69314718055994530942 * 0.69314718055994530942);
    const int32_t *__restrict__ order = A.order;
    const uint32_t nTot = order ? (uint32_t)A.w.sc->ord_n : (uint32_t)n;  // (n < 2^31: 32-bit positions keep the bookkeeping on the scalar unit)
    // Eight queue heads, 64 bytes apart: chunk c = 8 k + h is the k-th chunk taken from head h.  One head is one hot word
    // (~90 dequeues per us for the whole GPU) and an IRLS row lasts ~6 us, which had forced chunks of 128 rows; eight heads
    // take 64-row chunks without queueing up, and the waves' ends balance better.  A wave starts at the head of its XCD
    // (workgroups are dealt round-robin) and moves on when a head runs dry; the schedule's slow-first order holds up to that interleaving.
    unsigned long long *heads = A.w.queue + 32;
    int cur_head = blockIdx.x & 7;
    unsigned int heads_left = 0xffu;
    const uint32_t nchunks = (nTot + (uint32_t)A.chunk - 1u) / (uint32_t)A.chunk;
    const int64_t rstride = row_stride(S);

    int spread_lg = A.spread ? 1 : -1;  // log2(lanes per row) of the samples-across-lanes layout; -1 = never
    while (spread_lg >= 0 && (1 << spread_lg) < S) spread_lg++;

    bool need = true, done = false, queue_empty = false;
    // the open chunk (wave-uniform): entries [chunk_pos, chunk_len) are still to be handed out; lane l holds the row of entry l
    uint32_t chunk_pos = 0, chunk_len = 0, rest_base = 0, rest_end = 0;  // [rest_base, rest_end): what is left of the last dequeue behind them
    int ord_reg = 0;
    int row = -1;
    int k = 0;
    double b0 = 0, b1 = 0, alpha = 0, size = 0, crow = 0, dev_old = 0, syA = 0, syB = 0;

    WDIAG(unsigned long long cyw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_eval_end = 0, wr_load_end = 0, g_open_cy = 0, g_open_n = 0;)
    for (;;) {
        WDIAG(const int sec = 0; const bool sec_bulk = !queue_empty; const unsigned long long wt0 = __builtin_amdgcn_s_memtime();)
        // refill (the scheme of disp_fit_kernel's: scalar chunk bookkeeping, a chunk's schedule entries read once and handed out by
        // lane permute, the all-zero flag and the whole record in one round trip)
        unsigned long long needmask = __ballot(need && !done);
#pragma unroll 1
        for (int attempt = 0; needmask != 0ull && attempt < 4; attempt++) {
            WDIAG(const unsigned long long wo0 = __builtin_amdgcn_s_memtime(); const bool opened = chunk_pos >= chunk_len;)
            if (chunk_pos >= chunk_len) {  // the lanes' entries are used up ...
                if (rest_base >= rest_end) {  // ... and nothing left of the last dequeue: one atomic per A.chunk rows
                    if (queue_empty) {
                        if (need) done = true;
                        break;
                    }
                    if (heads_left == 0u) {
                        queue_empty = true;
                        continue;
                    }
                    unsigned int kq = 0;
                    if (lane == 0) kq = (unsigned int)atomicAdd(heads + 8 * cur_head, 1ull);
                    kq = __builtin_amdgcn_readfirstlane(kq);
                    const uint32_t cq = kq * 8u + (uint32_t)cur_head;
                    if (cq >= nchunks) {  // this head is dry: on to the next one that is not known to be (uses up one attempt)
                        heads_left &= ~(1u << cur_head);
                        for (int t = 1; t <= 8; t++) {
                            const int hn = (cur_head + t) & 7;
                            if (heads_left & (1u << hn)) { cur_head = hn; break; }
                        }
                        continue;
                    }
                    rest_base = cq * (uint32_t)A.chunk;
                    rest_end = rest_base + (uint32_t)A.chunk < nTot ? rest_base + (uint32_t)A.chunk : nTot;
                }
                // the next (up to) 64 entries of the dequeued run: one per lane.  (Measured and dropped, round 4: the 64 entries after
                // them read at the same time and every record's first line touched a piece ahead, as the line search warms its rows:
                // the refill's 5 600 cycles of an IRLS tick's 16 500 — in-kernel timers, make DIAG=1 + CHICDIFF_IRLS_STAMPS — stayed
                // 5 300, the launch 0.349 ms: the refill is not waiting for cold lines.)
                const uint32_t b = rest_base;
                chunk_len = rest_end - rest_base < 64u ? rest_end - rest_base : 64u;
                rest_base += chunk_len;
                chunk_pos = 0;
                ord_reg = (int)b + lane;
                if (order && (uint32_t)lane < chunk_len) ord_reg = order[b + lane];
            }
            WDIAG(if (opened && !queue_empty) { g_open_cy += __builtin_amdgcn_s_memtime() - wo0; g_open_n++; })
            const uint32_t cnt = (uint32_t)__popcll(needmask), avail = chunk_len - chunk_pos;
            const uint32_t take = cnt < avail ? cnt : avail;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(needmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needmask, 0u));
            const int r = __shfl(ord_reg, (int)((chunk_pos + rank) & 63u));
            chunk_pos += take;
            WDIAG(const unsigned long long wr0 = __builtin_amdgcn_s_memtime();)
            if (need && !done && rank < take) {
                int iyA = 0, iyB = 0;
                const double2 *stp = reinterpret_cast<const double2 *>(A.w.start) + 2 * (int64_t)r;  // (wald_prep_kernel)
                const double2 st0 = stp[0], st1 = stp[1];
                load_row_sums(A.w.rowpack + (int64_t)r * rstride, S, s_nf, s_y, lane, gmask, iyA, iyB);
                WDIAG(const unsigned long long wr1 = __builtin_amdgcn_s_memtime(); if (!queue_empty) { cyw[4] += wr0 - wt0; cyw[5] += wr1 - wr0; cyw[7]++; } wr_load_end = wr1;)
                if (st0.x != st0.x) {  // alpha = NaN: an all-zero row (wald_prep_kernel); its values in the lane's LDS column are never used
                    A.w.beta0[r] = NAN;
                    A.w.beta1[r] = NAN;
                    A.w.betaIter[r] = 0;
                } else {
                    row = r;
                    syA = (double)iyA;
                    syB = (double)iyB;
                    alpha = st0.x;
                    size = 1.0 / alpha;
                    crow = st0.y;  // row constant + sum_j y_j (log alpha + log nf_j), from wald_prep
                    b0 = st1.x;
                    b1 = st1.y;
                    k = 0;
                    dev_old = 0;
                    need = false;
                }
            }
            needmask = __ballot(need && !done);
            WDIAG(if (wr_load_end) { if (!queue_empty) cyw[6] += __builtin_amdgcn_s_memtime() - wr_load_end; wr_load_end = 0; })
        }
        if (__ballot(!done) == 0ull) break;
        WDIAG(const unsigned long long wt1 = __builtin_amdgcn_s_memtime();)
        const bool active = !need && !done;
        const unsigned long long actmask = __ballot(active);
        double wA = 0, wB = 0, uA = 0, uB = 0, Dl = 0, zcA = 0, zcB = 0, Dc = 0;
        const bool spread_now = queue_empty && spread_lg >= 0 && (__popcll(actmask) << spread_lg) <= 64;
        if (spread_now) {
            // Samples across lanes for the end of the launch (as disp_kernels.hip eval_point_spread): the rows still running are
            // the ones that need 20-100 steps, a handful per wave, and a row-per-lane tick costs the same ~800 instructions for
            // one busy lane as for 64.  The g-th live row is handled by lanes G g .. G g + G - 1 (G = 2^spread_lg >= S), one
            // sample each; every lane of the group then folds the S samples' terms in sample order with the very operations of
            // the loop below (same values, same order: same bits), and the lane that owns the row picks the sums up.
            const int grp = lane >> spread_lg, jj = lane & ((1 << spread_lg) - 1);
            int owner = 0, nact = 0;
            for (unsigned long long m = actmask; m; m &= m - 1ull, nact++)
                if (grp == nact) owner = __ffsll((long long)m) - 1;
            const bool mine = grp < nact && jj < S;
            const double b0o = __shfl(b0, owner), b1o = __shfl(b1, owner), alo = __shfl(alpha, owner), szo = __shfl(size, owner);
            const double etaAo = b0o, etaBo = b0o + b1o;
            const double E0 = texp(etaAo, s_exptab), E1 = texp(etaBo, s_exptab);
            const double nfj = mine ? s_nf[jj * 64 + owner] : 1.0;
            const double y = mine ? (double)s_y[jj * 64 + owner] : 0.0;
            const bool g = (gmask >> jj) & 1;
            const double raw = nfj * (g ? E1 : E0);
            const bool floored = raw < o.minmu;
            const double mu = floored ? o.minmu : raw;
            const double ma = alo * mu;
            const double t1 = 1.0 + ma;
            const double rt = rcp(t1);
            const double wj = mu * rt;
            const double yr = y * rt;
            const double sy = szo + y;
            const double L = tlog1p_from(ma, t1, rt, s_logtab);
            double de = 0.0;
            if (floored) de = (tlog(o.minmu, s_logtab) - tlog(nfj, s_logtab)) - (g ? etaBo : etaAo);
            const int fl = floored ? 1 : 0;
            const int base = grp << spread_lg;
            for (int j0 = 0; j0 < S; j0 += 4) {
                double q_wj[4], q_yr[4], q_sy[4], q_L[4], q_de[4], q_y[4];
                int q_fl[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {  // four samples' terms in flight at a time
                    const int src = base + (j0 + t < S ? j0 + t : j0);
                    q_wj[t] = __shfl(wj, src);
                    q_yr[t] = __shfl(yr, src);
                    q_sy[t] = __shfl(sy, src);
                    q_L[t] = __shfl(L, src);
                    q_fl[t] = __shfl(fl, src);
                    q_de[t] = __shfl(de, src);
                    q_y[t] = __shfl(y, src);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (j0 + t >= S) continue;
                    const bool gj = (gmask >> (j0 + t)) & 1;
                    const double gB = gj ? 1.0 : 0.0, gA = gj ? 0.0 : 1.0;
                    wA = fma(q_wj[t], gA, wA);
                    wB = fma(q_wj[t], gB, wB);
                    uA = fma(q_yr[t], gA, uA);
                    uB = fma(q_yr[t], gB, uB);
                    Dl = fma(q_sy[t], q_L[t], Dl);
                    if (q_fl[t]) {
                        if (gj) zcB = fma(q_wj[t], q_de[t], zcB); else zcA = fma(q_wj[t], q_de[t], zcA);
                        Dc = fma(-q_y[t], q_de[t], Dc);
                    }
                }
            }
            // an active lane's group is its rank among the active lanes
            const int src = (active ? __popcll(actmask & ((1ull << lane) - 1ull)) : 0) << spread_lg;
            wA = __shfl(wA, src);
            wB = __shfl(wB, src);
            uA = __shfl(uA, src);
            uB = __shfl(uB, src);
            Dl = __shfl(Dl, src);
            zcA = __shfl(zcA, src);
            zcB = __shfl(zcB, src);
            Dc = __shfl(Dc, src);
        }
        if (active) {  // lanes without a row sit this tick out; the wave as a whole goes on

            // z_j = eta_j + (y_j - mu_j)/mu_j and w_j = mu_j/(1 + alpha mu_j) give w_j z_j = w_j (eta_j - 1) + y_j/(1 + alpha mu_j):
            // no 1/mu, and eta is the group's constant unless mu was floored at minmu (rare, handled in the branch)
            const double etaA = b0, etaB = b0 + b1;
            double E0 = 0, E1 = 0;
            if (!spread_now) {  // (wave-uniform)
                E0 = texp(etaA, s_exptab);
                E1 = texp(etaB, s_exptab);
            }
            for (int j = 0; j < (spread_now ? 0 : S); j++) {
                const double nfj = s_nf[j * 64 + lane];
                const double y = (double)s_y[j * 64 + lane];
                const bool g = (gmask >> j) & 1;
                const double raw = nfj * (g ? E1 : E0);
                const bool floored = raw < o.minmu;
                const double mu = floored ? o.minmu : raw;
                const double ma = alpha * mu;
                const double t1 = 1.0 + ma;
                const double rt = rcp(t1);
                const double wj = mu * rt;
                const double gB = g ? 1.0 : 0.0, gA = g ? 0.0 : 1.0;  // wave-uniform: exact 1/0 factors instead of register selects
                const double yr = y * rt;
                wA = fma(wj, gA, wA);
                wB = fma(wj, gB, wB);
                uA = fma(yr, gA, uA);
                uB = fma(yr, gB, uB);
                // -log dnbinom's mu-dependent part: (size+y) log1p(alpha mu) - y log(alpha mu)
                Dl = fma(size + y, tlog1p_from(ma, t1, rt, s_logtab), Dl);
                if (floored) {  // eta_j = log(minmu) - log(nf_j) instead of the group's eta
                    const double de = (tlog(o.minmu, s_logtab) - tlog(nfj, s_logtab)) - (g ? etaB : etaA);
                    if (g) zcB = fma(wj, de, zcB); else zcA = fma(wj, de, zcA);
                    Dc = fma(-y, de, Dc);  // log mu_j = log nf_j + eta_group + de
                }
            }
            WDIAG(const unsigned long long wt2 = __builtin_amdgcn_s_memtime(); if (sec_bulk) { cyw[sec + 0] += wt1 - wt0; cyw[sec + 1] += wt2 - wt1; cyw[sec + 3]++; wt_eval_end = wt2; })
            const double zA = fma(etaA - 1.0, wA, uA) + zcA, zB = fma(etaB - 1.0, wB, uB) + zcB;
            const double D = (Dl - fma(etaA, syA, etaB * syB)) + Dc;  // `crow` carries the per-row constant
            bool stop = false;
            int iter_out = k;
            if (k >= 1) {
                const double dev = 2.0 * (D - crow);
                // DESeq2: conv_test = |dev - dev_old| / (|dev| + 0.1); NaN -> give up, < tol (from the 2nd step on) -> done.
                // Compared without the division.
                const double diff = fabs(dev - dev_old), scale = fabs(dev) + 0.1;
                if (!(diff == diff) || !(scale == scale) || (isinf(diff) && isinf(scale))) {  // the quotient would be NaN
                    stop = true;
                    iter_out = o.betaMaxit;
                } else if (k >= 2 && diff < o.betaTol * scale) {
                    stop = true;
                } else if (k >= o.betaMaxit) {
                    stop = true;
                }
                dev_old = dev;
            }
            if (!stop) {
                const double m00 = wA + wB + lambda, m01 = wB, m11 = wB + lambda;
                const double r0 = zA + zB, r1 = zB;
                const double idet = rcp(fma(m00, m11, -(m01 * m01)));
                b0 = fma(m11, r0, -(m01 * r1)) * idet;
                b1 = fma(m00, r1, -(m01 * r0)) * idet;
                k++;
                if (fabs(b0) > 30.0 || fabs(b1) > 30.0) {
                    stop = true;
                    iter_out = o.betaMaxit;
                }
            }
            if (stop) {
                if (iter_out >= o.betaMaxit) {
                    // fitNbinomGLMsOptim: start again from the least-squares start values, keep the optimum whatever happens
                    b0 = A.w.binit0[row];
                    b1 = A.w.binit1[row];
                    A.w.optimConv[row] = optim_row(s_y + lane, s_nf + lane, 64, S, gmask, alpha, A.w.crow[row], b0, b1) ? 1 : 0;
                }
                A.w.beta0[row] = b0;
                A.w.beta1[row] = b1;
                A.w.betaIter[row] = iter_out;
                need = true;
            }
        }
        WDIAG(if (wt_eval_end) { cyw[sec + 2] += __builtin_amdgcn_s_memtime() - wt_eval_end; wt_eval_end = 0; })
    }
    WDIAG(if (lane == 0) { for (int q = 0; q < 8; q++) atomicAdd(&g_irls_cy[q], cyw[q]); atomicAdd(&g_irls_open[0], g_open_cy); atomicAdd(&g_irls_open[1], g_open_n); })
}

// (Measured and dropped, round 2: the same IRLS with the row held in registers — exec-masked refill loads, 64-row chunks,
// no LDS staging.  Bit-identical, but slower at every S: 0.60 against 0.50 ms at 2 M x 8, 0.48 against 0.34 ms at S = 4,
// 1.21 against 0.83 ms at S = 16.  The kernel is issue-bound at ~6.6 cycles per wave instruction like the line
// searches; what idles half its lanes is the end of each wave's allotment — a third of the waves carry one row that
// needs 20-100 ticks — not the refill, and refilling on every tick only adds instructions.)

__device__ __forceinline__ int trim_lo(int n) {
    // trimratio c(1/3, 1/4, 1/8) on bins (0,3.5], (3.5,23.5], (23.5,Inf)
    const double tr = n <= 3 ? 1.0 / 3 : (n <= 23 ? 1.0 / 4 : 1.0 / 8);
    return (int)floor(n * tr);
}
__device__ __forceinline__ double trim_scale(int n) { return n <= 3 ? 2.04 : (n <= 23 ? 1.86 : 1.51); }

// R mean(x, trim) over the samples of cell `c`: drop the `lo` smallest and `lo` largest values.
// The row's values sit in LDS as s_q[j * T + tid]; `sq` selects (q - cm)^2 instead of q.  An
// element is kept when its rank (ties broken by index) lies in [lo, nc - lo): O(nc^2) LDS reads,
// no per-thread arrays (so no scratch), exact and order-independent.
__device__ __forceinline__ double cell_trimmed_mean(const double *s_q, int T, int tid, int S, uint64_t gmask, int c,
                                                    int nc, int lo, bool sq, double cm) {
    double sum = 0;
    if (lo == 1) {
        // cells of 3 .. 7 samples (every design up to 7 v 7): one value dropped at each end.  The element of rank 0 is the smallest
        // value, lowest index among equals; the element of rank nc - 1 the largest, highest index among equals: two linear passes
        // instead of nc^2 comparisons, and the kept values are added in index order as below — the same bits
        int imin = -1, imax = -1;
        double vmin = INFINITY, vmax = -INFINITY;
        for (int j = 0; j < S; j++) {
            if ((int)((gmask >> j) & 1) != c) continue;
            double vj = s_q[j * T + tid];
            if (sq) vj = (vj - cm) * (vj - cm);
            if (vj < vmin) { vmin = vj; imin = j; }
            if (vj >= vmax) { vmax = vj; imax = j; }
        }
        for (int j = 0; j < S; j++) {
            if ((int)((gmask >> j) & 1) != c || j == imin || j == imax) continue;
            double vj = s_q[j * T + tid];
            if (sq) vj = (vj - cm) * (vj - cm);
            sum += vj;
        }
        return sum / (nc - 2);
    }
    for (int j = 0; j < S; j++) {
        if ((int)((gmask >> j) & 1) != c) continue;
        double vj = s_q[j * T + tid];
        if (sq) vj = (vj - cm) * (vj - cm);
        int rank = 0;
        for (int k = 0; k < S; k++) {
            if ((int)((gmask >> k) & 1) != c) continue;
            double vk = s_q[k * T + tid];
            if (sq) vk = (vk - cm) * (vk - cm);
            rank += (vk < vj || (vk == vj && k < j)) ? 1 : 0;
        }
        if (rank >= lo && rank < nc - lo) sum += vj;
    }
    return sum / (nc - 2 * lo);
}

__global__ __launch_bounds__(256) void wald_final_kernel(const int32_t *__restrict__ counts,
                                                         const double *__restrict__ nf, FitDims d, FitWork w, Opts o,
                                                         chicdiff_nbglm_out out) {
    extern __shared__ double s_q[];  // [S][blockDim.x] normalised counts of this thread's row
    __shared__ LogEntry s_lt[64];
    __shared__ ExpEntry s_et[64];
    exp_table_to_lds(s_et);
    log_table_to_lds(s_lt);  // (ends with the barrier)
    const int T = blockDim.x, tid = threadIdx.x;
    const int64_t n = d.n;
    const int S = d.S;
    const double lambda = 1e-6 / (0.69314718055994530942 * 0.69314718055994530942);
    const bool want_cooks = out.maxCooks && (d.nA >= 3 || d.nB >= 3);
    double v[3] = {0, 0, 0};  // sum deviance, non-converged rows, all-zero rows
    for (int64_t i = blockIdx.x * (int64_t)T + tid; i < n; i += (int64_t)gridDim.x * T) {
        const bool az = w.allZero[i];
        double B0 = NAN, B1 = NAN, s0 = NAN, s1 = NAN, st = NAN, pv = NAN, dv = NAN, mc = NAN;
        int bconv = 0, biter = 0, amax = -1;
        if (!az) {
            const double alpha = w.disp[i], size = rcp(alpha);
            const double b0 = w.beta0[i], b1 = w.beta1[i];
            biter = w.betaIter[i];
            // rows the IRLS gave up on went through the optim fallback, which left 1 (reached the mode) or 0 there; -1 = IRLS converged
            const int oc = biter < o.betaMaxit ? -1 : w.optimConv[i];
            bconv = (biter < o.betaMaxit) || oc == 1;
            // (round 4: table-driven exp / log as in the IRLS, reciprocals where a quotient is not part of a decision — the kernel was
            // 2 400 instructions per row, a third of them IEEE divisions and polynomial logarithms)
            const double E0 = texp(b0, s_et), E1 = texp(b0 + b1, s_et);
            const double la = tlog(alpha, s_lt);
            double wA = 0, wB = 0, ll = w.crow[i], m = 0;
            int y_next = counts[i];
            double nf_next = nf[i];
            for (int j = 0; j < S; j++) {
                const bool g = (d.gmask >> j) & 1;
                const double y = (double)y_next;
                const double nfj = nf_next;
                if (j + 1 < S) {  // the next sample's loads are in flight while this one is worked on
                    y_next = counts[(int64_t)(j + 1) * n + i];
                    nf_next = nf[(int64_t)(j + 1) * n + i];
                }
                const double muf = nfj * (g ? E1 : E0);  // no floor for the likelihood
                const double mu = fmax(muf, o.minmu);
                const double wj = mu * rcp(fma(alpha, mu, 1.0));
                if (g) wB += wj; else wA += wj;
                // log dnbinom(y; size, mu) = crow_j - (size+y) log1p(alpha mu) + y log(alpha mu); DESeq2's optim
                // path evaluates it after flooring mu, the IRLS path before
                const double mul = oc >= 0 ? mu : muf;
                const double ma = alpha * mul, t = 1.0 + ma;
                ll -= (size + y) * tlog1p_from(ma, t, rcp(t), s_lt);
                if (y > 0) ll += y * (la + tlog(mul, s_lt));
                if (want_cooks) {
                    const double q = y / nfj;
                    s_q[j * T + tid] = q;
                    m += q;
                }
            }
            const double m00 = wA + wB + lambda, m01 = wB, m11 = wB + lambda, det = m00 * m11 - m01 * m01;
            const double idet = rcp_or_div(det);
            const double i00 = m11 * idet, i01 = -m01 * idet, i11 = m00 * idet;
            const double a00 = wA + wB, a01 = wB, a11 = wB;
            const double t00 = i00 * a00 + i01 * a01, t01 = i00 * a01 + i01 * a11;
            const double t10 = i01 * a00 + i11 * a01, t11 = i01 * a01 + i11 * a11;
            const double v0 = t00 * i00 + t01 * i01, v1 = t10 * i01 + t11 * i11;
            B0 = kLog2e * b0;
            B1 = kLog2e * b1;
            s0 = kLog2e * sqrt(fmax(v0, 0.0));
            s1 = kLog2e * sqrt(fmax(v1, 0.0));
            st = B1 / s1;
            pv = pnorm_two_sided(st);
            dv = -2.0 * ll;
            if (!bconv || !(v0 > 0) || !(v1 > 0)) v[1] += 1;
            if (want_cooks) {
                // robustMethodOfMomentsDisp: max over cells (>= 3 samples) of the scaled trimmed variance
                m /= S;
                double vmax = -INFINITY;
                for (int c = 0; c < 2; c++) {
                    const int nc = c ? d.nB : d.nA;
                    if (nc < 3) continue;
                    const int lo = trim_lo(nc);
                    const double cm = cell_trimmed_mean(s_q, T, tid, S, d.gmask, c, nc, lo, false, 0.0);
                    const double vv = trim_scale(nc) * cell_trimmed_mean(s_q, T, tid, S, d.gmask, c, nc, lo, true, cm);
                    if (vv > vmax) vmax = vv;
                }
                const double arob = fmax((vmax - m) / (m * m), 0.04);
                mc = -INFINITY;
                double call = -INFINITY;  // which.max(cooks[i, ]) runs over ALL samples
                int yc_next = counts[i];
                double nfc_next = nf[i];
                for (int j = 0; j < S; j++) {
                    const bool g = (d.gmask >> j) & 1;
                    const double nfj = nfc_next;
                    const double yc = (double)yc_next;
                    if (j + 1 < S) {
                        yc_next = counts[(int64_t)(j + 1) * n + i];
                        nfc_next = nf[(int64_t)(j + 1) * n + i];
                    }
                    const double muf = nfj * (g ? E1 : E0);
                    const double mu = fmax(muf, o.minmu);
                    const double wj = mu * rcp(fma(alpha, mu, 1.0));
                    const double h = wj * (g ? (i00 + 2 * i01 + i11) : i00);
                    const double V = muf + arob * muf * muf;
                    const double ck = ((yc - muf) * (yc - muf) * 0.5 * h) * rcp_or_div(V * ((1 - h) * (1 - h)));  // (y - mu)^2 / V / 2 h / (1 - h)^2
                    if (ck > call) { call = ck; amax = j; }
                    if ((g ? d.nB : d.nA) >= 3 && ck > mc) mc = ck;
                }
            }
            v[0] += dv;
        } else {
            v[2] += 1;
        }
        if (out.log2FoldChange) out.log2FoldChange[i] = B1;
        if (out.lfcSE) out.lfcSE[i] = s1;
        if (out.stat) out.stat[i] = st;
        if (out.pvalue) out.pvalue[i] = pv;
        if (out.intercept) out.intercept[i] = B0;
        if (out.interceptSE) out.interceptSE[i] = s0;
        if (out.deviance) out.deviance[i] = dv;
        if (out.maxCooks) out.maxCooks[i] = mc;
        if (out.betaConv) out.betaConv[i] = bconv;
        if (out.betaIter) out.betaIter[i] = biter;
        if (out.cooksArgmax) out.cooksArgmax[i] = amax;
    }
    // block partials: 3 values (blocks of 64..256 threads)
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = T >> 6;
    __syncthreads();
    for (int k = 0; k < 3; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double acc = 0;
        for (int q = 0; q < nw; q++) acc += red[threadIdx.x][q];
        w.partials[(size_t)blockIdx.x * 3 + threadIdx.x] = acc;
    }
}

// design ~1: fitNbinomGLMs' intercept-only shortcut (no IRLS).  This is the Wald stage of every theta-grid fit (chicdiff.R:1641-1647),
// where nothing overlaps it (the line-search launches of the other thetas fill the chip: small kernels wait for them), so it is
// written as lean as wald_prep: log y! from the table, table-driven logarithms, one reciprocal per offset instead of two divisions
// (round 3: IEEE divisions, polynomial logs, a second Stirling evaluation for log y!: 0.33 ms per fit at 2 M x 8).
__global__ __launch_bounds__(256) void wald_intercept_kernel(const int32_t *__restrict__ counts,
                                                             const double *__restrict__ nf, FitDims d, FitWork w,
                                                             chicdiff_nbglm_out out) {
    __shared__ double s_logfact[kLogFactN];
    __shared__ LogEntry s_lt[64];
    for (int k = threadIdx.x; k < kLogFactN; k += 256) s_logfact[k] = w.logfact[k];
    log_table_to_lds(s_lt);  // (ends with the barrier)
    const int64_t n = d.n;
    const int S = d.S;
    double v[3] = {0, 0, 0};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double B0 = NAN, s0 = NAN, st = NAN, pv = NAN, dv = NAN;
        const bool az = w.allZero[i];
        if (!az) {
            const double alpha = w.disp[i], size = rcp(alpha);
            // the row is read ONCE (S <= 16: counts and offsets stay in registers between the two passes over the samples; round 3 read
            // both matrices twice: 384 MB at 2 M x 8, which is what the kernel's 0.33 ms were)
            double f[16];
            int yv[16];
            double bm = 0;
            if (S <= 16) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    f[j] = 1.0;
                    yv[j] = 0;
                    if (j < S) {
                        yv[j] = counts[(int64_t)j * n + i];
                        f[j] = nf[(int64_t)j * n + i];
                        bm += (double)yv[j] / f[j];  // (as prep: the same baseMean)
                    }
                }
            } else {
                for (int j = 0; j < S; j++) bm += (double)counts[(int64_t)j * n + i] / nf[(int64_t)j * n + i];
            }
            bm /= S;
            B0 = log2(bm);
            const double e = exp2(B0);
            const LgrCtx cs = lgr_make_t(size, s_lt), c1 = lgr_one();
            const double la = tlog(alpha, s_lt);
            double ll = 0, xtwx = 0;
            auto sample = [&](double nfj, int yi) {
                const double mu = nfj * e;
                const double y = (double)yi, ma = alpha * mu, t = 1.0 + ma, rt = rcp(t);
                ll -= (size + y) * tlog1p_from(ma, t, rt, s_lt);
                if (yi > 0) ll += lgr_eval_t(cs, yi, s_lt) - (yi < kLogFactN ? s_logfact[yi] : lgr_eval(c1, yi)) + y * (la + tlog(mu, s_lt));
                xtwx += mu * rt;
            };
            if (S <= 16) {
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (j < S) sample(f[j], yv[j]);
            } else {
                for (int j = 0; j < S; j++) sample(nf[(int64_t)j * n + i], counts[(int64_t)j * n + i]);
            }
            s0 = kLog2e * sqrt(1.0 / xtwx);
            st = B0 / s0;
            pv = pnorm_two_sided(st);
            dv = -2.0 * ll;
            v[0] += dv;
        } else {
            v[2] += 1;
        }
        if (out.log2FoldChange) out.log2FoldChange[i] = NAN;
        if (out.lfcSE) out.lfcSE[i] = NAN;
        if (out.stat) out.stat[i] = st;
        if (out.pvalue) out.pvalue[i] = pv;
        if (out.intercept) out.intercept[i] = B0;
        if (out.interceptSE) out.interceptSE[i] = s0;
        if (out.deviance) out.deviance[i] = dv;
        if (out.maxCooks) out.maxCooks[i] = NAN;
        if (out.betaConv) out.betaConv[i] = az ? 0 : 1;
        if (out.betaIter) out.betaIter[i] = az ? 0 : 1;
        if (out.cooksArgmax) out.cooksArgmax[i] = -1;
    }
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 3; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        w.partials[(size_t)blockIdx.x * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// partials (kRedBlocks x 3) -> sc->final_sums[0..7): deviance sum, non-converged rows, all-zero rows, and four verdicts of this
// rank (1 = yes): trend-barrier timeout, negative counts, select overflow in this fit, select overflow in the size-factor select
// before it; plus the size factors of that select -> sc->final_sf, so that one read of the scalars brings everything to the host
__global__ void dev_sum_kernel(FitWork w, const int32_t *carry, const double *sf_dev, int S) {
    __shared__ double red[256];
    double *sums = w.sc->final_sums;
    for (int k = 0; k < 3; k++) {
        double acc = 0;
        for (int b = threadIdx.x; b < kRedBlocks; b += 256) acc += w.partials[(size_t)b * 3 + k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[k] = red[0];
        __syncthreads();
    }
    // a rank whose persistent trend kernel lost its grid barrier (sharded fits with the gathered trend): summed over the
    // ranks with the three above, so that every rank learns of it and all of them refit the same way
    if (threadIdx.x == 0) {
        sums[3] = w.sc->failed == 3 ? 1.0 : 0.0;
        sums[4] = w.sc->neg_counts ? 1.0 : 0.0;  // a rank that saw a negative / NA count: every rank must refuse the fit (api.hip)
        sums[5] = w.sc->sel_overflow ? 1.0 : 0.0;  // a candidate list of a sharded select did not fit: every rank refits (api.hip)
        sums[6] = (carry && *carry) ? 1.0 : 0.0;
        sums[7] = 0.0;
    }
    if (sf_dev && (int)threadIdx.x < S) w.sc->final_sf[threadIdx.x] = sf_dev[threadIdx.x];
}

// The IRLS's schedule (slow rows first) decides when a row STARTS; in a fit of n <= 131 072 rows — two thirds of the lanes three waves per
// SIMD hold — every row starts at once whatever the order, and building the order costs a launch and the class counts (round 6, 100 k x 8:
// wald_prep 0.028 -> 0.018 ms, IRLS 0.147 -> 0.135; 250 k x 8 still wants it: gene-wise and IRLS + 5 ... 13 % without)
static bool irls_schedule(const FitDims &d, const Opts &o) { return o.schedule && d.n > 131072; }
void launch_wald_prep(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    o.schedule = irls_schedule(d, o) ? o.schedule : 0;
    int64_t nblk, tile;
    order_tiles(d.n, nblk, tile);
    if (o.schedule && tile == 256) {  // at most one row per thread: the schedule's class counts ride along (no order_hist launch)
        wald_prep_kernel<<<(unsigned)nblk, 256, 0, st>>>(counts, nf, d, w, o.schedule, tile, reinterpret_cast<unsigned int *>(w.hist));
        launch_order_build(d, w, 0, true, st);
        return;
    }
    wald_prep_kernel<<<1536, 256, 0, st>>>(counts, nf, d, w, o.schedule, 0, nullptr);  // one resident round: 80 VGPRs = 6 workgroups per CU
    if (o.schedule) launch_order_build(d, w, 0, false, st);
}
void launch_wald_irls(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    WaldArgs A{counts, nf, d, w, o, 64, irls_schedule(d, o) ? w.order : nullptr, o.spread};
    const size_t lds_per_wave = (size_t)d.S * 64 * 12;
    int threads = 256;
    while (threads > 64 && lds_per_wave * (threads / 64) > 40 * 1024) threads >>= 1;
    const size_t lds = lds_per_wave * (threads / 64);
    int64_t blocks = ((d.n + 63) / 64 + threads / 64 - 1) / (threads / 64);
    int64_t per_cu = (int64_t)(160 * 1024 / lds) < 8 ? (int64_t)(160 * 1024 / lds) : 8;
    const int64_t by_regs = 4 * WALD_MINW / (threads / 64);  // workgroups per CU the registers allow (WALD_MINW waves per SIMD)
    if (per_cu > by_regs) per_cu = by_regs;                   // one resident round: a workgroup that starts late finds the queue empty
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    if (blocks < 1) blocks = 1;
    // Chunk = rows a wave takes from a queue head per atomic.  Small chunks balance the waves' ends but cost refill retries at
    // their boundaries (and, on ONE head, queue up: ~90 dequeues per us for the whole GPU — 64 / 96 / 128 / 256 / 512 rows:
    // 0.49 / 0.42 / 0.40 / 0.44 / 0.61 ms at 2 M x 8).  On the eight heads of the kernel: 32 / 64 / 128 rows 0.455 / 0.410 /
    // 0.377 ms at 2 M x 8 (S = 4: 0.37 / 0.34 / 0.32; S = 16: - / 0.56 / 0.54), 500 k x 8: 0.217 / 0.195 / 0.210
    const int64_t per_wave = d.n / (blocks * (threads / 64));
    A.chunk = per_wave >= 256 ? 128 : 64;
#ifdef CHICDIFF_DIAG
    const bool stamps = getenv("CHICDIFF_IRLS_STAMPS") != nullptr;  // diagnostic build only: blocking, never timed
    if (stamps) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_irls_cy), z, sizeof z);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_irls_open), z, 16);
    }
#endif
    wald_irls_kernel<<<(unsigned)blocks, threads, lds, st>>>(A);
#ifdef CHICDIFF_DIAG
    if (stamps) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_irls_cy), sizeof h);
        {
            const double t = h[3] ? (double)h[3] : 1.0, r = h[7] ? (double)h[7] : 1.0;
            printf("  IRLS ticks while the queue has rows (lane 0 of every wave): %.0f per wave; s_memtime cycles per tick: refill %.0f, evaluate %.0f, solve + state %.0f\n",
                   t / (double)(blocks * (threads / 64)), h[0] / t, h[1] / t, h[2] / t);
            unsigned long long ho[2];
            (void)hipMemcpyFromSymbol(ho, HIP_SYMBOL(g_irls_open), sizeof ho);
            printf("    pieces of the schedule opened: %.1f per wave, %.0f cycles each\n", (double)ho[1] / (double)(blocks * (threads / 64)), ho[1] ? (double)ho[0] / (double)ho[1] : 0.0);
            printf("    inside a refill that lane 0 took part in (%.0f per wave): tick start -> loads issued %.0f, loads issued -> row in LDS %.0f, -> end of the attempt %.0f cycles\n",
                   r / (double)(blocks * (threads / 64)), h[4] / r, h[5] / r, h[6] / r);
        }
    }
#endif
}
void launch_wald_final(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o,
                       const chicdiff_nbglm_out &out, hipStream_t st) {
    const int threads = d.S <= 16 ? 256 : 64;
    wald_final_kernel<<<kRedBlocks, threads, (size_t)d.S * threads * sizeof(double), st>>>(counts, nf, d, w, o, out);
}
void launch_wald_intercept(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts,
                           const chicdiff_nbglm_out &out, hipStream_t st) {
    wald_intercept_kernel<<<kRedBlocks, 256, 0, st>>>(counts, nf, d, w, out);
}
void launch_dev_sum_finish(FitDims d, FitWork w, const int32_t *carry, const double *sf_dev, hipStream_t st) {
    dev_sum_kernel<<<1, 256, 0, st>>>(w, carry, sf_dev, d.S);
}

}  // namespace cd
