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
//     Armijo step, the start point, or a grid point of the fitDispGrid fallback — so lanes in
//     different phases still execute the same instruction stream;
//   * iteration counts are heavy-tailed (median ~7, 1.7 % of rows run 100 + 40 grid points),
//     so a finished lane immediately pulls the next row from a global queue (one atomic per
//     wave per refill) instead of idling until its 63 neighbours finish.
#include "common.h"
#include "devmath.h"

namespace cd {

// ------------------------------------------------------------------------------------------
// prep: per-row moments of the normalised counts (getBaseMeansAndVariances, roughDispEstimate,
// linearModelMu group means).  One thread per row, sample-major loads are coalesced.
__global__ __launch_bounds__(256) void prep_kernel(const int32_t *__restrict__ counts,
                                                   const double *__restrict__ nf, FitDims d, FitWork w) {
    const int64_t n = d.n;
    const int S = d.S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0, g0 = 0, g1 = 0;
        int64_t tot = 0;
        for (int j = 0; j < S; j++) {
            const int32_t k = counts[(int64_t)j * n + i];
            const double q = (double)k / nf[(int64_t)j * n + i];
            tot += k;
            s += q;
            if ((d.gmask >> j) & 1) g1 += q; else g0 += q;
        }
        const double bm = s / S;
        g0 /= d.nA;
        if (d.p == 2) g1 /= d.nB;
        const double m0 = fmax(1.0, g0), m1 = fmax(1.0, g1);
        double v = 0, est = 0;
        for (int j = 0; j < S; j++) {
            const double q = (double)counts[(int64_t)j * n + i] / nf[(int64_t)j * n + i];
            v += (q - bm) * (q - bm);
            const double mj = ((d.gmask >> j) & 1) ? m1 : m0;
            est += ((q - mj) * (q - mj) - mj) / (mj * mj);
        }
        w.baseMean[i] = bm;
        w.baseVar[i] = v / (S - 1);
        w.gm0[i] = g0;
        w.gm1[i] = g1;
        w.rough[i] = fmax(est / (S - d.p), 0.0);
        w.allZero[i] = (tot == 0);
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

__global__ void colsum_finish_kernel(FitDims d, FitWork w, int nblk) {
    const int j = threadIdx.x;
    if (j > d.S) return;
    double s = 0;
    for (int b = 0; b < nblk; b++) s += w.partials[(int64_t)j * nblk + b];
    if (j < d.S) w.sc->colsum[j] = s; else w.sc->nnz = s;
}

__global__ void xim_kernel(FitDims d, FitWork w) {
    if (threadIdx.x || blockIdx.x) return;
    double x = 0;
    for (int j = 0; j < d.S; j++) x += 1.0 / (w.sc->colsum[j] / w.sc->nnz);
    w.sc->xim = x / d.S;
}

void launch_prep(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts, hipStream_t st) {
    prep_kernel<<<kRedBlocks, 256, 0, st>>>(counts, nf, d, w);
    colsum_kernel<<<dim3(64, d.S + 1), 256, 0, st>>>(nf, d, w);
}
void launch_prep_finish(FitDims d, FitWork w, hipStream_t st) { colsum_finish_kernel<<<1, 128, 0, st>>>(d, w, 64); }
void launch_xim(FitDims d, FitWork w, hipStream_t st) { xim_kernel<<<1, 64, 0, st>>>(d, w); }

// ------------------------------------------------------------------------------------------
enum Phase : int { PH_NEED = 0, PH_INIT = 1, PH_SEARCH = 2, PH_GRID1 = 3, PH_GRID2 = 4, PH_DONE = 5 };

struct DispArgs {
    const int32_t *counts;
    const double *nf;
    FitDims d;
    FitWork w;
    Opts o;
};

// log posterior of a = log(alpha) and its derivative for one row held in LDS (A2.6).
// mu_j = max(nf_j * groupmean_g, minmu) is rebuilt on the fly; log(mu + 1/alpha) is folded
// into log(1 + mu*alpha) - a so each sample costs one log besides the lgamma/digamma pair.
__device__ __forceinline__ void eval_point(const double *s_nf, const int *s_y, int lane, int S, uint64_t gmask,
                                           bool p2, double gm0, double gm1, double minmu, double a,
                                           bool use_prior, double prior_mean, double prior_isig,
                                           double &lp, double &dlp) {
    const double alpha = exp(a);
    const double r = 1.0 / alpha;
    double lg_r, dg_r;
    lgamma_digamma(r, lg_r, dg_r);
    double ll = 0, sd = 0, wA = 0, wB = 0, dA = 0, dB = 0;
    for (int j = 0; j < S; j++) {
        const double nfj = s_nf[j * 64 + lane];
        const double y = (double)s_y[j * 64 + lane];
        const bool g = (gmask >> j) & 1;
        const double mu = fmax(nfj * (g ? gm1 : gm0), minmu);
        const double ma = mu * alpha;
        const double t = 1.0 + ma;
        const double rt = rcp(t);
        const double L = log(t);
        const double wj = mu * rt;  // 1 / (1/mu + alpha)
        if (g) { wB += wj; dB -= wj * wj; } else { wA += wj; dA -= wj * wj; }
        double lg, dg;
        lgamma_digamma(y + r, lg, dg);
        ll += (lg - lg_r) - y * (L - a) - r * L;
        sd += (dg_r - dg) + L - ma * rt + y * alpha * rt;
    }
    double cr, dcr;
    if (p2) {
        cr = -0.5 * log(wA * wB);
        dcr = -0.5 * (dA / wA + dB / wB);
    } else {
        cr = -0.5 * log(wA);
        dcr = -0.5 * (dA / wA);
    }
    double pr = 0, dpr = 0;
    if (use_prior) {
        const double dd = a - prior_mean;
        pr = -0.5 * dd * dd * prior_isig;
        dpr = -dd * prior_isig;
    }
    lp = ll + pr + cr;
    dlp = (r * r * sd + dcr) * alpha + dpr;
}

template <bool MAP>
__global__ __launch_bounds__(256) void disp_fit_kernel(DispArgs A) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = A.d.S;
    const int64_t n = A.d.n;
    double *s_nf = smem + (size_t)wave * S * 96;  // per wave: S*64 doubles + S*64 ints = S*96 doubles' worth
    int *s_y = reinterpret_cast<int *>(s_nf + S * 64);
    const uint64_t gmask = A.d.gmask;
    const bool p2 = A.d.p == 2;
    const Opts o = A.o;
    const double min_log_alpha = log(o.minDisp / 10.0);
    const double glo = log(1e-8), ghi = log(o.maxDisp), gstep = (ghi - glo) / 19.0;
    FitScalars *sc = A.w.sc;
    // fit-wide scalars (uniform)
    const double xim = sc->xim;
    double c0 = 0, c1 = 0, prior_isig = 0, out_thr = 0;
    if (MAP) {
        c0 = sc->coefs[0];
        c1 = sc->coefs[1];
        prior_isig = 1.0 / sc->dispPriorVar;
        out_thr = o.outlierSD * sqrt(sc->varLogDispEsts);
    }
    unsigned long long *queue = A.w.queue + (MAP ? 1 : 0);

    int phase = PH_NEED, iter = 0, iacc = 0, gt = 0, gbi = 0;
    int64_t row = -1;
    double a = 0, lp = 0, dlp = 0, kappa = 0, init_lp = 0, a0 = 0, gm0 = 0, gm1 = 0, prior_mean = 0;
    double gbest = 0, ghat = 0, dgene = 0, a_new = 0;
    bool queue_empty = false;

    for (;;) {
        // ---- refill: lanes without a row pull the next ones from the queue -----------------
        for (int attempt = 0; attempt < 4; attempt++) {
            const unsigned long long needmask = __ballot(phase == PH_NEED);
            if (!needmask) break;
            if (queue_empty) {
                if (phase == PH_NEED) phase = PH_DONE;
                break;
            }
            const int cnt = __popcll(needmask);
            const int leader = __ffsll((long long)needmask) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(queue, (unsigned long long)cnt);
            base = __shfl(base, leader);
            if (base + cnt >= (unsigned long long)n) queue_empty = true;
            if (phase == PH_NEED) {
                const int rank = __popcll(needmask & ((1ull << lane) - 1ull));
                const int64_t r = (int64_t)base + rank;
                if (r >= n) {
                    phase = PH_DONE;
                } else if (A.w.allZero[r]) {
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
                    row = r;
                    for (int j = 0; j < S; j++) {
                        s_nf[j * 64 + lane] = A.nf[(int64_t)j * n + r];
                        s_y[j * 64 + lane] = A.counts[(int64_t)j * n + r];
                    }
                    gm0 = A.w.gm0[r];
                    gm1 = A.w.gm1[r];
                    const double bm = A.w.baseMean[r];
                    if (!MAP) {
                        const double moments = (A.w.baseVar[r] - xim * bm) / (bm * bm);
                        a0 = fmin(fmax(o.minDisp, fmin(A.w.rough[r], moments)), o.maxDisp);
                        a = log(a0);
                    } else {
                        dgene = A.w.dispGene[r];
                        const double df = c0 + c1 / bm;
                        A.w.dispFit[r] = df;
                        prior_mean = log(df);
                        a = log(dgene > 0.1 * df ? dgene : df);
                    }
                    phase = PH_INIT;
                }
            }
        }
        if (__ballot(phase != PH_DONE) == 0ull) break;

        // ---- choose this tick's evaluation point -------------------------------------------
        double a_eval = a;
        if (phase == PH_SEARCH) {
            iter++;
            const double a_prop = a + kappa * dlp;
            if (a_prop < -30.0) kappa = (-30.0 - a) / dlp;
            if (a_prop > 10.0) kappa = (10.0 - a) / dlp;
            a_new = a + kappa * dlp;
            a_eval = a_new;
        } else if (phase == PH_GRID1) {
            a_eval = (gt == 19) ? ghi : glo + gt * gstep;
        } else if (phase == PH_GRID2) {
            a_eval = (gt == 19) ? ghat + gstep : (ghat - gstep) + gt * (2.0 * gstep / 19.0);
        }

        // ---- evaluate -------------------------------------------------------------------------
        double l_new = 0, dl_new = 0;
        if (phase != PH_DONE && phase != PH_NEED)
            eval_point(s_nf, s_y, lane, S, gmask, p2, gm0, gm1, o.minmu, a_eval, MAP, prior_mean, prior_isig, l_new,
                       dl_new);

        // ---- advance the per-lane state machine ---------------------------------------------
        bool finished = false;  // line search over: decide between result and grid fallback
        double result = 0;
        bool have_result = false;
        if (phase == PH_INIT) {
            lp = l_new;
            dlp = dl_new;
            init_lp = l_new;
            kappa = o.kappa0;
            iter = 0;
            iacc = 0;
            phase = PH_SEARCH;
        } else if (phase == PH_SEARCH) {
            const double theta_kappa = -l_new;
            const double theta_hat_kappa = -lp - kappa * 1.0e-4 * dlp * dlp;
            if (theta_kappa <= theta_hat_kappa) {
                iacc++;
                a = a_new;
                const double change = l_new - lp;
                if (change < o.dispTol) {
                    lp = l_new;
                    finished = true;
                } else if (a < min_log_alpha) {
                    finished = true;
                } else {
                    lp = l_new;
                    dlp = dl_new;
                    kappa = fmin(kappa * 1.1, o.kappa0);
                    if (iacc % 5 == 0) kappa *= 0.5;
                }
            } else {
                kappa *= 0.5;
            }
            if (!finished && iter >= o.maxit) finished = true;
            if (finished) {
                bool grid;
                if (!MAP) {
                    double dd = fmin(exp(a), o.maxDisp);
                    if (lp < init_lp + fabs(init_lp) / 1e6) dd = a0;  // noIncrease: keep alpha_init
                    const bool conv = (iter < o.maxit) && (iter != 1);
                    grid = !conv && dd > o.minDisp * 10;
                    result = dd;
                } else {
                    grid = !(iter < o.maxit);
                    result = exp(a);
                }
                if (grid) {
                    phase = PH_GRID1;
                    gt = 0;
                    gbest = -INFINITY;
                    gbi = 0;
                } else {
                    have_result = true;
                }
            }
        } else if (phase == PH_GRID1 || phase == PH_GRID2) {
            if (l_new > gbest) {
                gbest = l_new;
                gbi = gt;
            }
            gt++;
            if (gt == 20) {
                if (phase == PH_GRID1) {
                    ghat = (gbi == 19) ? ghi : glo + gbi * gstep;
                    phase = PH_GRID2;
                    gt = 0;
                    gbest = -INFINITY;
                    gbi = 0;
                } else {
                    const double fa = (gbi == 19) ? ghat + gstep : (ghat - gstep) + gbi * (2.0 * gstep / 19.0);
                    result = exp(fa);
                    have_result = true;
                }
            }
        }
        if (have_result) {
            const double dd = fmin(fmax(result, o.minDisp), o.maxDisp);
            if (!MAP) {
                A.w.dispGene[row] = dd;
                A.w.geneIter[row] = iter;
            } else {
                const bool outl = log(dgene) > prior_mean + out_thr;
                A.w.dispMAP[row] = dd;
                A.w.disp[row] = outl ? dgene : dd;
                A.w.outlier[row] = outl;
                A.w.mapIter[row] = iter;
            }
            phase = PH_NEED;
        }
    }
}

static void launch_disp(bool map, const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o,
                        hipStream_t st) {
    DispArgs A{counts, nf, d, w, o};
    const size_t lds_per_wave = (size_t)d.S * 64 * 12;
    // 256-thread blocks while four waves' rows fit comfortably in LDS, else 64-thread blocks
    int threads = 256;
    while (threads > 64 && lds_per_wave * (threads / 64) > 40 * 1024) threads >>= 1;
    const size_t lds = lds_per_wave * (threads / 64);
    // persistent grid: enough waves to fill 256 CUs; rows are pulled from the queue
    int64_t waves_needed = (d.n + 63) / 64;
    int64_t blocks = (waves_needed + threads / 64 - 1) / (threads / 64);
    const int64_t max_blocks = 256 * (int64_t)(160 * 1024 / (lds > 0 ? lds : 1) < 8 ? 160 * 1024 / lds : 8);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    if (map)
        disp_fit_kernel<true><<<(unsigned)blocks, threads, lds, st>>>(A);
    else
        disp_fit_kernel<false><<<(unsigned)blocks, threads, lds, st>>>(A);
}

void launch_disp_gene(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    launch_disp(false, counts, nf, d, w, o, st);
}
void launch_disp_map(const int32_t *counts, const double *nf, FitDims d, FitWork w, Opts o, hipStream_t st) {
    launch_disp(true, counts, nf, d, w, o, st);
}

}  // namespace cd
