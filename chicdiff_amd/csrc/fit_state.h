// fit_state.h — the fit's global state machines, shared between device and host.
//
// Everything global in one fit is a SUM over rows followed by a small scalar
// decision: Gamma-GLM trend IRLS (DESeq2 parametricDispersionFit + glm.fit, Appendix A3), exact
// medians by radix select (size factors A1, MAD A3), prior variance (A4).  The decisions live
// here as plain inline functions over `FitScalars`, so that
//   * the GPU runs them in one-thread kernels (no host round trip per IRLS step), and
//   * the row-sharded path (sum-all-reduce between the per-rank partial sums and the decision)
//     can be exercised on CPU, world_size 2 over gloo, with exactly this code
//     (tests/harness/shard_harness.cpp).
// No HIP types here: compiles with hipcc and with g++.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CD_HD __host__ __device__ inline
#else
#define CD_HD inline
#endif

namespace cd {

constexpr int kMaxS = 64;          // samples per fit (group mask is one 64-bit word)
constexpr int kSelBits = 12;       // radix-select digit width
constexpr int kSelBins = 1 << kSelBits;
constexpr int kTrendSums = 8;      // dev, sw, swx, swxx, swy, swxy, count, invalid

// Device-resident scalars of one fit; the host reads the struct back once, at the end.
struct FitScalars {
    double colsum[kMaxS];  // column sums of nf over non-all-zero rows (then all-reduced)
    double nnz;            // number of non-all-zero rows (double: goes through the f64 all-reduce)
    double xim;            // mean_j 1/colMeans(nf)_j                  (DESeq2 momentsDispEstimate)
    // trend state machine (parametricDispersionFit + glm.fit)
    double coefs[2];       // outer-loop coefficients (define the `good` set)
    double b[2];           // inner IRLS iterate
    double devold;
    int32_t inner_it, outer_it, phase, finished, failed, conv, neg_counts /* a count < 0 (NA_integer_) was seen by prep */,
        trend_local /* dispFit[] holds the local-regression trend (DESeq2 fitType "local"): coefs are not used */;
    double nfit;
    // MAD / prior
    double med, mad, varLogDispEsts, dispPriorVar;
    double nres;
    double sumDeviance;
    double nonconv;
    // radix-select state: up to kMaxS columns x 2 ranks (lower / upper median)
    uint64_t sel_prefix[kMaxS * 2];
    double sel_rank[kMaxS * 2];  // remaining 0-based rank inside the current prefix
    double sel_count[kMaxS];     // population per column
    double sel_value[kMaxS * 2]; // selected order statistics
    uint32_t sel_cnt[kMaxS * 2]; // single-rank shortcut: candidates left after two rounds, per (column, slot)
    int32_t sel_fast_done, _pad2; // 1 = the shortcut resolved every order statistic of the running select
    // schedule of the gene-wise line search (disp_kernels.hip: order_*): rows of order[0, ord_na) are dealt out statically,
    // rows of order[ord_na, ord_n) through the queue; all-zero rows are in neither
    int64_t ord_na, ord_n;
    // sharded selects run optimistically (no host look at sel_fast_done): a candidate list that did not fit (massive ties)
    // leaves this set, the host sees it with the fit's final scalars and refits with every histogram round
    int32_t sel_overflow, _pad3;
    // what the host reads when a fit ends, gathered here by the fit's last kernel so that ONE copy brings everything back:
    // [0] deviance sum, [1] non-converged rows, [2] all-zero rows, then four verdicts of this rank (all seven summed over the ranks
    // of a sharded fit): [3] trend kernel's grid barrier timed out, [4] negative / NA count, [5] a select's candidate list overflowed
    // in this fit, [6] ... in the size-factor select the caller ran before it; and the size factors of that select
    double final_sums[8];
    double final_sf[kMaxS];
};

// order-preserving map double -> uint64 (NaN never passed in)
CD_HD uint64_t key_of(double x) {
    union { double d; uint64_t u; } c;
    c.d = x;
    return (c.u & 0x8000000000000000ull) ? ~c.u : (c.u | 0x8000000000000000ull);
}
CD_HD double value_of(uint64_t k) {
    union { double d; uint64_t u; } c;
    c.u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return c.d;
}

// ---- trend ---------------------------------------------------------------------------------
enum { TR_INNER_START = 0, TR_INNER_ITER = 1 };

CD_HD void trend_init(FitScalars *sc) {
    sc->coefs[0] = 0.1;  // parametricDispersionFit: coefs <- c(.1, 1)
    sc->coefs[1] = 1.0;
    sc->b[0] = 0.1;
    sc->b[1] = 1.0;
    sc->devold = 0;
    sc->inner_it = 0;
    sc->outer_it = 0;
    sc->phase = TR_INNER_START;
    sc->finished = 0;
    sc->failed = 0;
    sc->conv = 0;
}

// One row's contribution to the fused pass: is it in the current `good` set, and if so its
// deviance at the iterate b and the weighted-LS sums for the next iterate.
CD_HD void trend_row(const FitScalars *sc, double baseMean, double disp, double *v /*[kTrendSums]*/) {
    const double r = disp / (sc->coefs[0] + sc->coefs[1] / baseMean);
    if (!(r > 1e-4 && r < 15)) return;
    const double x = 1.0 / baseMean;
    const double mu = sc->b[0] + sc->b[1] * x;
    if (!(mu > 0) || !isfinite(mu)) {
        v[7] += 1;
        return;
    }
    v[0] += -2.0 * (log(disp / mu) - (disp - mu) / mu);  // Gamma deviance residual
    const double wt = 1.0 / (mu * mu);                   // glm.fit weight for Gamma/identity
    v[1] += wt;
    v[2] += wt * x;
    v[3] += wt * x * x;
    v[4] += wt * disp;
    v[5] += wt * x * disp;
    v[6] += 1;
}

// Consume the (all-reduced) sums of one pass: R's glm.fit bookkeeping (epsilon 1e-8, maxit 25)
// inside parametricDispersionFit's outer loop (<= 10 re-selections, stop on sum(log(c/c_old)^2) < 1e-6).
CD_HD void trend_step(FitScalars *sc, const double *s /*[kTrendSums]*/) {
    if (sc->finished) return;
    const double dev = s[0], sw = s[1], swx = s[2], swxx = s[3], swy = s[4], swxy = s[5], cnt = s[6], bad = s[7];
    bool inner_done = false, conv = false;
    if (bad > 0 || cnt < 2) {  // invalid mu (R would step-halve; DESeq2 ends in "fit failed") or no data
        sc->failed = 1;
        sc->finished = 1;
        return;
    }
    if (sc->phase == TR_INNER_START) {
        sc->devold = dev;
        sc->inner_it = 0;
        sc->phase = TR_INNER_ITER;
    } else {
        if (fabs(dev - sc->devold) / (fabs(dev) + 0.1) < 1e-8) {
            inner_done = true;
            conv = true;
        } else {
            sc->devold = dev;
            if (sc->inner_it >= 25) inner_done = true;  // glm.fit maxit, not converged
        }
    }
    if (!inner_done) {
        const double det = sw * swxx - swx * swx;
        const double nb0 = (swxx * swy - swx * swxy) / det, nb1 = (sw * swxy - swx * swy) / det;
        if (!isfinite(nb0) || !isfinite(nb1)) {
            sc->failed = 1;
            sc->finished = 1;
            return;
        }
        sc->b[0] = nb0;
        sc->b[1] = nb1;
        sc->inner_it++;
        return;
    }
    const double o0 = sc->coefs[0], o1 = sc->coefs[1];
    sc->coefs[0] = sc->b[0];
    sc->coefs[1] = sc->b[1];
    if (!(sc->coefs[0] > 0 && sc->coefs[1] > 0)) {  // "parametric dispersion fit failed"
        sc->failed = 1;
        sc->finished = 1;
        return;
    }
    const double l0 = log(sc->coefs[0] / o0), l1 = log(sc->coefs[1] / o1);
    if ((l0 * l0 + l1 * l1 < 1e-6) && conv) {
        sc->finished = 1;
        sc->conv = 1;
        return;
    }
    sc->outer_it++;
    if (sc->outer_it > 10) {  // "dispersion fit did not converge"
        sc->failed = 2;
        sc->finished = 1;
        return;
    }
    sc->phase = TR_INNER_START;  // next glm() call: new `good` set, start = coefs
}

// ---- radix select ----------------------------------------------------------------------------
// ranks of the two middle order statistics (R median(): mean of the two for even counts)
CD_HD void sel_begin(FitScalars *sc, int col, double population) {
    sc->sel_count[col] = population;
    const int64_t mi = (int64_t)population;
    sc->sel_rank[2 * col] = (double)((mi - 1) / 2);
    sc->sel_rank[2 * col + 1] = (double)(mi / 2);
    sc->sel_prefix[2 * col] = 0;
    sc->sel_prefix[2 * col + 1] = 0;
}
// does `key` fall under prefix `p` on the bits above `hi`?
CD_HD bool sel_match(uint64_t key, uint64_t p, int hi) { return hi >= 64 || (key >> hi) == (p >> hi); }

// given the (all-reduced) digit histogram `g[0..nb)` for one slot, extend its prefix
CD_HD void sel_pick(FitScalars *sc, int col, int slot, const double *g, int nb, int shift, uint64_t prefix) {
    const double rank = sc->sel_rank[2 * col + slot];
    double cum = 0;
    int b = 0;
    for (; b < nb - 1; b++) {
        if (cum + g[b] > rank) break;
        cum += g[b];
    }
    sc->sel_prefix[2 * col + slot] = prefix | ((uint64_t)b << shift);
    sc->sel_rank[2 * col + slot] = rank - cum;
}
CD_HD double sel_median(const FitScalars *sc, int col) {
    const double lo = value_of(sc->sel_prefix[2 * col]), hi = value_of(sc->sel_prefix[2 * col + 1]);
    return (sc->sel_count[col] > 0) ? (lo + hi) / 2.0 : NAN;
}
static const int kSelShifts[6] = {52, 40, 28, 16, 4, 0};  // 5 x 12 bits + 4 bits
CD_HD bool sel_first_round(int shift) { return shift == 52; }
CD_HD int sel_bits(int shift) { return shift == 0 ? 4 : kSelBits; }

// ---- gathering the candidates after two rounds (sharded shortcut) ------------------------------------
// After two 12-bit rounds only the keys sharing 24 bits with a median are left — a few hundred among millions.
// Sharded, every rank knows how many of them it holds (its own round-2 histogram at the chosen digit), so
// instead of four more histogram rounds: one all-reduce of the per-rank counts (each rank fills its own row),
// every rank writes its candidates at its offset into a zeroed buffer, one all-reduce (a sum of disjoint
// entries with zeros = a gather), and every rank sorts the same list.
constexpr int kSelCap = 4096;      // candidates per (column, slot): what the histogram buffer holds
constexpr int kSelMaxWorld = 64;   // rows of the count buffer
// this rank's candidates of (col, slot): the entry of its own round-2 histogram at the digit the (global) step chose
CD_HD double sel_local_count(const FitScalars *sc, const double *local_hist, int col, int slot) {
    const uint64_t p0 = sc->sel_prefix[2 * col], p1 = sc->sel_prefix[2 * col + 1];
    if (slot == 1 && p0 == p1) return 0.0;  // the upper middle shares the lower one's candidates
    const uint64_t p = slot ? p1 : p0;
    const int hslot = (slot == 1 && (p0 >> 52) != (p1 >> 52)) ? 1 : 0;  // which round-2 histogram counted them
    return local_hist[((size_t)col * 2 + hslot) * kSelBins + (size_t)((p >> 40) & (kSelBins - 1))];
}
// offset of `rank`'s candidates of entry q (= 2*col + slot) and the total over ranks, from the all-reduced counts
CD_HD void sel_gather_layout(const double *cnt, int world, int rank, int nq, int q, double *base, double *total) {
    double b = 0, t = 0;
    for (int r = 0; r < world; r++) {
        const double c = cnt[(size_t)r * nq + q];
        if (r < rank) b += c;
        t += c;
    }
    *base = b;
    *total = t;
}

// ---- prior variance (estimateDispersionsPriorVar, closed-form branch) ---------------------------
CD_HD double trigamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r += 1.0 / (x * x);
        x += 1.0;
    }
    const double xi = 1.0 / x, x2 = xi * xi;
    double s = 691.0 / 2730.0 - x2 * (7.0 / 6.0);
    s = 5.0 / 66.0 - x2 * s;
    s = 1.0 / 30.0 - x2 * s;
    s = 1.0 / 42.0 - x2 * s;
    s = 1.0 / 30.0 - x2 * s;
    s = 1.0 / 6.0 - x2 * s;
    return r + xi * (1.0 + 0.5 * xi + x2 * s);
}
CD_HD void prior_var(FitScalars *sc, int S, int p, double prior_in) {
    const double v = sc->mad * sc->mad;
    sc->varLogDispEsts = v;
    sc->dispPriorVar = (prior_in == prior_in) ? prior_in : fmax(v - trigamma_pos((S - p) / 2.0), 0.25);
}

}  // namespace cd
