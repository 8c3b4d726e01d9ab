// shard_harness.cpp — TEST-ONLY CPU backend for chicdiff_amd/csrc/fit_driver.h.
//
// Runs the product's own driver (fit_driver.h) and state machines (fit_state.h) over a ROW SHARD
// held in host memory, with the per-row passes written as plain loops and every global sum
// going through the caller's sum-all-reduce callback — the same protocol the HIP library uses
// (include/chicdiff_hip.h: chicdiff_allreduce_fn), here over host buffers so that
// tests/test_distributed_gloo.py can run it with world_size 2 on the gloo backend.
// It is not part of the product and is never loaded by chicdiff_amd.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../chicdiff_amd/csrc/fit_driver.h"

using namespace cd;

typedef int (*allreduce_fn)(void *user, void *buf, int64_t count);

struct CpuBackend {
    int world_ = 1;
    allreduce_fn cb = nullptr;
    void *user = nullptr;
    // trend inputs (row shard)
    const double *bm = nullptr, *dg = nullptr;
    const int32_t *az = nullptr;
    int64_t n = 0;
    double minDisp = 1e-8;
    // select inputs
    std::vector<double> resid;
    const int32_t *counts = nullptr;
    std::vector<double> lgm;
    int S = 0;
    FitScalars sc;
    std::vector<double> sums_, hist_;
    CpuBackend() : sums_(128, 0.0), hist_((size_t)kMaxS * 2 * kSelBins, 0.0) { memset(&sc, 0, sizeof sc); }

    int world() const { return world_; }
    int allreduce(double *buf, int64_t cnt) { return world_ > 1 ? cb(user, buf, cnt) : 0; }
    double *sums() { return sums_.data(); }
    int64_t sums_len() const { return cd::kTrendSums; }
    double *hist() { return hist_.data(); }
    void trend_init() { cd::trend_init(&sc); }
    void trend_pass(bool fused) {
        if (sc.finished) return;
        double v[kTrendSums] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t i = 0; i < n; i++) {
            if (az[i]) continue;
            if (!(dg[i] > 100 * minDisp)) continue;
            trend_row(&sc, bm[i], dg[i], v);
        }
        for (int k = 0; k < kTrendSums; k++) sums_[k] = v[k];
        if (fused) cd::trend_step(&sc, sums_.data());
    }
    void trend_step() { cd::trend_step(&sc, sums_.data()); }
    const FitScalars *sync_scalars() { return &sc; }

    bool key(const SelSpec &a, int col, int64_t i, uint64_t &k) const {
        double x;
        if (a.mode == SEL_SIZEFACTOR) {
            const double lg = lgm[i];
            const int32_t c = counts[(int64_t)col * n + i];
            if (!isfinite(lg) || c <= 0) return false;
            x = log((double)c) - lg;
        } else {
            x = resid[i];
            if (x != x) return false;
            if (a.mode == SEL_ABSDEV) x = fabs(x - sc.med);
        }
        k = key_of(x);
        return true;
    }
    void sel_hist(const SelSpec &a, int shift) {
        const int bits = sel_bits(shift), hi = shift + bits;
        const uint64_t mask = (1ull << bits) - 1ull;
        std::fill(hist_.begin(), hist_.begin() + (size_t)a.ncol * 2 * kSelBins, 0.0);
        uint64_t k;
        for (int c = 0; c < a.ncol; c++) {
            const bool first = sel_first_round(shift);  // stale prefixes of an earlier select must not matter
            const uint64_t p0 = first ? 0 : sc.sel_prefix[2 * c], p1 = first ? 0 : sc.sel_prefix[2 * c + 1];
            double *g = hist_.data() + (size_t)c * 2 * kSelBins;
            for (int64_t i = 0; i < n; i++) {
                if (!key(a, c, i, k)) continue;
                const unsigned dig = (unsigned)((k >> shift) & mask);
                if (sel_match(k, p0, hi)) g[dig] += 1;
                else if (p0 != p1 && sel_match(k, p1, hi)) g[kSelBins + dig] += 1;
            }
        }
    }
    void sel_step(const SelSpec &a, int shift) {
        const int nb = 1 << sel_bits(shift);
        for (int c = 0; c < a.ncol; c++) {
            if (sel_first_round(shift)) {  // population = sum of the first histogram
                double total = 0;
                for (int b = 0; b < nb; b++) total += hist_[(size_t)c * 2 * kSelBins + b];
                cd::sel_begin(&sc, c, total);
            }
            const uint64_t p0 = sc.sel_prefix[2 * c], p1 = sc.sel_prefix[2 * c + 1];
            for (int slot = 0; slot < 2; slot++) {
                const int hslot = (slot == 1 && p0 != p1) ? 1 : 0;
                sel_pick(&sc, c, slot, hist_.data() + ((size_t)c * 2 + hslot) * kSelBins, nb, shift, slot ? p1 : p0);
            }
        }
    }
    bool sel_shortcut(const SelSpec &) { return false; }  // the sharded path always runs all six rounds
    void sel_finish(const SelSpec &a) {
        for (int c = 0; c < a.ncol; c++) {
            const double med = sel_median(&sc, c);
            if (a.mode == SEL_RESID) sc.med = med;
            else if (a.mode == SEL_ABSDEV) sc.mad = 1.4826 * med;
            else sc.sel_value[2 * c] = exp(med);
        }
    }
};

extern "C" {

// trend + MAD + prior variance over a row shard.  out: c0, c1, varLogDispEsts, dispPriorVar, outer_it, failed, med, mad
int harness_trend_mad(const double *baseMean, const double *dispGene, const int32_t *allZero, int64_t n, double minDisp,
                      int32_t S, int32_t p, double prior_in, int32_t world, allreduce_fn cb, void *user, double *out) {
    CpuBackend be;
    be.world_ = world;
    be.cb = cb;
    be.user = user;
    be.bm = baseMean;
    be.dg = dispGene;
    be.az = allZero;
    be.n = n;
    be.minDisp = minDisp;
    int rc = drive_trend(be);
    if (rc) return rc;
    be.resid.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double r = NAN;
        if (!allZero[i] && dispGene[i] >= 100 * minDisp)
            r = log(dispGene[i]) - log(be.sc.coefs[0] + be.sc.coefs[1] / baseMean[i]);
        be.resid[(size_t)i] = r;
    }
    SelSpec a{SEL_RESID, 1};
    if ((rc = drive_select(be, a))) return rc;
    a.mode = SEL_ABSDEV;
    if ((rc = drive_select(be, a))) return rc;
    prior_var(&be.sc, S, p, prior_in);
    out[0] = be.sc.coefs[0];
    out[1] = be.sc.coefs[1];
    out[2] = be.sc.varLogDispEsts;
    out[3] = be.sc.dispPriorVar;
    out[4] = be.sc.outer_it;
    out[5] = be.sc.failed;
    out[6] = be.sc.med;
    out[7] = be.sc.mad;
    return 0;
}

// median-of-ratios size factors over a row shard (counts column-major n x S)
int harness_size_factors(const int32_t *counts, int64_t n, int32_t S, int32_t world, allreduce_fn cb, void *user, double *sf) {
    CpuBackend be;
    be.world_ = world;
    be.cb = cb;
    be.user = user;
    be.n = n;
    be.S = S;
    be.counts = counts;
    be.lgm.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double s = 0;
        for (int j = 0; j < S; j++) s += log((double)counts[(int64_t)j * n + i]);
        be.lgm[(size_t)i] = s / S;
    }
    SelSpec a{SEL_SIZEFACTOR, S};
    const int rc = drive_select(be, a);
    if (rc) return rc;
    for (int j = 0; j < S; j++) sf[j] = be.sc.sel_value[2 * j];
    return 0;
}
}
