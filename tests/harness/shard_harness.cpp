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

#include <algorithm>
#include <vector>

#include "../../chicdiff_amd/csrc/fit_driver.h"

using namespace cd;

typedef int (*allreduce_fn)(void *user, void *buf, int64_t count);

struct CpuBackend {
    int world_ = 1, rank_ = 0;
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
    std::vector<double> sums_, hist_, hist_local_, cnt_;
    CpuBackend() : sums_(128, 0.0), hist_((size_t)kMaxS * 2 * kSelBins, 0.0), hist_local_((size_t)kMaxS * 2 * kSelBins, 0.0),
                   cnt_((size_t)kSelMaxWorld * 2 * kMaxS, 0.0) { memset(&sc, 0, sizeof sc); }

    int world() const { return world_; }
    int world_size() const { return world_; }
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
    bool sel_shortcut(const SelSpec &) { return false; }
    // sharded shortcut (fit_state.h), plain-loop twin of the HIP kernels
    bool sel_can_gather() const { return world_ <= kSelMaxWorld && !no_gather; }
    bool sel_gather_done() { return sync_scalars()->sel_fast_done != 0; }
    bool no_gather = false;
    double *sel_counts() { return cnt_.data(); }
    void sel_keep_local_hist(const SelSpec &a) { std::copy(hist_.begin(), hist_.begin() + (size_t)a.ncol * 2 * kSelBins, hist_local_.begin()); }
    void sel_gather_counts(const SelSpec &a) {
        const int nq = 2 * a.ncol;
        std::fill(cnt_.begin(), cnt_.begin() + (size_t)world_ * nq, 0.0);
        for (int q = 0; q < nq; q++) cnt_[(size_t)rank_ * nq + q] = sel_local_count(&sc, hist_local_.data(), q / 2, q % 2);
    }
    bool gather_fits(const SelSpec &a) const {
        for (int q = 0; q < 2 * a.ncol; q++) {
            double b, t;
            sel_gather_layout(cnt_.data(), world_, rank_, 2 * a.ncol, q, &b, &t);
            if (t > kSelCap) return false;
        }
        return true;
    }
    void sel_gather_place(const SelSpec &a) {
        std::fill(hist_.begin(), hist_.begin() + (size_t)a.ncol * 2 * kSelCap, 0.0);
        if (!gather_fits(a)) return;
        uint64_t k;
        for (int c = 0; c < a.ncol; c++) {
            const uint64_t p0 = sc.sel_prefix[2 * c], p1 = sc.sel_prefix[2 * c + 1];
            double pos[2], tot;
            sel_gather_layout(cnt_.data(), world_, rank_, 2 * a.ncol, 2 * c, &pos[0], &tot);
            sel_gather_layout(cnt_.data(), world_, rank_, 2 * a.ncol, 2 * c + 1, &pos[1], &tot);
            for (int64_t i = 0; i < n; i++) {
                if (!key(a, c, i, k)) continue;
                int slot = -1;
                if (sel_match(k, p0, 40)) slot = 0;
                else if (p0 != p1 && sel_match(k, p1, 40)) slot = 1;
                if (slot < 0) continue;
                hist_[((size_t)2 * c + slot) * kSelCap + (size_t)pos[slot]] = value_of(k);
                pos[slot] += 1;
            }
        }
    }
    void sel_gather_finish(const SelSpec &a) {
        sc.sel_fast_done = 0;
        if (!gather_fits(a)) return;
        for (int c = 0; c < a.ncol; c++) {
            const uint64_t p0 = sc.sel_prefix[2 * c], p1 = sc.sel_prefix[2 * c + 1];
            uint64_t res[2] = {p0, p1};
            for (int slot = 0; slot < 2; slot++) {
                const int hs = (slot == 1 && p0 != p1) ? 1 : 0;
                double b, t;
                sel_gather_layout(cnt_.data(), world_, rank_, 2 * a.ncol, 2 * c + hs, &b, &t);
                const int m = (int)t;
                if (m == 0) continue;
                std::vector<uint64_t> ks((size_t)m);
                for (int e = 0; e < m; e++) ks[(size_t)e] = key_of(hist_[((size_t)2 * c + hs) * kSelCap + (size_t)e]);
                std::sort(ks.begin(), ks.end());
                const int r = (int)sc.sel_rank[2 * c + slot];
                res[slot] = ks[(size_t)(r < m ? r : m - 1)];
            }
            sc.sel_prefix[2 * c] = res[0];
            sc.sel_prefix[2 * c + 1] = res[1];
        }
        sc.sel_fast_done = 1;
    }
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
                      int32_t S, int32_t p, double prior_in, int32_t world, int32_t rank, int32_t six_rounds, allreduce_fn cb,
                      void *user, double *out) {
    CpuBackend be;
    be.world_ = world;
    be.rank_ = rank;
    be.no_gather = six_rounds != 0;
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
int harness_size_factors(const int32_t *counts, int64_t n, int32_t S, int32_t world, int32_t rank, int32_t six_rounds,
                         allreduce_fn cb, void *user, double *sf) {
    CpuBackend be;
    be.world_ = world;
    be.rank_ = rank;
    be.no_gather = six_rounds != 0;
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
