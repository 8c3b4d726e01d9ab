// fit_driver.h — launch order of the fit's global steps, written once against a small backend
// interface.  The HIP backend (api.hip) enqueues kernels on a stream and all-reduces device
// buffers through the caller's callback; the CPU test backend (tests/harness/shard_harness.cpp)
// runs plain loops over a row shard and all-reduces host buffers over gloo.  Same driver, same
// state machines (fit_state.h), so the sharded control flow is exercised without a GPU.
//
// Backend concept:
//   int  world() const;                       // ranks sharing the fit
//   int  allreduce(double *buf, int64_t n);   // sum-all-reduce in place (backend memory); 0 = ok
//   double *sums();  int64_t sums_len();      // per-pass trend partials handed to allreduce (HIP: one row of 8 per block)
//   double *hist();                           // select histograms handed to allreduce
//   void trend_init(); void trend_pass(bool fused_step); void trend_step();
//   const FitScalars *sync_scalars();         // make the scalars host-visible (may block)
//   void sel_hist(const SelSpec&, int shift);   // digit histograms for the live prefixes (first round: all keys)
//   void sel_step(const SelSpec&, int shift);   // first round also derives the populations and ranks
//   void sel_finish(const SelSpec&);
//   bool sel_shortcut(const SelSpec&);          // single rank, optional: finish after the second round; true = done
//   bool sel_can_gather();  double *sel_counts(); int rank();   // sharded shortcut (fit_state.h): per-rank count rows,
//   void sel_keep_local_hist(const SelSpec&);   //   this rank's round-2 histogram kept aside before the all-reduce,
//   void sel_gather_counts(const SelSpec&);     //   own row of the count buffer,
//   void sel_gather_place(const SelSpec&);      //   own candidates at their offset in the zeroed hist() buffer,
//   void sel_gather_finish(const SelSpec&);     //   sort + pick; sets sel_fast_done when every list fitted
//   bool sel_gather_done();                     //   did they?  (may answer "yes" without looking, if an overflow is caught later)
#pragma once
#include "fit_state.h"

namespace cd {

enum SelMode { SEL_RESID = 0, SEL_ABSDEV = 1, SEL_SIZEFACTOR = 2 };
struct SelSpec {
    int mode;
    int ncol;  // columns selected simultaneously (1, or S for size factors)
};

// returns 0, or -1 (all-reduce failed) / -2 (state machine did not finish)
template <class B>
int drive_trend(B &be) {
    be.trend_init();
    int passes = 0;
    for (;;) {
        const int batch = passes == 0 ? 24 : 8;  // IRLS passes between two looks at the finished flag (typical total ~20)
        for (int k = 0; k < batch; k++) {
            const bool single = be.world() <= 1;
            be.trend_pass(single);  // single rank: the reducing block also advances the state machine
            if (!single) {
                if (be.allreduce(be.sums(), be.sums_len())) return -1;
                be.trend_step();  // sums the (all-reduced) partials in a fixed order and advances the state machine
            }
        }
        passes += batch;
        if (be.sync_scalars()->finished) return 0;
        if (passes > 11 * 27 + 16) return -2;
    }
}

// exact medians (lower/upper middle order statistics) of `ncol` columns by 12-bit radix select
template <class B>
int drive_select(B &be, const SelSpec &a) {
    // round 0 doubles as the population count: the sum of its (all-reduced) histogram
    for (int r = 0; r < 6; r++) {
        const bool gather = r == 1 && be.world() > 1 && be.sel_can_gather();
        be.sel_hist(a, kSelShifts[r]);
        if (gather) be.sel_keep_local_hist(a);
        if (be.allreduce(be.hist(), (int64_t)a.ncol * 2 * kSelBins)) return -1;
        be.sel_step(a, kSelShifts[r]);
        if (r == 1 && be.world() <= 1 && be.sel_shortcut(a)) break;  // single-rank HIP backend: finishes from the few candidates left
        if (gather) {  // 2 collectives + 1 pass instead of 4 + 4; falls through to the remaining rounds if a list overflows
            be.sel_gather_counts(a);
            if (be.allreduce(be.sel_counts(), (int64_t)be.world_size() * 2 * a.ncol)) return -1;
            be.sel_gather_place(a);
            if (be.allreduce(be.hist(), (int64_t)a.ncol * 2 * kSelCap)) return -1;
            be.sel_gather_finish(a);
            if (be.sel_gather_done()) break;  // HIP backend: assumed (checked with the fit's final scalars: sel_overflow); test backend: looked up
        }
    }
    be.sel_finish(a);
    return 0;
}

}  // namespace cd
