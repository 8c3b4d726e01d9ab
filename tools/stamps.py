"""Diagnostic: per-wave timestamps and tick counts of the two line-search launches (needs the DIAG build:
make -C chicdiff_amd/csrc DIAG=1).  Usage on the GPU box: python tools/stamps.py [rows] [samples]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["CHICDIFF_DISP_STAMPS"] = "gpurun_out/stamps.bin"
os.environ["CHICDIFF_HIP_LIB"] = os.environ.get("STAMPS_LIB", "chicdiff_amd/lib/libchicdiff_hip_diag.so")
from chicdiff_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.make(n, S)
print('n', n, 'S', S)
ctx = hip.HipContext(0)
for kv in sys.argv[3:]:  # option=value pairs of chicdiff_hip_set_option
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
ctx.nbglm_fit(dk, dn, d["group"])
raw = np.fromfile("gpurun_out/stamps.bin", dtype=np.uint64)
pos = 0
K = 34
while pos < len(raw):
    kind, nw = int(raw[pos]), int(raw[pos + 1]); pos += 2
    st = raw[pos:pos + nw * K].reshape(nw, K).astype(np.int64); pos += nw * K
    t0 = st[:, 0].min()
    us = lambda x: (x - t0) / 100.0  # s_memrealtime: 100 MHz
    ran = st[:, 2] > 0
    print("kernel", "MAP" if kind else "gene", "waves", nw, "ran", ran.sum())
    print("  start  us: min %.1f med %.1f max %.1f" % tuple(np.percentile(us(st[ran, 0]), [0, 50, 100])))
    qe = st[:, 1] > 0
    print("  q-empty us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(us(st[qe, 1]), [0, 10, 50, 90, 100])))
    print("  exit   us: min %.1f p10 %.1f med %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(us(st[ran, 2]), [0, 10, 50, 90, 99, 100])))
    print("  live rows at q-empty: mean %.1f max %d" % (st[qe, 3].mean(), st[qe, 3].max()))
    late = np.argsort(-st[:, 2])[:8]  # the waves that leave last: what they did after their queue ran dry
    for wv in late:
        print("    late wave %4d: q-empty %.1f exit %.1f us, live rows at q-empty %d, ticks after q-empty row/spread/burst %d/%d/%d, all ticks %d, lean %d" %
              (wv, us(st[wv, 1]), us(st[wv, 2]), st[wv, 3], st[wv, 4], st[wv, 5], st[wv, 6], st[wv, 7], st[wv, 33]))
    dr = us(st[qe, 2]) - us(st[qe, 1])
    print("  drain (exit - q-empty) us: med %.1f p90 %.1f max %.1f" % tuple(np.percentile(dr, [50, 90, 100])))
    tk = st[qe, 4:7]
    print("  ticks after q-empty (row-per-lane / spread / burst): mean %.1f / %.1f / %.1f, max %d / %d / %d" % (*tk.mean(0), *tk.max(0)))
    tot = tk.sum(1)
    ok = tot > 20
    print("  us per drain tick: med %.2f (waves with > 20 drain ticks)" % np.median(dr[ok] / tot[ok]))
    # least squares: drain = a*row + b*spread + c*burst
    coef = np.linalg.lstsq(tk[ok].astype(float), dr[ok], rcond=None)[0]
    print("  fitted us per tick: row-per-lane %.2f, spread %.2f, burst %.2f" % tuple(coef))
    cy = st[qe, 8:11].astype(float)
    with np.errstate(divide="ignore", invalid="ignore"):
        per = np.nansum(cy, 0) / np.maximum(tk.sum(0), 1)
    print("  s_memtime cycles per tick (tick start to next tick start): row-per-lane %.0f, spread %.0f, burst %.0f  (100 MHz clock? ratio to us: %.1f)" % (*per, np.nansum(cy) / max(dr.sum(), 1e-9)))
    bulk = us(st[qe, 1]) / np.maximum(st[qe, 7] - tot, 1)
    print("  us per tick before q-empty: med %.2f; total ticks per wave: med %d" % (np.median(bulk), np.median(st[qe, 7])))
    if os.environ.get("STAMPS_SPLIT3"):  # (experiment: every third workgroup's waves run at s_setprio(3): option line_search_prio = 100)
        wv = np.nonzero(qe)[0]
        hi = (wv // 2) % 3 == 0
        for nm, m in (("workgroups 0 mod 3", hi), ("the others", ~hi)):
            print("    %s: us per bulk tick med %.2f, ticks per wave med %d, q-empty med %.1f us, exit med %.1f max %.1f" %
                  (nm, np.median(bulk[m]), np.median(st[qe, 7][m]), np.median(us(st[qe, 1])[m]), np.median(us(st[qe, 2])[m]), us(st[qe, 2])[m].max()))
    sec = st[qe, 11:16].astype(float)
    tot_sec = sec[:, :4].sum()
    print("  bulk ticks (queue not empty), s_memtime cycles per tick: refill %.0f, choose point %.0f, evaluate %.0f, state machine %.0f (of %.0f); shares %.1f / %.1f / %.1f / %.1f %%" % (
        *(sec[:, :4].sum(0) / max(sec[:, 4].sum(), 1)), tot_sec / max(sec[:, 4].sum(), 1), *(100 * sec[:, :4].sum(0) / max(tot_sec, 1))))
    ev = st[qe, 16:20].astype(float)
    print("    inside evaluate (row per lane): row constants %.0f, prefix table %.0f, samples %.0f, finish %.0f cycles per tick" % tuple(ev.sum(0) / max(sec[:, 4].sum(), 1)))
    sp = st[qe, 20:29].astype(float)
    nsp = max(sp[:, 7].sum(), 1)
    print("    samples-across-lanes ticks (%d per wave), cycles per tick: tick start -> evaluate %.0f | owner walk %.0f, row constants %.0f, prefix + sample + exchange store %.0f, fold %.0f, finish %.0f, pick-up %.0f | evaluate end -> tick end %.0f" % (
        nsp / max(qe.sum(), 1), sp[:, 6].sum() / nsp, *(sp[:, :6].sum(0) / nsp), sp[:, 8].sum() / nsp))
    rj = st[:, 29:33].astype(float).sum(0)
    print("    line-search steps rejected (Armijo): %.1f %% of %d in samples-across-lanes ticks, %.1f %% of %d in bulk ticks" % (100 * rj[0] / max(rj[1], 1), rj[1], 100 * rj[2] / max(rj[3], 1), rj[3]))
