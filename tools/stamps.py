import sys, numpy as np
sys.path.insert(0, '.')
import os
os.environ["CHICDIFF_DISP_STAMPS"] = "gpurun_out/stamps.bin"
from chicdiff_amd import hip, synth
d = synth.make(2_000_000, 8)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
ctx.nbglm_fit(dk, dn, d["group"])
raw = np.fromfile("gpurun_out/stamps.bin", dtype=np.uint64)
pos = 0
while pos < len(raw):
    kind, nw = int(raw[pos]), int(raw[pos + 1]); pos += 2
    st = raw[pos:pos + nw * 4].reshape(nw, 4).astype(np.int64); pos += nw * 4
    t0 = st[:, 0].min()
    us = lambda x: (x - t0) / 100.0  # s_memrealtime: 100 MHz
    ran = st[:, 2] > 0
    print("kernel", "MAP" if kind else "gene", "waves", nw, "ran", ran.sum())
    print("  start  us: min %.1f med %.1f max %.1f" % tuple(np.percentile(us(st[ran, 0]), [0, 50, 100])))
    qe = st[:, 1] > 0
    print("  q-empty us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(us(st[qe, 1]), [0, 10, 50, 90, 100])))
    print("  exit   us: min %.1f p10 %.1f med %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(us(st[ran, 2]), [0, 10, 50, 90, 99, 100])))
    print("  active lanes at q-empty: mean %.1f" % st[qe, 3].mean())
    dr = us(st[qe, 2]) - us(st[qe, 1])
    print("  drain (exit - q-empty) us: med %.1f p90 %.1f max %.1f" % tuple(np.percentile(dr, [50, 90, 100])))
