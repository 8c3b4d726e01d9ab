"""Diagnostic: per-row iteration counts and estimates of two library builds on the same matrix.
usage: python tools/cmp_iters.py <rows> <samples> lib_a.so lib_b.so"""
import os, subprocess, sys
import numpy as np
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, '.')
from chicdiff_amd import hip, synth
n, S, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
want = ["dispGeneEst", "dispGeneIter", "dispMAP", "dispIter", "dispFit", "dispersion", "dispOutlier"]
o, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
np.savez(out, **{k: v.cpu().numpy() for k, v in o.items()}, trend=np.array(sc["trendCoef"]))
'''
n, S, la, lb = sys.argv[1:5]
res = []
for i, lib in enumerate((la, lb)):
    out = f"/tmp/cmp_iters_{i}.npz"
    subprocess.run([sys.executable, "-c", CHILD, n, S, out], env=dict(os.environ, CHICDIFF_HIP_LIB=os.path.abspath(lib)), check=True)
    res.append(np.load(out))
a, b = res
print("trend", a["trend"], b["trend"])
for it, est in (("dispGeneIter", "dispGeneEst"), ("dispIter", "dispMAP")):
    d = b[it].astype(np.int64) - a[it].astype(np.int64)
    print(it, "rows with other counts:", int((d != 0).sum()), "max +", int(d.max()), "max -", int(d.min()), "sum a", int(a[it].sum()), "sum b", int(b[it].sum()))
    for i in np.argsort(-np.abs(d))[:8]:
        print("   row", int(i), it, int(a[it][i]), "->", int(b[it][i]), est, a[est][i], "->", b[est][i], "dispFit", a["dispFit"][i], "gene", a["dispGeneEst"][i], "outlier", int(a["dispOutlier"][i]))
    rel = np.abs(a[est] - b[est]) / np.maximum(np.abs(a[est]), 1e-300)
    print("  ", est, "rows beyond 1e-9:", int(np.nansum(rel > 1e-9)), "beyond 1e-6:", int(np.nansum(rel > 1e-6)))
