import sys, time, numpy as np, cProfile, pstats
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n, S = 250000, 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk = ctx.to_device(d["counts"], np.int32)
dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"]
outs = {}
for _ in range(5): ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want, outputs=outs)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want, outputs=outs)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)
