"""Wall clock of the theta scan (5 design-~1 fits, chicdiff.R:1619-1662) under library options.
usage: python tools/theta_grid_timing.py <rows> <samples> [option=value ...]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n, S = int(sys.argv[1]), int(sys.argv[2])
d = synth.make(n, S)
ctx = hip.HipContext(0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
dk = ctx.to_device(d["counts"], np.int32)
dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
sf = ctx.size_factors(dk)
grid = [0.0, 0.25, 0.5, 0.75, 1.0]
dev0 = ctx.theta_grid(dk, dfm, sf, grid)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev = ctx.theta_grid(dk, dfm, sf, grid)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(sys.argv[1:]), "theta grid ms: median %.3f" % np.median(ts), ["%.2f" % t for t in ts], "deviances", [float(x) for x in dev])
