import sys, time, numpy as np
import torch
from chicdiff_amd import hip, synth
n, S = 2_000_000, 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk = ctx.to_device(d["counts"], np.int32)
dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
sf = ctx.size_factors(dk)
grid = [0.0, 0.25, 0.5, 0.75, 1.0]
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.theta_grid(dk, dfm, sf, grid)
    torch.cuda.synchronize(); print("theta grid ms", (time.perf_counter() - t0) * 1e3)
