"""Wall time of the 5-point theta grid (chicdiff.R:1619-1662) at 2 M x 8 for several numbers of concurrent fits."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk = ctx.to_device(d["counts"], np.int32)
dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
sf = ctx.size_factors(dk)
grid = [0.0, 0.25, 0.5, 0.75, 1.0]
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
for lanes in (1, 2, 3, 5):
    ctx.set_option("theta_grid_concurrency", lanes)
    ctx.theta_grid(dk, dfm, sf, grid)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.theta_grid(dk, dfm, sf, grid)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"n={n} S={S} theta grid, {lanes} fit(s) in flight: min {min(ts):.2f} ms  ({' '.join('%.2f' % t for t in ts)})")
