import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n, S = 2_000_000, 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk = ctx.to_device(d["counts"], np.int32); dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
out = {}
for timing in (False, True, False):
    ctx.enable_timing(timing)
    for _ in range(2): ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    torch.cuda.synchronize(); print("timing", timing, "ms/step", (time.perf_counter() - t0) / 10 * 1e3)
