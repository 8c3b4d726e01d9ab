"""a9 / f1: results() post-processing at the benchmark's scale (2 M rows) — Cook's-free path: independent filtering
(50 filtered BH rejection counts + lowess + final BH) and a plain BH; library timers, for rocprofv3 --kernel-trace."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chicdiff_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = hip.HipContext(0)
g = torch.Generator(device=ctx.device)
g.manual_seed(1)
pv = torch.rand(n, dtype=torch.float64, device=ctx.device, generator=g)
pv[::10] = pv[::10] ** 6  # some real signal, so that the rejection counts are not all zero
bm = torch.exp(torch.randn(n, dtype=torch.float64, device=ctx.device, generator=g) * 1.4 + 2.9)
ctx.enable_timing(True)
for name, fn in (("bh_adjust", lambda: ctx.bh_adjust(pv)), ("independent_filtering", lambda: ctx.independent_filtering(bm, pv))):
    fn()
    ts, wall = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = __import__("time").perf_counter()
        fn()
        torch.cuda.synchronize()
        wall.append((__import__("time").perf_counter() - t0) * 1e3)
        ts.append(ctx.kernel_times()[name][0])
    print(f"{name}: n = {n}  device {np.median(ts):.4f} ms  wall {np.median(wall):.4f} ms")
