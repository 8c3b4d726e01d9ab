"""a1 count join at the benchmark's scale (22 M sorted RU rows against a 10 M-row count table), timed by the library's
HIP events; for rocprofv3 --pmc passes (python3 tools/join_timing.py [reps] [keys_per_query])."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chicdiff_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 5 / 11
ctx = hip.HipContext(0)
dev = ctx.device
g = torch.Generator(device=dev)
g.manual_seed(1)
n, F = 2_000_000, 11
keys = torch.unique(torch.randint(0, 2 ** 40, (int(n * F * dens),), dtype=torch.int64, device=dev, generator=g))
vals = torch.randint(1, 100, (keys.numel(),), dtype=torch.int32, device=dev, generator=g)
if len(sys.argv) > 3 and sys.argv[3] == "regions":  # RU as Chicdiff builds it: runs of F consecutive other-end IDs around a peak
    peak = keys[torch.randint(0, keys.numel(), (n,), device=dev, generator=g)]
    qk = torch.sort((peak[:, None] + torch.arange(-(F // 2), F // 2 + 1, device=dev)[None, :]).reshape(-1)).values
else:
    qk = keys[torch.randint(0, keys.numel(), (n * F,), device=dev, generator=g)]
    qk = torch.sort(torch.where(torch.rand(n * F, device=dev, generator=g) < 0.5, qk + 1, qk)).values
bait, oe = (qk >> 32).to(torch.int32), (qk & 0xFFFFFFFF).to(torch.int32)
ctx.enable_timing(True)
ctx.count_join(bait, oe, keys, vals)
ts = []
for _ in range(reps):
    ctx.count_join(bait, oe, keys, vals)
    ts.append(ctx.kernel_times()["count_join"][0])
nbytes = 12 * n * F + 12 * keys.numel()
ms = float(np.median(ts))
print(f"count_join: {n * F} queries x {keys.numel()} keys  {ms:.4f} ms  {nbytes / ms / 1e6:.0f} GB/s of algorithmic bytes")
