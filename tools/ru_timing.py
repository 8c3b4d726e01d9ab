"""f4 region universe at the benchmark's scale (2 M peaks, RUexpand 5): library timers of the two entry points."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chicdiff_amd import hip

n = 2_000_000
ctx = hip.HipContext(0)
g = torch.Generator(device=ctx.device)
g.manual_seed(1)
pb = torch.randint(1000, 800000, (n,), dtype=torch.int32, device=ctx.device, generator=g)
po = pb + torch.randint(2, 60, (n,), dtype=torch.int32, device=ctx.device, generator=g)
chr_of = (torch.arange(0, 840001, device=ctx.device) // 35000).to(torch.int32)
ctx.enable_timing(True)
ctx.region_universe(pb, po, 5, chr_of)
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ru = ctx.region_universe(pb, po, 5, chr_of)
    torch.cuda.synchronize()
    print(f"wall {(time.perf_counter() - t0) * 1e3:.3f} ms, last entry point's kernels: {ctx.kernel_times()}  rows {ru['baitID'].numel()}")
