"""Time the HBM-bound kernels (a2 window sums, a4 offsets, a1 count join) at C3 scale; print achieved GB/s."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip
n, S, F = 2_000_000, 8, 11
ctx = hip.HipContext(0)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(1)
fragN = torch.randint(0, 50, (S, n * F), dtype=torch.int32, device=dev, generator=g)
fragFM = torch.rand((S, n * F), dtype=torch.float64, device=dev, generator=g) + 0.1
rp = torch.arange(0, (n + 1) * F, F, dtype=torch.int64, device=dev)
ctx.enable_timing(True)
def run(name, fn, bytes_):
    fn(); fn()
    ts = []
    for _ in range(5):
        fn(); ts.append(ctx.kernel_times()[name][0])
    ms = float(np.median(ts))
    print(f"{name}: {ms:.3f} ms  {bytes_ / ms / 1e6:.0f} GB/s  ({bytes_/1e6:.0f} MB algorithmic)")
run("window_sums", lambda: ctx.window_sums(fragN, fragFM, rp), (12 * n * F * S + 12 * n * S + 8 * n))
N, FM = ctx.window_sums(fragN, fragFM, rp)
sf = np.ones(S)
out = torch.empty_like(FM)
run("offsets", lambda: ctx.offsets(FM, sf, 0.5, out=out), 16 * n * S)
# count join: 22M RU rows against a 10M-key table
nk = 10_000_000
keys = torch.sort(torch.randint(0, 2**40, (nk,), dtype=torch.int64, device=dev, generator=g)).values
keys = torch.unique(keys); vals = torch.randint(1, 100, (keys.numel(),), dtype=torch.int32, device=dev, generator=g)
qi = torch.randint(0, keys.numel(), (n * F,), device=dev, generator=g)
qk = keys[qi]; miss = torch.rand(n * F, device=dev, generator=g) < 0.5
qk = torch.where(miss, qk + 1, qk)
bait = (qk >> 32).to(torch.int32); oe = (qk & 0xFFFFFFFF).to(torch.int32)
run("count_join", lambda: ctx.count_join(bait, oe, keys, vals), 12 * n * F)

# RU is keyed by baitID (chicdiff.R:425): queries arrive (nearly) sorted, which is what a binary search's caches see
order = torch.argsort(qk)
bait2, oe2 = bait[order].contiguous(), oe[order].contiguous()
run("count_join", lambda: ctx.count_join(bait2, oe2, keys, vals), 12 * n * F)
