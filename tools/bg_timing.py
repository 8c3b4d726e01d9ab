"""Time the fragment-background kernel (a3) alone at the end-to-end leg's scale: 2 M regions x 11 fragments, 8 replicates, an
840 001-fragment map; FullMean only and all three outputs.  usage: python tools/bg_timing.py [S] [reps]"""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, F, nid = 2_000_000, 11, 840_001
ctx = hip.HipContext(0)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(3)
pb = torch.sort(torch.randint(1000, 800000, (n,), dtype=torch.int64, device=dev, generator=g)).values
po = pb + torch.randint(2, 60, (n,), dtype=torch.int64, device=dev, generator=g)
bait = pb.repeat_interleave(F).to(torch.int32)
oe = (po.repeat_interleave(F) + torch.arange(-5, 6, device=dev).repeat(n)).to(torch.int32)
sj = torch.exp(torch.randn((S, nid), dtype=torch.float64, device=dev, generator=g) * 0.3)
si = torch.exp(torch.randn((S, nid), dtype=torch.float64, device=dev, generator=g) * 0.3)
tblb = torch.randint(0, 6, (S, nid), dtype=torch.int32, device=dev, generator=g)
tlb = torch.randint(0, 6, (S, nid), dtype=torch.int32, device=dev, generator=g)
T = torch.exp(torch.randn((S, 6, 6), dtype=torch.float64, device=dev, generator=g) * 0.5 - 3.0)
midsum = torch.arange(nid, device=dev, dtype=torch.int64) * 8000 + 4000
distfun = np.zeros((S, 10))
for j in range(S):
    fit = np.array([14.0 + 0.1 * j, -1.6, 0.05, -0.003]); ends = np.array([np.log(10000.0), np.log(1.5e6)])
    beta = fit[1] + 2 * fit[2] * ends + 3 * fit[3] * ends ** 2
    alpha = fit[0] + (fit[1] - beta) * ends + fit[2] * ends ** 2 + fit[3] * ends ** 3
    distfun[j] = [*fit, alpha[0], beta[0], alpha[1], beta[1], ends[0], ends[1]]
ctx.enable_timing(True)
nru = bait.numel()
for only in (True, False):
    ts = []
    for _ in range(reps + 2):
        out = ctx.fragment_background(bait, oe, 0, midsum, sj, si, tblb, tlb, T, distfun, only_fullmean=only)
        ts.append(ctx.kernel_times()["fragment_background"][0])
        del out
    ms = float(np.median(ts[2:]))
    by = nru * (8 + 8 * S * (1 if only else 3))
    print(f"fragment_background S={S} rows={nru} {'FullMean only' if only else 'Bmean, Tmean, FullMean'}: {ms:.3f} ms, {by / ms / 1e6:.0f} GB/s algorithmic ({by / 1e6:.0f} MB)")
