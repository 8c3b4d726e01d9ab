import sys, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip
ctx = hip.HipContext(0)
rng = np.random.default_rng(0)
x = np.exp(rng.uniform(-30, 30, 1000000))
dx = torch.as_tensor(x).to(ctx.device)
for op in (6, 7, 2):
    r = ctx.selftest_math(op, dx).cpu().numpy()
    err = np.abs(r * x - 1)
    print("op", op, "max |x*rcp-1|", err.max(), "median", np.median(err))
