"""Quick A/B timing of one fit on the GPU box: prints per-kernel HIP-event times (ms)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
for kv in sys.argv[3:]:  # option=value pairs of chicdiff_hip_set_option
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
out = {}
for _ in range(2):
    ctx.nbglm_fit(dk, dn, d["group"], outputs=out)
ctx.enable_timing(True)
acc = {}
reps = 3
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctx.nbglm_fit(dk, dn, d["group"], outputs=out)
    for k, (ms, c) in ctx.kernel_times().items():
        acc[k] = acc.get(k, 0) + ms / reps
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"n={n} S={S} fit wall {dt*1e3:.3f} ms  ->", " ".join(f"{k}={v:.3f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
