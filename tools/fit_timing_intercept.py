"""Per-stage HIP-event times of one design-~1 fit (what each theta of the grid runs): python tools/fit_timing_intercept.py [rows] [samples]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
g = np.zeros(S, dtype=np.int32)
out = {}
for _ in range(2):
    ctx.nbglm_fit(dk, dn, g, want=["deviance"], outputs=out)
ctx.enable_timing(True)
acc = {}
reps = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctx.nbglm_fit(dk, dn, g, want=["deviance"], outputs=out)
    for k, (ms, c) in ctx.kernel_times().items():
        acc[k] = acc.get(k, 0) + ms / reps
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"n={n} S={S} design ~1 fit wall {dt*1e3:.3f} ms  ->", " ".join(f"{k}={v:.3f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
