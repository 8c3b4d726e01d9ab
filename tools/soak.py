"""Soak: many fits of changing shape on one context (workspace regrowth, theta grid lanes, host path), watching the
device's free memory for leaks and the results for run-to-run identity."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
ctx = hip.HipContext(0)
rng = np.random.default_rng(0)
free0 = None
ref = {}
shapes = [(5000, 4), (120000, 8), (300, 6), (64, 16), (40000, 3), (250000, 8), (1, 8), (70000, 33)]
for it in range(120):
    n, S = shapes[it % len(shapes)]
    d = synth.make(n, S)
    g = d["group"] if S > 3 else np.zeros(S, np.int32) if S == 3 else d["group"]
    dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
    try:
        out, sc = ctx.nbglm_fit(dk, dn, g, want=["pvalue", "dispersion"])
    except hip.ChicdiffHipError as e:
        print("shape", (n, S), "->", e)
        continue
    key = (n, S)
    p = out["dispersion"].cpu().numpy()
    if key in ref:
        assert np.array_equal(ref[key], p, equal_nan=True), key
    ref[key] = p
    if it % 8 == 3 and n > 1000:
        fm = d["nf"] * (d["mu"][:, None] / S)
        keep = d["counts"].sum(1) > 0
        dev = ctx.theta_grid(ctx.to_device(d["counts"][keep], np.int32), ctx.to_device(fm[keep], np.float64), np.ones(S), [0.0, 0.3, 0.6, 1.0])
        assert np.all(np.isfinite(dev)), dev
    if it % 8 == 5 and n > 1000:
        r, _ = ctx.nbglm_fit_host(d["counts"], d["nf"], g, want=["dispersion"])
        assert np.array_equal(r["dispersion"], p, equal_nan=True)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 40:
        free0 = free
    if it > 40 and it % 20 == 0:
        print(f"iteration {it}: free device memory {free / 2**20:.0f} MiB (at iteration 40: {free0 / 2**20:.0f} MiB)")
assert free0 - free < 64 * 2**20, "device memory keeps shrinking"
print("soak OK")
