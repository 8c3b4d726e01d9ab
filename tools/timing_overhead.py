"""What the per-stage HIP events of chicdiff_hip_enable_timing cost per fused call: off / every stage (1) / the three fit kernels
only (2, what bench.py's timed region uses)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
for n in (2000000, 250000):
    S = 8
    d = synth.make(n, S)
    ctx = hip.HipContext(0)
    dk = ctx.to_device(d["counts"], np.int32)
    dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
    want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"]
    outs = {}
    for _ in range(3): ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want, outputs=outs)
    for rep in range(2):
        for timing in (0, 1, 2):
            ctx.enable_timing(timing)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want, outputs=outs)
                if timing: ctx.kernel_times()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
            print(f"n={n} timing={timing}: {dt*1e3:.3f} ms/step")
    del ctx
