import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
S = 8
ctx = hip.HipContext(0)
for maxit in (100, 30, 12):
    for n in (1_000_000, 4_000_000):
        d = synth.make(n, S)
        dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
        opts = hip.default_opts(maxit=maxit)
        out = {}
        ctx.enable_timing(False)
        ctx.nbglm_fit(dk, dn, d["group"], outputs=out, opts=opts)
        ctx.enable_timing(True)
        acc = {}
        for _ in range(3):
            ctx.nbglm_fit(dk, dn, d["group"], outputs=out, opts=opts)
            for k, (ms, c) in ctx.kernel_times().items():
                acc[k] = acc.get(k, 0) + ms / 3
        print(f"maxit={maxit} n={n}: disp_gene={acc['disp_gene']:.3f} disp_map={acc['disp_map']:.3f} wald_irls={acc['wald_irls']:.3f}")
