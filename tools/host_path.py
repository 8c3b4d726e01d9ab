"""Wall time of the host-buffer entry point (what the R shim calls) against the resident path."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n, S = 2_000_000, 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
counts = np.asfortranarray(d["counts"].astype(np.int32))
nf = np.asfortranarray(d["nf"])
want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"]
for rep in range(3):
    t0 = time.perf_counter()
    out, sc = ctx.nbglm_fit_host(counts, nf, d["group"], want=want)
    print("host-buffer fit, 6 output columns: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.nbglm_fit(dk, dn, d["group"], want=want)
    torch.cuda.synchronize(); print("resident fit: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
