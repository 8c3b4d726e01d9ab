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
import ctypes as C
k = counts; f = nf
res = {name: np.empty(n) for name in want}
out = hip.Out()
for name in want:
    setattr(out, name, res[name].ctypes.data)
g = (C.c_int32 * S)(*[int(x) for x in d["group"]])
sc = hip.Scalars()
for threads in (8, 4, 16, 12):
    ctx.set_option("host_copy_threads", threads)
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        rc = ctx.lib.chicdiff_hip_nbglm_fit(ctx.h, k.ctypes.data, f.ctypes.data, n, S, g, None, C.byref(out), C.byref(sc))
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    print("host-buffer fit (C ABI, 6 output columns), %2d copy threads: min %.1f ms, median %.1f ms  (%s)" % (threads, min(ts), np.median(ts), " ".join("%.1f" % t for t in ts)))
print("cpu count", __import__("os").cpu_count())
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.nbglm_fit(dk, dn, d["group"], want=want)
    torch.cuda.synchronize(); print("resident fit: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
