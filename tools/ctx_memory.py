"""Diagnostic: free device memory across context lifetimes (create / fit / results() / destroy) — does destroying a context give everything back?  usage on the GPU box: python tools/ctx_memory.py"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
free = lambda: (torch.cuda.synchronize(), torch.cuda.mem_get_info()[0])[1]
L = hip.load_library()
d = synth.make(3000, 8)
print("start", free() >> 10)
for it in range(10):
    c = hip.HipContext(0, use_torch_stream=False)
    a = free()
    if it >= 2:
        dk, dn = c.to_device(d["counts"], np.int32), c.to_device(d["nf"], np.float64)
        out, sc = c.nbglm_fit(dk, dn, d["group"])
        if it >= 4:
            padj, info = c.independent_filtering(out["baseMean"], out["pvalue"])
            del padj
        del dk, dn, out
        torch.cuda.empty_cache()
    b = free()
    c.close()
    print(it, "after create", a >> 10, "after work", b >> 10, "after close", free() >> 10, "KiB")
