import sys, numpy as np
sys.path.insert(0, '.')
from chicdiff_amd import hip, synth
from oracle import oracle
d = synth.make(20000, 8)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
nz = ref["allZero"] == 0
prev = None
for rep in range(4):
    out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=["dispGeneEst", "dispGeneIter", "dispersion", "pvalue"])
    g = {k: v.cpu().numpy().copy() for k, v in out.items()}
    bad = np.nonzero(nz & (np.abs(g["dispGeneEst"] - ref["dispGeneEst"]) > 1e-6 * ref["dispGeneEst"]))[0]
    same = None if prev is None else all(np.array_equal(prev[k], g[k], equal_nan=True) for k in g)
    print("rep", rep, "rows off vs oracle:", len(bad), "identical to previous run:", same)
    if rep == 0:
        for i in bad[:8]:
            print("   ", i, g["dispGeneEst"][i], ref["dispGeneEst"][i], g["dispGeneIter"][i], ref["dispGeneIter"][i])
    prev = g
