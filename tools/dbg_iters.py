import sys, numpy as np
sys.path.insert(0, '.')
from chicdiff_amd import hip, synth
from oracle import oracle
d = synth.make(20000, 8)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=["dispGeneIter","dispGeneEst","dispIter","betaIter","dispersion"])
got = {k: v.cpu().numpy() for k, v in out.items()}
ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
nz = ref["allZero"] == 0
for k in ["dispGeneIter", "dispIter", "betaIter"]:
    diff = got[k][nz] - ref[k][nz]
    u, c = np.unique(diff, return_counts=True)
    print(k, dict(zip(u.tolist(), c.tolist())))
bad = np.nonzero(nz & (got["dispGeneIter"] != ref["dispGeneIter"]))[0][:15]
for i in bad:
    print(i, got["dispGeneIter"][i], ref["dispGeneIter"][i], got["dispGeneEst"][i], ref["dispGeneEst"][i], ref["dispInit"][i], d["counts"][i])
