import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from chicdiff_amd import hip, synth
from oracle import oracle
d = synth.make(2500, 8)
rng = np.random.default_rng(0)
k = d["counts"].copy()
spike = rng.choice(2500, 40, replace=False)
k[spike, rng.integers(0, 8, 40)] += rng.integers(300, 3000, 40).astype(np.int32)
ctx = hip.HipContext(0)
want = ["dispGeneEst","dispGeneIter","dispersion","dispIter","log2FoldChange","pvalue","betaIter","betaConv","deviance","dispMAP","dispOutlier"]
out, sc = ctx.nbglm_fit(ctx.to_device(k, np.int32), ctx.to_device(d["nf"], np.float64), d["group"], want=want)
got = {a: b.cpu().numpy() for a, b in out.items()}
ref = oracle.nbglm_fit(k, d["nf"], d["group"])
nz = ref["allZero"] == 0
bad = np.nonzero(nz & (np.abs(got["pvalue"] - ref["pvalue"]) > 1e-6 * np.abs(ref["pvalue"])))[0]
print("bad rows", bad, "spiked?", np.isin(bad, spike))
for i in bad:
    print(i, k[i], {a: (got[a][i], ref[a][i]) for a in want})
print(sc, ref["trendCoef"], ref["status"])
