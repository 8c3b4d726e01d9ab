import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from chicdiff_amd import hip, synth
from oracle import oracle
rng = np.random.default_rng(17)
n, S = 4000, 6
d = synth.make(n, S)
counts, nf = d["counts"].copy(), d["nf"].copy()
big = rng.choice(n, 300, replace=False)
counts[big] = rng.integers(2 ** 20, 2 ** 30, size=(300, S))
counts[big[:100], 0] = 0
nf[big[100:200]] *= np.exp(rng.normal(0, 2.0, size=(100, S)))
nf /= np.exp(np.log(nf).mean(axis=1, keepdims=True))
group = d["group"]
ctx = hip.HipContext(0)
want = ["dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispersion", "dispIter", "dispOutlier", "baseMean", "log2FoldChange", "betaIter", "betaConv", "intercept", "lfcSE", "deviance"]
out, sc = ctx.nbglm_fit(ctx.to_device(counts, np.int32), ctx.to_device(nf, np.float64), group, want=want)
got = {k: v.cpu().numpy() for k, v in out.items()}
ref = oracle.nbglm_fit(counts, nf, group)
print("trend", sc["trendCoef"], ref["trendCoef"], "priorvar", sc["dispPriorVar"], ref["dispPriorVar"], sc["varLogDispEsts"], ref["varLogDispEsts"])
nz = ref["allZero"] == 0
rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
isbig = np.zeros(n, bool); isbig[big] = True
for k in ["dispGeneEst", "dispFit", "dispMAP", "dispersion"]:
    r = rel(got[k], ref[k]); off = nz & (r > 1e-6)
    print(k, "off", off.sum(), "of which big", (off & isbig).sum(), "max", np.nanmax(r[nz]))
off = nz & (rel(got["dispGeneEst"], ref["dispGeneEst"]) > 1e-6)
for i in np.nonzero(off)[0][:10]:
    print(i, isbig[i], counts[i], "gene", got["dispGeneEst"][i], ref["dispGeneEst"][i], "iter", got["dispGeneIter"][i], ref["dispGeneIter"][i])
np.savez("gpurun_out/extreme.npz", counts=counts, nf=nf, group=group, big=big, **{"g_"+k: v for k, v in got.items()})
