import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from chicdiff_amd import hip, synth
from oracle import oracle
import np_twin
n, S = 2_000_000, 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=["dispGeneEst", "baseMean", "dispGeneIter"])
got = {k: v.cpu().numpy() for k, v in out.items()}
ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=16)
nz = ref["allZero"] == 0
print("GPU trend", sc["trendCoef"], sc["trendOuterIter"], "oracle", ref["trendCoef"], ref["trendOuterIter"])
rel = np.abs(got["dispGeneEst"] - ref["dispGeneEst"]) / ref["dispGeneEst"]
use = nz & (ref["dispGeneEst"] > 1e-6)
print("dispGeneEst rel diff quantiles (useForFit rows):", np.quantile(rel[use], [0.5, 0.99, 0.9999, 1.0]))
print("rows with rel diff > 1e-9:", (rel[use] > 1e-9).sum(), " > 1e-6:", (rel[use] > 1e-6).sum(), " > 1e-3:", (rel[use] > 1e-3).sum())
bad = np.nonzero(use & (rel > 1e-3))[0][:10]
for i in bad: print("  ", i, got["dispGeneEst"][i], ref["dispGeneEst"][i], got["dispGeneIter"][i], ref["dispGeneIter"][i], d["counts"][i])
useg = nz & (got["dispGeneEst"] > 1e-6)
print("useForFit membership differs in", (use != useg).sum(), "rows")
# oracle's trend routine on the GPU's gene estimates and vice versa
cg, itg, _ = oracle.parametric_dispersion_fit(got["baseMean"][useg], got["dispGeneEst"][useg])
co, ito, _ = oracle.parametric_dispersion_fit(ref["baseMean"][use], ref["dispGeneEst"][use])
print("oracle trend on GPU estimates:", cg, itg, " on oracle estimates:", co, ito)
