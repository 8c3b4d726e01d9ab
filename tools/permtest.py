import sys, numpy as np
sys.path.insert(0,'.')
import torch
from chicdiff_amd import hip, synth
n,S=int(sys.argv[1]),8
d=synth.make(n,S)
ctx=hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
want=["dispGeneEst","dispersion","pvalue","baseMean"]
out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
perm = torch.randperm(n, device=ctx.device, generator=torch.Generator(device=ctx.device).manual_seed(0))
out2, sc2 = ctx.nbglm_fit(dk[:, perm].contiguous(), dn[:, perm].contiguous(), d["group"], want=want)
print("trend", sc["trendCoef"], sc2["trendCoef"], np.abs(sc["trendCoef"]-sc2["trendCoef"])/sc["trendCoef"])
for k in want:
    a=out[k][perm].cpu().numpy(); b=out2[k].cpu().numpy()
    ok=~np.isnan(a)
    r=np.abs(a[ok]-b[ok])/np.maximum(np.abs(a[ok]),1e-300)
    print(k, "max rel", r.max(), "n>1e-12", (r>1e-12).sum(), "n>1e-9", (r>1e-9).sum(), "n>1e-6", (r>1e-6).sum())
