"""Survey of the full-size parity residuals (GPU box): which rows differ between the HIP path and the oracle at 2 M x 8, and why.
Usage: python tools/parity_survey.py [rows] [samples] — prints the tables tests/test_gpu_parity.py::test_full_size_* assert on."""
import os, sys, json
import numpy as np
sys.path.insert(0, '.')
from chicdiff_amd import hip, synth
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
want = ["log2FoldChange", "pvalue", "dispersion", "dispGeneEst", "dispMAP", "dispFit", "dispOutlier", "betaConv", "betaIter", "dispIter", "lfcSE"]
T = min(16, os.cpu_count() or 1)
ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=T)
out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
got = {k: v.cpu().numpy() for k, v in out.items()}
opts = hip.default_opts(trendCoef=ref["trendCoef"], dispPriorVar=ref["dispPriorVar"])
out2, sc2 = ctx.nbglm_fit(dk, dn, d["group"], want=want, opts=opts)
pin = {k: v.cpu().numpy() for k, v in out2.items()}
live = ref["allZero"] == 0
print("varLogDispEsts gpu/pinned/oracle", sc["varLogDispEsts"], sc2["varLogDispEsts"], ref["varLogDispEsts"])
print("betaConv==0 rows: oracle", int((live & (ref["betaConv"] == 0)).sum()), "gpu", int((live & (got["betaConv"] == 0)).sum()), "both", int((live & (ref["betaConv"] == 0) & (got["betaConv"] == 0)).sum()))
# (A) pinned comparison: every row
for k in ("dispGeneEst", "dispMAP", "dispersion", "log2FoldChange", "lfcSE", "pvalue"):
    r = rel(pin[k][live], ref[k if k != "log2FoldChange" else "log2FoldChange"][live])
    print(f"pinned {k}: >1e-6: {(r > 1e-6).sum()}  >1e-4: {(r > 1e-4).sum()}  >1e-2: {(r > 1e-2).sum()}  max {np.nanmax(r):.3e}")
rg = rel(pin["dispGeneEst"], ref["dispGeneEst"]); rm = rel(pin["dispMAP"], ref["dispMAP"]); rd = rel(pin["dispersion"], ref["dispersion"])
gene_off = np.flatnonzero(live & (rg > 1e-6))
map_off = np.flatnonzero(live & (rm > 1e-6))
disp_off = np.flatnonzero(live & (rd > 1e-6))
print("gene-wise off rows", len(gene_off), "MAP off rows", len(map_off), "of which also gene-off", len(np.intersect1d(gene_off, map_off)), "final dispersion off", len(disp_off))
only_map = np.setdiff1d(map_off, gene_off)
# arbitrate the MAP stage for rows whose gene-wise estimates agree
if len(only_map):
    arb = oracle.arbitrate_disp(d["counts"], d["nf"], d["group"], only_map, dict(dispGeneEst=ref["dispGeneEst"], dispFit=ref["dispFit"], dispPriorVar=ref["dispPriorVar"]), stage="map")
    eg, eo = rel(pin["dispMAP"][only_map], arb), rel(ref["dispMAP"][only_map], arb)
    print("MAP-only off rows arbitrated: gpu right", int((eg <= 1e-6).sum()), "oracle right", int((eo <= 1e-6).sum()), "neither", int(((eg > 1e-6) & (eo > 1e-6)).sum()))
    for i, a, x, y in list(zip(only_map, arb, eg, eo))[:12]:
        print(f"   row {i}: gpu {pin['dispMAP'][i]:.9e} oracle {ref['dispMAP'][i]:.9e} arb {a:.9e} iters gpu {pin['dispIter'][i]} oracle {ref['dispIter'][i]}")
# lfc / p off rows in the pinned comparison: are they all in disp_off?
for k in ("log2FoldChange", "pvalue"):
    r = rel(pin[k], ref[k])
    off = np.flatnonzero(live & (r > 1e-6) & (ref["betaConv"] == 1) & (pin["betaConv"] == 1))
    print(f"pinned {k} off (>1e-6, both converged): {len(off)}; inside dispersion-off set: {np.isin(off, disp_off).sum()}; abs diff max {np.abs(pin[k][off] - ref[k][off]).max() if len(off) else 0:.3e}")
    rest = off[~np.isin(off, disp_off)]
    for i in rest[:10]:
        print(f"   row {i}: {k} gpu {pin[k][i]:.9e} oracle {ref[k][i]:.9e} disp rel {rd[i]:.2e} betaIter {pin['betaIter'][i]}/{ref['betaIter'][i]} counts {d['counts'][i]}")
small = live & (np.abs(ref["log2FoldChange"]) <= 1e-2) & (ref["betaConv"] == 1) & (pin["betaConv"] == 1)
print("small-lfc rows", int(small.sum()), "max abs diff", np.abs(pin["log2FoldChange"][small] - ref["log2FoldChange"][small]).max(), "outside disp_off:",
      np.abs(pin["log2FoldChange"][small & ~np.isin(np.arange(n), disp_off)] - ref["log2FoldChange"][small & ~np.isin(np.arange(n), disp_off)]).max())
# betaConv == 0 rows
nc = np.flatnonzero(live & ((ref["betaConv"] == 0) | (pin["betaConv"] == 0) | (ref["betaIter"] >= 100) | (pin["betaIter"] >= 100)))
print("rows through the optim fallback on either side:", len(nc))
for i in nc[:40]:
    print(f"   row {i}: lfc gpu {pin['log2FoldChange'][i]:.10e} oracle {ref['log2FoldChange'][i]:.10e} rel {rel(pin['log2FoldChange'][i], ref['log2FoldChange'][i]):.2e} p rel {rel(pin['pvalue'][i], ref['pvalue'][i]):.2e} "
          f"SE rel {rel(pin['lfcSE'][i], ref['lfcSE'][i]):.2e} conv {pin['betaConv'][i]}/{ref['betaConv'][i]} iter {pin['betaIter'][i]}/{ref['betaIter'][i]} disp rel {rd[i]:.1e}")
# (B) free vs free: rows beyond 1e-3 -> distance from the outlier threshold
rf = rel(got["dispersion"], ref["dispersion"])
far = np.flatnonzero(live & (rf > 1e-3))
thr_g = np.log(got["dispFit"]) + 2 * np.sqrt(sc["varLogDispEsts"]); thr_o = np.log(ref["dispFit"]) + 2 * np.sqrt(ref["varLogDispEsts"])
shift = np.abs(thr_g - thr_o)
print("free vs free: rows beyond 1e-3:", len(far), " trend-threshold shift: median", np.median(shift[live]), "max", shift[live].max())
for i in far[:30]:
    lg_g, lg_o = np.log(got["dispGeneEst"][i]), np.log(ref["dispGeneEst"][i])
    print(f"   row {i}: disp gpu {got['dispersion'][i]:.6e} oracle {ref['dispersion'][i]:.6e} outlier gpu/oracle {got['dispOutlier'][i]}/{ref['dispOutlier'][i]} "
          f"log dispGene - thr: gpu {lg_g - thr_g[i]:+.3e} oracle {lg_o - thr_o[i]:+.3e} shift {shift[i]:.2e} gene rel {rel(got['dispGeneEst'][i], ref['dispGeneEst'][i]):.1e}")
flip = live & (got["dispOutlier"] != ref["dispOutlier"])
print("outlier flag flips:", int(flip.sum()), "of them within the shift:", int((np.minimum(np.abs(np.log(got['dispGeneEst']) - thr_g), np.abs(np.log(ref['dispGeneEst']) - thr_o))[flip] <= 2 * shift[flip] + 1e-9).sum()))
