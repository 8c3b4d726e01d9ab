"""Second half of tools/profile_round.sh: SQ / fp64 counter passes -> per-stage entries of pmc_traffic.json (what bench.py replays for
roofline.valu_busy / fp64_frac).  A stage's counters are its line-search launch PLUS the fitDispGrid launch behind it (round 6: both sit
inside the HIP-event scope bench.py times).  usage: python tools/pmc_post.py <outdir> <tag>"""
import glob, json, sys
import pandas as pd
OUT, TAG = sys.argv[1], sys.argv[2]

def table(sub):
    fs = glob.glob(f"{OUT}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None
    df = pd.concat([pd.read_csv(f) for f in fs])
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("cd::", "").str.replace("void ", "")
    return df.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()

def pick(index, prefix):  # the variant that ran (disp_fit_kernel<false, 2>, <true, 3>, ...)
    hit = [k for k in index if k.startswith(prefix)]
    return hit[0] if hit else None

g = table("pmc_sq")
g.to_csv(f"{OUT}/{TAG}_pmc_sq_bench_2Mx8.csv")
print(g.to_string())
tj = json.load(open(f"{OUT}/pmc_traffic.json"))
stages = {"disp_gene": ["disp_fit_kernel<false", "disp_grid_kernel<false"], "disp_map": ["disp_fit_kernel<true", "disp_grid_kernel<true"], "wald_irls": ["wald_irls_kernel"]}
for k, prefixes in stages.items():
    key = k + ":2000000x8"
    rows = [pick(g.index, p) for p in prefixes]
    rows = [r for r in rows if r]
    if key in tj and rows:
        tot = g.loc[rows].sum()
        tj[key]["valu"] = {"SQ_INSTS_VALU": float(tot["SQ_INSTS_VALU"]), "SQ_ACTIVE_INST_VALU": float(tot["SQ_ACTIVE_INST_VALU"]),
                           "active_lanes_per_inst": round(float(tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_INSTS_VALU"]), 1),
                           "kernels": rows, "source": f"profiles/{TAG}_pmc_sq_bench_2Mx8.csv"}
# fp64 flops per launch: wave-level instruction counts (FMA = 2 flops) x the lanes active per VALU instruction of the same kernels
g6 = table("pmc_f64")
if g6 is not None:
    g6.to_csv(f"{OUT}/{TAG}_pmc_f64_bench_2Mx8.csv")
    print(g6.to_string())
    for k, prefixes in stages.items():
        key = k + ":2000000x8"
        rows = [pick(g6.index, p) for p in prefixes]
        rows = [r for r in rows if r]
        if key in tj and rows and "valu" in tj[key]:
            tot = g6.loc[rows].sum()
            mix = {c: float(tot[c]) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64") if c in tot}
            wave_flops = 2 * mix.get("SQ_INSTS_VALU_FMA_F64", 0) + mix.get("SQ_INSTS_VALU_MUL_F64", 0) + mix.get("SQ_INSTS_VALU_ADD_F64", 0) + mix.get("SQ_INSTS_VALU_TRANS_F64", 0)
            tj[key]["valu"]["fp64_insts"] = mix
            tj[key]["valu"]["fp64_flops_per_launch"] = wave_flops * tj[key]["valu"]["active_lanes_per_inst"]
            tj[key]["valu"]["fp64_source"] = f"profiles/{TAG}_pmc_f64_bench_2Mx8.csv"
json.dump(tj, open(f"{OUT}/pmc_traffic.json", "w"), indent=1)
