#!/bin/bash
# End-of-round evidence run (GPU box): kernel-trace stats, separate PMC passes, plain bench line.
# Usage (from the repo root on the GPU box): bash tools/profile_round.sh r01_e
set -e
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-kernels"
P="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-kernels"
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $B > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python3 $P > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python3 $P > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/pmc_sq --output-format csv -- python3 $P > /dev/null 2> $OUT/pmc_sq.err
# fp64 instruction mix of the fit kernels (their own pass: the SQ block holds few counters at once); not fatal if a name is missing
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $OUT/pmc_f64 --output-format csv -- python3 $P > /dev/null 2> $OUT/pmc_f64.err || echo "fp64 counter pass failed (see pmc_f64.err)"
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT $TAG 2000000 8 > $OUT/pmc_traffic.log
python3 - <<PY
import glob, pandas as pd
fs = glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True)
df = pd.concat([pd.read_csv(f) for f in fs])
df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("cd::", "").str.replace("void ", "")
g = df.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
g.to_csv("$OUT/${TAG}_pmc_sq_bench_2Mx8.csv")
print(g.to_string())
import json
tj = json.load(open("$OUT/pmc_traffic.json"))
def pick(index, prefix):  # the line-search variant that ran (disp_fit_kernel<false, 2>, <true, 3>, ...)
    hit = [k for k in index if k.startswith(prefix)]
    return hit[0] if hit else prefix
names = {"disp_gene": pick(g.index, "disp_fit_kernel<false"), "disp_map": pick(g.index, "disp_fit_kernel<true"), "wald_irls": "wald_irls_kernel"}
for k, kn in names.items():
    key = k + ":2000000x8"
    if key in tj and kn in g.index:
        r = g.loc[kn]
        tj[key]["valu"] = {"SQ_INSTS_VALU": float(r["SQ_INSTS_VALU"]), "SQ_ACTIVE_INST_VALU": float(r["SQ_ACTIVE_INST_VALU"]),
                           "active_lanes_per_inst": round(float(r["SQ_THREAD_CYCLES_VALU"] / r["SQ_INSTS_VALU"]), 1),
                           "source": "profiles/${TAG}_pmc_sq_bench_2Mx8.csv"}
# fp64 flops per launch: wave-level instruction counts (FMA = 2 flops) x the lanes active per VALU instruction of the same kernel
fs = glob.glob("$OUT/pmc_f64/**/*counter_collection.csv", recursive=True)
if fs:
    d6 = pd.concat([pd.read_csv(f) for f in fs])
    d6["k"] = d6["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("cd::", "").str.replace("void ", "")
    g6 = d6.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
    g6.to_csv("$OUT/${TAG}_pmc_f64_bench_2Mx8.csv")
    print(g6.to_string())
    for k, kn in names.items():
        key = k + ":2000000x8"
        if key in tj and kn in g6.index and "valu" in tj[key]:
            r6 = g6.loc[kn]
            mix = {c: float(r6[c]) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64") if c in r6}
            wave_flops = 2 * mix.get("SQ_INSTS_VALU_FMA_F64", 0) + mix.get("SQ_INSTS_VALU_MUL_F64", 0) + mix.get("SQ_INSTS_VALU_ADD_F64", 0) + mix.get("SQ_INSTS_VALU_TRANS_F64", 0)
            tj[key]["valu"]["fp64_insts"] = mix
            tj[key]["valu"]["fp64_flops_per_launch"] = wave_flops * tj[key]["valu"]["active_lanes_per_inst"]
            tj[key]["valu"]["fp64_source"] = "profiles/${TAG}_pmc_f64_bench_2Mx8.csv"
json.dump(tj, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
tail -1 $OUT/bench.json
