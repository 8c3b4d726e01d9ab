# every configuration of DESIGN.md section 5 on one box: bash tools/bench_configs.sh <tag>
set -e
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for cfg in "30000 4" "200000 4" "250000 8" "500000 8" "1000000 8" "2000000 4" "2000000 8" "2000000 16" "20000000 16"; do
  set -- $cfg
  python bench.py --rows $1 --samples $2 --no-cpu-baseline --no-hbm-kernels > $OUT/${TAG}_bench_${1}x${2}.json 2> $OUT/bench_${1}x${2}.err
  python - "$OUT/${TAG}_bench_${1}x${2}.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(j["config"]["rows_per_gpu"], "x", j["config"]["samples"], "ms_per_step", j["ms_per_step"], "M/s %.0f" % (j["value"] / 1e6), {k: v[0] for k, v in j["kernels_ms"].items()})
PY
done
for rows in 250000 500000 1000000 2000000; do
  CHICDIFF_BENCH_FORCE_DIST=1 python bench.py --rows $rows --no-cpu-baseline --no-hbm-kernels > $OUT/${TAG}_bench_forcedist_${rows}.json 2> $OUT/bench_forcedist_${rows}.err
  python - "$OUT/${TAG}_bench_forcedist_${rows}.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("forcedist", j["config"]["rows_per_gpu"], "ms_per_step", j["ms_per_step"])
PY
done
for w in 2 4 8; do
  CHICDIFF_BENCH_FORCE_DIST=1 CHICDIFF_BENCH_FAKE_WORLD=$w python bench.py --rows 2000000 --no-cpu-baseline --no-hbm-kernels > $OUT/${TAG}_bench_fakeworld_${w}.json 2> $OUT/bench_fakeworld_${w}.err
  python - "$OUT/${TAG}_bench_fakeworld_${w}.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("fakeworld", j["config"]["rows_per_gpu"], "ms_per_step", j["ms_per_step"], "value", j["value"], "projected", j.get("projected_value"), {k: v[0] for k, v in j["kernels_ms"].items() if k in ("trend_fit",)})
PY
done
