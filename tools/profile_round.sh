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
python3 tools/pmc_post.py $OUT $TAG
tail -1 $OUT/bench.json
