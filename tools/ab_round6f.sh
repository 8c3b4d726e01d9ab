set -e
TAG=r06_h
LIB=ablibs/lib_r06e.so
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fit_parity or layouts_agree or all_zero_rows or fit_edge or extreme or fuzz or row_queue or theta_grid or fused_wald or intercept" > gpurun_out/$TAG/pytest_fit.log 2>&1 || { tail -40 gpurun_out/$TAG/pytest_fit.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest_fit.log
for cfg in "2000000 8" "1000000 8" "500000 8" "250000 8" "2000000 4" "2000000 16" "200000 4" "30000 4"; do
  python tools/ab_libs.py $cfg ablibs/lib_r06a.so
  python tools/ab_libs.py $cfg $LIB
done > gpurun_out/$TAG/ab.txt 2>&1
grep -v digest gpurun_out/$TAG/ab.txt | cut -c1-150
R=$GRAFT_REPO_ROOT
CHICDIFF_HIP_LIB=$R/$LIB rocprofv3 --kernel-trace --stats -d /tmp/prof_new --output-format csv -- python3 tools/fit_timing.py 2000000 8 > $R/gpurun_out/$TAG/prof_new.log 2>&1
find /tmp/prof_new -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/$TAG/kernel_stats_2M_new.csv
head -16 $R/gpurun_out/$TAG/kernel_stats_2M_new.csv | cut -c1-130
