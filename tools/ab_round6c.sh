set -e
TAG=r06_e
mkdir -p gpurun_out/$TAG
for mw in 2 3; do
  python tools/stamps.py 2000000 8 line_search_min_waves=$mw > gpurun_out/$TAG/stamps_2M_mw$mw.txt 2>&1
done
python tools/stamps.py 1000000 8 line_search_min_waves=2 > gpurun_out/$TAG/stamps_1M_mw2.txt 2>&1
python tools/stamps.py 1000000 8 line_search_min_waves=3 > gpurun_out/$TAG/stamps_1M_mw3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for L in lib_r06a lib_r06d; do
  CHICDIFF_HIP_LIB=$R/ablibs/$L.so rocprofv3 --kernel-trace --stats -d /tmp/prof_$L --output-format csv -- python3 tools/fit_timing.py 500000 8 > $R/gpurun_out/$TAG/prof_$L.log 2>&1
  find /tmp/prof_$L -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/$TAG/kernel_stats_500k_$L.csv
  head -12 $R/gpurun_out/$TAG/kernel_stats_500k_$L.csv | cut -c1-150
done
