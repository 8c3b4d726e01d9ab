set -e
TAG=r06_d
LIB=ablibs/lib_r06d.so
mkdir -p gpurun_out/$TAG
for cfg in "2000000 8" "1000000 8" "2000000 4"; do
  python tools/ab_libs.py $cfg ablibs/lib_r06a.so
  python tools/ab_libs.py $cfg $LIB
  for pr in 0 8 16 25 33; do
    python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3 line_search_prio=$pr
  done
  python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=2 line_search_prio=25
done > gpurun_out/$TAG/ab.txt 2>&1
grep -v digest gpurun_out/$TAG/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in lib_r06a lib_r06d; do
  CHICDIFF_HIP_LIB=$R/ablibs/$L.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof_$L --output-format csv -- python3 $R/tools/fit_timing.py 500000 8 > $R/gpurun_out/$TAG/prof_$L.log 2>&1
  find $R/gpurun_out/$TAG/prof_$L -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/$TAG/kernel_stats_500k_$L.csv
  head -12 $R/gpurun_out/$TAG/kernel_stats_500k_$L.csv | cut -c1-150
done
