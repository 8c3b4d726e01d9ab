set -e
TAG=r06_j
LIB=ablibs/lib_r06f.so
mkdir -p gpurun_out/$TAG
for cfg in "500000 8" "250000 8" "1000000 8"; do
  python tools/ab_libs.py $cfg ablibs/lib_r06a.so
  python tools/ab_libs.py $cfg $LIB
  python tools/ab_libs.py $cfg $LIB -- line_search_spread=3
  python tools/ab_libs.py $cfg $LIB -- line_search_spread=2
  python tools/ab_libs.py $cfg ablibs/lib_r06a.so -- line_search_spread=2
done > gpurun_out/$TAG/ab.txt 2>&1
grep -v digest gpurun_out/$TAG/ab.txt | cut -c1-150
