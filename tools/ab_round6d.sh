set -e
TAG=r06_f
LIB=ablibs/lib_r06d.so
mkdir -p gpurun_out/$TAG
cfg="2000000 8"
{
python tools/ab_libs.py $cfg ablibs/lib_r06a.so
python tools/ab_libs.py $cfg $LIB
for ca in 1 2 3 4 5; do
  python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3 line_search_classes_a=$ca
done
for dl in 1 2 4 8; do
  python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3 line_search_classes_a=3 line_search_deal=$dl
done
for ch in 16 32; do
  python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3 line_search_chunk=$ch
done
python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3 line_search_schedule=0
} > gpurun_out/$TAG/ab.txt 2>&1
grep -v digest gpurun_out/$TAG/ab.txt | cut -c1-110
