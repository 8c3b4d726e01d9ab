set -e
TAG=${1:-r06_c}
LIB=${2:-ablibs/lib_r06c.so}
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fit_parity or layouts_agree or all_zero_rows or fit_edge or extreme or fuzz or row_queue or theta_grid or fused_wald" > gpurun_out/$TAG/pytest_fit.log 2>&1 || { tail -40 gpurun_out/$TAG/pytest_fit.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest_fit.log
for cfg in "2000000 8" "250000 8" "2000000 4" "2000000 16" "200000 4" "500000 8"; do
  python tools/ab_libs.py $cfg ablibs/lib_r06a.so
  python tools/ab_libs.py $cfg $LIB
  python tools/ab_libs.py $cfg $LIB -- line_search_min_waves=3
done > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
