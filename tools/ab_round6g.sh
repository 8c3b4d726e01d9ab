set -e
TAG=r06_k
mkdir -p gpurun_out/$TAG
for i in 1 2; do
STAMPS_LIB=ablibs/lib_r06a_diag.so python tools/stamps.py 500000 8 > gpurun_out/$TAG/stamps_500k_old_$i.txt 2>&1
STAMPS_LIB=ablibs/lib_r06f_diag.so python tools/stamps.py 500000 8 > gpurun_out/$TAG/stamps_500k_new_$i.txt 2>&1
done
