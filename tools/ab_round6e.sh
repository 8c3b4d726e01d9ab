set -e
TAG=r06_g
mkdir -p gpurun_out/$TAG
STAMPS_LIB=ablibs/lib_r06a_diag.so python tools/stamps.py 2000000 8 > gpurun_out/$TAG/stamps_2M_old.txt 2>&1
STAMPS_LIB=ablibs/lib_r06d_diag.so python tools/stamps.py 2000000 8 > gpurun_out/$TAG/stamps_2M_new_mw2.txt 2>&1
