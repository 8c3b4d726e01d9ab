"""Diagnostic: where a pass of the persistent trend kernel spends its time (s_memrealtime stamps of workgroup 0, printed by the
kernel when it ends; profiles/r04_trend_pass_stamps.txt).  Needs a library whose global_kernels.hip was built with
-DCHICDIFF_TREND_STAMPS (never the product library):

    make -C chicdiff_amd/csrc && mkdir -p chicdiff_amd/lib/obj_ts chicdiff_amd/lib/ab && cp chicdiff_amd/lib/obj/*.o chicdiff_amd/lib/obj_ts/
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DCHICDIFF_TREND_STAMPS -c chicdiff_amd/csrc/global_kernels.hip -o chicdiff_amd/lib/obj_ts/global_kernels.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o chicdiff_amd/lib/ab/trend_stamps.so chicdiff_amd/lib/obj_ts/*.o

usage (GPU box): python tools/trend_stamps.py <rows>"""
import os, sys
sys.path.insert(0, '.')
os.environ["CHICDIFF_HIP_LIB"] = "chicdiff_amd/lib/ab/trend_stamps.so"
import numpy as np
from chicdiff_amd import hip, synth
n = int(sys.argv[1]); S = 8
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
for _ in range(3):
    ctx.nbglm_fit(dk, dn, d["group"])
