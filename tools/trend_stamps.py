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
