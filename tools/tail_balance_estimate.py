"""Diagnostic (DIAG build): what balancing the end of the gene-wise launch over groups of neighbouring waves could give.
From the wave stamps: the launch ends with its last wave; if the waves of a group of G (consecutive wave numbers = one workgroup of G
waves) shared their rows perfectly, a group would end at the mean of its waves' ends (never before the longest single chain: the
median wave's end is taken as that bound).  usage: python tools/tail_balance_estimate.py <rows> <S>"""
import os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["CHICDIFF_DISP_STAMPS"] = "gpurun_out/stamps_tb.bin"
os.environ["CHICDIFF_HIP_LIB"] = "chicdiff_amd/lib/libchicdiff_hip_diag.so"
from chicdiff_amd import hip, synth
n, S = int(sys.argv[1]), int(sys.argv[2])
d = synth.make(n, S)
ctx = hip.HipContext(0)
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
ctx.nbglm_fit(dk, dn, d["group"])
raw = np.fromfile("gpurun_out/stamps_tb.bin", dtype=np.uint64)
K = 34
kind, nw = int(raw[0]), int(raw[1])  # first record: the gene-wise launch
st = raw[2:2 + nw * K].reshape(nw, K).astype(np.int64)
st = st[st[:, 2] > 0]
nw = len(st)
t0 = st[:, 0].min()
ex = (st[:, 2] - t0) / 100.0
qe = (np.where(st[:, 1] > 0, st[:, 1], st[:, 2]) - t0) / 100.0
print(f"n {n} S {S}: {nw} waves; exit median {np.median(ex):.0f} us, p90 {np.percentile(ex, 90):.0f}, max {ex.max():.0f}; queue empty median {np.median(qe):.0f}")
for G in (2, 4, 8, 16):
    m = nw // G * G
    g = ex[:m].reshape(-1, G).mean(1)
    bound = max(g.max(), np.median(ex))
    print(f"  groups of {G:2d} waves sharing perfectly: slowest group's mean end {g.max():.0f} us -> launch ~{bound:.0f} us ({100 * (1 - bound / ex.max()):.0f} % less)")
