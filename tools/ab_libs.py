"""A/B of library builds on the GPU box: for each library given, one child process runs tools/fit_timing.py's fit with
CHICDIFF_HIP_LIB pointing at it and prints per-kernel HIP-event times and a digest of every output column (equal digests =
bit-identical results).  usage: python tools/ab_libs.py <rows> <samples> lib_a.so lib_b.so ... [-- option=value ...]"""
import hashlib, os, subprocess, sys

CHILD = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
import torch
from chicdiff_amd import hip, synth
n, S = int(sys.argv[1]), int(sys.argv[2])
d = synth.make(n, S)
ctx = hip.HipContext(0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
want = ["baseMean", "dispGeneEst", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue", "deviance", "maxCooks", "betaIter"]
for _ in range(2):
    out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
ctx.enable_timing(True)
acc = {}
reps = 5
for _ in range(reps):
    out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
    for k, (ms, c) in ctx.kernel_times().items():
        acc[k] = acc.get(k, 0) + ms / reps
h = hashlib.sha256()
for k in want:
    h.update(out[k].cpu().numpy().tobytes())
print("  total %.3f ms  " % sum(acc.values()) + " ".join(f"{k}={v:.3f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
print("  digest", h.hexdigest()[:16], "sumDeviance", repr(sc.get("sumDeviance")))
'''

def main():
    args = sys.argv[1:]
    opts = []
    if "--" in args:
        k = args.index("--"); opts = args[k + 1:]; args = args[:k]
    n, S, libs = args[0], args[1], args[2:]
    for lib in libs:
        env = dict(os.environ, CHICDIFF_HIP_LIB=os.path.abspath(lib))
        print(f"{lib}  n={n} S={S} {' '.join(opts)}", flush=True)
        r = subprocess.run([sys.executable, "-c", CHILD, n, S] + opts, env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode:
            print("  FAILED", r.stderr[-2000:])
        sys.stdout.flush()

main()
