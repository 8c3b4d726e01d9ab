"""Write the synthetic matrices of chicdiff_amd/synth.py as raw binaries an R box can read (tools/make_golden.R):
<out>/<tag>.counts.i32 (n x S, column-major), <tag>.nf.f64 (n x S, column-major), <tag>.meta.txt (n, S, group).
Default set: C2-shaped 2v2 (the reference's own design), a 4v4, a design-~1 slice and a heterogeneous 2v2 whose prior
variance lands above DESeq2's 0.25 floor (so that the set.seed(2) simulation decides the value)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chicdiff_amd import synth  # noqa: E402


def heterogeneous_counts(n, S, sdlog, seed=7):
    rng = np.random.default_rng(seed)
    mu = np.exp(rng.normal(np.log(60.0), 1.0, n))
    alpha = (0.05 + 2.0 / mu) * np.exp(rng.normal(0.0, sdlog, n))
    nf = np.exp(rng.normal(0.0, 0.2, (n, S)))
    nf /= np.exp(np.log(nf).mean(1, keepdims=True))
    size = (1.0 / alpha)[:, None]
    k = rng.negative_binomial(size, size / (size + mu[:, None] * nf)).astype(np.int32)
    return k, nf


def write(out, tag, counts, nf, group):
    n, S = counts.shape
    np.asfortranarray(counts.astype(np.int32)).T.tofile(os.path.join(out, f"{tag}.counts.i32"))  # column-major bytes
    np.asfortranarray(nf.astype(np.float64)).T.tofile(os.path.join(out, f"{tag}.nf.f64"))
    with open(os.path.join(out, f"{tag}.meta.txt"), "w") as f:
        f.write(f"{n} {S}\n" + " ".join(str(int(g)) for g in group) + "\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="golden_inputs")
    ap.add_argument("--rows", type=int, default=20000)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for S in (4, 8):
        d = synth.make(a.rows, S)
        write(a.out, f"synth_{a.rows}x{S}", d["counts"], d["nf"], d["group"])
    d = synth.make(a.rows, 4)
    keep = d["counts"].sum(1) > 0
    write(a.out, f"synth_{a.rows}x4_intercept", d["counts"][keep], d["nf"][keep], np.zeros(4, int))
    k, nf = heterogeneous_counts(6000, 4, 1.3)
    write(a.out, "hetero_6000x4", k, nf, [0, 0, 1, 1])
    print("wrote", sorted(os.listdir(a.out)))
