#!/usr/bin/env python3
"""Lift the reference's chr19 design files into a small committed fixture (data, not code).

Source: ``ChicdiffData/inst/extdata/designDir/chr19_GRCh37_HindIII.{rmap,baitmap}`` under ``/root/reference``
(11 542 restriction fragments, 1 052 baits).  Together with ``tests/golden/chr19_results.npz`` (baitID, minOE, maxOE,
avDist of the reference's own run) the map pins IHWcorrection's covariate ``avDist = mean(distSign)`` by region
(chicdiff.R:1965-1967 with the distSign of :868-882): tests/test_results_postprocessing.py.

Run here (authoring container only; the reference does not exist on the GPU box):

    python tools/make_golden_design.py

Writes ``tests/golden/chr19_design.npz`` (rmap: chr code, start, end, ID; baitmap: ID, start, end).
"""
import os

import numpy as np

REF = "/root/reference/ChicdiffData/inst/extdata/designDir"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chr19_design.npz")


def main():
    rmap = np.loadtxt(os.path.join(REF, "chr19_GRCh37_HindIII.rmap"), dtype=str)
    bm = np.loadtxt(os.path.join(REF, "chr19_GRCh37_HindIII.baitmap"), dtype=str, usecols=(0, 1, 2, 3))
    chr_names = np.unique(np.char.strip(rmap[:, 0], '"'))
    np.savez_compressed(
        OUT,
        rmap_chr=np.char.strip(rmap[:, 0], '"'), rmap_start=rmap[:, 1].astype(np.int32), rmap_end=rmap[:, 2].astype(np.int32),
        rmap_id=rmap[:, 3].astype(np.int32), bait_chr=np.char.strip(bm[:, 0], '"'), bait_start=bm[:, 1].astype(np.int32),
        bait_end=bm[:, 2].astype(np.int32), bait_id=bm[:, 3].astype(np.int32))
    print("wrote", OUT, len(rmap), "fragments,", len(bm), "baits; chromosomes", chr_names)


if __name__ == "__main__":
    main()
