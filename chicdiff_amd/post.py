"""Host-side mirrors of the steps either side of the test, over the HIP entry points.

* ``getRegionUniverse``  — chicdiff.R:369-426 (window mode) after the peak matrix has been read and filtered
  (``readAndFilterPeakMatrix`` stays reference R): expand each (baitID, oeID) call, clip to the restriction map
  and to the bait's chromosome, return RU in the reference's own row order.
* ``applyIHWweights``    — chicdiff.R:2038-2049, the part of ``IHWcorrection`` after ``ihw()`` has been trained
  on the control set (training stays reference R): breaks from ``distLookup``, group cut, weight lookup and
  renormalisation, weighted p-values and their BH adjustment.

Both keep their arrays on the GPU; no CPU fallback (the HIP library raises when it is missing).
"""
from __future__ import annotations

import numpy as np


def rmap_chr_codes(rmap_chr, rmap_id):
    """chr_of[0..maxfrag] (int32, -1 = ID not on the map) from the rmap's chromosome and ID columns
    (``fread(rmapfile)``, chicdiff.R:382, :388)."""
    ids = np.asarray(rmap_id, dtype=np.int64)
    names, codes = np.unique(np.asarray(rmap_chr).astype(str), return_inverse=True)
    chr_of = np.full(int(ids.max()) + 1, -1, dtype=np.int32)
    chr_of[ids] = codes.astype(np.int32)
    return chr_of, names


def getRegionUniverse(ctx, baitID, oeID, RUexpand, rmap_chr, rmap_id):
    """Window-mode region universe.  Returns a dict of device tensors: ``baitID, regionID, otherEndID`` in RU.DT's
    order (keyed by baitID after otherEndID, chicdiff.R:389/:393), plus the region-major CSR view
    (``region_ptr, minOE, maxOE`` and ``csr_*`` rows) that the window sums consume."""
    torch = ctx.torch
    chr_of, _ = rmap_chr_codes(rmap_chr, rmap_id)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(ctx.device)
    r = ctx.region_universe(dev(baitID), dev(oeID), int(RUexpand), dev(chr_of))
    # setkey(otherEndID) then setkey(baitID): two stable sorts of the (regionID, otherEndID)-ordered rows
    o1 = torch.sort(r["otherEndID"], stable=True).indices
    o2 = o1[torch.sort(r["baitID"][o1], stable=True).indices]
    return dict(baitID=r["baitID"][o2], regionID=r["regionID"][o2], otherEndID=r["otherEndID"][o2],
                region_ptr=r["region_ptr"], minOE=r["minOE"], maxOE=r["maxOE"],
                csr_baitID=r["baitID"], csr_regionID=r["regionID"], csr_otherEndID=r["otherEndID"])


def ihw_breaks(minLogDist, maxLogDist):
    """chicdiff.R:2030-2031, :2039: minLogDist[1] <- 0; maxLogDist[last] <- Inf;
    breaks <- (c(minLogDist, Inf) + c(0, maxLogDist)) / 2."""
    lo = np.asarray(minLogDist, dtype=np.float64).copy()
    hi = np.asarray(maxLogDist, dtype=np.float64).copy()
    lo[0] = 0.0
    hi[-1] = np.inf
    return (np.concatenate([lo, [np.inf]]) + np.concatenate([[0.0], hi])) / 2.0


def applyIHWweights(ctx, avDist, pvalue, minLogDist, maxLogDist, avWeights):
    """Columns ``group, weight, weighted_pvalue, weighted_padj`` (device tensors) for the test set, given the
    distance dependency learned on the control set (``distLookup``, chicdiff.R:2012-2031)."""
    torch = ctx.torch
    dev = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx.device)
    return ctx.ihw_apply(dev(avDist), dev(pvalue), ihw_breaks(minLogDist, maxLogDist), avWeights)
