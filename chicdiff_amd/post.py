"""Host-side mirrors of the steps either side of the test, over the HIP entry points.

* ``getRegionUniverse``  — chicdiff.R:369-426 (window mode) after the peak matrix has been read and filtered
  (``readAndFilterPeakMatrix`` stays reference R): expand each (baitID, oeID) call, clip to the restriction map
  and to the bait's chromosome, return RU in the reference's own row order.
* ``applyIHWweights``    — chicdiff.R:2038-2049, the part of ``IHWcorrection`` after ``ihw()`` has been trained
  on the control set (training stays reference R): breaks from ``distLookup``, group cut, weight lookup and
  renormalisation, weighted p-values and their BH adjustment.

* ``getFullRegionDataHip`` — the chinput branch of ``getFullRegionData1`` (chicdiff.R:577-945) without the long
  "recast" table: per replicate chinput text -> key table -> count join into one column of the fragment matrix;
  per-fragment Chicago tables -> Bmean + Tmean = FullMean per RU row and replicate.  Returns the device-resident
  block ``DESeq2Wrap`` takes in place of ``FullRegionData`` (tested twin of r/R/getFullRegionData_hip.R).

All keep their arrays on the GPU; no CPU fallback (the HIP library raises when it is missing).
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


class HipRegionData(dict):
    """Device-resident fragment block: ``samples, condition, S, n, fragN (S, nfrag) int32, fragFullMean (S, nfrag)
    float64, region_ptr (n + 1) int64`` — what r/R/getFullRegionData_hip.R calls a chicdiffHipRegionData object."""


def getFullRegionDataHip(ctx, RU, chinput_files, condition, background):
    """``RU``: dict / DataFrame with baitID, regionID (1..n), otherEndID; ``chinput_files``: one .chinput path per
    replicate (chicdiff.R:811-831), ``condition``: one label per replicate (:921-923); ``background``: the dense
    per-fragment tables of the Chicago data sets over IDs id_min .. id_min + nid - 1 —
    ``id_min, midsum (nid,), sj, si (S, nid), tblb, tlb (S, nid; -1 = NA), T (S, ntblb, ntlb), distfun (S, 10)``
    (what .hipBackgroundTables builds in R from ``x@x`` and ``.chicEstimateDistFun``)."""
    torch = ctx.torch
    S = len(chinput_files)
    assert len(condition) == S
    bait = np.asarray(RU["baitID"], dtype=np.int32)
    region = np.asarray(RU["regionID"], dtype=np.int64)
    oe = np.asarray(RU["otherEndID"], dtype=np.int32)
    order = np.lexsort((oe, region))                      # (regionID, otherEndID): a region's fragments are consecutive
    bait, region, oe = bait[order], region[order], oe[order]
    ids, starts = np.unique(region, return_index=True)
    if not np.array_equal(ids, np.arange(1, len(ids) + 1)):
        raise ValueError("RU: regionID must be 1..n without gaps")
    nfrag, n = len(bait), len(ids)
    dev = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(ctx.device)
    d_bait, d_oe = dev(bait, np.int32), dev(oe, np.int32)
    flags = np.zeros(int(bait.max()) + 1, dtype=np.uint8)
    flags[np.unique(bait)] = 1                            # baits <- sort(unique(RU$baitID)), chicdiff.R:775
    d_flags = torch.from_numpy(flags).to(ctx.device)
    fragN = torch.empty((S, nfrag), dtype=torch.int32, device=ctx.device)
    for s, path in enumerate(chinput_files):
        keys, vals, _ = ctx.read_chinput(path, d_flags)   # fread + x[J(baits)] + setkey(baitID, otherEndID)
        fragN[s] = ctx.count_join(d_bait, d_oe, keys, vals)   # merge(all.x = TRUE); N[is.na(N)] <- 0
    b = background
    _, _, fragFM = ctx.fragment_background(d_bait, d_oe, int(b["id_min"]), dev(b["midsum"], np.int64), dev(b["sj"], np.float64),
                                           dev(b["si"], np.float64), dev(b["tblb"], np.int32), dev(b["tlb"], np.int32),
                                           dev(b["T"], np.float64), np.asarray(b["distfun"], dtype=np.float64))
    return HipRegionData(samples=[str(p) for p in chinput_files], condition=list(condition), S=S, n=n, fragN=fragN,
                         fragFullMean=fragFM, region_ptr=dev(np.concatenate([starts, [nfrag]]), np.int64))
