"""Host-side mirror of the reference's operator for the hot path:

    DESeq2Wrap(chicdiff.settings, RU, FullRegionData, suffix = "", theta = NULL)   (chicdiff.R:1494)

Same name, argument meaning, messages, warnings and error behaviour; the DESeq2 hand-off
(chicdiff.R:1557-1691) and the long-table group-by (:1540-1556) are replaced by the HIP library
behind ``include/chicdiff_hip.h``.  R is not installed in this image, so the host language here
is Python with pandas standing in for data.table; the R wrapper a Chicdiff maintainer would drop
in is r/R/DESeq2Wrap_hip.R.  There is no CPU fallback: a missing HIP
library or GPU raises.

Inputs
  chicdiff_settings : dict with the reference's keys (chicdiff.R:3-24); used here: norm, theta,
                      theta_grid, rmapfile, saveAuxData, outprefix.
  RU                : DataFrame baitID, regionID (1..n), otherEndID               (chicdiff.R:392-425)
  FullRegionData    : long "recast" DataFrame, one row per (region, fragment, sample) with columns
                      baitID, otherEndID, regionID, sample, N, FullMean, condition, ... (:912-925)
Output
  DataFrame: baseMean, log2FoldChange, lfcSE, stat, pvalue, padj, baitID, maxOE, minOE, regionID,
  OEchr, OEstart, OEend, baitchr, baitstart, baitend — one row per regionID ascending (:1752-1757);
  ``out.attrs["theta"]`` set whenever the combined branch ran (:1759-1760).
"""
from __future__ import annotations

import sys
import warnings

import numpy as np



def message(*a):
    """R message(): progress lines go to stderr (users grep them)."""
    print(*a, sep="", file=sys.stderr)


def _read_rmap(path):
    import pandas as pd

    rmap = pd.read_csv(path, sep=r"\s+", header=None, quotechar='"', engine="python")
    if rmap.shape[1] != 4:
        raise ValueError("rmap file should have 4 columns: <chr> <start> <end> <id>")
    rmap.columns = ["OEchr", "OEstart", "OEend", "otherEndID"]
    return rmap


def _dense_fragments(FullRegionData):
    """Long table -> per-sample fragment arrays in (regionID, otherEndID) order + CSR region offsets.

    Mirrors setkey(fragData, otherEndID) + by=(baitID, regionID, sample) (chicdiff.R:1526, :1540-1547):
    within a region fragments are summed in ascending otherEndID order."""
    fd = FullRegionData.sort_values("otherEndID", kind="stable")
    samples = list(dict.fromkeys(fd["sample"].tolist()))  # unique(), order of first appearance
    S = len(samples)
    cond = fd["condition"].to_numpy()[:S]                 # colData: fragData$condition[1:ncol] (:1556)
    first = fd["sample"].to_numpy()[:S]
    if list(first) != samples:
        raise ValueError("FullRegionData: the first rows do not hold one row per sample (recast layout expected)")
    cond_of = dict(zip(samples, cond))
    per = []
    key = None
    for s in samples:
        sub = fd[fd["sample"] == s].sort_values(["regionID", "otherEndID"], kind="stable")
        k = (sub["regionID"].to_numpy(), sub["otherEndID"].to_numpy())
        if key is None:
            key = k
        elif not (np.array_equal(k[0], key[0]) and np.array_equal(k[1], key[1])):
            raise ValueError("FullRegionData: samples do not cover the same (regionID, otherEndID) rows")
        per.append(sub)
    region = key[0]
    ids, starts = np.unique(region, return_index=True)
    if not np.array_equal(ids, np.arange(1, len(ids) + 1)):
        # stopifnot(identical(1:nrow(annoData), annoData$regionID))  (chicdiff.R:1717)
        raise ValueError("regionID must be 1..n without gaps")
    region_ptr = np.concatenate([starts, [len(region)]]).astype(np.int64)
    if np.any([p["N"].isna().any() for p in per]):
        raise ValueError("NA counts in FullRegionData$N")
    fragN = np.stack([p["N"].to_numpy(dtype=np.int32) for p in per], axis=1)
    fragFM = np.stack([p["FullMean"].to_numpy(dtype=np.float64) for p in per], axis=1)
    return samples, [cond_of[s] for s in samples], fragN, fragFM, region_ptr


def DESeq2Wrap(chicdiff_settings, RU, FullRegionData, suffix="", theta=None, ctx=None):
    import pandas as pd

    from . import hip

    from .settings import asChicdiffSettings, hipDevice

    if chicdiff_settings.get("norm") is not None:
        nm = chicdiff_settings["norm"]
        nm = nm[0] if isinstance(nm, (list, tuple)) and len(nm) == 1 else nm
        if str(nm) not in ("standard", "fullmean", "combined"):
            raise ValueError("DESeq2Wrap error: Unknown normalisation method.")   # chicdiff.R:1507-1509
    # the reference's list, unchanged: every one of its 17 keys keeps its meaning (`device` = plot device, chicdiff.R:20)
    chicdiff_settings = asChicdiffSettings({k: v for k, v in chicdiff_settings.items()})
    Grid = list(chicdiff_settings["theta_grid"])
    rmapfile = chicdiff_settings["rmapfile"]
    save_rds = chicdiff_settings.get("saveAuxData", False)
    outprefix = chicdiff_settings.get("outprefix", "")

    if theta is None and chicdiff_settings.get("theta") is not None:
        theta = chicdiff_settings["theta"]

    norm = chicdiff_settings["norm"]

    if theta is not None:
        if theta == 1 and norm != "standard":
            warnings.warn('Mixing parameter theta set to 1, equivalent to norm = "standard". '
                          "The norm method has been reset accordingly.")
            norm = "standard"
        if not theta and norm != "fullmean":
            warnings.warn('Mixing parameter theta set to 0, equivalent to norm = "fullmean". '
                          "The norm method has been reset accordingly.")
            norm = "fullmean"

    own_ctx = ctx is None
    if own_ctx:
        ctx = hip.HipContext(hipDevice(chicdiff_settings))   # the new key `hipDevice`; never the reference's `device`
    torch = ctx.torch
    try:
        from .post import HipRegionData
        if isinstance(FullRegionData, HipRegionData):  # the device-resident block of post.getFullRegionDataHip
            samples, conds, region_ptr = FullRegionData["samples"], FullRegionData["condition"], None
        else:
            samples, conds, fragN, fragFM, region_ptr = _dense_fragments(FullRegionData)
        S = len(samples)
        levels = sorted(set(conds))  # character -> factor: alphabetical levels, first = reference (A0)
        if len(levels) != 2:
            raise ValueError(f"design ~condition needs exactly two conditions, got {levels}")
        group = np.array([levels.index(c) for c in conds], dtype=np.int32)

        # window sums (a2) -> counts and FullMean matrices, resident in HBM from here on
        if region_ptr is None:
            d_N, d_FM = ctx.window_sums(FullRegionData["fragN"], FullRegionData["fragFullMean"], FullRegionData["region_ptr"])
        else:
            d_N, d_FM = ctx.window_sums(ctx.to_device(fragN, np.int32), ctx.to_device(fragFM, np.float64),
                                        torch.as_tensor(region_ptr).to(ctx.device))
        n = d_N.shape[1]
        null_sf = ctx.size_factors(d_N)  # estimateSizeFactors (a5)

        want = ["baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "maxCooks", "cooksArgmax"]
        tt = None
        if norm == "standard":
            d_nf = torch.as_tensor(null_sf, device=ctx.device)[:, None].expand(S, n).contiguous()
            label = "Standard DESeq2 normalisation"
        elif norm == "fullmean":
            d_nf = ctx.offsets(d_FM, null_sf, None)
            label = "Chicago full mean-based normalisation"
        else:
            tt = theta
            if tt is None:
                message("Optimising scaling factors...")
                deviances = ctx.theta_grid(d_N, d_FM, null_sf, Grid)
                message("Total deviances by theta (Fullmean --> Standard):")
                print(" ".join(f"{x:f}" for x in deviances), file=sys.stderr)
                if np.any(np.isnan(deviances)):
                    # sum(mcols$deviance) without na.rm (chicdiff.R:1647): an all-zero row makes every
                    # deviance NA and the reference then fails on an empty `tt`
                    raise ValueError("theta grid: a region has zero counts in every sample, total deviance is NA")
                sel = [g for g, dv in zip(Grid, deviances) if dv == deviances.min()]  # which(deviances == min)
                tt = sel[0] if len(sel) == 1 else sel
                if isinstance(tt, list):
                    raise ValueError(f"theta grid: ties at theta = {tt}")
            message("Theta=", tt)
            d_nf = ctx.offsets(d_FM, null_sf, tt)
            label = "combined normalisation"

        out, sc = ctx.nbglm_fit(d_N, d_nf, group, want=want)
        if sc["status"] & hip.ST_TREND_LOCAL:
            # DESeq2's own reaction to a failed parametric fit, and its message (estimateDispersionsFit)
            message("-- note: fitType='parametric', but the dispersion trend was not well captured by the\n"
                    "   function: y = a/x + b, and a local regression fit was automatically substituted.\n"
                    "   specify fitType='local' or 'mean' to avoid this message next time.")
        if sc["status"] & hip.ST_TREND_FAILED:
            # not even the local regression could be fitted (fewer than four rows with a usable dispersion estimate);
            # DESeq2's other documented alternative, fitType = "mean", is available on request (new optional setting)
            if chicdiff_settings.get("trendFallback") != "mean":
                raise RuntimeError('no dispersion trend could be fitted (parametric and local fits both failed): set '
                                   'chicdiff_settings["trendFallback"] = "mean" to refit with fitType = "mean"')
            message('-- note: no dispersion trend could be fitted; fitType = "mean" was used instead.')
            out, sc = ctx.nbglm_fit(d_N, d_nf, group, want=want, opts=hip.default_opts(fitType=1))

        message("Processing model output")
        rmap = _read_rmap(rmapfile)
        if isinstance(RU, dict) and "region_ptr" in RU:   # pipeline.RegionUniverse: the CSR view already holds min / max per region
            mn, mx = RU["minOE"].cpu().numpy(), RU["maxOE"].cpu().numpy()
            ru = pd.DataFrame({"regionID": np.arange(1, len(mn) + 1), "baitID": RU["peak_baitID"], "minOE": mn, "maxOE": mx})
            ru = ru[ru["minOE"] != np.iinfo(np.int32).min]   # a region without any fragment left has no RU row
        else:
            ru = RU.groupby("regionID", sort=True).agg(baitID=("baitID", "first"), minOE=("otherEndID", "min"),
                                                       maxOE=("otherEndID", "max")).reset_index()
        anno = ru.merge(rmap[["otherEndID", "OEchr", "OEstart"]], left_on="minOE", right_on="otherEndID").drop(columns="otherEndID")
        anno = anno.merge(rmap[["otherEndID", "OEend"]], left_on="maxOE", right_on="otherEndID").drop(columns="otherEndID")
        bait = rmap.rename(columns={"OEchr": "baitchr", "OEstart": "baitstart", "OEend": "baitend", "otherEndID": "baitID"})
        anno = anno.merge(bait, on="baitID").sort_values("regionID", kind="stable").reset_index(drop=True)
        if len(anno) != n or not np.array_equal(anno["regionID"].to_numpy(), np.arange(1, n + 1)):
            raise AssertionError("identical(1:nrow(annoData), annoData$regionID) is not TRUE")

        # results(): Cook's cutoff, independent filtering, BH (a9) — on the device
        sizes = [int((group == 0).sum()), int((group == 1).sum())]
        if S > 2 and max(sizes) >= 3:  # DESeq2 applies the Cook's cutoff only when some group has >= 3 replicates
            from scipy import stats
            ctx.cooks_filter(d_N, group, out["maxCooks"], out["cooksArgmax"], out["pvalue"], stats.f.ppf(0.99, 2, S - 2))
        d_padj, _info = ctx.independent_filtering(out["baseMean"], out["pvalue"])
        host = {k: v.cpu().numpy() for k, v in out.items()}
        pvalue, padj = host["pvalue"], d_padj.cpu().numpy()
        message(f"{label}: # unweighted interactions with padj<0.05: ", int(np.sum(padj < 0.05)))
        if save_rds:
            message("Saving the DESeq object")
            np.savez(f"{outprefix}_DESeqObj{suffix}.npz", trendCoef=sc["trendCoef"], dispPriorVar=sc["dispPriorVar"],
                     varLogDispEsts=sc["varLogDispEsts"], theta=np.nan if tt is None else tt, **host)

        res = pd.DataFrame({"baseMean": host["baseMean"], "log2FoldChange": host["log2FoldChange"],
                            "lfcSE": host["lfcSE"], "stat": host["stat"], "pvalue": pvalue, "padj": padj})
        cols = ["baitID", "maxOE", "minOE", "regionID", "OEchr", "OEstart", "OEend", "baitchr", "baitstart", "baitend"]
        result = pd.concat([res, anno[cols]], axis=1)
        if norm == "combined":
            result.attrs["theta"] = tt
        return result
    finally:
        if own_ctx:
            ctx.close()
