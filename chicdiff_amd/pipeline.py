"""Host-side mirror of the reference's pipeline driver and of the stages either side of ``DESeq2Wrap`` as they run
with the device path behind them — the tested twin of r/R/getFullRegionData_hip.R and r/R/post_hip.R (R is not installed
here; the reference's own chicdiffPipeline() needs no change).

    chicdiffPipeline(chicdiff.settings)                                         chicdiff.R:301-347
      getRegionUniverse(chicdiff.settings)                                       :369-426   -> device expansion (f4)
      getControlRegionUniverse(chicdiff.settings, RU)                            :456-511   -> host draws, device expansion
      getFullRegionData(chicdiff.settings, RU, RUcontrol, suffix = "")           :1460-1478 -> device blocks, never the long table
      DESeq2Wrap(chicdiff.settings, RU, FullRegionData[[1]])                     :1494      -> chicdiff_amd.deseq2wrap
      DESeq2Wrap(chicdiff.settings, RUcontrol, FullRegionData[[2]], suffix = "Control", theta = attributes(DESeqOut)$theta)
      IHWcorrection(chicdiff.settings, DESeqOut, FullRegionData[[1]], DESeqOutControl, FullRegionData[[2]],
                    countput = FullRegionData[[3]])                              :1956-2065 -> avDist from the device, ihw() stays R

``chicdiff_settings`` is the reference's 17-key list as a dict, passed through UNCHANGED (chicdiff_amd/settings.py);
in particular ``device`` keeps meaning the plot device.  What stays outside this module, as in SURVEY.md §2: reading
Chicago's .Rds/.RDa objects (``read_chicago`` hands over their table) and training ``ihw()`` (``ihw`` hands over its
result); both are R packages the reference calls.  There is no CPU fallback: the HIP library raises when it is missing.
"""
from __future__ import annotations

import sys

import numpy as np

from . import post
from .deseq2wrap import DESeq2Wrap, _read_rmap, message
from .settings import asChicdiffSettings, conditions_per_sample, sample_names

PEAK_KEY_COLUMNS = ["baitChr", "baitStart", "baitEnd", "baitID", "baitName", "oeChr", "oeStart", "oeEnd", "oeID", "oeName", "dist"]


# ---- readAndFilterPeakMatrix, chicdiff.R:218-277 (text I/O; kept on the host as in the reference) -----------------------
def readAndFilterPeakMatrix(peakFiles, targetColumns, chicagoData, conditions, score, outprefix=""):
    import pandas as pd

    read = lambda f: pd.read_csv(f, sep=None, engine="python")
    files = [peakFiles] if isinstance(peakFiles, str) else list(peakFiles)
    x = read(files[0])
    for f in files[1:]:                                    # .multimerge(): merge(all = TRUE) on the 11 key columns
        x = x.merge(read(f), on=PEAK_KEY_COLUMNS, how="outer")
    if not all(c in x.columns for c in targetColumns):
        raise ValueError("All specified targetColumns must be present in the peak file(s)")
    all_baits = pd.unique(x["baitID"])
    x = x[list(x.columns[:11]) + [c for c in x.columns[11:] if c in targetColumns]]
    sel = np.zeros(len(x), dtype=bool)
    for cl in targetColumns:                               # any score > threshold (NA never passes)
        sel |= (x[cl] > score).fillna(False).to_numpy()
    x = x[sel]
    if len(targetColumns) > len(conditions):               # at least 2 non-NA replicates in each condition
        sel2 = np.ones(len(x), dtype=bool)
        for cond in conditions:
            cols = [c for c in x.columns if c in chicagoData[cond]]
            sel2 &= x[cols].notna().sum(axis=1).to_numpy() >= 2
        x = x[sel2]
    x = x[x["dist"].notna()]                               # trans interactions out
    x = x[~((x["oeID"] == x["baitID"] + 1) | (x["oeID"] == x["baitID"] - 1))]   # directly adjacent fragments out
    filtered = np.asarray(all_baits)[~np.isin(all_baits, x["baitID"].to_numpy())]   # one %in%, as chicdiff.R:271
    # chicdiff.R:272-274: fwrite(list(filtered), file) — an UNNAMED list.  data.table's fwrite writes a header line only when the
    # names attribute is a character vector, so the reference's file holds the bare IDs, one per line (round 4 wrote a "V1" line
    # first on an unverified reading; no R here to settle it, so this follows fwriteR.c as recalled — ADVICE r04)
    with open(f"{outprefix}_filteredBaits.txt", "w") as f:
        f.write("".join(f"{b}\n" for b in filtered))
    return x.reset_index(drop=True)


def _rmap_tables(rmapfile):
    """Dense per-ID tables of the restriction map: id_min, midsum (start + end), chromosome codes (-1 = not on the map)."""
    rmap = _read_rmap(rmapfile)
    ids = rmap["otherEndID"].to_numpy(np.int64)
    if len(np.unique(ids)) != len(ids):
        raise ValueError("Duplicated fragment IDs found in rmapfile")
    id_min, nid = int(ids.min()), int(ids.max() - ids.min() + 1)
    midsum = np.zeros(nid, dtype=np.int64)
    midsum[ids - id_min] = rmap["OEstart"].to_numpy(np.int64) + rmap["OEend"].to_numpy(np.int64)
    names, codes = np.unique(rmap["OEchr"].astype(str).to_numpy(), return_inverse=True)
    chr_codes = np.full(nid, -1, dtype=np.int32)
    chr_codes[ids - id_min] = codes.astype(np.int32)
    return rmap, id_min, midsum, chr_codes, names


class RegionUniverse(dict):
    """RU as the device path carries it: ``baitID, regionID, otherEndID`` in RU.DT's order (device tensors) plus the
    region-major CSR view ``region_ptr, minOE, maxOE, csr_baitID, csr_regionID, csr_otherEndID``."""

    def to_frame(self):
        import pandas as pd
        return pd.DataFrame({k: self[k].cpu().numpy() for k in ("baitID", "regionID", "otherEndID")})


def _expand(ctx, baitID, oeID, RUexpand, rmap):
    ru = RegionUniverse(post.getRegionUniverse(ctx, baitID, oeID, RUexpand, rmap["OEchr"].astype(str).to_numpy(),
                                               rmap["otherEndID"].to_numpy()))
    ru["peak_baitID"] = np.asarray(baitID, dtype=np.int32)
    return ru


def getRegionUniverse(chicdiff_settings, ctx, suffix=""):
    """chicdiff.R:369-426 (window mode): peaks -> regions; the expansion, the clip to the map and the cis filter run on
    the device (chicdiff_hip_region_universe_*)."""
    s = asChicdiffSettings(chicdiff_settings)
    x = readAndFilterPeakMatrix(s["peakfiles"], s["targetColumns"], s["chicagoData"], list(s["chicagoData"]), s["score"],
                                s["outprefix"])
    ru = _expand(ctx, x["baitID"].to_numpy(np.int32), x["oeID"].to_numpy(np.int32), s["RUexpand"], _read_rmap(s["rmapfile"]))
    if s["saveAuxData"]:
        ru.to_frame().to_csv(f"{s['outprefix']}_RegionUniverse{suffix}.csv", index=False)
    return ru


def getControlRegionUniverse(chicdiff_settings, RU, ctx, rng=None):
    """chicdiff.R:456-511: as many control regions as test regions, around random baits at N(0, maxContact / 3) fragment
    offsets (giveDists / giveOneSeed, :430-449; the reference draws unseeded, so only the distribution is reproduced —
    pass ``rng`` for a repeatable run), expanded like the test regions."""
    import pandas as pd
    s = asChicdiffSettings(chicdiff_settings)
    rng = np.random.default_rng() if rng is None else rng
    rmap = _read_rmap(s["rmapfile"])
    bmap = pd.read_csv(s["baitmapfile"], sep=r"\s+", header=None, quotechar='"', engine="python").iloc[:, :4]
    bmap.columns = ["chr", "start", "end", "ID"]
    ru_b, ru_o = RU["baitID"].cpu().numpy(), RU["otherEndID"].cpu().numpy()
    # merge(RU, rmap[, c("chr", "ID")], by.x = "baitID", by.y = "ID") (:465): an inner join — RU rows whose bait is not on the
    # map drop out silently; dense chromosome codes per ID, -1 = not on the map
    ids = rmap["otherEndID"].to_numpy(np.int64)
    names, codes = np.unique(rmap["OEchr"].astype(str).to_numpy(), return_inverse=True)
    code_of = np.full(int(max(ids.max(), ru_b.max())) + 2, -1, dtype=np.int64)
    code_of[ids] = codes
    ru_code = code_of[ru_b.astype(np.int64)]
    on_map = ru_code >= 0
    contact = pd.DataFrame({"chr": names[ru_code[on_map]], "d": np.abs(ru_b.astype(np.int64) - ru_o)[on_map]}).groupby("chr")["d"].max()   # :466
    n_regions = len(np.unique(RU["regionID"].cpu().numpy()))
    draw = rng.choice(bmap["ID"].to_numpy(), size=n_regions, replace=True)                                           # :468
    ctrl = pd.DataFrame({"ID": draw}).merge(bmap[bmap["chr"].astype(str).isin(contact.index)][["chr", "ID"]], on="ID")
    ctrl["chr"] = ctrl["chr"].astype(str)
    seeds = np.zeros(len(ctrl), dtype=np.int64)
    for ch, idx in ctrl.groupby("chr").indices.items():
        ids = rmap.loc[rmap["OEchr"].astype(str) == ch, "otherEndID"]
        lo, hi, std = int(ids.min()), int(ids.max()), contact[ch] / 3.0
        bait = ctrl["ID"].to_numpy(np.int64)[idx]
        dist = np.zeros(len(bait), dtype=np.int64)
        todo = np.ones(len(bait), dtype=bool)
        while todo.any():                                                          # giveDists: redraw until on the chromosome and != 0
            d = np.rint(rng.normal(0.0, std, todo.sum())).astype(np.int64)
            ok = (((bait[todo] + np.abs(d)) < hi) | ((bait[todo] - np.abs(d)) > lo)) & (d != 0)
            where = np.flatnonzero(todo)
            dist[where[ok]] = d[ok]
            todo[where[ok]] = False
        fwd = bait + dist
        seeds[idx] = np.where((fwd < lo) | (fwd > hi), bait - dist, fwd)           # giveOneSeed
    ctrl["oeID"] = seeds
    ctrl = ctrl.sort_values(["ID", "oeID"], kind="stable").reset_index(drop=True)  # setkey(baitID, oeID); regionID := 1:nrow
    ru = _expand(ctx, ctrl["ID"].to_numpy(np.int32), ctrl["oeID"].to_numpy(np.int32), s["RUexpand"], rmap)
    if s["saveAuxData"]:
        ru.to_frame().to_csv(f"{s['outprefix']}_ControlRegionUniverse.csv", index=False)
    return ru


# ---- the Chicago side: per-fragment tables of one replicate (chicdiff.R:656-692, 538-573) -----------------------------
def chicEstimateDistFun(x, binsize=20000):
    """.chicEstimateDistFun, chicdiff.R:538-573: cubic lm() of log(refBinMean) on log(bin midpoint), linear head and
    tail by continuity of f and f'.  Returns the ten numbers chicdiff_hip_fragment_background_dev takes."""
    fd = x[["distbin", "refBinMean"]].drop_duplicates().dropna(subset=["refBinMean"])
    fd = fd.sort_values("refBinMean", ascending=False, kind="stable")
    mid = round(binsize / 2) + binsize * np.arange(len(fd), dtype=np.float64)
    lm, y = np.log(mid), np.log(fd["refBinMean"].to_numpy(np.float64))
    fit = np.linalg.lstsq(np.stack([np.ones_like(lm), lm, lm ** 2, lm ** 3], axis=1), y, rcond=None)[0]
    ends = np.array([lm.min(), lm.max()])
    beta = fit[1] + 2 * fit[2] * ends + 3 * fit[3] * ends ** 2
    alpha = fit[0] + (fit[1] - beta) * ends + fit[2] * ends ** 2 + fit[3] * ends ** 3
    return np.array([*fit, alpha[0], beta[0], alpha[1], beta[1], ends[0], ends[1]])


def background_tables(xs, id_min, nid):
    """Dense per-fragment tables of the S Chicago data sets ``xs`` (DataFrames with the columns of chicagoData@x):
    first s_j / tblb per bait, first s_i / tlb per other end (chicdiff.R:656-672), Tmean of every (tblb, tlb) pair
    (:676-681), the distance function (:538-573) — the ``background`` argument of post.getFullRegionDataHip."""
    S = len(xs)
    lev = lambda col: sorted({str(v) for x in xs for v in x[col].dropna().unique()})
    levB, levL = lev("tblb"), lev("tlb")
    codeB, codeL = {v: i for i, v in enumerate(levB)}, {v: i for i, v in enumerate(levL)}
    sj, si = np.full((S, nid), np.nan), np.full((S, nid), np.nan)
    tblb, tlb = np.full((S, nid), -1, dtype=np.int32), np.full((S, nid), -1, dtype=np.int32)
    T = np.full((S, max(len(levB), 1), max(len(levL), 1)), np.nan)
    distfun = np.zeros((S, 10))
    for s, x in enumerate(xs):
        x = x.sort_values(["baitID", "otherEndID"], kind="stable")               # setkey(x, baitID, otherEndID), :630
        b = x.drop_duplicates("baitID", keep="first")
        b = b[(b["baitID"] >= id_min) & (b["baitID"] < id_min + nid)]
        sj[s, b["baitID"].to_numpy(np.int64) - id_min] = b["s_j"].to_numpy(np.float64)
        tblb[s, b["baitID"].to_numpy(np.int64) - id_min] = [codeB.get(str(v), -1) if v == v and v is not None else -1 for v in b["tblb"]]
        o = x.drop_duplicates("otherEndID", keep="first")
        o = o[(o["otherEndID"] >= id_min) & (o["otherEndID"] < id_min + nid)]
        si[s, o["otherEndID"].to_numpy(np.int64) - id_min] = o["s_i"].to_numpy(np.float64)
        tlb[s, o["otherEndID"].to_numpy(np.int64) - id_min] = [codeL.get(str(v), -1) if v == v and v is not None else -1 for v in o["tlb"]]
        tm = x.dropna(subset=["tblb", "tlb"]).sort_values(["tlb", "tblb"], kind="stable").drop_duplicates(["tblb", "tlb"], keep="first")
        for tb, tl, v in zip(tm["tblb"], tm["tlb"], tm["Tmean"]):
            T[s, codeB[str(tb)], codeL[str(tl)]] = v
        distfun[s] = chicEstimateDistFun(x)
    return dict(sj=sj, si=si, tblb=tblb, tlb=tlb, T=T, distfun=distfun, levB=levB, levL=levL)


def _countput(xs, conditions, rmap):
    """countput, chicdiff.R:708-735 + :754-768: per condition the replicates' observed pairs (non-NA distSign) stacked,
    then Nav = mean(N), Bav = mean(Bmean), score = max(score), the other end's midpoint — what plotDiffBaits() draws."""
    import pandas as pd
    mid = pd.DataFrame({"otherEndID": rmap["otherEndID"], "midpoint": (rmap["OEstart"] + rmap["OEend"]) / 2})
    out = []
    for cond in dict.fromkeys(conditions):
        parts = []
        for x, c in zip(xs, conditions):
            if c != cond:
                continue
            sc = "newScore" if "newScore" in x.columns else "score"
            y = x.loc[x["distSign"].notna(), ["baitID", "otherEndID", "N", "Bmean", sc]].rename(columns={sc: "score"})
            parts.append(y.merge(mid, on="otherEndID"))
        z = pd.concat(parts).groupby(["baitID", "otherEndID"], sort=False).agg(
            Nav=("N", "mean"), Bav=("Bmean", "mean"), score=("score", "max"), oeID_mid=("midpoint", "first")).reset_index()
        z["condition"] = cond
        out.append(z)
    return pd.concat(out, ignore_index=True)


def getFullRegionData(chicdiff_settings, RU, RUcontrol, suffix="", ctx=None, read_chicago=None):
    """chicdiff.R:1460-1478 with the device path behind it: list(test block, control block, countput).  Every Chicago
    data set and every chinput file is read ONCE for both universes (what ``parallel = TRUE`` -> getFullRegionData2,
    :948-1456, does in the reference; the result does not depend on it).  The long "recast" table (one row per region,
    fragment and sample) is never built: a block holds the per-sample fragment columns N and FullMean on the device in
    (regionID, otherEndID) order, the region offsets, and IHWcorrection()'s per-region avDist."""
    s = asChicdiffSettings(chicdiff_settings)
    if read_chicago is None:
        raise ValueError("read_chicago: a reader of the Chicago data sets is required (readRDSorRDA stays R)")
    torch = ctx.torch
    chicagoData, countData = s["chicagoData"], s["countData"]
    names, conditions = sample_names(chicagoData), conditions_per_sample(chicagoData)
    paths = [p for reps in chicagoData.values() for p in reps.values()]
    S = len(paths)
    rmap, id_min, midsum, chr_codes, _ = _rmap_tables(s["rmapfile"])
    nid = len(midsum)
    dev = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(ctx.device)

    xs, dispersions = [], np.zeros(S)
    for i, p in enumerate(paths):
        message("\nReading Chicago dataset ", i + 1, " of ", S, " : ", names[i])
        x, dispersions[i] = read_chicago(p)
        xs.append(x)
    bg = background_tables(xs, id_min, nid)
    bg.update(id_min=id_min, midsum=midsum)
    message("Saving counts\n")
    countput = _countput(xs, conditions, rmap)
    countput.to_csv(f"{s['outprefix']}_countput.csv", index=False)                 # the reference: saveRDS(_countput.Rds), :769

    universes = [RU, RUcontrol]
    baits = np.unique(np.concatenate([u["csr_baitID"].cpu().numpy() for u in universes]))   # sort(unique(RU$baitID)), :775
    flags = np.zeros(int(baits.max()) + 1, dtype=np.uint8)
    flags[baits] = 1
    d_flags = torch.from_numpy(flags).to(ctx.device)
    tables = []
    if countData is not None:                                                      # chicdiff.R:811-858
        cpaths = [p for reps in countData.values() for p in reps.values()]
        cnames = sample_names(countData)
        for i, p in enumerate(cpaths):
            message("Reading count data for ", cnames[i])
            keys, vals, _ = ctx.read_chinput(p, d_flags)
            tables.append((keys, vals))
    else:                                                                          # chicdiff.R:742-747, 774-807
        message("Reconstructing countData")
        for x in xs:
            tables.append(ctx.count_table(dev(x["baitID"].to_numpy(), np.int32), dev(x["otherEndID"].to_numpy(), np.int32),
                                          dev(x["N"].to_numpy(), np.int32), d_flags))

    d_midsum, d_chr = dev(midsum, np.int64), dev(chr_codes, np.int32)
    d_bg = {k: dev(bg[k], t) for k, t in (("sj", np.float64), ("si", np.float64), ("tblb", np.int32), ("tlb", np.int32), ("T", np.float64))}
    blocks = []
    for u, is_control in zip(universes, (False, True)):
        message("Reading data for significant interactions" if not is_control else "\nReading data for control interactions")
        d_bait, d_oe, ptr = u["csr_baitID"], u["csr_otherEndID"], u["region_ptr"]
        nfrag, n = d_bait.numel(), ptr.numel() - 1
        if countData is not None:
            fragN = ctx.count_join_multi(d_bait, d_oe, tables)   # the replicate loop of :843-858 as one pass over the RU rows
        else:
            message("Merging countData")
            fragN = ctx.count_join_inner(d_bait, d_oe, tables)
        _, _, fragFM = ctx.fragment_background(d_bait, d_oe, id_min, d_midsum, d_bg["sj"], d_bg["si"], d_bg["tblb"], d_bg["tlb"],
                                               d_bg["T"], bg["distfun"], only_fullmean=True)
        avDist = ctx.region_avdist(d_bait, d_oe, ptr, id_min, d_midsum, d_chr)
        blocks.append(post.HipRegionData(samples=names, condition=list(conditions), S=S, n=n, fragN=fragN, fragFullMean=fragFM,
                                         region_ptr=ptr, avDist=avDist, dispersions=dispersions, is_control=is_control))
    return [blocks[0], blocks[1], countput]


# ---- IHWcorrection, chicdiff.R:1956-2065 ------------------------------------------------------------------------------
def _avdist(FullRegionData, ctx):
    if isinstance(FullRegionData, post.HipRegionData):
        return FullRegionData["avDist"]
    # the reference's long table: RU.recast[, list(avDist = mean(distSign)), by = "regionID"] (:1965)
    av = FullRegionData.groupby("regionID", sort=True)["distSign"].mean().to_numpy(np.float64)
    return ctx.torch.from_numpy(av).to(ctx.device)


def dist_lookup(ihw_df, ihw_weights):
    """distLookup, chicdiff.R:2012-2031, from ihw()'s result: ``ihw_df`` = ihwRes@df (columns covariate, group),
    ``ihw_weights`` = ihwRes@weights (groups x folds)."""
    df = ihw_df[ihw_df["group"].notna()]
    lc = np.log(df["covariate"].to_numpy(np.float64))
    g = df["group"].to_numpy().astype(np.int64)
    groups = np.unique(g)
    if not np.array_equal(groups, np.arange(1, len(groups) + 1)):
        raise ValueError("Assumption violated")                                    # :2023-2026
    lo = np.array([lc[g == k].min() for k in groups])
    hi = np.array([lc[g == k].max() for k in groups])
    w = np.asarray(ihw_weights, dtype=np.float64)
    avW = w.sum(axis=1) / w.shape[1]                                               # rowSums(w) / ncol(w)
    lo[0], hi[-1] = 0.0, np.inf                                                    # :2030-2031
    return dict(group=groups, minLogDist=lo, maxLogDist=hi, avWeights=avW)


def IHWcorrection(chicdiff_settings, DESeqOut, FullRegionData, DESeqOutControl, FullControlRegionData, countput=None,
                  DiagPlot=True, diffbaitPlot=True, suffix="", ctx=None, ihw=None, rng=None):
    """chicdiff.R:1956-2065.  ``ihw(pvalue, covariate, alpha)`` stands for IHW::ihw(pvalue ~ abs(avDist), data =
    out.control, alpha = 0.05) (:1994; an R package the reference calls — SURVEY.md §2 row 10) and returns
    ``(df, weights)`` = (ihwRes@df with columns covariate and group, ihwRes@weights).  The covariate comes from the
    device blocks, the application side (:2038-2049) runs on the device; the diagnostic plots stay R."""
    import pandas as pd
    s = asChicdiffSettings(chicdiff_settings)
    if ihw is None:
        raise ValueError("ihw: the trained weights are required (IHW::ihw stays R)")
    rng = np.random.default_rng() if rng is None else rng
    out, ctl = DESeqOut.copy(), DESeqOutControl.copy()
    d_av = _avdist(FullRegionData, ctx)
    out["avDist"] = d_av.cpu().numpy()                                             # by position = regionID order (:1967)
    out["uniform"] = rng.uniform(size=len(out))                                    # :1970-1971 (unseeded in the reference)
    out["shuff"] = rng.permutation(out["pvalue"].to_numpy())
    message("Comparison against p-vals for out")
    ctl["avDist"] = _avdist(FullControlRegionData, ctx).cpu().numpy()
    ctl["uniform"] = rng.uniform(size=len(ctl))
    ctl["shuff"] = rng.permutation(ctl["pvalue"].to_numpy())
    message("Comparison against p-vals for outcontrol")
    ihw_df, ihw_w = ihw(ctl["pvalue"].to_numpy(), np.abs(ctl["avDist"].to_numpy()), 0.05)
    message("Trained weights on the control sample")
    look = dist_lookup(ihw_df, ihw_w)
    message("Learned distance dependency")
    w = post.applyIHWweights(ctx, d_av, ctx.torch.from_numpy(out["pvalue"].to_numpy(np.float64)).to(ctx.device),
                             look["minLogDist"], look["maxLogDist"], look["avWeights"])
    out["avgLogDist"] = np.log(np.abs(out["avDist"].to_numpy()))
    group = w["group"].cpu().numpy()
    out.insert(0, "group", np.where(group == np.iinfo(np.int32).min, -1, group))
    out["avWeights"] = np.where(group > 0, look["avWeights"][np.clip(group, 1, None) - 1], np.nan)
    out["weight"] = w["weight"].cpu().numpy()
    out["weighted_pvalue"] = w["weighted_pvalue"].cpu().numpy()
    out["weighted_padj"] = w["weighted_padj"].cpu().numpy()
    message("applied to test data")
    attrs = dict(out.attrs)
    out = out.sort_values("group", kind="stable").reset_index(drop=True)           # merge(out, distLookup, by = "group") re-sorts
    out.attrs.update(attrs)
    out.to_csv(f"{s['outprefix']}_results{suffix}.csv", index=False)               # the reference: saveRDS(_results.Rds), :2062
    return out


def chicdiffPipeline(chicdiff_settings, ctx=None, read_chicago=None, ihw=None, rng=None):
    """chicdiff.R:301-347, same stage order and messages."""
    from . import hip
    from .settings import hipDevice
    own = ctx is None
    if own:
        ctx = hip.HipContext(hipDevice(chicdiff_settings))
    try:
        s = asChicdiffSettings(chicdiff_settings)
        message("\n*** Running getRegionUniverse\n")
        RU = getRegionUniverse(chicdiff_settings, ctx)
        message("\n*** Running getControlRegionUniverse\n")
        RUcontrol = getControlRegionUniverse(chicdiff_settings, RU, ctx, rng=rng)
        message("\n*** Running getFullRegionData\n")
        FullRegionData = getFullRegionData(chicdiff_settings, RU, RUcontrol, suffix="", ctx=ctx, read_chicago=read_chicago)
        message("\n*** Running DESeq2Wrap for FullRegion\n")
        DESeqOut = DESeq2Wrap(chicdiff_settings, RU, FullRegionData[0], ctx=ctx)
        message("\n*** Running DESeq2Wrap for FullControlRegion\n")
        if s["norm"] == "combined" and DESeqOut.attrs.get("theta") is None and s["theta"] is None:
            import warnings
            warnings.warn("Normalisation weight theta is not defined and its inference may fail on control regions")
        DESeqOutControl = DESeq2Wrap(chicdiff_settings, RUcontrol, FullRegionData[1], suffix="Control",
                                     theta=DESeqOut.attrs.get("theta"), ctx=ctx)
        message("\n*** Running IHWcorrection\n")
        output = IHWcorrection(chicdiff_settings, DESeqOut, FullRegionData[0], DESeqOutControl, FullRegionData[1],
                               countput=FullRegionData[2], ctx=ctx, ihw=ihw, rng=rng)
        print({k: v for k, v in s.items()}, file=sys.stderr)
        return output
    finally:
        if own:
            ctx.close()
