"""A synthetic Chicdiff experiment on the reference's real chr19 HindIII design (tests/golden/chr19_design.npz), written
as the files chicdiffPipeline() reads: design files, a peak matrix, one Chicago table and one .chinput per replicate.
Used by the pipeline tests (GPU) and by the host-logic tests of chicdiff_amd/pipeline.py (CPU)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPLICATES = {"CD4": ["NCD4_22", "NCD4_23"], "Mono": ["Mon_2", "Mon_3"]}   # the conditions / replicates of the reference's own run


def golden_settings():
    """The reference's settings list exactly as its own chr19 run saved it (test_settings.Rds -> chr19_settings.json)."""
    with open(os.path.join(HERE, "golden", "chr19_settings.json")) as f:
        return json.load(f)


def write_design(tmp):
    d = np.load(os.path.join(HERE, "golden", "chr19_design.npz"))
    rmapfile, baitmapfile = os.path.join(tmp, "chr19.rmap"), os.path.join(tmp, "chr19.baitmap")
    with open(rmapfile, "w") as f:                               # space-separated, quoted chromosome: the reference's own format
        for c, s, e, i in zip(d["rmap_chr"], d["rmap_start"], d["rmap_end"], d["rmap_id"]):
            f.write(f'"{c}" {s} {e} {i}\n')
    with open(baitmapfile, "w") as f:
        for c, s, e, i in zip(d["bait_chr"], d["bait_start"], d["bait_end"], d["bait_id"]):
            f.write(f'"{c}" {s} {e} {i} "gene{i}"\n')
    return rmapfile, baitmapfile, d


def make_experiment(tmp, seed=7, npeaks=2500, with_chinput=True):
    """Returns (settings, truth): ``settings`` = the golden settings with only the file entries replaced;
    ``truth`` = what the generator knows (the peaks that survive the filter, per-replicate tables, counts)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    tmp = str(tmp)
    rmapfile, baitmapfile, d = write_design(tmp)
    ids = d["rmap_id"].astype(np.int64)
    id_min, nid = int(ids[0]), len(ids)
    start, end = d["rmap_start"].astype(np.int64), d["rmap_end"].astype(np.int64)
    baits = d["bait_id"].astype(np.int64)
    names = [r for reps in REPLICATES.values() for r in reps]
    S = len(names)

    # ---- peak matrix (Chicago makePeakMatrix layout: 11 key columns + one score column per replicate) ----------------
    pb = rng.choice(baits, npeaks)
    off = rng.integers(2, 60, npeaks) * rng.choice([-1, 1], npeaks)
    off[: npeaks // 10] = rng.choice([-1, 1], npeaks // 10)                      # directly adjacent: filtered out
    po = np.clip(pb + off, ids[0], ids[-1])
    keep = po != pb
    pb, po = pb[keep], po[keep]
    pair = np.unique(np.stack([pb, po], 1), axis=0)
    pb, po = pair[:, 0], pair[:, 1]
    m = len(pb)
    scores = rng.gamma(2.0, 3.0, (m, S))
    scores[rng.random((m, S)) < 0.15] = np.nan
    scores[: m // 8] = np.minimum(np.nan_to_num(scores[: m // 8], nan=1.0), 4.0)  # nothing above the threshold: filtered out
    mid = (start + end) / 2.0
    dist = mid[po - id_min] - mid[pb - id_min]
    dist_col = dist.copy()
    trans = rng.random(m) < 0.03
    dist_col[trans] = np.nan                                                      # trans calls carry dist = NA: filtered out
    peak = pd.DataFrame({"baitChr": "19", "baitStart": start[pb - id_min], "baitEnd": end[pb - id_min], "baitID": pb,
                         "baitName": [f"gene{b}" for b in pb], "oeChr": "19", "oeStart": start[po - id_min],
                         "oeEnd": end[po - id_min], "oeID": po, "oeName": ".", "dist": dist_col})
    for j, nm in enumerate(names):
        peak[nm] = scores[:, j]
    peakfile = os.path.join(tmp, "peakMatrix.txt")
    peak.to_csv(peakfile, sep="\t", index=False, na_rep="NA")
    # the filter of readAndFilterPeakMatrix (chicdiff.R:250-270), restated
    ok = (np.nan_to_num(scores, nan=-1.0) > 5.0).any(1)
    for cond, reps in REPLICATES.items():
        cols = [names.index(r) for r in reps]
        ok &= (~np.isnan(scores[:, cols])).sum(1) >= 2
    ok &= ~trans & (np.abs(po - pb) != 1)
    truth = dict(peak_bait=pb[ok].astype(np.int32), peak_oe=po[ok].astype(np.int32), names=names, id_min=id_min, nid=nid,
                 midsum=start + end)

    # ---- per-replicate Chicago tables + chinputs ------------------------------------------------------------------------
    levB = ["(0,25]", "(25,60]", "(60,150]"]
    levL = ["(0,3]", "(3,9]", "(9,30]", "(30,100]"]
    nbin = 75
    refmean = np.exp(3.0 - 0.9 * np.log(np.arange(1, nbin + 1)))                # decreasing in distance
    bait_tblb = {b: levB[k] for b, k in zip(baits, rng.integers(0, len(levB), len(baits)))}
    files_rds, files_chin, xs, tables, chin_cols = {}, {}, [], [], []
    depth = rng.lognormal(0, 0.15, S)
    group = np.array([0, 0, 1, 1])
    for j, nm in enumerate(names):
        # observed pairs: every bait sees a window of other ends around it, dense enough that control regions carry counts
        if with_chinput:
            ob = np.repeat(baits, 140)
            oo = ob + rng.integers(-70, 71, len(ob))
        else:   # without chinputs a pair must be in EVERY replicate's table to count (Reduce(merge)): cover the whole window
            ob = np.repeat(baits, 141)
            oo = ob + np.tile(np.arange(-70, 71), len(baits))
        okp = (oo >= ids[0]) & (oo <= ids[-1]) & (oo != ob)
        key = np.unique((ob[okp] << 32) | oo[okp])
        key = key[rng.random(len(key)) < (0.8 if with_chinput else 0.99)]   # (without chinputs a pair must be in EVERY table to count: keep regions non-empty)
        ob, oo = key >> 32, key & 0xFFFFFFFF
        dd = mid[oo - id_min] - mid[ob - id_min]
        mu = 40.0 * depth[j] / (1.0 + np.abs(dd) / 2e4)
        lfc = np.where((ob % 7 == 0), 1.2, 0.0) * group[j]
        N = rng.negative_binomial(8.0, 8.0 / (8.0 + mu * 2.0 ** lfc)).astype(np.int32) + 1   # Chicago holds observed pairs: N >= 1
        s_j_all = np.exp(rng.normal(0, 0.25, nid))
        s_j_all[(baits[::29]) - id_min] = np.nan                                 # baits Chicago filtered out
        s_i_all = np.exp(rng.normal(0, 0.25, nid))
        tlb_all = np.array(levL + [None], dtype=object)[np.where(rng.random(nid) < 0.9, rng.integers(0, len(levL), nid), len(levL))]
        Tm = np.exp(rng.normal(-2.5, 0.4, (len(levB), len(levL))))
        binno = np.minimum((np.abs(dd) // 20000).astype(np.int64), 10 ** 9)
        inrange = binno < nbin
        x = pd.DataFrame({"baitID": ob.astype(np.int32), "otherEndID": oo.astype(np.int32), "N": N,
                          "distSign": np.rint(dd), "s_j": s_j_all[ob - id_min], "s_i": s_i_all[oo - id_min],
                          "tblb": [bait_tblb[b] for b in ob], "tlb": tlb_all[oo - id_min]})
        x["Tmean"] = [Tm[levB.index(tb), levL.index(tl)] if tl is not None else np.nan for tb, tl in zip(x["tblb"], x["tlb"])]
        x["Bmean"] = x["s_j"] * x["s_i"] * 0.01
        x["score"] = rng.gamma(2.0, 2.0, len(x))
        x["distbin"] = [f"bin{b:03d}" if ir else None for b, ir in zip(binno, inrange)]
        x["refBinMean"] = np.where(inrange, refmean[np.minimum(binno, nbin - 1)], np.nan)
        path = os.path.join(tmp, f"{nm}.chicago.pkl")
        x.sample(frac=1.0, random_state=j).to_pickle(path)                       # any row order
        files_rds[nm] = path
        xs.append(x)
        tables.append(dict(sj=s_j_all, si=s_i_all, tlb=tlb_all, T=Tm, levB=levB, levL=levL, bait_tblb=bait_tblb))
        if with_chinput:
            from test_chinput import write_chinput
            cpath = os.path.join(tmp, f"{nm}.chinput")
            extra = rng.integers(ids[0], ids[-1], (4000, 2))                     # reads of non-bait fragments as well
            kb = np.concatenate([ob, extra[:, 0]]).astype(np.int32)
            ko = np.concatenate([oo, extra[:, 1]]).astype(np.int32)
            kn = np.concatenate([N, rng.integers(1, 9, len(extra)).astype(np.int32)])
            uk, first = np.unique((kb.astype(np.int64) << 32) | ko, return_index=True)
            sh = rng.permutation(len(uk))
            write_chinput(cpath, kb[first][sh], ko[first][sh], kn[first][sh])
            chin_cols.append((kb[first][sh], ko[first][sh], kn[first][sh]))
            files_chin[nm] = cpath
    truth.update(xs=xs, tables=tables, group=group, chinput=chin_cols)

    settings = golden_settings()                                                  # the reference's list, then only its file entries
    settings["peakfiles"] = [peakfile]
    settings["chicagoData"] = {c: {r: files_rds[r] for r in reps} for c, reps in REPLICATES.items()}
    settings["countData"] = {c: {r: files_chin[r] for r in reps} for c, reps in REPLICATES.items()} if with_chinput else None
    settings["rmapfile"] = [rmapfile]
    settings["baitmapfile"] = [baitmapfile]
    settings["outprefix"] = [os.path.join(tmp, "test")]
    return settings, truth


def read_chicago_pickle(path):
    """Stand-in for readRDSorRDA() + x@x / x@params$dispersion (chicdiff.R:614-626) on the tables make_experiment wrote."""
    import pandas as pd
    return pd.read_pickle(path), 2.5


def quantile_ihw(ngroups=5, nfolds=3):
    """Stand-in for IHW::ihw(pvalue ~ abs(avDist), alpha = 0.05) (chicdiff.R:1994): groups = quantile bins of the
    covariate (what ihw() does with a numeric covariate), weights fixed and decreasing with distance.  Returns
    (ihwRes@df, ihwRes@weights)."""
    def ihw(pvalue, covariate, alpha):
        import pandas as pd
        cov = np.asarray(covariate, dtype=np.float64)
        edges = np.quantile(cov[~np.isnan(cov)], np.linspace(0, 1, ngroups + 1))
        group = np.clip(np.searchsorted(edges, cov, side="right"), 1, ngroups).astype(float)
        group[np.isnan(cov)] = np.nan
        w = np.linspace(2.0, 0.4, ngroups)[:, None] * (1.0 + 0.05 * np.arange(nfolds))[None, :]
        return pd.DataFrame({"pvalue": pvalue, "covariate": cov, "group": group}), w
    return ihw
