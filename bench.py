#!/usr/bin/env python3
"""bench.py — interactions/sec of one complete NB-GLM Wald test on MI355X.

One "step" = one pass of the hot path over one resident batch (BASELINE.json):
    size factors (a5)  ->  offsets sc(theta) from FullMean (a4)  ->  estimateDispersions +
    nbinomWaldTest equivalents, design ~condition (a6 + a7)
on a synthetic n x S count matrix (default: configs[2] of BASELINE.json, 2 M x 8, 4v4 — the
configuration the metric is quoted on; it fits one GPU).  Inputs are resident in HBM when the
timed region starts; outputs stay in HBM.

N > 1: one rank per GPU, rows sharded in contiguous blocks (chicdiff_amd/dist.py:shard_bounds), the
global statistics (size-factor medians, trend sums, MAD, deviance) go through RCCL sum-all-reduces.
`python bench.py --gpus N` from a bare shell starts its own ranks (a child `python -m
torch.distributed.run`, spawned before this process touches a GPU); under an external
torch.distributed.run (WORLD_SIZE set) it is a rank.  --scaling strong (default: BASELINE.json
configs[3], "2 M x 8 sharded across 8 MI355X": --rows is the GLOBAL row count, rows/N per rank) or
weak (--rows per rank); for N > 1 the other of the two is measured as well, after the first, and reported
under "other_scaling" (same K and W; `value` is always the --scaling one).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)
FP64_VALU_PEAK_TFLOPS = 78.6  # vector fp64, = 1/2 of the 157.3 TF fp32 figure in the guide


def algorithmic_bytes(S):
    """counts 4S + offsets 8S in, six fp64 results out."""
    return 12 * S + 48


def cpu_baseline(S, theta, sample_rows, threads):
    """Oracle (CPU restatement, kind='port') on a bounded sample of the same workload."""
    from chicdiff_amd import synth
    from oracle import oracle

    d = synth.make(sample_rows, S)
    fm = d["nf"] * (d["mu"][:, None] / S)
    t0 = time.perf_counter()
    sf = oracle.size_factors(d["counts"])
    nf = oracle.offsets(fm, sf, theta)
    oracle.nbglm_fit(d["counts"], nf, d["group"], nthreads=threads)
    dt = time.perf_counter() - t0
    return sample_rows / dt, dt


def hbm_kernels(ctx, torch, n, S, F=11):
    """The HBM-bound rows of the path (a2 window sums, a4 offsets, a1 count join) at the same scale,
    outside the timed region: achieved GB/s on their algorithmic bytes vs 8 TB/s."""
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    fragN = torch.randint(0, 50, (S, n * F), dtype=torch.int32, device=dev, generator=g)
    fragFM = torch.rand((S, n * F), dtype=torch.float64, device=dev, generator=g) + 0.1
    rp = torch.arange(0, (n + 1) * F, F, dtype=torch.int64, device=dev)
    out = {}

    def run(name, fn, nbytes):
        fn()
        ts = []
        for _ in range(5):
            fn()
            ts.append(ctx.kernel_times()[name][0])
        ms = float(np.median(ts))
        out[name] = {"ms": round(ms, 4), "achieved_GBs": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes": nbytes}

    ctx.enable_timing(True)
    run("window_sums", lambda: ctx.window_sums(fragN, fragFM, rp), 12 * n * F * S + 12 * n * S)  # 48*S B/region at F=11 (+outputs)
    _, FM = ctx.window_sums(None, fragFM, rp)
    del fragN, fragFM
    o = torch.empty_like(FM)
    run("offsets", lambda: ctx.offsets(FM, np.ones(S), 0.5, out=o), 16 * n * S)
    nk = 5 * n
    keys = torch.unique(torch.randint(0, 2 ** 40, (nk,), dtype=torch.int64, device=dev, generator=g))
    vals = torch.randint(1, 100, (keys.numel(),), dtype=torch.int32, device=dev, generator=g)
    qk = keys[torch.randint(0, keys.numel(), (n * F,), device=dev, generator=g)]
    qk = torch.sort(torch.where(torch.rand(n * F, device=dev, generator=g) < 0.5, qk + 1, qk)).values  # RU is keyed by baitID
    bait, oe = (qk >> 32).to(torch.int32), (qk & 0xFFFFFFFF).to(torch.int32)
    run("count_join", lambda: ctx.count_join(bait, oe, keys, vals), 12 * n * F + 12 * keys.numel())
    del keys, vals, qk, bait, oe
    # f4: region universe (2 x int32 per peak in, 3 x int32 per RU row + CSR out) and f1/f3: BH over n p-values
    pb = torch.randint(1000, 800000, (n,), dtype=torch.int32, device=dev, generator=g)
    po = pb + torch.randint(2, 60, (n,), dtype=torch.int32, device=dev, generator=g)
    chr_of = (torch.arange(0, 840001, device=dev) // 35000).to(torch.int32)
    ru = ctx.region_universe(pb, po, 5, chr_of)
    nrow = ru["baitID"].numel()
    del ru
    def run_ru():  # one API call since round 5 (scan, fill, one read-back): its kernel time (HIP events on the library's stream)
        ctx.region_universe(pb, po, 5, chr_of)
        ts = []
        for _ in range(5):
            ctx.region_universe(pb, po, 5, chr_of)
            ts.append(ctx.last_region_universe_ms)
        ms = float(np.median(ts))
        nbytes = 2 * (8 * n) + 16 * n + 12 * nrow
        out["region_universe"] = {"ms": round(ms, 4), "achieved_GBs": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes": nbytes, "timing": "kernel time of the one call (scan + fill)"}

    run_ru()
    pv = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    run("bh_adjust", lambda: ctx.bh_adjust(pv), 16 * n)
    bmv = torch.exp(torch.randn(n, dtype=torch.float64, device=dev, generator=g) * 1.4 + 2.9)
    run("independent_filtering", lambda: ctx.independent_filtering(bmv, pv), 24 * n)  # a9: baseMean + p in, padj out
    ctx.enable_timing(False)
    return out


class EndToEndError(RuntimeError):
    pass


def end_to_end(ctx, torch, synth, n, S, F=11, reps=3):
    """The drop-in's whole resident path in the reference's default mode (chicdiffPipeline, chicdiff.R:301-347: norm = "combined",
    theta = NULL), one stage after the other on `n` peaks x `S` replicates, everything resident in HBM: what a user of
    chicdiffPipeline(backend = "hip") waits for between "Chicago objects read" and "ihw() trained" — NOT part of `value`.

      test set   : region universe (:353-426) -> count join x S (:843-858) -> Bmean / Tmean / FullMean (:628-703, 894-896)
                   -> window sums (:1540-1556) -> size factors (:1561-1562) -> theta grid, 5 design-~1 fits (:1619-1662)
                   -> final fit at the chosen theta (:1666-1674) -> results(): Cook's cutoff, independent filtering, BH (:1720-1762)
      control set: the same up to the window sums, then ONE fit at the inherited theta (:331-332) and results()
      IHW        : avDist per region (:1965) and the application block (:2038-2049) with fixed weights (training = IHW::ihw stays R)

    Synthetic inputs: peaks on a 840 001-fragment map, RU rows from the device's own region universe, the synthetic count matrix
    (chicdiff_amd/synth.py; rows [0, n) test, [n, 2n) control) split over each region's fragments by random weights and turned into
    one sorted (baitID, otherEndID) -> N table per replicate (what a .chinput becomes), random Chicago tables (s_j, s_i, Tmean
    bins, distance function).  A region whose counts are all zero gets one read (the reference's theta scan needs every total
    deviance finite: sum() without na.rm, :1647)."""
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    maxfrag = 840000
    chr_of = (torch.arange(0, maxfrag + 1, device=dev) // 35000).to(torch.int32)
    nid = maxfrag + 1
    group = synth.groups(S)
    grid = [0.0, 0.25, 0.5, 0.75, 1.0]

    def peaks():
        pb = torch.randint(1000, 800000, (n,), dtype=torch.int64, device=dev, generator=g)
        dd = torch.randint(2, 60, (n,), dtype=torch.int64, device=dev, generator=g) * (torch.randint(0, 2, (n,), device=dev, generator=g) * 2 - 1)
        # every region keeps fragments on its bait's chromosome (a region the cis filter empties, :406-419, trips the reference's own
        # stopifnot at :1717): where the window around bait + d would leave the chromosome, the peak goes to the other side of the bait
        off_chr = ((pb + dd + 5) // 35000 != pb // 35000) | ((pb + dd - 5) // 35000 != pb // 35000)
        po = pb + torch.where(off_chr, -dd, dd)
        key = torch.sort(pb * (1 << 32) + po).values      # setkey(baitID, oeID): regions bait-major, as the reference numbers them
        return (key >> 32).to(torch.int32), (key & 0xFFFFFFFF).to(torch.int32)

    wrng = np.random.Generator(np.random.PCG64(20190123 + 7))

    def split_counts(ru, k):
        """k (S, n) region counts -> (S, nfrag) fragment counts that sum back to k over each region's rows.  The fragment weights are
        INTEGERS drawn on the host from a seeded PCG64 (Gamma(0.3), scaled, + 1): round 5 drew them with torch._standard_gamma, which
        took no seeded generator (the inputs differed from run to run), and formed a region's shares from differences of a GLOBAL fp64
        running sum — a one-fragment region whose weight fell below the sum's ulp got 0 / 0 and with it an all-zero row (BENCH_r05).
        With integer weights the running sum is exact in int64 and each region's shares are exact ratios: first lo = 0, last hi = 1."""
        ptr = ru["region_ptr"]
        nfrag = int(ru["baitID"].numel())
        cnt = ptr[1:] - ptr[:-1]
        rid = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
        w = torch.from_numpy(np.minimum(wrng.standard_gamma(0.3, nfrag) * 4096.0, 2.0 ** 30).astype(np.int64) + 1).to(dev)
        cum = torch.cumsum(w, 0)
        cum_excl = cum - w
        start = cum_excl[ptr[:-1].clamp(max=nfrag - 1)]
        tot = (cum[(ptr[1:] - 1).clamp(min=0)] - start).to(torch.float64)
        hi = (cum - start[rid]).to(torch.float64) / tot[rid]
        lo = (cum_excl - start[rid]).to(torch.float64) / tot[rid]
        out = torch.empty((S, nfrag), dtype=torch.int32, device=dev)
        for j in range(S):
            kj = k[j][rid].to(torch.float64)
            out[j] = (torch.floor(kj * hi + 1e-9) - torch.floor(kj * lo + 1e-9)).to(torch.int32)
        return out

    # ---- set-up (not timed): both universes, the replicates' count tables, the Chicago tables ----
    sets = {}
    for name, start_row in (("test", 0), ("control", n)):
        pb, po = peaks()
        for _ in range(4):   # (a handful of peaks per million still lose every fragment to the clipping: they take their neighbour's place)
            ru = ctx.region_universe(pb, po, 5, chr_of)
            empty = torch.nonzero((ru["region_ptr"][1:] - ru["region_ptr"][:-1]) == 0).flatten()
            if empty.numel() == 0:
                break
            src = (empty - 1).clamp(min=0)
            pb[empty], po[empty] = pb[src], po[src]
        d = synth.make(n, S, start=start_row)
        k = torch.from_numpy(np.ascontiguousarray(d["counts"].T)).to(dev)
        k[0, k.sum(0) == 0] = 1
        sets[name] = dict(pb=pb, po=po, frag=split_counts(ru, k), key=ru["baitID"].to(torch.int64) * (1 << 32) + ru["otherEndID"].to(torch.int64))
        del ru, k, d
    tables = []
    allkeys = torch.cat([sets["test"]["key"], sets["control"]["key"]])
    for j in range(S):
        v = torch.cat([sets["test"]["frag"][j], sets["control"]["frag"][j]])
        sel = v > 0
        ks, order = torch.sort(allkeys[sel])
        vs = v[sel][order]
        first = torch.ones_like(ks, dtype=torch.bool)
        first[1:] = ks[1:] != ks[:-1]                      # a pair that sits in several regions keeps one count, as in a chinput
        tables.append((ks[first].contiguous(), vs[first].contiguous()))
    for st in sets.values():
        del st["frag"]
    del allkeys

    # the one-read guard, checked where it matters: on the JOINED matrix (after the split and the de-duplication).  A region that the
    # join leaves without a read in every replicate gets one: its first fragment's key goes into replicate 0's table.  (The reference's
    # theta scan needs every total deviance finite: sum() without na.rm, chicdiff.R:1647, 1660.)
    def joined_all_zero(st):
        ru = ctx.region_universe(st["pb"], st["po"], 5, chr_of)
        ptr = ru["region_ptr"].to(torch.int64)
        tot = torch.zeros(n, dtype=torch.int64, device=dev)
        for kk, vv in tables:
            cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ctx.count_join(ru["baitID"], ru["otherEndID"], kk, vv).to(torch.int64), 0)])
            tot += cs[ptr[1:]] - cs[ptr[:-1]]
        return torch.nonzero(tot == 0).flatten(), ptr

    repaired = 0
    for _ in range(3):
        extra = []
        for st in sets.values():
            zero, ptr = joined_all_zero(st)
            if zero.numel():
                extra.append(st["key"][ptr[zero]])
        if not extra:
            break
        extra = torch.unique(torch.cat(extra))
        repaired += int(extra.numel())
        ks, order = torch.sort(torch.cat([tables[0][0], extra]), stable=True)     # (an existing key sorts first and keeps its count)
        vs = torch.cat([tables[0][1], torch.ones_like(extra, dtype=tables[0][1].dtype)])[order]
        first = torch.ones_like(ks, dtype=torch.bool)
        first[1:] = ks[1:] != ks[:-1]
        tables[0] = (ks[first].contiguous(), vs[first].contiguous())
    else:
        raise RuntimeError("end_to_end set-up: regions without a read remain after three repair passes")
    nkeys = int(np.mean([t[0].numel() for t in tables]))
    for st in sets.values():
        del st["key"]
    sj = torch.exp(torch.randn((S, nid), dtype=torch.float64, device=dev, generator=g) * 0.3)
    si = torch.exp(torch.randn((S, nid), dtype=torch.float64, device=dev, generator=g) * 0.3)
    si[torch.rand((S, nid), device=dev, generator=g) < 0.02] = float("nan")   # other ends Chicago never saw: s_i NA -> 1 (:668-672)
    ntblb, ntlb = 6, 6
    tblb = torch.randint(0, ntblb, (S, nid), dtype=torch.int32, device=dev, generator=g)
    tlb = torch.randint(0, ntlb, (S, nid), dtype=torch.int32, device=dev, generator=g)
    T = torch.exp(torch.randn((S, ntblb, ntlb), dtype=torch.float64, device=dev, generator=g) * 0.5 - 3.0)
    midsum = (torch.arange(nid, device=dev, dtype=torch.int64) * 8000 + 4000)
    distfun = np.zeros((S, 10))
    for j in range(S):
        fit = np.array([14.0 + 0.1 * j, -1.6, 0.05, -0.003])
        ends = np.array([np.log(10000.0), np.log(1.5e6)])
        beta = fit[1] + 2 * fit[2] * ends + 3 * fit[3] * ends ** 2
        alpha = fit[0] + (fit[1] - beta) * ends + fit[2] * ends ** 2 + fit[3] * ends ** 3
        distfun[j] = [*fit, alpha[0], beta[0], alpha[1], beta[1], ends[0], ends[1]]
    from scipy import stats
    cutoff = float(stats.f.ppf(0.99, 2, S - 2)) if S - S // 2 >= 3 else None
    breaks = np.array([0.0, 10.5, 11.5, 12.5, np.inf])
    weights = np.array([1.8, 1.3, 0.7, 0.2])
    want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue", "maxCooks", "cooksArgmax"]

    def run(sync, checks=None):
        """one pass; sync=True: a device synchronisation after every stage (the per-stage split), False: only at the end"""
        times = {}
        torch.cuda.synchronize()
        t_all = time.perf_counter()

        def stage(name, fn):
            t0 = time.perf_counter()
            out = fn()
            if sync:
                torch.cuda.synchronize()
                times[name] = times.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            return out

        res = {}
        theta = None
        for name in ("test", "control"):
            st = sets[name]
            ru = stage("region_universe", lambda: ctx.region_universe(st["pb"], st["po"], 5, chr_of))
            nfrag = ru["baitID"].numel()
            fragN = torch.empty((S, nfrag), dtype=torch.int32, device=dev)

            # all S replicates' joins from one read of the RU rows (round 6; before: S joins, each re-reading them)
            stage("count_join", lambda: ctx.count_join_multi(ru["baitID"], ru["otherEndID"], tables, out=fragN))
            fragFM = stage("fragment_background", lambda: ctx.fragment_background(ru["baitID"], ru["otherEndID"], 0, midsum, sj, si, tblb, tlb, T, distfun, only_fullmean=True)[2])
            N, FM = stage("window_sums", lambda: ctx.window_sums(fragN, fragFM, ru["region_ptr"]))
            if checks is not None:  # (warm-up pass only: what the synthetic inputs look like after the join and the sums)
                cnt = ru["region_ptr"][1:] - ru["region_ptr"][:-1]
                checks[name] = dict(regions_without_fragments=int((cnt == 0).sum()), all_zero_rows=int((N.sum(0) == 0).sum()),
                                    rows_with_na_fullmean=int(torch.isnan(FM).any(0).sum()), mean_count=float(N.double().mean()),
                                    joined_nonzero_fraction=float((fragN > 0).double().mean()))
            del fragN, fragFM
            if name == "test":
                sf = stage("size_factors", lambda: ctx.size_factors(N))
                dv = stage("theta_grid", lambda: ctx.theta_grid(N, FM, sf, grid))
                res["deviances"] = [float(x) for x in dv]
                if not np.isfinite(dv).all():
                    # tt <- Grid[which(deviances == min(deviances))], :1660: with an NA total the reference selects nothing and fails at
                    # the next line; so does this leg (round 5 fell back to theta = 0.5 silently: BENCH_r05's leg ran that way)
                    raise EndToEndError("theta grid: total deviances not all finite (an all-zero row in the joined matrix?): %r" % (res["deviances"],))
                theta = grid[int(np.argmin(dv))]
            out, sc = stage("final_fit" if name == "test" else "control_fit", lambda: ctx.wald_test(N, FM, group, theta=theta, want=want))

            def results():
                if cutoff is not None:
                    ctx.cooks_filter(N, group, out["maxCooks"], out["cooksArgmax"], out["pvalue"], cutoff)
                return ctx.independent_filtering(out["baseMean"], out["pvalue"], 0.1)
            padj, info = stage("results", results)
            if name == "test":
                def ihw():
                    av = ctx.region_avdist(ru["baitID"], ru["otherEndID"], ru["region_ptr"], 0, midsum, chr_of)
                    if checks is not None:
                        checks["avDist"] = dict(na=int(torch.isnan(av).sum()), min_abs=float(av.abs().min()), max_abs=float(av.abs().max()))
                    return ctx.ihw_apply(av, out["pvalue"], breaks, weights)
                w = stage("ihw_covariate_and_application", ihw)
                res.update(nfrag=int(nfrag), rejections_padj_0_05=int((padj < 0.05).sum()), weighted_rejections=int((w["weighted_padj"] < 0.05).sum()),
                           status=int(sc["status"]))
            del ru, N, FM, out, padj
        torch.cuda.synchronize()
        return (time.perf_counter() - t_all) * 1e3, times, theta, res

    checks = {"reads_added_by_the_guard": repaired}
    try:
        run(True, checks)                                       # warm-up: workspaces, the theta grid's child contexts
        totals, splits = [], []
        for _ in range(reps):
            tot, _, theta, res = run(False)
            totals.append(tot)
        for _ in range(reps):
            _, tms, _, _ = run(True)
            splits.append(tms)
    except EndToEndError as e:   # no total_ms: a pass that selected no theta is not a measurement
        return {"error": str(e), "input_checks": checks}
    stages = {k: round(float(np.median([sp[k] for sp in splits])), 3) for k in splits[0]}
    nfrag = res["nfrag"]
    hbm = {  # algorithmic bytes of the HBM-bound stages (both sets), SURVEY.md 8(d): per RU row / fragment / region and replicate
        "count_join": 2 * (8 * nfrag + S * (4 * nfrag + 12 * nkeys)),   # one pass for all replicates: the RU rows read once (S separate joins: 2 S (12 nfrag + 12 nkeys))
        "fragment_background": 2 * (8 * nfrag + 8 * S * nfrag),
        "window_sums": 2 * (12 * S * nfrag + 12 * S * n),
        "region_universe": 2 * (16 * n + 12 * nfrag + 16 * n),
    }
    return {"total_ms": round(float(np.median(totals)), 3), "total_runs_ms": [round(t, 3) for t in totals],
            "sum_of_stages_ms": round(sum(stages.values()), 3), "stages_ms": stages,
            "hbm_stage_fraction_of_peak": {k: round(v / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in hbm.items()},
            "count_join_fraction_by_the_bytes_of_S_separate_joins": round(2 * S * (12 * nfrag + 12 * nkeys) / (stages["count_join"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "input_checks": checks, "theta_chosen": theta, "peaks_per_set": n, "ru_rows_per_set": nfrag, "keys_per_replicate_table": nkeys, **{k: v for k, v in res.items() if k != "nfrag"},
            "what": "resident default-mode pipeline, test + control sets, chicdiff.R:301-347 order: total_ms = one pass without intermediate "
                    "synchronisation (median of %d); stages_ms = a pass with a device synchronisation after every stage (host-side Python / ctypes "
                    "overhead of ~40 calls included in both)" % reps}


def theta_grid_time(ctx, torch, dk, dfm, S):
    """a8: the reference's default mode first scans theta over a 5-point grid, one design-~1 fit per theta
    (chicdiff.R:1619-1662); wall clock of that scan on the benchmark matrix."""
    sf = ctx.size_factors(dk)
    grid = [0.0, 0.25, 0.5, 0.75, 1.0]
    ctx.theta_grid(dk, dfm, sf, grid)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev = ctx.theta_grid(dk, dfm, sf, grid)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del dev  # (all-zero rows of the synthetic matrix make every total deviance NA, as in the reference: sum() without na.rm)
    return {"ms": round(float(np.median(ts)), 3), "thetas": len(grid), "runs_ms": [round(t, 3) for t in ts]}


def theta_grid_replicas_time(hip, synth, torch, dist, local_rank, n_global, S, share_gpu=False):
    """a8 with N > 1 ranks as a REPLICA problem (chicdiff_amd.dist.theta_grid_replicas): every rank holds all `n_global` rows on
    its own GPU in a context WITHOUT a process group, fits its share of the 5 grid points, one all-gather of 5 doubles follows.
    Wall clock between barriers, max over ranks."""
    from chicdiff_amd.dist import theta_grid_replicas, theta_replica_plan
    c2 = hip.HipContext(local_rank)
    try:
        if share_gpu and dist.get_world_size() > 1:  # one-GPU rehearsal: every rank's persistent trend kernel must be resident at once
            c2.set_option("trend_persistent_blocks", max(1, 256 // dist.get_world_size()))
        d = synth.make(n_global, S)
        dk = c2.to_device(d["counts"], np.int32)
        dfm = c2.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
        sf = c2.size_factors(dk)
        grid = [0.0, 0.25, 0.5, 0.75, 1.0]
        for _ in range(2):  # (the first calls create the grid's child contexts and their workspaces)
            theta_grid_replicas(c2, dk, dfm, sf, grid)
        ts = []
        for _ in range(3):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            theta_grid_replicas(c2, dk, dfm, sf, grid)
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if "nccl" in str(dist.get_backend()) else "cpu")
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            ts.append(float(dt.item()) * 1e3)
        world = dist.get_world_size()
        return {"ms": round(float(np.median(ts)), 3), "thetas": len(grid), "runs_ms": [round(t, 3) for t in ts], "rows_per_rank": n_global,
                "points_per_rank": [len(x) for x in theta_replica_plan(len(grid), world)], "refits_last_call": c2.last_refits(),
                "what": "every rank holds all rows and fits theta k for k = rank mod world; one all-gather of 5 doubles (max over ranks)"}
    finally:
        c2.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=2_000_000, help="interactions: global (strong scaling) or per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-kernels", action="store_true", help="skip the window-sum / offsets / count-join side measurement")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the resident default-mode pipeline leg (region universe ... IHW application)")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--one-mode", action="store_true", help="N > 1: skip the second measurement (the other of strong / weak scaling)")
    args = ap.parse_args()

    force_dist = os.environ.get("CHICDIFF_BENCH_FORCE_DIST") == "1"  # rehearse the N > 1 code path with a 1-rank group
    share_gpu = os.environ.get("CHICDIFF_BENCH_SHARE_GPU") == "1"    # rehearsal on a one-GPU box: all ranks on device 0, gloo transport
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        # bare shell: start the ranks ourselves, BEFORE anything here initialises a GPU (never exec from a process
        # that has), and hand their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch

    from chicdiff_amd import hip, synth
    from chicdiff_amd.dist import shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not share_gpu and torch.cuda.device_count() < world:  # device_count() does not initialise the GPU
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    S = args.samples
    ctx = hip.HipContext(local_rank)
    # CHICDIFF_BENCH_FAKE_WORLD=N (with CHICDIFF_BENCH_FORCE_DIST=1, one GPU): rehearse ONE rank's step of an N-GPU strong-scaling run —
    # rows / N on this rank, the whole sharded protocol over a 1-rank RCCL group, and the trend's rows gathered as if N ranks had
    # each sent this rank's block, so that the single-launch trend + MAD kernel runs on all N x (rows / N) rows as it does on every
    # rank of the real thing (library option "bench_fake_world").  `value` is then rows / (this rank's step): what N GPUs would
    # deliver BEFORE any inter-GPU latency — a projection, labelled as such in `config`.
    fake_world = int(os.environ.get("CHICDIFF_BENCH_FAKE_WORLD", "0") or 0)
    if fake_world > 1:
        if not (force_dist and world == 1):
            raise SystemExit("CHICDIFF_BENCH_FAKE_WORLD needs CHICDIFF_BENCH_FORCE_DIST=1 and --gpus 1")
        ctx.set_option("bench_fake_world", fake_world)
    if share_gpu and world > 1:  # rehearsal: all ranks' persistent trend kernels must be resident on the ONE GPU at the same time
        ctx.set_option("trend_persistent_blocks", max(1, 256 // world))
    collectives, comm_ranks = "none (single rank)", 1
    if dist is not None:
        comm_ranks = dist.get_world_size()
        # A rank that cannot bring its communicator up must end the job, not leave its peers waiting inside ncclCommInitRank:
        # init_rccl() decides collectively (every rank raises or none does), and a watchdog ends THIS process with a non-zero
        # code if the whole attempt does not return in time — torch.distributed.run then tears the other ranks down.
        import threading
        limit = float(os.environ.get("CHICDIFF_RCCL_INIT_TIMEOUT", "300"))
        done = threading.Event()

        def watchdog():
            if not done.wait(limit):
                print(f"rank {rank}: communicator setup did not finish within {limit:.0f} s — giving up (exit 3)", file=sys.stderr, flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:  # the library's own RCCL communicator: ncclAllReduce / ncclAllGather issued from C++ on the fit's stream
            if share_gpu:
                raise RuntimeError("CHICDIFF_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device")
            if os.environ.get("CHICDIFF_BENCH_COLLECTIVES") == "torch":
                raise RuntimeError("CHICDIFF_BENCH_COLLECTIVES=torch")
            ctx.init_rccl()
            collectives = "RCCL, called by the library (ncclAllReduce / ncclAllGather on device buffers)"
        except Exception as e:  # same protocol through torch.distributed (one Python callback per collective) — on EVERY rank: the verdict is collective
            print(f"rank {rank}: direct RCCL unavailable ({e}); using the torch.distributed hook", file=sys.stderr, flush=True)
            if os.environ.get("CHICDIFF_BENCH_COLLECTIVES") == "rccl":  # asked for the direct path only: fail loudly, every rank alike
                done.set()
                dist.destroy_process_group()
                raise SystemExit(4)
            ctx.set_process_group(memory="device_via_host" if share_gpu else "device")
            collectives = ("gloo through torch.distributed, device buffers staged through the host (one-GPU rehearsal)" if share_gpu
                           else "RCCL through torch.distributed.all_reduce / all_gather_into_tensor (host callback)")
        done.set()
    # the six result columns of SURVEY.md 8d plus what nbinomWaldTest always computes and results() reads for a 4v4 design:
    # Cook's distances (chicdiff.R:1674, 1739) — inside the timed step since round 4
    want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue", "maxCooks", "cooksArgmax"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, breakdown=True):
        """W warm-up steps, then exactly K timed steps between two barriers; the MAX over ranks."""
        if scaling == "strong":
            lo, hi = shard_bounds(args.rows, fake_world if fake_world > 1 else world, rank)
            n, n_global = hi - lo, args.rows
        else:
            lo, n, n_global = rank * args.rows, args.rows, world * args.rows
        d = synth.make(n, S, start=lo)
        dk = ctx.to_device(d["counts"], np.int32)
        dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)  # region-level FullMean (window sums)
        group = d["group"]
        outputs = {}

        def step():
            return ctx.wald_test(dk, dfm, group, theta=args.theta, want=want, outputs=outputs)

        for _ in range(args.warmup):
            step()
        # HIP events on the library's stream around the gene-wise line search only — the dominant kernel of every configuration
        # measured (checked against the second pass below) and the one the roofline is quoted on: an event is a packet of its own
        # on the stream, each bracketed stage costs the step ~12 us (rocprofv3 kernel trace, profiles/r05_kernel_gaps_250000x8.txt;
        # round 4 bracketed three stages here, all ~20 cost 0.09 ms: tools/timing_overhead.py), so the full breakdown comes from a
        # second pass of the same steps
        ctx.enable_timing(3)
        ktimes = {}
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, sc = step()
            for k, (ms, cnt) in ctx.kernel_times().items():
                a = ktimes.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
        barrier()
        elapsed = time.perf_counter() - t0
        ctx.enable_timing(False)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kfull, coll = {}, {}
        if breakdown:  # the same K steps again, every stage and every collective bracketed (not part of `value`)
            ctx.enable_timing(1)
            for _ in range(args.steps):
                step()
                for k, (ms, cnt) in ctx.kernel_times().items():
                    a = kfull.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
                for k, (cnt, ms, nbytes) in ctx.collective_stats().items():
                    a = coll.setdefault(k, [0, 0.0, 0.0]); a[0] += cnt; a[1] += ms; a[2] += nbytes
            ctx.enable_timing(False)
            barrier()
        return dict(elapsed=elapsed, n=n, n_global=n_global, ktimes=ktimes, kfull=kfull, coll=coll, sc=sc, d=d, dk=dk, dfm=dfm, group=group)

    m = measure(args.scaling)
    elapsed, n, n_global, ktimes, kfull, sc, d, dk, dfm, group = (m[k] for k in ("elapsed", "n", "n_global", "ktimes", "kfull", "sc", "d", "dk", "dfm", "group"))

    ms_per_step = elapsed / args.steps * 1e3
    value = n_global / (elapsed / args.steps)
    projected_value = None
    if fake_world > 1:  # a rehearsal: `value` stays what was MEASURED (this rank's rows / its step); the x N projection has a key of its own
        projected_value = value
        value = n / (elapsed / args.steps)

    # dominant kernel = largest total HIP-event time inside the timed region
    dom = max(((k, v) for k, v in ktimes.items() if k != "allreduce"), key=lambda kv: kv[1][0])
    dom_name, (dom_ms, dom_launches) = dom
    avg_ms = dom_ms / dom_launches
    if kfull:  # the stage bracketed in the timed region must be the largest of the fully bracketed pass
        top = max(((k, v) for k, v in kfull.items() if k not in ("allreduce", "allgather")), key=lambda kv: kv[1][0])[0]
        if top != dom_name:
            print(f"bench.py: WARNING the largest stage of the breakdown pass is {top}, the roofline is quoted on {dom_name}", file=sys.stderr)
    alg_bytes = algorithmic_bytes(S) * n
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic = None
    valu = None
    ent = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pj = json.load(open(pmc_path))
            ent = pj.get(f"{dom_name}:{n}x{S}")
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                if "valu" in ent:  # what actually bounds the kernel: wave-level VALU instructions against the issue slots of the launch
                    slots = avg_ms * 1e-3 * 1024 * 2.4e9 / 4  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz
                    valu = {"insts_per_launch": ent["valu"]["SQ_INSTS_VALU"], "issue_slot_fraction": round(ent["valu"]["SQ_INSTS_VALU"] / slots, 3),
                            "active_lanes_per_inst": ent["valu"]["active_lanes_per_inst"], "source": ent["valu"]["source"]}
                    if "SQ_ACTIVE_INST_VALU" in ent["valu"]:
                        # SQ_ACTIVE_INST_VALU counts, in units of four cycles, the cycles a SIMD spends executing VALU instructions: against
                        # the launch's duration x 1024 SIMDs at the 2.4 GHz peak clock (a lower bound: under this load s_memtime shows ~2.15 GHz)
                        valu["valu_busy_fraction"] = round(ent["valu"]["SQ_ACTIVE_INST_VALU"] * 4 / (avg_ms * 1e-3 * 1024 * 2.4e9), 3)
        except Exception:
            traffic = None
    # per-collective figures of one step on EVERY rank: a slow link or a straggling rank shows as one rank's stream time
    my_coll = {k: {"count": v[0] // args.steps, "ms": round(v[1] / args.steps, 4), "bytes": int(v[2] / args.steps)} for k, v in m["coll"].items()}
    coll_all = [my_coll]
    if dist is not None:
        print(f"rank {rank}: rows {n}, step {ms_per_step:.3f} ms (max over ranks), collectives per step {json.dumps(my_coll)}, refits {ctx.last_refits()}", file=sys.stderr, flush=True)
        coll_all = [None] * world
        dist.all_gather_object(coll_all, my_coll)
    valu_busy = valu.get("valu_busy_fraction") if valu else None
    fp64_frac = None
    if valu and ent.get("valu", {}).get("fp64_flops_per_launch"):
        # the roof that binds: counted fp64 flops of the launch (wave-level FMA x 2 + MUL + ADD + transcendental, x the lanes
        # active per VALU instruction: profiles/pmc_traffic.json, its own --pmc pass) / this run's launch time / the 78.6 TF vector-fp64 peak
        valu["fp64_flops_per_launch"] = ent["valu"]["fp64_flops_per_launch"]
        valu["fp64_insts"] = ent["valu"].get("fp64_insts")
        valu["fp64_source"] = ent["valu"].get("fp64_source")
        fp64_frac = round(ent["valu"]["fp64_flops_per_launch"] / (avg_ms * 1e-3) / (FP64_VALU_PEAK_TFLOPS * 1e12), 4)
    roofline = {"bound": "valu", "valu_busy": valu_busy, "fp64_frac": fp64_frac, "fp64_peak_tflops": FP64_VALU_PEAK_TFLOPS, "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "frac_of_measured_stream_6290": round(achieved / 6290.0, 5), "valu": valu,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                "note": "the fit kernels are bound by fp64 VALU issue (~4e4 flop per interaction), hence bound = 'valu' and valu_busy = fraction of the "
                        "launch the SIMDs spend executing VALU instructions (SQ_ACTIVE_INST_VALU x 4 cycles / (duration x 1024 SIMDs x 2.4 GHz), "
                        "replayed from profiles/pmc_traffic.json: counters cannot be read inside a timed run); achieved / peak / frac stay the HBM "
                        "figures BASELINE.json's north_star asks for (algorithmic bytes / launch duration against 8 TB/s)"}
    result = {
        "metric": "interactions/sec NB-GLM Wald test", "value": round(value, 1), "unit": "interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {n_global} interactions x {S} samples ({S // 2}v{S - S // 2}), "
                               f"size factors + offsets(theta={args.theta}) + dispersions + Wald, design ~condition",
                   "rows_per_gpu": n, "samples": S, "global_rows": n_global,
                   "parallelism": (f"rows-sharded x{world}" if fake_world <= 1 else
                                   f"REHEARSAL on one GPU of one rank's step of a x{fake_world} run: {n} rows here, trend + MAD on all {fake_world} x {n} gathered rows, "
                                   f"1-rank RCCL group; value = THIS rank's rows / its step (measured); projected_value = global rows / this step, a projection before inter-GPU latency"),
                   "collectives": collectives, "ranks_in_communicator": comm_ranks},
        "roofline": roofline,
        "kernels_ms": {k: [round(v[0] / args.steps, 4), v[1] // args.steps] for k, v in sorted(kfull.items(), key=lambda kv: -kv[1][0]) if k != "allreduce"},
        "kernels_ms_note": "second pass of the same K steps with every stage bracketed by HIP events (costs 0.09 ms per step); the timed region brackets only the gene-wise line search (roofline.avg_launch_ms); since round 5 the offsets are formed inside prep (no stage of their own)",
        "collectives_per_step": ({"count": kfull["allreduce"][1] // args.steps, "ms": round(kfull["allreduce"][0] / args.steps, 4),
                                  "note": "sum-all-reduces of one fit on this rank and their summed duration on the stream (already inside the stages' kernels_ms)"}
                                 if "allreduce" in kfull else None),
        "collectives_per_rank": coll_all if dist is not None else None,
        "timed_outputs": want,
        "fit_status": int(sc["status"]),
    }
    if projected_value is not None:
        result["projected_value"] = round(projected_value, 1)
    if world > 1 and not args.one_mode:
        # the other way of scaling, measured in the same run with the same K / W: `value` follows --scaling (default strong =
        # BASELINE.json configs[3], 2 M rows in all); this is the same metric with the per-GPU work fixed instead
        other = "weak" if args.scaling == "strong" else "strong"
        del m, dk, dfm, d
        mo = measure(other, breakdown=False)
        result["other_scaling"] = {"scaling": other, "value": round(mo["n_global"] / (mo["elapsed"] / args.steps), 1), "unit": "interactions/s",
                                   "ms_per_step": round(mo["elapsed"] / args.steps * 1e3, 3), "rows_per_gpu": mo["n"], "global_rows": mo["n_global"],
                                   "steps": args.steps, "warmup": args.warmup}
    if world > 1 and not args.no_hbm_kernels:  # (every rank takes part; rank 0 reports)
        result["theta_grid_replicas"] = theta_grid_replicas_time(hip, synth, torch, dist, local_rank, args.rows, S, share_gpu)
    if rank == 0 and world == 1 and not args.no_hbm_kernels:
        result["hbm_kernels"] = hbm_kernels(ctx, torch, n, S)
        result["theta_grid"] = theta_grid_time(ctx, torch, dk, dfm, S)
    if rank == 0 and world == 1 and not args.no_hbm_kernels and not args.no_end_to_end:
        del dk, dfm
        torch.cuda.empty_cache()
        result["end_to_end"] = end_to_end(ctx, torch, synth, n, S)
        dk = ctx.to_device(d["counts"], np.int32)
        dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
    if rank == 0 and world == 1 and not args.no_hbm_kernels:
        # what an R caller sees: chicdiff_hip_nbglm_fit on HOST buffers (INTEGER(counts), REAL(nf) in, six columns out) — staging
        # through pinned slices + PCIe both ways included.  Reported beside `value`, never as `value`.
        nf_host = np.asfortranarray(ctx.offsets(dfm, sc["sizeFactors"], args.theta).T.cpu().numpy())   # column-major, as R holds them
        k_host = np.asfortranarray(d["counts"].astype(np.int32))
        ctx.nbglm_fit_host(k_host, nf_host, group, want=want)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.nbglm_fit_host(k_host, nf_host, group, want=want)
            ts.append((time.perf_counter() - t0) * 1e3)
        hm = float(np.median(ts))
        result["host_buffer_entry"] = {"ms": round(hm, 3), "interactions_per_s": round(n / hm * 1e3, 1),
                                       "what": "chicdiff_hip_nbglm_fit: dispersions + Wald on caller-owned host matrices, PCIe-inclusive (no size factors / offsets)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = 1  # DESeq2 as Chicdiff calls it is single-threaded: 1 of the box's cores, stated
        v1, dt1 = cpu_baseline(S, args.theta, args.cpu_sample_rows, cores)
        result["cpu_baseline"] = {
            "value": round(v1, 1), "unit": "interactions/s", "cores": cores, "kind": "port",
            "sample": f"first {args.cpu_sample_rows} rows of the same synthetic matrix, oracle/ C restatement (not R/DESeq2: no R on the box), "
                      f"{dt1:.1f} s on 1 of {os.cpu_count() or 1} cores",
        }
        result["gpu_over_cpu_1core"] = round(value / v1, 1)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
