"""GPU parity: the HIP path (through the C ABI, chicdiff_amd.hip) against the CPU oracle on the
same seeded inputs, and against the reference's golden table where it pins something.

Bar (BASELINE.json north_star): log2FoldChange and Wald p within 1e-6 relative; integer work
bit-exact.  The dispersion line searches and the IRLS stop on data-dependent tolerance tests, so
a last-bit difference between libm and the device math can flip one stopping decision in a rare
row.  There is NO blanket allowance for that any more: every fit is compared with the oracle run
under the same global scalars (`explain_fit` -> `assert_rows_explained`), where EVERY row must be
within the bounds or on a list whose every entry a referee explains (binary128 re-run of the line
search, trace of the IRLS); `check_close` requires every row within its tolerance."""
import os

import numpy as np
import pytest

from chicdiff_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as g
    g.build()  # no-op when the in-tree library and the oracle are up to date (they travel with the snapshot)
    from chicdiff_amd import hip
    c = hip.HipContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


PARITY_LOG = []  # every comparison of this module; tests/conftest.py writes it to gpurun_out/parity_gpu.json


def check_close(name, got, ref, mask, tol=1e-6):
    """EVERY row of `mask` within `tol` (relative).  For columns without data-dependent stopping decisions, or for rows
    that `assert_rows_explained` has already found free of them."""
    import inspect
    r = rel(got[mask], ref[mask])
    n = int(mask.sum())
    ok = r <= tol
    test = next((f.function for f in inspect.stack() if f.function.startswith("test_")), "?")
    PARITY_LOG.append(dict(test=test, column=name, rows=n, tol=tol, max_rel=float(r.max()) if n else 0.0, rows_off=int((~ok).sum()),
                           frac_within=float(ok.mean()) if n else 1.0, required_frac=1.0, loose=None, rows_beyond_loose=0,
                           noise_rows_allowed=0))
    print(f"{name}: n={n} max rel {r.max() if n else 0.0:.3e} within {tol:g}: {ok.mean() if n else 1.0:.6f} ({(~ok).sum()} rows off)")
    if os.environ.get("CHICDIFF_PARITY_RECORD_ONLY") == "1":  # survey run: collect the numbers, judge nothing
        return
    assert ok.all(), (name, float(r.max()), np.flatnonzero(mask)[~ok][:10])


def record_unconditioned(tag, got, ref, sc, ref_sc):
    """The figure BASELINE.json's north_star states literally — a FREE GPU fit against a FREE reference (here: oracle) fit, every
    non-all-zero row — counted and logged, NOT asserted: rows outside 1e-6 for the dispersion, for lfc
    (1e-6 max(|lfc|, 1e-2)) and for p (1e-6 max(1, z^2)), the same at 1e-5 / 1e-4 / 1e-3, and the trend coefficients' relative
    distance.  Every conditioned comparison of this module (assert_rows_explained) explains these rows one by one; this is the
    number a user who runs DESeq2 beside the library sees.  -> PARITY_LOG -> gpurun_out/parity_gpu.json -> profiles/."""
    import inspect
    live = ref["allZero"] == 0
    z2 = np.maximum(1.0, ref["stat"] ** 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        err = {
            "dispersion": np.abs(got["dispersion"] - ref["dispersion"]) / np.maximum(np.abs(ref["dispersion"]), 1e-300),
            "log2FoldChange": np.abs(got["log2FoldChange"] - ref["log2FoldChange"]) / np.maximum(np.abs(ref["log2FoldChange"]), 1e-2),
            "pvalue": np.abs(got["pvalue"] - ref["pvalue"]) / np.maximum(np.abs(ref["pvalue"]) * z2, 1e-300),
        }
    rec = dict(test=next((f.function for f in inspect.stack() if f.function.startswith("test_")), "?"), column="UNCONDITIONED free GPU fit vs free oracle fit: " + tag,
               rows=int(live.sum()), tol=1e-6, required_frac=None, asserted=False,
               trend_coef_gpu=[float(x) for x in sc["trendCoef"]], trend_coef_oracle=[float(x) for x in ref_sc["trendCoef"]],
               trend_coef_rel_distance=[float(x) for x in rel(np.asarray(sc["trendCoef"], dtype=np.float64), np.asarray(ref_sc["trendCoef"], dtype=np.float64))],
               dispPriorVar_gpu=float(sc["dispPriorVar"]), dispPriorVar_oracle=float(ref_sc["dispPriorVar"]))
    for k, e in err.items():
        e = np.where(np.isnan(e), np.where(np.isnan(got[k]) & np.isnan(ref[k]), 0.0, np.inf), e)[live]
        rec[k] = dict(rows_outside_1e6=int((e > 1e-6).sum()), rows_outside_1e5=int((e > 1e-5).sum()), rows_outside_1e4=int((e > 1e-4).sum()),
                      rows_outside_1e3=int((e > 1e-3).sum()), max_scaled_error=float(e.max()), median_scaled_error=float(np.median(e)))
        print(f"UNCONDITIONED {tag}: {k}: rows outside 1e-6 / 1e-5 / 1e-4 / 1e-3: {rec[k]['rows_outside_1e6']} / {rec[k]['rows_outside_1e5']} / "
              f"{rec[k]['rows_outside_1e4']} / {rec[k]['rows_outside_1e3']} of {rec['rows']}, median {rec[k]['median_scaled_error']:.2e}, max {rec[k]['max_scaled_error']:.2e}")
    print(f"UNCONDITIONED {tag}: trend coefficients rel. distance {rec['trend_coef_rel_distance']}")
    PARITY_LOG.append(rec)
    return rec


WANT = ["baseMean", "baseVar", "dispGeneEst", "dispFit", "dispMAP", "dispersion", "log2FoldChange", "lfcSE", "stat",
        "pvalue", "intercept", "interceptSE", "deviance", "maxCooks", "dispGeneIter", "dispIter", "dispOutlier",
        "betaConv", "betaIter", "allZero"]


def run_fit(ctx, d, group, **optkw):
    from chicdiff_amd import hip
    dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
    opts = hip.default_opts(**optkw) if optkw else None
    out, sc = ctx.nbglm_fit(dk, dn, group, want=WANT, opts=opts)
    return {k: v.cpu().numpy() for k, v in out.items()}, sc


@pytest.mark.parametrize("n,S", [(20000, 8), (6000, 16), (5000, 6)])
def test_fit_parity_two_groups(ctx, oracle, n, S):
    d = synth.make(n, S)
    got, sc = run_fit(ctx, d, d["group"])
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
    nz = ref["allZero"] == 0
    assert np.array_equal(got["allZero"], ref["allZero"])
    assert sc["status"] & 1 == 0 and sc["trendOuterIter"] == ref["trendOuterIter"]
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-7)
    assert np.isclose(sc["varLogDispEsts"], ref["varLogDispEsts"], rtol=1e-7)
    assert np.isclose(sc["dispPriorVar"], ref["dispPriorVar"], rtol=1e-7)
    check_close("baseMean", got["baseMean"], ref["baseMean"], nz, 1e-13)
    check_close("baseVar", got["baseVar"], ref["baseVar"], nz & (ref["baseVar"] > 0), 1e-11)
    # gene-wise estimates at the floor (alpha < 1e-6, i.e. 1/alpha > 1e6): the profile likelihood is flat to
    # within its own rounding noise there (DESeq2 included) and the grid argmax is decided by that noise;
    # such rows are excluded from the trend (alpha > 1e-6 rule) and restart from the trend in the MAP step,
    # so only "both at the floor" is required of them
    floor = ref["dispGeneEst"] < 1e-6
    assert np.all(got["dispGeneEst"][nz & floor] < 1e-6)
    # every row of the fit against the oracle under the GPU's trend: within the bounds or refereed
    ref_g, listed = explain_fit(f"{n} x {S}", oracle, d["counts"], d["nf"], d["group"], got, sc)
    clean = nz & ~listed
    check_close("dispFit", got["dispFit"], ref_g["dispFit"], nz, 1e-12)
    check_close("lfcSE", got["lfcSE"], ref_g["lfcSE"], clean, 1e-6)
    check_close("deviance", got["deviance"], ref_g["deviance"], clean, 1e-6)
    mc = nz & np.isfinite(ref_g["maxCooks"])
    assert mc.sum() == nz.sum()
    check_close("maxCooks", got["maxCooks"], ref_g["maxCooks"], clean & (ref_g["maxCooks"] > 1e-12), 1e-5)
    # rows sitting at alpha = minDisp (1/alpha = 1e8) search on pure cancellation noise in DESeq2 as well:
    # their iteration count is not reproducible across libm implementations, their estimate (1e-8) is.
    interior = clean & (ref_g["dispGeneEst"] > 1e-6)
    same_it = got["dispGeneIter"][interior] == ref_g["dispGeneIter"][interior]
    # (step counts are not results: a search that stops one step apart at a point where the steps have become smaller than 1e-6
    # gives the same estimate — counted, for the record)
    print(f"unlisted rows: gene-wise iteration counts equal on {same_it.sum()} of {interior.sum()} interior rows, MAP on "
          f"{int((got['dispIter'][clean] == ref_g['dispIter'][clean]).sum())} and IRLS on {int((got['betaIter'][clean] == ref_g['betaIter'][clean]).sum())} of {int(clean.sum())}")
    assert np.array_equal(got["dispOutlier"][clean], ref_g["dispOutlier"][clean])
    assert np.all(np.isnan(got["pvalue"][~nz])) and np.all(np.isnan(got["log2FoldChange"][~nz]))
    assert np.isnan(sc["sumDeviance"]) == bool((~nz).any())


@pytest.mark.parametrize("S", [5, 12])
def test_all_zero_rows_reach_the_row_queue_kernels_as_a_flag_in_the_record(ctx, oracle, S):
    """The line searches and the IRLS learn that a row is all zero from the sign bit prep leaves in the row record's header
    (round 4; before: a load of its own per row).  Rows zeroed at the positions that matter to a wave — first and last of the
    matrix, either side of a 64-row chunk boundary, a run longer than a chunk — at S = 12 (record read by the 16-byte loader,
    three quads) and S = 5 (the generic loader): NaN results and zero step counts exactly there, every other row against the
    oracle as in every parity test."""
    n = 4000
    d = synth.make(n, S)
    counts = d["counts"].copy()
    zero = np.zeros(n, bool)
    zero[[0, 1, 63, 64, 65, 127, n - 1]] = True
    zero[1000:1100] = True
    counts[zero] = 0
    got, sc = run_fit(ctx, dict(d, counts=counts), d["group"])
    az = counts.sum(1) == 0
    assert zero[az].sum() == zero.sum() and np.array_equal(got["allZero"] != 0, az)
    for k in ("dispGeneEst", "dispFit", "dispMAP", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"):
        assert np.all(np.isnan(got[k][az])), k
        assert not np.any(np.isnan(got[k][~az])), k
    for k in ("dispGeneIter", "dispIter", "betaIter"):
        assert np.all(got[k][az] == 0) and np.all(got[k][~az] > 0), k
    explain_fit(f"{n} x {S} with rows zeroed", oracle, counts, d["nf"], d["group"], got, sc)


def test_fit_parity_2v2_with_prior(ctx, oracle):
    d = synth.make(20000, 4)
    got, sc = run_fit(ctx, d, d["group"], dispPriorVar=0.8)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], dispPriorVar=0.8)
    assert sc["dispPriorVar"] == 0.8 and np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6)
    explain_fit("20000 x 4 (2v2), dispPriorVar given", oracle, d["counts"], d["nf"], d["group"], got, sc, dispPriorVar=0.8)
    assert np.all(np.isnan(got["maxCooks"]))
    # without a caller-supplied prior the closed form is used and flagged
    _, sc2 = run_fit(ctx, d, d["group"])
    assert sc2["status"] & 2


@pytest.mark.parametrize("S", [4, 8])
def test_fit_parity_intercept_only(ctx, oracle, S):
    d = synth.make(8000, S)
    g = np.zeros(S, dtype=np.int32)
    # drop all-zero rows so that sum(deviance) is finite, as the reference requires (chicdiff.R:1647)
    keep = d["counts"].sum(1) > 0
    d = {k: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == keep.shape else v) for k, v in d.items()}
    got, sc = run_fit(ctx, d, g, dispPriorVar=0.6)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], g, dispPriorVar=0.6)
    nz = np.ones(len(d["counts"]), bool)
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6)
    ref_g, listed = explain_fit(f"design ~1, S = {S}", oracle, d["counts"], d["nf"], g, got, sc, dispPriorVar=0.6)
    check_close("intercept", got["intercept"], ref_g["beta0"], nz, 1e-12)
    check_close("deviance", got["deviance"], ref_g["deviance"], nz & ~listed, 1e-6)
    assert np.isclose(sc["sumDeviance"], ref["sumDeviance"], rtol=1e-7)
    assert np.all(np.isnan(got["log2FoldChange"]))


def test_size_factors(ctx, oracle):
    for n, S in [(50000, 8), (1001, 4), (7, 3)]:
        d = synth.make(n, S)
        dk = ctx.to_device(d["counts"], np.int32)
        got = ctx.size_factors(dk)
        ref = oracle.size_factors(d["counts"])
        print(n, S, np.max(rel(got, ref)))
        assert np.allclose(got, ref, rtol=1e-13)


def test_size_factors_with_massive_ties(ctx, oracle):
    """More identical ratios than the select shortcut can hold (4096 candidates per median): the
    on-device tail rounds take over (global_kernels.hip: sel_tail_kernel)."""
    rng = np.random.default_rng(5)
    n, S = 60000, 4
    counts = rng.poisson(30, size=(n, S)).astype(np.int32)
    counts[: 2 * n // 3] = [10, 20, 30, 41]  # two thirds of the rows share every ratio, so the medians sit in the tie
    counts = counts[rng.permutation(n)]
    got = ctx.size_factors(ctx.to_device(counts, np.int32))
    ref = oracle.size_factors(counts)
    assert np.allclose(got, ref, rtol=1e-13), (got, ref)
    # and an even count of rows whose two middle order statistics differ
    counts2 = np.concatenate([np.tile(np.array([[8, 16, 24, 33]], np.int32), (9000, 1)), np.tile(np.array([[9, 15, 25, 30]], np.int32), (9000, 1))])
    got2, ref2 = ctx.size_factors(ctx.to_device(counts2, np.int32)), oracle.size_factors(counts2)
    assert np.allclose(got2, ref2, rtol=1e-13), (got2, ref2)


def test_offsets_and_window_sums(ctx, oracle):
    import torch
    d = synth.make(4000, 8, fragments=11)
    dN = ctx.to_device(d["fragN"], np.int32)
    dF = ctx.to_device(d["fragFullMean"], np.float64)
    rp = torch.as_tensor(d["region_ptr"]).to(ctx.device)
    N, FM = ctx.window_sums(dN, dF, rp)
    Nr, FMr = oracle.window_sums(d["fragN"], d["fragFullMean"], d["region_ptr"])
    assert np.array_equal(N.cpu().numpy().T, Nr)  # integer window sums: bit-exact
    assert np.array_equal(np.isnan(FM.cpu().numpy().T), np.isnan(FMr))
    assert np.allclose(FM.cpu().numpy().T, FMr, rtol=1e-15, equal_nan=True)
    sf = oracle.size_factors(Nr)
    for theta in (None, 0.0, 0.25, 1.0):
        got = ctx.offsets(FM, sf, theta).cpu().numpy().T
        ref = oracle.offsets(FMr, sf, theta)
        assert np.allclose(got, ref, rtol=1e-13), theta
    # ragged windows, including empty ones
    rng = np.random.default_rng(2)
    lens = rng.integers(0, 12, 500)
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nfrag = int(rp2[-1])
    fN = rng.integers(0, 50, (nfrag, 4)).astype(np.int32)
    N2, _ = ctx.window_sums(ctx.to_device(fN, np.int32), None, torch.as_tensor(rp2).to(ctx.device))
    N2r, _ = oracle.window_sums(fN, None, rp2)
    assert np.array_equal(N2.cpu().numpy().T, N2r)


def test_count_join(ctx, oracle):
    import torch
    rng = np.random.default_rng(9)
    bait = rng.integers(1000, 1400, 200000).astype(np.int32)
    oe = rng.integers(0, 30000, 200000).astype(np.int32)
    keys = np.unique((bait.astype(np.int64) << 32) | oe)[::2]
    vals = rng.integers(1, 500, len(keys)).astype(np.int32)
    t = lambda a: torch.as_tensor(a).to(ctx.device)
    got = ctx.count_join(t(bait), t(oe), t(keys), t(vals)).cpu().numpy()
    ref = oracle.count_join(bait, oe, keys, vals)
    assert np.array_equal(got, ref)                      # unsorted queries: global-memory search path
    order = np.argsort((bait.astype(np.int64) << 32) | oe, kind="stable")
    got2 = ctx.count_join(t(bait[order]), t(oe[order]), t(keys), t(vals)).cpu().numpy()
    assert np.array_equal(got2, ref[order])              # RU keyed by baitID: narrow key range per tile
    assert (got2 == 0).any() and (got2 > 0).any()
    e = ctx.count_join(t(bait[:5]), t(oe[:5]), t(keys[:0]), t(vals[:0])).cpu().numpy()
    assert np.array_equal(e, np.zeros(5, dtype=np.int32))  # empty count table: every N is 0
    sb, so = bait[order], oe[order]
    for nq in (1, 7, 511, 512, 513, 1025):               # ragged last tile, single lanes
        assert np.array_equal(ctx.count_join(t(sb[:nq]), t(so[:nq]), t(keys), t(vals)).cpu().numpy(), ref[order][:nq]), nq
    # pointers that are not 16-byte aligned (a caller's slice): the scalar load / store variant
    db, do = t(sb), t(so)
    for off in (1, 2, 3):
        assert np.array_equal(ctx.count_join(db[off:], do[off:], t(keys), t(vals)).cpu().numpy(), ref[order][off:]), off
    # a count table far denser than the queries (chinput holds every observed pair of the RU baits): the 16 keys a
    # lane holds in registers do not reach its next query, the search continues in what is left of the range;
    # and one far sparser; and regions as Chicdiff builds them (runs of consecutive IDs, gaps between them)
    allk = np.unique((rng.integers(1000, 1400, 3000000).astype(np.int64) << 32) | rng.integers(0, 30000, 3000000))
    for kk in (allk, allk[::97]):
        vv = rng.integers(1, 500, len(kk)).astype(np.int32)
        assert np.array_equal(ctx.count_join(db, do, t(kk), t(vv)).cpu().numpy(), oracle.count_join(sb, so, kk, vv))
    rb = np.repeat(rng.integers(1000, 1400, 20000), 11).astype(np.int32)
    ro = (np.repeat(rng.integers(20, 29000, 20000), 11) + np.tile(np.arange(11), 20000)).astype(np.int32)
    o2 = np.argsort(rb, kind="stable")                   # setkey(RU, baitID): sorted by bait only
    rb, ro = rb[o2], ro[o2]
    vv = rng.integers(1, 500, len(allk)).astype(np.int32)
    got3 = ctx.count_join(t(rb), t(ro), t(allk), t(vv)).cpu().numpy()
    assert np.array_equal(got3, oracle.count_join(rb, ro, allk, vv)) and (got3 > 0).sum() > 1000
    # extreme keys: INT32_MAX / negative IDs never match a table of valid IDs and must not disturb their neighbours
    xb = np.array([2**31 - 1, 1000, -5, 1200, 1200], dtype=np.int32)
    xo = np.array([-1, 5, 7, 2**31 - 1, 17], dtype=np.int32)
    assert np.array_equal(ctx.count_join(t(xb), t(xo), t(keys), t(vals)).cpu().numpy(), oracle.count_join(xb, xo, keys, vals))


def test_count_join_randomised_shapes(ctx, oracle):
    """The join kernel's paths (coarse level, LDS window, global search, ragged tail, unaligned pointers) under random
    table / query shapes: tiny and empty tables, queries entirely below or above the table, long runs of duplicates,
    clustered keys (one tile spanning thousands of keys), sorted and unsorted callers.  Bit-exact against the oracle."""
    import torch
    rng = np.random.default_rng(123)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(ctx.device)
    for trial in range(60):
        nk = int(rng.choice([0, 1, 2, 63, 64, 65, 511, 513, 4097, 20000]))
        nq = int(rng.choice([1, 5, 64, 500, 512, 520, 3000, 9000]))
        nb = int(rng.choice([1, 3, 50]))                                   # baits in play
        span = int(rng.choice([40, 2000, 200000]))                         # other-end IDs per bait
        kb = rng.integers(100, 100 + nb, nk).astype(np.int64)
        ko = rng.integers(0, span, nk).astype(np.int64)
        keys = np.unique((kb << 32) | ko)
        vals = rng.integers(1, 1000, len(keys)).astype(np.int32)
        mode = trial % 4
        if mode == 0 or len(keys) == 0:                                    # random queries around the table
            qb = rng.integers(99, 101 + nb, nq).astype(np.int32)
            qo = rng.integers(0, span, nq).astype(np.int32)
        elif mode == 1:                                                    # mostly hits, with duplicates
            pick = keys[rng.integers(0, len(keys), nq)]
            qb, qo = (pick >> 32).astype(np.int32), (pick & 0xFFFFFFFF).astype(np.int32)
        elif mode == 2:                                                    # everything below / above the table
            qb = np.full(nq, 99 if trial % 8 < 4 else 100 + nb + 5, dtype=np.int32)
            qo = rng.integers(0, span, nq).astype(np.int32)
        else:                                                              # two far-apart clusters in one tile
            qb = np.where(rng.uniform(size=nq) < 0.5, 100, 100 + nb - 1).astype(np.int32)
            qo = rng.integers(0, span, nq).astype(np.int32)
        if trial % 3:                                                      # RU order (sorted), else an arbitrary caller
            order = np.lexsort((qo, qb))
            qb, qo = qb[order], qo[order]
        ref = oracle.count_join(qb, qo, keys, vals)
        off = int(rng.integers(0, 4)) if nq > 8 else 0                     # now and then pointers that are not 16-byte aligned
        got = ctx.count_join(t(qb)[off:], t(qo)[off:], t(keys), t(vals)).cpu().numpy()
        assert np.array_equal(got, ref[off:]), (trial, nk, nq, nb, span, mode, off)


def test_count_join_window_boundary(ctx, oracle):
    """The join's LDS key window holds 768 keys (global_kernels.hip kJoinCap); a tile whose key range is wider goes through the global
    search.  One tile of 512 queries over a dense run of keys, the run's width swept across the limit (the range a tile takes is
    rounded up to the table's coarse level, every 64th key, so the sweep covers every residue): both paths, bit-exact, on either side
    of the boundary and exactly on it."""
    import torch
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(ctx.device)
    rng = np.random.default_rng(77)
    base = np.int64(700) << 32
    keys = base + np.arange(5000, dtype=np.int64) * 2                     # even other-end IDs 0 .. 9998 of bait 700
    vals = rng.integers(1, 1000, len(keys)).astype(np.int32)
    dk, dv = t(keys), t(vals)
    for start in (0, 37, 1000):
        for width in list(range(640, 900, 7)) + [703, 704, 705, 767, 768, 769, 831, 832, 833]:
            # 512 sorted queries between key `start` and key `start + width - 1`: hits (even IDs) and misses (odd IDs) mixed
            oe = np.sort(rng.integers(2 * start, 2 * (start + width), 512)).astype(np.int32)
            oe[0], oe[-1] = 2 * start, 2 * (start + width - 1)          # the tile spans the whole run
            qb = np.full(512, 700, dtype=np.int32)
            got = ctx.count_join(t(qb), t(oe), dk, dv).cpu().numpy()
            assert np.array_equal(got, oracle.count_join(qb, oe, keys, vals)), (start, width)


def test_count_join_all_replicates_in_one_pass(ctx, oracle):
    """chicdiff_hip_count_join_multi_dev (the replicate loop of chicdiff.R:843-858 as ONE pass over the RU rows) against S single joins and
    the oracle, bit for bit: row counts in every residue mod 4 (a replicate's column of the S x nru result starts 16-byte aligned for one
    nru in four: the shifted store), ragged last tiles, tables of very different sizes incl. an empty one, more replicates than one
    launch takes (16), unaligned query pointers, unsorted callers, and tiles on both sides of the 768-key window."""
    import torch
    rng = np.random.default_rng(4242)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(ctx.device)
    rb = np.repeat(rng.integers(1000, 1400, 6000), 11).astype(np.int32)
    ro = (np.repeat(rng.integers(20, 29000, 6000), 11) + np.tile(np.arange(11), 6000)).astype(np.int32)
    o2 = np.argsort(rb, kind="stable")                                     # setkey(RU, baitID)
    rb, ro = rb[o2], ro[o2]
    allk = np.unique((rng.integers(1000, 1400, 1500000).astype(np.int64) << 32) | rng.integers(0, 30000, 1500000))

    def tables(S):
        out = []
        for s in range(S):
            kk = allk[rng.uniform(size=len(allk)) < (0.9, 0.5, 0.02, 0.0, 0.3)[s % 5]]   # dense (global-search tiles), medium, sparse, EMPTY
            out.append((kk, rng.integers(1, 500, len(kk)).astype(np.int32)))
        return out

    for S, nq, off in ((8, len(rb), 0), (8, len(rb) - 1, 0), (8, len(rb) - 2, 0), (8, len(rb) - 3, 0), (3, 513, 0), (2, 1, 0), (1, 7, 0),
                       (20, 4099, 0), (5, 30001, 1), (5, 30001, 3), (4, 512, 0), (4, 1024, 2)):
        tabs = tables(S)
        qb, qo = rb[off:nq], ro[off:nq]
        db, do = t(rb)[off:nq], t(ro)[off:nq]
        dt = [(t(k), t(v)) for k, v in tabs]
        got = ctx.count_join_multi(db, do, dt).cpu().numpy()
        assert got.shape == (S, len(qb))
        for s, (k, v) in enumerate(tabs):
            assert np.array_equal(got[s], oracle.count_join(qb, qo, k, v)), (S, nq, off, s)
            assert np.array_equal(got[s], ctx.count_join(db, do, dt[s][0], dt[s][1]).cpu().numpy()), (S, nq, off, s)
        assert (got > 0).any() or len(qb) < 8
    # an unsorted caller (every tile spans the whole table: the global search) and extreme IDs
    perm = rng.permutation(20000)
    tabs = tables(3)
    dt = [(t(k), t(v)) for k, v in tabs]
    got = ctx.count_join_multi(t(rb[perm]), t(ro[perm]), dt).cpu().numpy()
    for s, (k, v) in enumerate(tabs):
        assert np.array_equal(got[s], oracle.count_join(rb[perm], ro[perm], k, v)), s
    xb = np.array([2**31 - 1, 1000, -5, 1200, 1200], dtype=np.int32)
    xo = np.array([-1, 5, 7, 2**31 - 1, 17], dtype=np.int32)
    got = ctx.count_join_multi(t(xb), t(xo), dt).cpu().numpy()
    for s, (k, v) in enumerate(tabs):
        assert np.array_equal(got[s], oracle.count_join(xb, xo, k, v)), s


def test_theta_grid(ctx, oracle):
    d = synth.make(6000, 8, fragments=3)
    keep = d["counts"].sum(1) > 0
    counts = d["counts"][keep]
    _, FM = oracle.window_sums(None, d["fragFullMean"], d["region_ptr"])
    FM = FM[keep]
    sf = oracle.size_factors(counts)
    thetas = [0.0, 0.25, 0.5, 0.75, 1.0]
    dk, dF = ctx.to_device(counts, np.int32), ctx.to_device(FM, np.float64)
    got = ctx.theta_grid(dk, dF, sf, thetas)
    g0 = np.zeros(8, dtype=np.int32)
    ref = np.array([oracle.nbglm_fit(counts, oracle.offsets(FM, sf, th), g0)["sumDeviance"] for th in thetas])
    print(got, ref)
    assert np.allclose(got, ref, rtol=1e-7)
    assert np.argmin(got) == np.argmin(ref)
    # the fits of the grid run concurrently (child contexts, one stream each) by default: same numbers one after the
    # other, with 2 and with 7 thetas in flight
    thetas7 = [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0]
    base = None
    for lanes in (1, 2, 5, 7):
        ctx.set_option("theta_grid_concurrency", lanes)
        g = ctx.theta_grid(dk, dF, sf, thetas7)
        base = g if base is None else base
        assert np.allclose(g, base, rtol=1e-10), (lanes, g, base)
    ctx.set_option("theta_grid_concurrency", 5)
    assert np.allclose(base[[0, 2, 3, 4, 6]], got, rtol=1e-10)
    # design ~1 with 4 samples: residual d.f. 3, the simulated prior variance (prior_mc.h) inside every fit of the grid
    d4 = synth.make(5000, 4, fragments=3)
    keep4 = d4["counts"].sum(1) > 0
    _, FM4 = oracle.window_sums(None, d4["fragFullMean"], d4["region_ptr"])
    c4, FM4 = d4["counts"][keep4], FM4[keep4]
    sf4 = oracle.size_factors(c4)
    got4 = ctx.theta_grid(ctx.to_device(c4, np.int32), ctx.to_device(FM4, np.float64), sf4, thetas)
    ref4 = np.array([oracle.nbglm_fit(c4, oracle.offsets(FM4, sf4, th), np.zeros(4, np.int32))["sumDeviance"] for th in thetas])
    assert np.allclose(got4, ref4, rtol=1e-7), (got4, ref4)


def test_pvalues_against_reference_golden_table(ctx, golden):
    import torch
    p = ctx.wald_pvalues(torch.as_tensor(golden["stat"]).to(ctx.device)).cpu().numpy()
    r = rel(p, golden["pvalue"])
    print("max rel err vs reference pvalue column", r.max())
    assert r.max() < 1e-13


def test_host_entry_point_and_errors(ctx, oracle):
    from chicdiff_amd import hip
    d = synth.make(3000, 8)
    res, sc = ctx.nbglm_fit_host(d["counts"], d["nf"], d["group"])
    explain_fit("host entry point, 3000 x 8", oracle, d["counts"], d["nf"], d["group"], res, sc)
    with pytest.raises(hip.ChicdiffHipError):
        ctx.nbglm_fit_host(d["counts"], d["nf"], [0, 0, 1, 1, 2, 2, 0, 1])
    with pytest.raises(hip.ChicdiffHipError):
        ctx.nbglm_fit_host(d["counts"], d["nf"], [1] * 8)
    bad = d["counts"].copy()
    bad[5, 2] = np.iinfo(np.int32).min  # NA_integer_
    with pytest.raises(hip.ChicdiffHipError, match="negative or NA_integer_"):
        ctx.nbglm_fit_host(bad, d["nf"], d["group"])
    # the device-pointer entry points refuse such counts too (prep raises a flag that comes back with the scalars)
    dkb, dnb = ctx.to_device(bad, np.int32), ctx.to_device(d["nf"], np.float64)
    with pytest.raises(hip.ChicdiffHipError, match="negative"):
        ctx.nbglm_fit(dkb, dnb, d["group"])
    with pytest.raises(hip.ChicdiffHipError, match="negative"):
        ctx.wald_test(dkb, dnb, d["group"], theta=0.5)
    with pytest.raises(hip.ChicdiffHipError, match="negative"):
        ctx.theta_grid(dkb, dnb, np.ones(8), [0.0, 0.5])
    # a larger host-buffer call (several staging slices, several copy threads) and its repeat on the warm arena
    d2 = synth.make(700_000, 8)
    want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue", "betaConv", "allZero"]
    r1, s1 = ctx.nbglm_fit_host(d2["counts"], d2["nf"], d2["group"], want=want)
    r2, s2 = ctx.nbglm_fit_host(d2["counts"], d2["nf"], d2["group"], want=want)
    o3, s3 = ctx.nbglm_fit(ctx.to_device(d2["counts"], np.int32), ctx.to_device(d2["nf"], np.float64), d2["group"], want=want)
    for k in want:
        assert np.array_equal(r1[k], r2[k], equal_nan=True) and np.array_equal(r1[k], o3[k].cpu().numpy(), equal_nan=True), k
    bad2 = d2["counts"].copy()
    bad2[654_321, 7] = -1
    with pytest.raises(hip.ChicdiffHipError, match=r"counts\[%d\]" % (7 * 700_000 + 654_321)):
        ctx.nbglm_fit_host(bad2, d2["nf"], d2["group"])


def test_allreduce_hook_on_device_single_rank(ctx, oracle):
    """The sharded protocol end to end on one GPU: a 1-rank RCCL group, every global sum goes
    through chicdiff_amd.dist.AllReduceHook on DEVICE memory (identity all-reduce), results must
    equal the fused single-process path."""
    import torch.distributed as dist
    from chicdiff_amd import hip
    d = synth.make(20000, 8)
    dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
    base, sc0 = ctx.nbglm_fit(dk, dn, d["group"])
    base = {k: v.clone() for k, v in base.items()}
    sf0 = ctx.size_factors(dk)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        c2 = hip.HipContext(0)
        c2.set_process_group()
        out, sc1 = c2.nbglm_fit(dk, dn, d["group"])
        sf1 = c2.size_factors(dk)
        assert c2._hook.error is None and c2._hook.calls >= 7  # consensus, nf column sums, the size-factor select (2 histogram rounds + counts + candidates), trend rows, final sums
        # (size factors, column sums, the trend's rows gathered in two collectives, four select rounds with their
        # candidate gathers, the final sums; with option sharded_trend_gather = 0 the trend alone makes ~20 calls)
        # under the default sharded_trend_gather = 1 the sharded protocol's trend, prior and everything
        # downstream are the single-rank fit's TO THE LAST BIT — exact medians, the nf column sums as correctly rounded
        # double-double sums, the trend fitted by the single-rank kernel on the gathered rows.  So: equality, not a tolerance.
        for k in base:
            a, b = base[k].cpu().numpy(), out[k].cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), (k, "hook path vs single process under sharded_trend_gather = 1 must be bit-identical",
                                                          int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()))
        assert np.array_equal(sc0["trendCoef"], sc1["trendCoef"]) and sc0["dispPriorVar"] == sc1["dispPriorVar"]
        assert np.array_equal(sf0, sf1)
        # the one documented exception: sharded_trend_gather = 0 (one all-reduce per IRLS pass of the trend: the same sums in another
        # order) — coefficients equal to ~1e-15, and the rows whose stopping decisions hang on that digit move
        c2.set_option("sharded_trend_gather", 0)
        out_pp, sc_pp = c2.nbglm_fit(dk, dn, d["group"])
        c2.set_option("sharded_trend_gather", 1)
        assert np.allclose(sc0["trendCoef"], sc_pp["trendCoef"], rtol=1e-12), "per-pass all-reduce trend (sharded_trend_gather = 0): another summation order, coefficients to 1e-12"
        for k in base:
            a, b = base[k].cpu().numpy(), out_pp[k].cpu().numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)), k
            ok = ~np.isnan(a)
            r = rel(b[ok], a[ok])
            assert np.mean(r < 1e-9) > 0.999 and r.max() < 1e-5, (k, r.max(), "tolerance kept ONLY for sharded_trend_gather = 0: the trend's sums in another order")
        c2.close()
    finally:
        dist.destroy_process_group()


def _make_tables(n, S, F, tmp_path, seed=0):
    """Synthetic RU + long FullRegionData in the reference's schema (chicdiff.R:392-425, :912-925) + an rmap file."""
    import pandas as pd
    d = synth.make(n, S, fragments=F)
    keep = d["counts"].sum(1) > 0  # every region has a count somewhere (the reference needs that for the theta grid)
    idx = np.nonzero(keep)[0]
    n2 = len(idx)
    rng = np.random.default_rng(seed)
    bait = 1000 + (np.arange(n2) // 7) * 40                       # a bait every 40 fragments, 7 regions per bait
    oe0 = bait + 2 + (np.arange(n2) % 7) * F
    region = np.repeat(np.arange(1, n2 + 1), F)
    oe = (oe0[:, None] + np.arange(F)[None, :]).ravel()
    RU = pd.DataFrame({"baitID": np.repeat(bait, F), "regionID": region, "otherEndID": oe})
    samples = [f"s{j}" for j in range(S)]
    conds = ["Mono" if g else "CD4" for g in d["group"]]
    fN = d["fragN"].reshape(n, F, S)[idx].reshape(n2 * F, S).copy()
    spike = rng.choice(n2, size=min(40, n2 // 10), replace=False)  # single-sample count outliers -> Cook's flags
    fN[spike * F, rng.integers(0, S, len(spike))] += rng.integers(300, 3000, len(spike)).astype(np.int32)
    fM = d["fragFullMean"].reshape(n, F, S)[idx].reshape(n2 * F, S)
    long = pd.DataFrame({
        "baitID": np.tile(RU["baitID"].to_numpy(), S), "otherEndID": np.tile(oe, S), "regionID": np.tile(region, S),
        "sample": np.repeat(samples, n2 * F), "N": fN.T.ravel(), "FullMean": fM.T.ravel(),
        "condition": np.repeat(conds, n2 * F)})
    # recast layout: rows of one (region, fragment) are sample-major; then shuffle regions to exercise the sort
    long = long.sort_values(["regionID", "otherEndID"], kind="stable").reset_index(drop=True)
    maxid = int(oe.max()) + 50
    rmap = pd.DataFrame({"chr": ['"19"'] * maxid, "start": np.arange(maxid) * 1000 + 1, "end": np.arange(maxid) * 1000 + 1000,
                         "id": np.arange(1, maxid + 1)})
    path = tmp_path / "test.rmap"
    rmap.to_csv(path, sep=" ", header=False, index=False, quoting=3)
    counts = fN.reshape(n2, F, S).sum(axis=1).astype(np.int32)
    _, FM = None, fM.reshape(n2, F, S).sum(axis=1)
    return RU, long, str(path), counts, FM, d["group"], fN, fM


@pytest.mark.parametrize("norm", ["combined", "fullmean", "standard"])
def test_deseq2wrap_mirror(ctx, oracle, tmp_path, norm):
    """The host mirror of DESeq2Wrap() (chicdiff.R:1494) against the same pipeline composed from the oracle."""
    import results_twin as results
    from chicdiff_amd.deseq2wrap import DESeq2Wrap
    RU, long, rmapfile, counts, _, group, fN, fM = _make_tables(2500, 8, 5, tmp_path)
    n = len(counts)
    settings = {"norm": norm, "theta": None, "theta_grid": [0, 0.25, 0.5, 0.75, 1], "rmapfile": rmapfile,
                "saveAuxData": False, "outprefix": str(tmp_path / "x")}
    out = DESeq2Wrap(settings, RU, long, ctx=ctx)
    assert list(out.columns) == ["baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj", "baitID", "maxOE",
                                 "minOE", "regionID", "OEchr", "OEstart", "OEend", "baitchr", "baitstart", "baitend"]
    assert np.array_equal(out["regionID"].to_numpy(), np.arange(1, n + 1))
    # oracle pipeline
    rp = np.arange(0, (n + 1) * 5, 5, dtype=np.int64)
    N, FM = oracle.window_sums(fN, fM, rp)
    assert np.array_equal(N, counts)
    sf = oracle.size_factors(N)
    if norm == "standard":
        nf = np.tile(sf, (n, 1))
    elif norm == "fullmean":
        nf = oracle.offsets(FM, sf, None)
    else:
        g0 = np.zeros(8, dtype=np.int32)
        devs = [oracle.nbglm_fit(N, oracle.offsets(FM, sf, th), g0)["sumDeviance"] for th in settings["theta_grid"]]
        tt = settings["theta_grid"][int(np.argmin(devs))]
        assert out.attrs["theta"] == tt
        nf = oracle.offsets(FM, sf, tt)
    # the table's numbers are those of the library's fit on the same matrices (bit for bit: the same calls the mirror makes),
    # and that fit is the oracle's: every row within the bounds or refereed
    import torch
    dN, dFM = ctx.window_sums(ctx.to_device(fN, np.int32), ctx.to_device(fM, np.float64), torch.as_tensor(rp).to(ctx.device))
    sf_dev = ctx.size_factors(dN)
    assert np.allclose(sf_dev, sf, rtol=1e-13)
    if norm == "standard":
        dnf = torch.as_tensor(sf_dev, device=ctx.device)[:, None].expand(8, n).contiguous()
    else:
        dnf = ctx.offsets(dFM, sf_dev, None if norm == "fullmean" else out.attrs["theta"])
    assert np.allclose(dnf.cpu().numpy().T, nf, rtol=1e-12)
    fit, sc = ctx.nbglm_fit(dN, dnf, group, want=WANT + ["cooksArgmax"])
    got = {k: v.cpu().numpy() for k, v in fit.items()}
    ref, listed = explain_fit(f"DESeq2Wrap mirror, norm = {norm}", oracle, N, dnf.cpu().numpy().T, group, got, sc)
    for col in ("baseMean", "log2FoldChange", "lfcSE", "stat"):
        assert np.array_equal(out[col].to_numpy(), got[col], equal_nan=True), col
    pv, nout = results.cooks_filter(got["pvalue"], got["maxCooks"], got["cooksArgmax"], lambda idx: N[idx], group)
    pv_o, nout_o = results.cooks_filter(ref["pvalue"], ref["maxCooks"], ref["cooksArgmax"], lambda idx: N[idx], group)
    got_p = out["pvalue"].to_numpy()
    assert np.array_equal(np.isnan(got_p), np.isnan(pv)) and nout > 0  # Cook's outliers flagged (4v4) ...
    assert np.array_equal(np.isnan(pv)[~listed], np.isnan(pv_o)[~listed])  # ... as the oracle flags them
    assert np.array_equal(got_p, pv, equal_nan=True)
    # rows whose IRLS diverged on the injected outliers go through the optim fallback on both sides (the
    # posterior mode DESeq2's L-BFGS-B call targets); they must be re-fitted, not left flagged
    assert (ref["betaIter"] >= 100).sum() > 0 and np.all(ref["betaConv"][ref["allZero"] == 0] == 1)
    # results(): independent filtering + BH of the table's own p-values by the host restatement the golden table pins
    padj_ref, _ = results.independent_filtering(out["baseMean"].to_numpy(), got_p)
    gpadj = out["padj"].to_numpy()
    assert np.array_equal(np.isnan(gpadj), np.isnan(padj_ref))
    okp = ~np.isnan(padj_ref)
    assert np.allclose(gpadj[okp], padj_ref[okp], rtol=1e-12)
    # annotation: window bounds and rmap coordinates (chicdiff.R:1703-1714)
    r0 = out.iloc[0]
    assert r0["minOE"] == RU[RU.regionID == 1].otherEndID.min() and r0["OEstart"] == (r0["minOE"] - 1) * 1000 + 1
    assert r0["OEend"] == r0["maxOE"] * 1000 and r0["baitstart"] == (r0["baitID"] - 1) * 1000 + 1


def test_deseq2wrap_argument_handling(ctx, tmp_path):
    from chicdiff_amd.deseq2wrap import DESeq2Wrap
    RU, long, rmapfile, *_ = _make_tables(1500, 8, 2, tmp_path)
    base = {"theta": None, "theta_grid": [0, 0.5, 1], "rmapfile": rmapfile, "saveAuxData": False, "outprefix": ""}
    with pytest.raises(ValueError, match="Unknown normalisation method"):
        DESeq2Wrap(dict(base, norm="median"), RU, long, ctx=ctx)
    with pytest.warns(UserWarning, match='equivalent to norm = "standard"'):
        out = DESeq2Wrap(dict(base, norm="combined"), RU, long, theta=1, ctx=ctx)
    assert "theta" not in out.attrs
    out = DESeq2Wrap(dict(base, norm="combined", theta=0.25), RU, long, ctx=ctx)  # theta from the settings list
    assert out.attrs["theta"] == 0.25


def test_fused_wald_test_equals_composed_calls(ctx, oracle):
    d = synth.make(30000, 8, fragments=2)
    _, FM = oracle.window_sums(None, d["fragFullMean"], d["region_ptr"])
    dk, dF = ctx.to_device(d["counts"], np.int32), ctx.to_device(FM, np.float64)
    sf = ctx.size_factors(dk)
    ref, sc_ref = ctx.nbglm_fit(dk, ctx.offsets(dF, sf, 0.25), d["group"])
    ref = {k: v.clone() for k, v in ref.items()}
    out, sc = ctx.wald_test(dk, dF, d["group"], theta=0.25)
    assert np.array_equal(sc["sizeFactors"], sf)
    for k in ref:
        assert np.array_equal(ref[k].cpu().numpy(), out[k].cpu().numpy(), equal_nan=True), k
    assert np.array_equal(sc["trendCoef"], sc_ref["trendCoef"])


def test_device_math(ctx):
    """The special functions the fit kernels are built on, against scipy (fp64, ~1 ulp expected)."""
    import torch
    from scipy import special
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(-40, 40, 200000)), 1 + rng.uniform(-0.05, 0.05, 50000),
                        1 + rng.uniform(0, 1e-7, 1000), rng.uniform(0.5, 2.0, 50000)])
    dx = torch.as_tensor(x).to(ctx.device)
    for op, name in ((0, "flog"), (1, "tlog")):
        got = ctx.selftest_math(op, dx).cpu().numpy()
        ref = np.log(x)
        r = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
        print(name, "max rel", r.max())
        assert r.max() < 2e-15, name
    got = ctx.selftest_math(2, dx).cpu().numpy()
    assert np.max(np.abs(got * x - 1)) < 4e-16
    xg = np.exp(rng.uniform(-11, 21, 200000))
    dg = torch.as_tensor(xg).to(ctx.device)
    lg = ctx.selftest_math(3, dg).cpu().numpy()
    ref = special.gammaln(xg)
    assert np.max(np.abs(lg - ref) / np.maximum(1, np.abs(ref))) < 1e-14
    di = ctx.selftest_math(4, dg).cpu().numpy()
    ref = special.digamma(xg)
    assert np.max(np.abs(di - ref) / np.maximum(1, np.abs(ref))) < 4e-15
    # texp (table-driven exp of the line searches and the IRLS): against 50-digit arithmetic on a sample, numpy on the rest
    import mpmath as mp
    xe = np.concatenate([rng.uniform(-30, 10, 200000), rng.uniform(-700, 700, 50000), rng.uniform(-1e-3, 1e-3, 20000), [0.0, -745.0, 709.0]])
    ge = ctx.selftest_math(8, torch.as_tensor(xe).to(ctx.device)).cpu().numpy()
    re_ = np.exp(xe)
    ok = re_ > 1e-300   # (denormal results: ldexp rounds once more)
    r = np.abs(ge[ok] - re_[ok]) / re_[ok]
    print("texp max rel vs numpy", r.max())
    assert r.max() < 4e-16 and ge[np.flatnonzero(xe == 0.0)[0]] == 1.0
    mp.mp.dps = 50
    worst = max(abs(mp.mpf(float(g)) / mp.exp(mp.mpf(float(v))) - 1) for g, v in zip(ge[:2000], xe[:2000]))
    print("texp max rel vs 50 digits (2000 points)", float(worst))
    assert worst < 2.5e-16


def test_fragment_background(ctx, oracle):
    """a3 on device against the oracle: real chr19 HindIII fragment geometry, synthetic Chicago tables."""
    import torch
    from test_oracle import _a3_inputs
    a = _a3_inputs(seed=3, S=4)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(ctx.device)
    B, Tm, F = ctx.fragment_background(t(a["bait"]), t(a["oe"]), a["id_min"], t(a["midsum"]), t(a["sj"]), t(a["si"]),
                                       t(a["tblb"]), t(a["tlb"]), t(a["T"]), a["distfun"])
    Br, Tr, Fr = oracle.fragment_background(**a)
    assert np.array_equal(Tm.cpu().numpy(), Tr, equal_nan=True)          # pure lookup: bit-exact
    assert np.allclose(B.cpu().numpy(), Br, rtol=1e-13, equal_nan=True) and np.isnan(Br).any()
    assert np.allclose(F.cpu().numpy(), Fr, rtol=1e-13, equal_nan=True)


FULL_WANT = ["log2FoldChange", "intercept", "pvalue", "stat", "dispersion", "dispGeneEst", "dispMAP", "dispFit", "dispOutlier", "dispIter", "betaConv", "betaIter"]


def neither_side_right(gpu, ora, arb, floor=1e-6):
    """Rows of a refereed list on which NEITHER double-precision search ends where the binary128 one does.  A side counts as
    right when it is within 1e-6 of the referee, or when it and the referee both sit at or below the trend's floor
    (100 * minDisp = 1e-6): DESeq2 keeps such a row out of the trend fit and out of the MAD (estimateDispersionsFit, reached from
    chicdiff.R:1573), starts its MAP search from the fitted value and never calls it an outlier — where below the floor the
    gene-wise estimate lies changes nothing downstream (round 4's one case: row 1244683 of 2 M x 8, GPU 1.12e-8, referee 1e-8)."""
    gpu, ora, arb = (np.asarray(x, dtype=float) for x in (gpu, ora, arb))
    right = lambda x: (rel(x, arb) <= 1e-6) | ((x <= floor) & (arb <= floor))
    return ~right(gpu) & ~right(ora)


def assert_rows_explained(tag, oracle, d, group, got, ref, dispPriorVar, maxit=100, gene_listed=None):
    """Full-size comparison of two fits that share their global scalars (trend, prior variance): EVERY non-all-zero row
    must agree — dispersion to 1e-6, log2FoldChange to 1e-6 * max(|lfc|, 1e-2), p to 1e-6 * max(1, z^2) (a relative change
    eps of the Wald statistic z changes a tail p-value by ~ z^2 eps) — or be on the list this function builds, and every
    listed row has a referee that shows the difference to be a stopping decision inside rounding noise:
      gene : gene-wise line search ended elsewhere       -> binary128 re-run (oracle_arbitrate_disp, stage 0) sides with one side
      map  : MAP line search ended elsewhere              -> binary128 re-run (stage 1) sides with one side
      irls : the IRLS stopped one step apart              -> conv_test at the disputed step is within 1e-3 of betaTol, and each
                                                             side's estimate is the trace's iterate at its own step count
    Design ~1 (group all zero) has no IRLS: the intercept (log2 scale) takes the fold change's place.
    No blanket allowance: returns the list (written to gpurun_out/ for profiles/)."""
    n = len(ref["allZero"])
    live = ref["allZero"] == 0
    listed = []
    two_groups = bool(np.any(np.asarray(group) != 0))
    # gene-wise stage (the caller may have arbitrated these rows already)
    gene_off = np.flatnonzero(live & ((ref["dispGeneEst"] > 1e-6) | (got["dispGeneEst"] > 1e-6)) & (rel(got["dispGeneEst"], ref["dispGeneEst"]) > 1e-6))
    if gene_listed is None:
        arb = oracle.arbitrate_disp(d["counts"], d["nf"], group, gene_off, ref) if len(gene_off) else np.empty(0)
        eg, eo = rel(got["dispGeneEst"][gene_off], arb), rel(ref["dispGeneEst"][gene_off], arb)
        nb = neither_side_right(got["dispGeneEst"][gene_off], ref["dispGeneEst"][gene_off], arb)
        assert not nb.any(), (tag, "gene-wise rows neither side gets right", gene_off[nb])
        listed += [dict(row=int(i), kind="gene", gpu=float(got["dispGeneEst"][i]), oracle=float(ref["dispGeneEst"][i]), referee=float(a))
                   for i, a in zip(gene_off, arb)]
    else:
        assert set(gene_off.tolist()) <= set(gene_listed), (tag, "gene-wise rows off that the caller did not arbitrate")
    in_gene = np.zeros(n, dtype=bool)
    in_gene[gene_off] = True
    if gene_listed is not None:
        in_gene[np.asarray(sorted(gene_listed), dtype=np.int64)] = True
    assert len(gene_off) <= 3e-5 * live.sum() + 2, (tag, len(gene_off))   # (a sanity bound on the list's length: each entry is refereed)
    # MAP stage, rows whose gene-wise estimates agree
    assert np.allclose(got["dispFit"][live], ref["dispFit"][live], rtol=1e-12), tag   # same trend on both sides
    map_off = np.flatnonzero(live & ~in_gene & (rel(got["dispMAP"], ref["dispMAP"]) > 1e-6))
    if len(map_off):
        arb = oracle.arbitrate_disp(d["counts"], d["nf"], group, map_off,
                                    dict(dispGeneEst=ref["dispGeneEst"], dispFit=ref["dispFit"], dispPriorVar=dispPriorVar), stage="map")
        eg, eo = rel(got["dispMAP"][map_off], arb), rel(ref["dispMAP"][map_off], arb)
        assert np.all((eg <= 1e-6) | (eo <= 1e-6)), (tag, "MAP rows neither side gets right", map_off[(eg > 1e-6) & (eo > 1e-6)])
        listed += [dict(row=int(i), kind="map", gpu=float(got["dispMAP"][i]), oracle=float(ref["dispMAP"][i]), referee=float(a),
                        iters=[int(got["dispIter"][i]), int(ref["dispIter"][i])]) for i, a in zip(map_off, arb)]
    assert len(map_off) <= 1e-5 * live.sum() + 2, (tag, len(map_off))
    disp_listed = in_gene.copy()
    disp_listed[map_off] = True
    rd = rel(got["dispersion"], ref["dispersion"])
    assert not np.any(live & ~disp_listed & (rd > 1e-6)), (tag, "dispersion off on unlisted rows", np.flatnonzero(live & ~disp_listed & (rd > 1e-6))[:10])
    assert np.array_equal(got["dispOutlier"][live & ~disp_listed], ref["dispOutlier"][live & ~disp_listed]), tag
    # Wald stage
    optim = live & ((got["betaIter"] >= maxit) | (ref["betaIter"] >= maxit))
    flip = np.flatnonzero(live & ~disp_listed & ~optim & (got["betaIter"] != ref["betaIter"]))
    for i in flip:
        kg, ko = int(got["betaIter"][i]), int(ref["betaIter"][i])
        lfc, cv = oracle.irls_trace(d["counts"], d["nf"], group, i, ref["dispersion"][i], steps=max(kg, ko) + 1)
        t = min(kg, ko)
        assert abs(kg - ko) == 1 and abs(cv[t - 1] / 1e-8 - 1.0) < 1e-3, (tag, "IRLS steps differ but conv_test is not at the tolerance", int(i), kg, ko, cv[:t + 1])
        assert abs(got["log2FoldChange"][i] - lfc[kg - 1]) <= 1e-9 * max(abs(lfc[kg - 1]), 1e-2) and abs(ref["log2FoldChange"][i] - lfc[ko - 1]) <= 1e-9 * max(abs(lfc[ko - 1]), 1e-2), (tag, int(i))
        listed.append(dict(row=int(i), kind="irls", steps=[kg, ko], conv_test_at_disputed_step=float(cv[t - 1]), gpu=float(got["log2FoldChange"][i]),
                           oracle=float(ref["log2FoldChange"][i])))
    assert len(flip) <= 5e-6 * live.sum() + 2, (tag, len(flip))
    wald_listed = disp_listed.copy()
    wald_listed[flip] = True
    chk = live & ~wald_listed
    assert np.array_equal(got["betaConv"][chk], ref["betaConv"][chk]), tag
    coef_got, coef_ref = (got["log2FoldChange"], ref["log2FoldChange"]) if two_groups else (got["intercept"], ref["beta0"])
    dl = np.abs(coef_got - coef_ref)
    bad = chk & ~(dl <= 1e-6 * np.maximum(np.abs(coef_ref), 1e-2))
    z2 = np.maximum(1.0, ref["stat"] ** 2)
    badp = chk & ~(rel(got["pvalue"], ref["pvalue"]) <= 1e-6 * z2)
    # Rows that went through the optim fallback (IRLS gave up after maxit steps: DESeq2 calls optim(L-BFGS-B) there, whose own
    # stopping rule is a relative reduction of the objective by 1e7 * eps = 2e-9) are held to the same bounds; one that misses them is
    # refereed: both estimates must sit at the same mode to FAR better than that optimiser could tell them apart — the log2-scale
    # negative log posterior (evaluated here in numpy) agrees to 1e-10 relative, the coefficients to 1e-4 standard errors
    for i in np.flatnonzero((bad | badp) & optim):
        y, f, al = d["counts"][i].astype(np.float64), d["nf"][i], float(ref["dispersion"][i])

        def nlp(b0, b1):
            from scipy.special import gammaln
            mu, size = f * np.exp2(b0 + b1 * np.asarray(group)), 1.0 / al
            ll = gammaln(y + size) - gammaln(size) - gammaln(y + 1) + size * np.log(size / (size + mu)) + y * (np.log(mu) - np.log(size + mu))
            return -ll.sum() + 0.5 * 1e-6 * (b0 * b0 + b1 * b1)

        fg, fo = nlp(got["intercept"][i], got["log2FoldChange"][i]), nlp(ref["beta0"][i], ref["log2FoldChange"][i])
        se = abs(ref["log2FoldChange"][i] / ref["stat"][i]) if ref["stat"][i] != 0 else np.inf
        print(f"{tag}: optim-fallback row {i}: lfc gpu {got['log2FoldChange'][i]!r} oracle {ref['log2FoldChange'][i]!r} (|diff| {dl[i]:.2e} = {dl[i] / se:.1e} SE), "
              f"objective gpu {fg!r} oracle {fo!r}")
        assert abs(fg - fo) <= 1e-10 * (abs(fo) + 1.0) and dl[i] <= 1e-4 * se, (tag, "optim-fallback row at different modes", int(i))
        listed.append(dict(row=int(i), kind="optim", gpu=float(got["log2FoldChange"][i]), oracle=float(ref["log2FoldChange"][i]), objective=[float(fg), float(fo)]))
        wald_listed[i] = True
    bad &= ~wald_listed
    badp &= ~wald_listed
    for i in np.flatnonzero(bad)[:10]:
        print(f"{tag}: row {i}: coefficient gpu {coef_got[i]!r} oracle {coef_ref[i]!r} (|diff| {dl[i]:.3e}), dispersion gpu {got['dispersion'][i]!r} oracle {ref['dispersion'][i]!r} "
              f"(rel {rd[i]:.3e}), stat {ref['stat'][i]:.4f}, IRLS steps gpu {got['betaIter'][i]} oracle {ref['betaIter'][i]}, counts {d['counts'][i].tolist()}, nf {d['nf'][i].tolist()}")
    assert not bad.any(), (tag, "log2FoldChange off on unlisted rows", np.flatnonzero(bad)[:10], dl[bad][:10])
    assert not badp.any(), (tag, "pvalue off on unlisted rows", np.flatnonzero(badp)[:10])
    chk = live & ~wald_listed
    n_optim = int((chk & optim).sum())   # rows through the optim fallback are held to the same bounds (not masked)
    kinds = {k: sum(1 for x in listed if x["kind"] == k) for k in ("gene", "map", "irls", "optim")}
    print(f"{tag}: {int(live.sum())} rows, every one within bounds except {len(listed)} refereed rows {kinds}; {n_optim} optim-fallback rows inside the bounds; "
          f"max rel dispersion {rd[chk].max():.2e}, lfc {(dl[chk] / np.maximum(np.abs(coef_ref[chk]), 1e-2)).max():.2e}, "
          f"p/(z^2) {(rel(got['pvalue'], ref['pvalue'])[chk] / z2[chk]).max():.2e}")
    PARITY_LOG.append(dict(test="assert_rows_explained", column=tag, rows=int(live.sum()), tol=1e-6, max_rel=float(rd[chk].max()), rows_off=len(listed),
                           frac_within=1.0 - len(listed) / max(int(live.sum()), 1), required_frac=1.0, loose=None, rows_beyond_loose=0,
                           noise_rows_allowed=0, refereed=kinds))
    return listed, wald_listed


def explain_fit(tag, oracle, counts, nf, group, got, sc, rows=None, **oracle_kw):
    """A GPU fit (free: its own trend) against the oracle run under the GPU's trend — the two then share their global scalars
    up to the prior variance the oracle derives from (almost) the same residuals — through `assert_rows_explained`: every
    row within the bounds or refereed.  A local trend (status bit 16) is handed over as its fitted values (oracle option
    dispFitIn), a parametric one as its two coefficients.  `rows` (bool mask) restricts the row-level comparison.
    Returns (oracle fit under the GPU's trend, mask of rows on the refereed list)."""
    kw = dict(oracle_kw)
    if sc["status"] & 16:
        kw.update(fitType=2, dispFitIn=np.nan_to_num(got["dispFit"]))
    else:
        kw["trendCoef"] = sc["trendCoef"]
    ref_g = oracle.nbglm_fit(counts, nf, group, **kw)
    assert np.array_equal(got["allZero"], ref_g["allZero"]), tag
    assert np.isclose(sc["varLogDispEsts"], ref_g["varLogDispEsts"], rtol=1e-5), (tag, sc["varLogDispEsts"], ref_g["varLogDispEsts"])
    assert np.isclose(sc["dispPriorVar"], ref_g["dispPriorVar"], rtol=1e-5), (tag, sc["dispPriorVar"], ref_g["dispPriorVar"])
    if rows is None:
        listed, wl = assert_rows_explained(tag, oracle, dict(counts=counts, nf=nf), group, got, ref_g, ref_g["dispPriorVar"], maxit=oracle_kw.get("betaMaxit", 100))
        return ref_g, wl
    cut = lambda f: {k: (v[rows] if isinstance(v, np.ndarray) and v.shape[:1] == rows.shape else v) for k, v in f.items()}
    listed, wl_sub = assert_rows_explained(tag, oracle, dict(counts=counts[rows], nf=nf[rows]), group, cut(got), cut(ref_g), ref_g["dispPriorVar"],
                                           maxit=oracle_kw.get("betaMaxit", 100))
    wl = np.zeros(len(rows), dtype=bool)
    wl[np.flatnonzero(rows)[wl_sub]] = True
    return ref_g, wl


def test_full_size_2Mx8_against_oracle_and_permutation(ctx, oracle):
    """BASELINE.json configs[2] (2 M x 8, 4v4) at full size, every row against the oracle (run on the host
    cores), plus a size-independent property (row permutation).

    The line search stops on `change < 1e-6` in a log-likelihood of size 1e2..1e5 and accepts steps on an Armijo
    inequality: in ANY double-precision implementation a few of these decisions per million rows fall inside rounding
    noise (DESeq2 evaluates lgamma(y + 1/alpha) - lgamma(1/alpha), the oracle does the same with R's own lgammafn
    restated, the GPU uses a cancellation-free form), the row then ends somewhere else, and through the global trend
    fit such a row moves EVERY row's MAP dispersion in the 6th digit.  So the chain is checked link by link:
    (1) gene-wise estimates: all rows agree to 1e-6 except a listed handful (<= 3 per 100 000); each listed row is
        re-run in binary128 by the arbiter (oracle_arbitrate_disp), which must side with one of the two — the table
        (who was right where) is written to gpurun_out/arbiter_2Mx8.json;
    (2) trend | gene-wise estimates: the oracle's own trend routine, fed the GPU's estimates, returns the GPU's
        coefficients to 1e-10; fed the arbitrated estimates it lands within 2e-5 of both sides;
    (3) everything downstream | trend: the oracle, given the GPU's two trend coefficients, reproduces the GPU's FREE
        fit to 1e-6 on >= 99.99 % of the rows (dispersion, lfc, p); and the GPU, given the oracle's trend and prior
        variance (DESeq2 exposes both), reproduces the oracle's free fit to 1e-6;
    (4) permuting the rows permutes the results."""
    import json
    import torch
    from chicdiff_amd import hip
    n, S = 2_000_000, 8
    d = synth.make(n, S)
    dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
    want = FULL_WANT
    out, sc = ctx.nbglm_fit(dk, dn, d["group"], want=want)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    threads = min(16, os.cpu_count() or 1)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=threads)
    print("trend", sc["trendCoef"], ref["trendCoef"], sc["trendOuterIter"], ref["trendOuterIter"])
    record_unconditioned("2 M x 8 (4v4)", got, ref, sc, ref)
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=2e-5) and sc["trendOuterIter"] == ref["trendOuterIter"]
    # (1) rows that enter the trend on either side (alpha > 1e-6) and differ: list and arbitrate
    live = (ref["allZero"] == 0) & ((ref["dispGeneEst"] > 1e-6) | (got["dispGeneEst"] > 1e-6))
    rg = rel(got["dispGeneEst"], ref["dispGeneEst"])
    bad = np.nonzero(live & (rg > 1e-6))[0]
    print("gene-wise estimates off by > 1e-6:", len(bad), "of", int(live.sum()))
    assert len(bad) <= 3e-5 * live.sum()
    arb = oracle.arbitrate_disp(d["counts"], d["nf"], d["group"], bad, ref)
    e_gpu, e_ora = rel(got["dispGeneEst"][bad], arb), rel(ref["dispGeneEst"][bad], arb)
    table = [dict(row=int(i), gpu=float(got["dispGeneEst"][i]), oracle=float(ref["dispGeneEst"][i]), arbiter=float(a),
                  gpu_vs_arbiter=float(x), oracle_vs_arbiter=float(y)) for i, a, x, y in zip(bad, arb, e_gpu, e_ora)]
    for t in table:
        print("  row %(row)d: gpu %(gpu).9e oracle %(oracle).9e arbiter %(arbiter).9e  (gpu-arb %(gpu_vs_arbiter).1e, oracle-arb %(oracle_vs_arbiter).1e)" % t)
    gpu_right, ora_right = int((e_gpu <= 1e-6).sum()), int((e_ora <= 1e-6).sum())
    neither = int(((e_gpu > 1e-6) & (e_ora > 1e-6)).sum())
    print(f"binary128 arbiter: GPU right on {gpu_right}, oracle right on {ora_right}, neither on {neither} of {len(bad)} rows")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(n=n, S=S, rows_compared=int(live.sum()), disagreeing_rows=table, gpu_matches_arbiter=gpu_right,
                   oracle_matches_arbiter=ora_right, neither=neither), open("gpurun_out/arbiter_2Mx8.json", "w"), indent=1)
    # a row both double-precision paths miss would be a systematic error, not rounding noise (neither_side_right: within 1e-6 of the
    # referee, or below the trend's 1e-6 floor together with it)
    nb = neither_side_right(got["dispGeneEst"][bad], ref["dispGeneEst"][bad], arb)
    assert not nb.any(), ("gene-wise rows neither side gets right", bad[nb])
    # (2) trend | gene-wise estimates
    useg = (ref["allZero"] == 0) & (got["dispGeneEst"] > 1e-6)
    cg, itg, rc = oracle.parametric_dispersion_fit(ref["baseMean"][useg], got["dispGeneEst"][useg])
    assert rc == 0 and np.allclose(cg, sc["trendCoef"], rtol=1e-10)
    dg = ref["dispGeneEst"].copy()
    dg[bad] = arb
    use = (ref["allZero"] == 0) & (dg > 1e-6)
    c_arb, it_arb, rc = oracle.parametric_dispersion_fit(ref["baseMean"][use], dg[use])
    print("trend: arbitrated", c_arb, "GPU", sc["trendCoef"], "oracle", ref["trendCoef"])
    assert rc == 0 and np.allclose(c_arb, sc["trendCoef"], rtol=2e-5) and np.allclose(c_arb, ref["trendCoef"], rtol=2e-5)
    # (3a) the oracle under the GPU's trend against the GPU's free fit: every row, no blanket allowance
    ref_g = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=threads, trendCoef=sc["trendCoef"])
    assert np.isclose(sc["varLogDispEsts"], ref_g["varLogDispEsts"], rtol=1e-5)
    listed_a, wl_a = assert_rows_explained("2M x 8, GPU free fit vs oracle under the GPU's trend", oracle, d, d["group"], got, ref_g, ref_g["dispPriorVar"],
                                           gene_listed=[int(i) for i in bad])
    # (3b) the GPU under the oracle's global scalars against the oracle's free fit
    opts = hip.default_opts(trendCoef=ref["trendCoef"], dispPriorVar=ref["dispPriorVar"])
    out2, sc2 = ctx.nbglm_fit(dk, dn, d["group"], want=want, opts=opts)
    got2 = {k: v.cpu().numpy() for k, v in out2.items()}
    assert np.array_equal(sc2["trendCoef"], ref["trendCoef"]) and sc2["dispPriorVar"] == ref["dispPriorVar"]
    listed_b, wl_b = assert_rows_explained("2M x 8, GPU under the oracle's scalars vs oracle free fit", oracle, d, d["group"], got2, ref, ref["dispPriorVar"],
                                           gene_listed=[int(i) for i in bad])
    # The two FREE fits against each other = (3a) + what the 6th-digit trend shift does to the oracle itself (ref_g vs ref: the
    # same code, the same rounding, trend coefficients 2e-5 apart).  Nothing to allow here: every row is either refereed above
    # or within 1e-6 of the oracle under the same trend; what remains is the algorithm's own sensitivity, reported as such —
    # a row whose MAP search stops one step earlier under a prior mean shifted by 2e-6 lands up to a few per cent away.
    live = ref["allZero"] == 0
    rs = rel(ref_g["dispersion"], ref["dispersion"])
    shift_rows = np.flatnonzero(live & (rs > 1e-3))
    rf = rel(got["dispersion"], ref["dispersion"])
    far = np.flatnonzero(live & (rf > 1e-3))
    assert set(far.tolist()) <= set(shift_rows.tolist()) | set(np.flatnonzero(wl_a).tolist()), "free vs free: a row beyond 1e-3 that neither the trend shift nor a referee explains"
    assert np.array_equal(got["dispOutlier"][live & ~wl_a], ref_g["dispOutlier"][live & ~wl_a])
    print(f"free vs free: dispersion beyond 2e-5 on {int((live & (rf > 2e-5)).sum())} rows, beyond 1e-3 on {len(far)}; the oracle against ITSELF under the GPU's "
          f"trend coefficients (rel. shift {np.max(rel(sc['trendCoef'], ref['trendCoef'])):.1e}): beyond 2e-5 on {int((live & (rs > 2e-5)).sum())} rows, beyond 1e-3 on {len(shift_rows)}")
    assert len(shift_rows) <= 2e-5 * live.sum()   # (a property of DESeq2's stopping rule under a perturbed prior, not of the GPU path)
    json.dump(dict(n=n, S=S, same_trend_gpu_vs_oracle=listed_a, pinned_scalars_gpu_vs_oracle=listed_b,
                   trend_shift_rows=[dict(row=int(i), oracle_own_trend=float(ref["dispersion"][i]), oracle_gpu_trend=float(ref_g["dispersion"][i]),
                                          gpu=float(got["dispersion"][i]), map_iters=[int(ref["dispIter"][i]), int(ref_g["dispIter"][i])]) for i in shift_rows]),
              open("gpurun_out/refereed_rows_2Mx8.json", "w"), indent=1)
    # (4) permuting the rows permutes the results (order-free sums, exact medians)
    perm = torch.randperm(n, device=ctx.device, generator=torch.Generator(device=ctx.device).manual_seed(0))
    p1 = out["pvalue"][perm].cpu().numpy()
    out3, sc3 = ctx.nbglm_fit(dk[:, perm].contiguous(), dn[:, perm].contiguous(), d["group"], want=["pvalue"])
    assert np.allclose(sc3["trendCoef"], sc["trendCoef"], rtol=1e-12)
    p2 = out3["pvalue"].cpu().numpy()
    ok = ~np.isnan(p1)
    r = rel(p2[ok], p1[ok])
    print("permutation, free fits: max rel", r.max(), "frac within 1e-9", np.mean(r < 1e-9), "(the trend's sums in another order: its 13th digit, and the stopping decisions that hang on it)")
    assert np.array_equal(np.isnan(p1), np.isnan(p2)) and np.mean(r < 1e-9) > 0.999, \
        "free fits of a permuted matrix: tolerance kept — the trend kernel adds its rows' terms in row order, a permutation changes the order (coefficients to 1e-12)"
    # with the trend and the prior variance pinned to the first fit's: bit for bit
    out4, sc4 = ctx.nbglm_fit(dk[:, perm].contiguous(), dn[:, perm].contiguous(), d["group"], want=["pvalue"],
                              opts=hip.default_opts(trendCoef=sc["trendCoef"], dispPriorVar=sc["dispPriorVar"]))
    assert np.array_equal(out4["pvalue"].cpu().numpy(), p1, equal_nan=True) and sc4["varLogDispEsts"] == sc["varLogDispEsts"]


def test_full_size_C2_200k_x4_2v2_against_oracle(ctx, oracle):
    """BASELINE.json configs[1] at full size: 200 000 interactions x 4 samples, 2 v 2 — the reference's own design
    (Chicdiff.Rmd:42).  Residual d.f. = 2, so dispPriorVar comes from DESeq2's set.seed(2) simulation (prior_mc.h):
    the device histogram + KL + loess must land on the oracle's value exactly, and dispersions / lfc / p follow."""
    n, S = 200_000, 4
    d = synth.make(n, S)
    got, sc = run_fit(ctx, d, d["group"])
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=min(16, os.cpu_count() or 1))
    assert sc["status"] & 2 and ref["status"] & 2
    print("dispPriorVar", sc["dispPriorVar"], ref["dispPriorVar"], "trend", sc["trendCoef"], ref["trendCoef"])
    record_unconditioned("C2 200 k x 4 (2v2)", got, ref, sc, ref)
    assert sc["dispPriorVar"] == ref["dispPriorVar"]
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6) and sc["trendOuterIter"] == ref["trendOuterIter"]
    # every row, optim-fallback rows included: within bounds or refereed.  The two fits are free, but their global scalars
    # coincide (prior variance exactly — it is the argmin over a grid — and the trend to 1e-6), so the same-trend rule applies
    # after handing the oracle the GPU's coefficients
    ref_g = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=min(16, os.cpu_count() or 1), trendCoef=sc["trendCoef"])
    assert ref_g["dispPriorVar"] == sc["dispPriorVar"]
    assert_rows_explained("C2 (200 000 x 4, 2v2), GPU free fit vs oracle under the GPU's trend", oracle, d, d["group"], got, ref_g, sc["dispPriorVar"])
    assert np.all(np.isnan(got["maxCooks"]))  # no group with >= 3 replicates
    # a heterogeneous 2v2 matrix of the same size: the prior variance lands above DESeq2's 0.25 floor
    from test_oracle import heterogeneous_counts
    counts, nf = heterogeneous_counts(n, S, 1.3)
    got, sc = run_fit(ctx, dict(counts=counts, nf=nf), d["group"])
    ref = oracle.nbglm_fit(counts, nf, d["group"], nthreads=min(16, os.cpu_count() or 1))
    print("heterogeneous: dispPriorVar", sc["dispPriorVar"], ref["dispPriorVar"])
    record_unconditioned("C2 heterogeneous 200 k x 4 (2v2)", got, ref, sc, ref)
    assert sc["dispPriorVar"] == ref["dispPriorVar"] and sc["dispPriorVar"] > 0.3
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6)
    ref_g, _ = explain_fit("C2 heterogeneous (200 000 x 4, 2v2)", oracle, counts, nf, d["group"], got, sc, nthreads=min(16, os.cpu_count() or 1))
    assert ref_g["dispPriorVar"] == sc["dispPriorVar"]


def _two_rank_worker(rank, world, port, n, S, q):
    import os
    import torch.distributed as dist
    from chicdiff_amd import hip
    from chicdiff_amd.dist import shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth.make(n, S, fragments=2)
        fm = d["fragFullMean"].reshape(n, 2, S).sum(axis=1)
        lo, hi = shard_bounds(n, world, rank)
        c = hip.HipContext(0)
        c.set_process_group(memory="device_via_host")
        # both ranks' single-launch trend kernels must be resident on the ONE GPU at once (their grid barriers spin): half the
        # workgroups each; the single-rank reference of the test runs on the same number, so the trend's sums have the same order
        c.set_option("trend_persistent_blocks", 256 // world)
        dk, dF = c.to_device(d["counts"][lo:hi], np.int32), c.to_device(fm[lo:hi], np.float64)
        out, sc = c.wald_test(dk, dF, d["group"], theta=0.5)
        assert c._hook.error is None and c._hook.calls >= 7
        assert c.last_refits() == 0, "no refit expected (a trend-barrier timeout would refit with one launch per pass: another summation order)"
        # a shard one rank cannot fit (here: empty on rank 1, n < 1) must fail on EVERY rank, not hang the others in
        # their first collective
        msgs = []
        for call in (lambda a, b: c.wald_test(a, b, d["group"], theta=0.5), lambda a, b: c.nbglm_fit(a, b, d["group"])):
            try:
                if rank == 1:
                    e0 = c.torch.empty((S, 0), dtype=c.torch.int32, device=c.device)
                    call(e0, c.torch.empty((S, 0), dtype=c.torch.float64, device=c.device))
                else:
                    call(dk, dF)
                msgs.append("no error")
            except hip.ChicdiffHipError as e:
                msgs.append(str(e))
        assert all(("NULL" in m or "<= n" in m) if rank == 1 else ("rejected their arguments" in m) for m in msgs), msgs
        # a negative / NA count on ONE rank has gone into everybody's size factors, trend and prior by the time it is seen:
        # every rank must refuse the fit (round 2: the peers reported success on corrupted statistics)
        bad = dk.clone()
        if rank == 1:
            bad[2, 5] = -2147483648   # NA_integer_
        nf1 = c.offsets(dF, np.ones(S), 0.5)
        for call in (lambda: c.nbglm_fit(bad, nf1, d["group"]), lambda: c.nbglm_fit(bad, nf1, d["group"], opts=hip.default_opts(fitType=7))):
            try:
                call()
                raise AssertionError(f"rank {rank}: a fit with a negative count on rank 1 reported success")
            except hip.ChicdiffHipError as e:
                assert ("negative value or NA_integer_" in str(e)) or ("fitType" in str(e)) or ("rejected their arguments" in str(e)), str(e)
        out2, _ = c.wald_test(dk, dF, d["group"], theta=0.5)  # and the context still works afterwards
        assert all(c.torch.equal(out[k], out2[k]) or c.torch.allclose(out[k], out2[k], equal_nan=True, rtol=0, atol=0) for k in out)
        # the trend with one all-reduce per IRLS pass (the path before the rows were gathered; still the fallback when the
        # persistent kernel cannot run): same coefficients up to summation order
        c.set_option("sharded_trend_gather", 0)
        calls = c._hook.calls
        out4, sc4 = c.wald_test(dk, dF, d["group"], theta=0.5)
        per_pass_calls = c._hook.calls - calls
        c.set_option("sharded_trend_gather", 1)
        calls = c._hook.calls
        c.wald_test(dk, dF, d["group"], theta=0.5)
        gathered_calls = c._hook.calls - calls
        assert np.allclose(sc4["trendCoef"], sc["trendCoef"], rtol=1e-10) and sc4["trendOuterIter"] == sc["trendOuterIter"]
        assert gathered_calls + 10 < per_pass_calls, (gathered_calls, per_pass_calls)  # two collectives instead of ~20
        out3, sc3 = c.wald_test(dk, dF, d["group"], theta=0.5, opts=hip.default_opts(fitType=2))  # the local trend: its order
        q.put((rank, lo, hi, {k: v.cpu().numpy() for k, v in out.items()}, sc["trendCoef"], sc["sizeFactors"], sc["dispPriorVar"],  # statistics and sums
               out3["dispersion"].cpu().numpy(), sc3["status"]))                                                                   # are all-reduced too
        c.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_one_gpu_match_single_rank(ctx):
    """world_size 2 on the real kernels: two processes share the one GPU of this box (RCCL forbids that,
    so the hook stages the device buffers through gloo), each fits its half of the rows, every global
    statistic goes through the all-reduce hook; the concatenated result must match the one-rank fit."""
    import socket
    import torch.multiprocessing as mp
    n, S = 30000, 8
    d = synth.make(n, S, fragments=2)
    fm = d["fragFullMean"].reshape(n, 2, S).sum(axis=1)
    ctx.set_option("trend_persistent_blocks", 128)   # what each of the two ranks below uses (they share this GPU): same order of the trend's sums
    try:
        ref, sc0 = ctx.wald_test(ctx.to_device(d["counts"], np.int32), ctx.to_device(fm, np.float64), d["group"], theta=0.5)
    finally:
        ctx.set_option("trend_persistent_blocks", 0)
    ref = {k: v.cpu().numpy() for k, v in ref.items()}
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_worker, args=(r, 2, port, n, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the ranks gather the trend's rows and fit them with the single-rank kernel: the same coefficients to the last bit
    assert np.array_equal(res[0][4], sc0["trendCoef"]) and np.array_equal(res[0][4], res[1][4])
    assert np.array_equal(res[0][5], sc0["sizeFactors"]) and np.array_equal(res[0][5], res[1][5])   # exact medians
    assert res[0][6] == res[1][6] == sc0["dispPriorVar"]
    from chicdiff_amd import hip
    loc, scl = ctx.wald_test(ctx.to_device(d["counts"], np.int32), ctx.to_device(fm, np.float64), d["group"], theta=0.5, opts=hip.default_opts(fitType=2))
    loc = loc["dispersion"].cpu().numpy()
    got_loc = np.concatenate([res[0][7], res[1][7]])
    okl = ~np.isnan(loc)
    assert (scl["status"] & 16) and res[0][8] == res[1][8] == scl["status"] and np.array_equal(np.isnan(got_loc), ~okl)
    rl = rel(got_loc[okl], loc[okl])
    print("local trend, 2-rank vs 1-rank dispersion: max rel", rl.max())
    # the LOCAL trend (fitType = 2, locfit restated) is the documented exception: its weighted sums are all-reduced per rank, i.e. added in
    # another order than on one rank — fitted values equal to ~1e-13, and the rows whose stopping decisions hang on that digit move
    assert np.mean(rl < 1e-9) > 0.999 and rl.max() < 1e-4, "local trend: tolerance kept (per-rank partial sums, another summation order)"
    # the parametric trend (default): the ranks gather the trend's rows and every rank fits them with the single-rank kernel — the sharded fit IS
    # the single-rank fit, to the last bit, in every output column
    for k in ref:
        got = np.concatenate([res[0][3][k], res[1][3][k]])
        nd = int((~((got == ref[k]) | (np.isnan(got) & np.isnan(ref[k])))).sum())
        print(k, "2-rank vs 1-rank: rows that differ:", nd)
        assert np.array_equal(got, ref[k], equal_nan=True), (k, nd, "two ranks vs one rank must be bit-identical (gathered trend)")


@pytest.mark.parametrize("n,S,group", [(300000, 8, None), (40000, 5, [0, 0, 1, 1, 1]), (30000, 16, None), (20000, 3, [0, 0, 0])])
def test_line_search_layouts_agree_bit_for_bit(ctx, n, S, group):
    """At the end of a launch the line-search kernels and the IRLS evaluate stragglers with the samples spread across lanes
    (disp_kernels.hip: eval_point_spread; wald_kernels.hip: the spread_now branch).  Which ticks run in which layout depends
    on the schedule, so the two layouts must agree to the last bit — and repeated runs must be identical."""
    import os
    d = synth.make(n, S)
    group = np.asarray(d["group"] if group is None else group, dtype=np.int32)
    dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
    want = ["dispGeneEst", "dispGeneIter", "dispMAP", "dispIter", "dispersion", "log2FoldChange", "lfcSE", "pvalue", "betaIter", "deviance"]

    def run():
        out, _ = ctx.nbglm_fit(dk, dn, group, want=want)
        return {k: v.cpu().numpy().copy() for k, v in out.items()}

    a = run()
    b = run()
    ctx.set_option("line_search_spread", 0)
    try:
        c = run()
    finally:
        ctx.set_option("line_search_spread", 1)
    ctx.set_option("line_search_spread", 2)  # samples across lanes, but every tick of the launch's end through the general tick (no lean tick)
    try:
        g = run()
    finally:
        ctx.set_option("line_search_spread", 1)
    ctx.set_option("line_search_schedule", 0)  # natural row order through the queue instead of likely-long rows first
    try:
        e = run()
    finally:
        ctx.set_option("line_search_schedule", 1)
    # the static deal hands a wave 64 schedule entries at a time, 64 / deal groups of `deal` entries each (1 .. 8; lanes beyond
    # the last whole group stay empty when 64 is not a multiple): every group size gives the same bits
    deals = {}
    for deal in (1, 3, 8):
        ctx.set_option("line_search_deal", deal)
        try:
            deals[deal] = run()
        finally:
            ctx.set_option("line_search_deal", 0)
    # score classes dealt out statically (0-1 by default, all but the last when the launch has 1.8 .. 8 rows per lane): any number
    for ca in (1, 5, 6):
        ctx.set_option("line_search_classes_a", ca)
        try:
            deals[f"classes dealt {ca}"] = run()
        finally:
            ctx.set_option("line_search_classes_a", 0)
    # rows per dequeue (64; 16 when a wave's share of the rows is about one chunk): any size gives the same bits
    for ch in (8, 24):
        ctx.set_option("line_search_chunk", ch)
        try:
            deals[f"chunk {ch}"] = run()
        finally:
            ctx.set_option("line_search_chunk", 0)
    # round 6: waves per SIMD (the three-waves build — 168 registers, no scratch — runs the MAP search of >= 750 k rows; by option: both
    # searches at any size), the exchange layouts of <= 128 entries only, and issue priority by search age: the same bits
    for mw in (2, 3):
        ctx.set_option("line_search_min_waves", mw)
        try:
            deals[f"{mw} waves per SIMD"] = run()
        finally:
            ctx.set_option("line_search_min_waves", 0)
    for name, val in (("line_search_spread", 3), ("line_search_prio", 20)):
        ctx.set_option(name, val)
        try:
            deals[f"{name} = {val}"] = run()
        finally:
            ctx.set_option(name, 1 if name == "line_search_spread" else 0)
    assert (a["dispGeneIter"] >= 100).sum() > 10  # the stragglers this is about are present (their fitDispGrid stages run in disp_grid_kernel)
    for deal, f in deals.items():
        for k in a:
            assert np.array_equal(a[k], f[k], equal_nan=True), f"{k}: deal {deal} differs"
    if n >= 300000:
        assert (a["betaIter"] >= 20).sum() > 0  # ... and the IRLS's
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), f"{k}: two runs differ"
        assert np.array_equal(a[k], c[k], equal_nan=True), f"{k}: layouts differ in {np.sum(~((a[k] == c[k]) | (np.isnan(a[k]) & np.isnan(c[k]))))} rows"
        assert np.array_equal(a[k], g[k], equal_nan=True), f"{k}: lean and general ticks differ in {np.sum(~((a[k] == g[k]) | (np.isnan(a[k]) & np.isnan(g[k]))))} rows"
        assert np.array_equal(a[k], e[k], equal_nan=True), f"{k}: schedules differ in {np.sum(~((a[k] == e[k]) | (np.isnan(a[k]) & np.isnan(e[k]))))} rows"


def test_kernel_timing_modes(ctx):
    """chicdiff_hip_enable_timing: 1 brackets every stage of a call with HIP events, 2 only the three fit kernels, 3 the gene-wise
    line search alone (what bench.py's timed region uses), 0 nothing; the results do not depend on it.  Since round 5 the fused call
    forms the offsets inside `prep` for fits of up to 262 144 rows (option fuse_offsets = 0: always a launch of their own, as the
    composed calls make it; 2: always inside prep) — same bits."""
    d = synth.make(20000, 8)
    dk = ctx.to_device(d["counts"], np.int32)
    dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / 8), np.float64)
    want = ["dispersion", "pvalue"]
    try:
        ctx.enable_timing(1)
        a, _ = ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want)
        full = ctx.kernel_times()
        ctx.enable_timing(2)
        b, _ = ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want)
        fit = ctx.kernel_times()
        ctx.enable_timing(3)
        b3, _ = ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want)
        fit3 = ctx.kernel_times()
        ctx.enable_timing(1)
        ctx.set_option("fuse_offsets", 0)
        u, _ = ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want)
        unfused = ctx.kernel_times()
    finally:
        ctx.enable_timing(0)
        ctx.set_option("fuse_offsets", 1)
    c, _ = ctx.wald_test(dk, dfm, d["group"], theta=0.5, want=want)
    # (the median / MAD of the residuals is taken inside the trend kernel since round 4: no "mad_select" stage of its own; the offsets
    # inside prep since round 5: no "offsets" stage unless asked for)
    for name in ("size_factors", "prep", "disp_gene", "trend_fit", "disp_map", "wald_prep", "wald_irls", "wald_final"):
        assert name in full and full[name][0] > 0 and full[name][1] == 1, name
    assert full.get("offsets", (0.0, 0))[1] == 0 and unfused["offsets"][1] == 1 and unfused["offsets"][0] > 0
    timed = {k for k, (ms, launches) in fit.items() if launches > 0}
    assert timed == {"disp_gene", "disp_map", "wald_irls"}, timed
    assert all(fit[k][0] > 0 for k in timed)
    assert {k for k, (ms, launches) in fit3.items() if launches > 0} == {"disp_gene"} and fit3["disp_gene"][0] > 0
    # NA rows take the size factors (chicdiff.R:1588-1589): a few of them, through both forms
    fm_na = dfm.clone()
    fm_na[3, 17] = float("nan")
    fm_na[0, 19999] = float("nan")
    na1, _ = ctx.wald_test(dk, fm_na, d["group"], theta=0.25, want=want)
    ctx.set_option("fuse_offsets", 0)
    try:
        na0, _ = ctx.wald_test(dk, fm_na, d["group"], theta=0.25, want=want)
    finally:
        ctx.set_option("fuse_offsets", 1)
    for k in want:
        assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), equal_nan=True)
        assert np.array_equal(a[k].cpu().numpy(), b3[k].cpu().numpy(), equal_nan=True)
        assert np.array_equal(a[k].cpu().numpy(), c[k].cpu().numpy(), equal_nan=True)
        assert np.array_equal(a[k].cpu().numpy(), u[k].cpu().numpy(), equal_nan=True), f"{k}: offsets inside prep and as a launch of their own differ"
        assert np.array_equal(na1[k].cpu().numpy(), na0[k].cpu().numpy(), equal_nan=True), f"{k}: ... with NA rows"


def test_bh_on_device(ctx, oracle):
    """f1/f3: p.adjust(p, "BH") — device sort + suffix minimum against the oracle; NA, ties, p = 0/1, n = 1."""
    import torch
    rng = np.random.default_rng(11)
    for n in (1, 2, 777, 300000):
        p = rng.uniform(size=n) ** 3
        if n > 10:
            p[rng.integers(0, n, n // 10)] = np.nan
            p[rng.integers(0, n, n // 10)] = p[3]  # ties
            p[5], p[6] = 0.0, 1.0
        got = ctx.bh_adjust(torch.from_numpy(p).to(ctx.device)).cpu().numpy()
        ref = oracle.bh_adjust(p)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.array_equal(got[ok], ref[ok]), n  # same operations in the same order: identical bits
    allna = ctx.bh_adjust(torch.full((50,), float("nan"), dtype=torch.float64, device=ctx.device)).cpu().numpy()
    assert np.all(np.isnan(allna))


def test_bh_and_filtering_at_benchmark_sizes(ctx, oracle):
    """p.adjust / results() at 0.4 M - 3.2 M rows (the sizes bench.py's hbm_kernels quotes): BH equal to the oracle to the bit
    and independent filtering equal to the numpy restatement — uniform, spiked, sorted, reversed, constant, half-NA inputs,
    massive ties, and inputs with structure at evenly spaced positions (written for the sample sort measured and dropped in
    round 3, DESIGN.md §5: a sort that samples evenly spaced rows must not depend on what sits there)."""
    import torch
    import results_twin as results
    rng = np.random.default_rng(5)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx.device)

    def bh_check(p, what):
        got = ctx.bh_adjust(dev(p)).cpu().numpy()
        ref = oracle.bh_adjust(p)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), what
        ok = ~np.isnan(ref)
        assert np.array_equal(got[ok], ref[ok]), what

    for n in (399999, 400000, 1000003, 2000000, 3226387, 3226388):
        p = rng.uniform(size=n)
        p[rng.uniform(size=n) < 0.1] **= 8
        p[rng.integers(0, n, n // 20)] = np.nan
        bh_check(p, f"spiked uniform, n={n}")
    n = 1 << 20
    bh_check(np.sort(rng.uniform(size=n)), "sorted")
    bh_check(np.sort(rng.uniform(size=n))[::-1].copy(), "reversed")
    bh_check(np.full(n, 0.25), "constant")
    p = rng.uniform(size=n)
    p[: n // 2] = np.nan
    bh_check(p, "half NA, in one block")
    p = rng.uniform(size=n)
    p[rng.integers(0, n, n // 2)] = 0.5
    p[rng.integers(0, n, n // 4)] = 1.0
    p[rng.integers(0, n, n // 8)] = 0.0
    bh_check(p, "massive ties")
    # every 16 384th-quantile position small, everything else large
    p = rng.uniform(0.5, 1.0, size=n)
    pos = (np.arange(16384, dtype=np.int64) * n) // 16384
    p[pos] = rng.uniform(0.0, 0.1, size=16384)
    bh_check(p, "structure at evenly spaced rows")

    def if_check(bm, p, what):
        hp, hinfo = results.independent_filtering(bm, p)
        dp, dinfo = ctx.independent_filtering(dev(bm), dev(p))
        assert np.array_equal(dinfo["numRej"], hinfo["numRej"]) and dinfo["index"] == hinfo["index"], what
        assert np.isclose(dinfo["filterThreshold"], hinfo["filterThreshold"], rtol=1e-15), what
        assert np.allclose(dp.cpu().numpy(), hp, rtol=1e-13, equal_nan=True), what

    for n in (600000, 2000000):
        bm = rng.lognormal(np.log(19), 1.4, n)
        bm[rng.uniform(size=n) < 0.02] = 0.0
        p = rng.uniform(size=n)
        p[rng.uniform(size=n) < 0.15] **= 6
        p[bm == 0] = np.nan
        p[rng.integers(0, n, 500)] = p[11]
        if_check(bm, p, f"n={n}")
    n = 1 << 20
    bm = rng.lognormal(np.log(19), 1.4, n)
    p = rng.uniform(size=n) ** 2
    bm2 = bm.copy()
    bm2[pos] = 1e-3 * rng.uniform(size=16384)
    if_check(bm2, p, "baseMean: structure at evenly spaced rows")
    p2 = rng.uniform(0.5, 1.0, size=n)
    p2[pos] = rng.uniform(0.0, 0.01, size=16384)
    if_check(bm, p2, "p-values: structure at evenly spaced rows")


def test_ihw_application_on_device_reproduces_golden_table(ctx, golden, oracle):
    """f3: chicdiff.R:2038-2049 on device against the reference's result table and against the oracle."""
    import torch
    from post_inputs import ihw_tables_from_golden
    breaks, w = ihw_tables_from_golden(golden)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx.device)
    out = {k: v.cpu().numpy() for k, v in ctx.ihw_apply(dev(golden["avDist"]), dev(golden["pvalue"]), breaks, w).items()}
    assert np.array_equal(out["group"], golden["group"])
    for k in ("weight", "weighted_pvalue", "weighted_padj"):
        assert np.allclose(out[k], golden[k], rtol=1e-13), k
    # NA handling: a distance outside the breaks, an NA p-value
    av = golden["avDist"][:1000].copy()
    pv = golden["pvalue"][:1000].copy()
    pv[7] = np.nan
    ref = oracle.ihw_apply(av, pv, breaks, w)
    got = ctx.ihw_apply(dev(av), dev(pv), breaks, w)
    assert np.array_equal(got["group"].cpu().numpy(), ref[0])
    assert np.allclose(got["weighted_padj"].cpu().numpy(), ref[3], rtol=1e-13, equal_nan=True)
    av[3] = 0.5  # log|avDist| < 0: group NA -> mean(avWeights) NA -> everything NA
    got = ctx.ihw_apply(dev(av), dev(pv), breaks, w)
    assert got["group"][3].item() == np.iinfo(np.int32).min and torch.isnan(got["weighted_padj"]).all()
    with pytest.raises(Exception):
        ctx.ihw_apply(dev(av), dev(pv), breaks[::-1].copy(), w)


@pytest.mark.parametrize("s", [5, 0, 12, 15, 16, 20])
def test_region_universe_on_device(ctx, oracle, s):
    """f4: getRegionUniverse window mode, bit-exact against the oracle on the reference's chr19 fragment IDs (RUexpand 16 and 20: windows of
    more than 32 candidates, which the fill kernel walks instead of reading off a per-region bit mask)."""
    import torch
    from post_inputs import region_universe_case
    bait, oe, chr_of = region_universe_case()
    dev = lambda a: torch.from_numpy(a).to(ctx.device)
    got = ctx.region_universe(dev(bait), dev(oe), s, dev(chr_of))
    ptr, rb, rr, ro = oracle.region_universe(bait, oe, s, chr_of)
    assert np.array_equal(got["region_ptr"].cpu().numpy(), ptr)
    for k, ref in (("baitID", rb), ("regionID", rr), ("otherEndID", ro)):
        assert np.array_equal(got[k].cpu().numpy(), ref), k
    ln = np.diff(ptr)
    mn, mx = got["minOE"].cpu().numpy(), got["maxOE"].cpu().numpy()
    nz = ln > 0
    assert np.array_equal(mn[nz], np.minimum.reduceat(ro, ptr[:-1][nz])) and np.array_equal(mx[nz], np.maximum.reduceat(ro, ptr[:-1][nz]))
    assert np.all(mn[~nz] == np.iinfo(np.int32).min)
    # the case holds peaks right beside their bait on both sides; with RUexpand = 0 those are the regions of TWO rows (R's descending
    # (bait + 2):(oe + 0)) the single call's capacity n * max(2 RUexpand + 1, 2) is sized for — and the two-pass path agrees
    assert ((oe - bait) == 1).sum() > 20 and ((oe - bait) == -1).sum() > 20
    if s == 0:
        assert ln.max() == 2 and (ln == 2).sum() > 20 and len(ro) <= 2 * len(bait)
    import ctypes as C
    n, maxfrag = len(bait), len(chr_of) - 1
    db, do, dc = dev(bait), dev(oe), dev(chr_of)
    ptr2 = torch.empty(n + 1, dtype=torch.int64, device=ctx.device)
    mn2, mx2 = (torch.empty(n, dtype=torch.int32, device=ctx.device) for _ in range(2))
    total = C.c_int64(0)
    ctx._check(ctx.lib.chicdiff_hip_region_universe_count_dev(ctx.h, db.data_ptr(), do.data_ptr(), n, s, dc.data_ptr(), maxfrag, ptr2.data_ptr(), mn2.data_ptr(),
                                                             mx2.data_ptr(), C.byref(total)))
    assert total.value == len(ro) and np.array_equal(ptr2.cpu().numpy(), ptr)
    rb2, rr2, ro2 = (torch.empty(max(total.value, 1), dtype=torch.int32, device=ctx.device) for _ in range(3))
    ctx._check(ctx.lib.chicdiff_hip_region_universe_fill_dev(ctx.h, db.data_ptr(), do.data_ptr(), n, s, dc.data_ptr(), maxfrag, ptr2.data_ptr(), rb2.data_ptr(),
                                                            rr2.data_ptr(), ro2.data_ptr()))
    for a, k in ((rb2, "baitID"), (rr2, "regionID"), (ro2, "otherEndID")):
        assert torch.equal(a[: total.value], got[k]), ("count + fill vs the single call", k)
    with pytest.raises(Exception):
        ctx.region_universe(dev(np.array([5, 9], np.int32)), dev(np.array([7, 9], np.int32)), s, dev(chr_of))


def test_post_mirrors(ctx, golden, oracle):
    """Host mirrors over the f3 / f4 entry points: getRegionUniverse returns RU.DT's own row order,
    applyIHWweights builds the breaks the way chicdiff.R:2039 does."""
    from chicdiff_amd import post
    from post_inputs import region_universe_case, region_universe_literal
    bait, oe, chr_of = region_universe_case(n=1500)
    ids = np.nonzero(chr_of >= 0)[0]
    ru = post.getRegionUniverse(ctx, bait, oe, 5, chr_of[ids].astype(str), ids)
    lit = region_universe_literal(bait, oe, 5, chr_of)
    got = np.stack([ru[k].cpu().numpy() for k in ("baitID", "regionID", "otherEndID")], axis=1)
    assert np.array_equal(got, lit)
    # distLookup with group ranges that touch: the breaks fall on the group boundaries of the golden table
    g = golden["group"].astype(np.int64)
    x = np.log(np.abs(golden["avDist"]))
    ng = int(g.max())
    edges = [0.5 * (x[g == k].max() + x[g == k + 1].min()) for k in range(1, ng)]
    lo, hi = np.array([1.0] + edges), np.array(edges + [20.0])
    w = np.array([np.unique(golden["avWeights"][g == k])[0] for k in range(1, ng + 1)])
    out = post.applyIHWweights(ctx, golden["avDist"], golden["pvalue"], lo, hi, w)
    assert np.array_equal(out["group"].cpu().numpy(), golden["group"])
    assert np.allclose(out["weighted_padj"].cpu().numpy(), golden["weighted_padj"], rtol=1e-13)


def test_count_table_feeds_count_join(ctx, oracle):
    """f2: unsorted chinput columns -> key table (bait filter + sort) -> count join, bit-exact against the oracle."""
    import torch
    rng = np.random.default_rng(21)
    nrows = 200000
    pairs = rng.choice(3000 * 3000, nrows, replace=False)  # unique (baitID, otherEndID) pairs, as in a chinput
    bait, oe = (pairs // 3000 + 1).astype(np.int32), (pairs % 3000 + 1).astype(np.int32)
    N = rng.integers(1, 200, nrows).astype(np.int32)
    in_ru = (rng.uniform(size=3002) < 0.3).astype(np.uint8)
    dev = lambda a: torch.from_numpy(a).to(ctx.device)
    for flags in (in_ru, None):
        keys, vals = ctx.count_table(dev(bait), dev(oe), dev(N), dev(flags) if flags is not None else None)
        rk, rv = oracle.count_table(bait, oe, N, flags)
        assert np.array_equal(keys.cpu().numpy(), rk) and np.array_equal(vals.cpu().numpy(), rv)
    keys, vals = ctx.count_table(dev(bait), dev(oe), dev(N), dev(in_ru))
    qb, qo = rng.integers(1, 3001, 50000).astype(np.int32), rng.integers(1, 3001, 50000).astype(np.int32)
    order = np.lexsort((qo, qb))
    qb, qo = qb[order], qo[order]
    got = ctx.count_join(dev(qb), dev(qo), keys, vals).cpu().numpy()
    rk, rv = oracle.count_table(bait, oe, N, in_ru)
    assert np.array_equal(got, oracle.count_join(qb, qo, rk, rv)) and (got > 0).sum() > 100


def test_c_abi_without_torch_arrays(ctx, oracle):
    """What the R shim does: device buffers from the library's own malloc / memcpy entry points, no torch tensor
    anywhere on the data path (size factors, then BH on a vector of p-values)."""
    import ctypes as C
    L, h = ctx.lib, ctx.h
    d = synth.make(5000, 4)
    counts = np.asfortranarray(d["counts"].astype(np.int32))  # n x S column-major, as INTEGER(mat) in R
    dptr = C.c_void_p()
    assert L.chicdiff_hip_malloc(h, counts.nbytes, C.byref(dptr)) == 0 and dptr.value
    assert L.chicdiff_hip_memcpy_h2d(h, dptr, counts.ctypes.data, counts.nbytes) == 0
    sf = np.empty(4)
    assert L.chicdiff_hip_size_factors_dev(h, dptr, 5000, 4, sf.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.allclose(sf, oracle.size_factors(d["counts"]), rtol=1e-13)
    assert L.chicdiff_hip_free(h, dptr) == 0
    p = np.random.default_rng(2).uniform(size=4096)
    dp, dq = C.c_void_p(), C.c_void_p()
    assert L.chicdiff_hip_malloc(h, p.nbytes, C.byref(dp)) == 0 and L.chicdiff_hip_malloc(h, p.nbytes, C.byref(dq)) == 0
    assert L.chicdiff_hip_memcpy_h2d(h, dp, p.ctypes.data, p.nbytes) == 0
    assert L.chicdiff_hip_bh_adjust_dev(h, dp, len(p), dq) == 0
    q = np.empty_like(p)
    assert L.chicdiff_hip_memcpy_d2h(h, q.ctypes.data, dq, q.nbytes) == 0
    assert np.array_equal(q, oracle.bh_adjust(p))
    assert L.chicdiff_hip_free(h, dp) == 0 and L.chicdiff_hip_free(h, dq) == 0
    assert L.chicdiff_hip_memcpy_h2d(h, None, p.ctypes.data, 8) != 0  # NULL device pointer: an error, not a crash


def test_direct_rccl_single_rank_matches_hook(ctx):
    """The library's own RCCL communicator (dlopen'ed librccl, ncclAllReduce issued from C++) against the
    torch.distributed hook: same sharded protocol, same kernels, so identical bits.  World size 1 is what one
    GPU allows; it still goes through ncclCommInitRank / ncclAllReduce."""
    import socket
    import torch.distributed as dist
    from chicdiff_amd import hip
    created = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        d = synth.make(40000, 8)
        c2 = hip.HipContext(0)
        dk, dn = c2.to_device(d["counts"], np.int32), c2.to_device(d["nf"], np.float64)
        c2.set_process_group()
        a, sa = c2.nbglm_fit(dk, dn, d["group"])
        a = {k: v.cpu().numpy().copy() for k, v in a.items()}
        c2.init_rccl()
        b, sb = c2.nbglm_fit(dk, dn, d["group"])
        for k in a:
            assert np.array_equal(a[k], b[k].cpu().numpy(), equal_nan=True), k
        assert np.array_equal(sa["trendCoef"], sb["trendCoef"])
        sf_a = c2.size_factors(dk)
        assert np.all(np.isfinite(sf_a))
        c2.close()
    finally:
        if created:
            dist.destroy_process_group()


def _rccl_refusal_worker(rank, port, q):
    import os
    import torch.distributed as dist
    from chicdiff_amd import hip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        c = hip.HipContext(0)
        try:
            c.init_rccl()
            q.put((rank, "initialised"))
        except hip.ChicdiffHipError as e:
            q.put((rank, "raised " + str(e)))
        c.set_process_group(memory="device_via_host")  # what bench.py falls back to (there: the nccl hook)
        d = synth.make(2000, 4)
        lo, hi = (0, 1000) if rank == 0 else (1000, 2000)
        _, sc = c.nbglm_fit(c.to_device(d["counts"][lo:hi], np.int32), c.to_device(d["nf"][lo:hi], np.float64), d["group"])
        q.put((rank, tuple(sc["trendCoef"])))
        c.close()
    finally:
        dist.destroy_process_group()


def test_direct_rccl_failure_is_symmetric_and_falls_back():
    """Two ranks on this box's one GPU: the unique id travels, the bootstrap connects, and RCCL refuses the
    duplicate device.  init_rccl must then raise on BOTH ranks (a one-sided fallback would deadlock the next
    collective) and the hook path must still work."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rccl_refusal_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(4)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    first = {r: m for r, m in got if isinstance(m, str)}
    assert len(first) == 2 and all(m.startswith("raised") and "every rank" in m for m in first.values()), first
    trends = [m for _, m in got if not isinstance(m, str)]
    assert len(trends) == 2 and trends[0] == trends[1]


@pytest.mark.parametrize("n,S,nB", [(3000, 64, 32), (4000, 33, 3), (1, 8, 4), (63, 8, 4), (65, 4, 2), (129, 10, 7)])
def test_fit_edge_shapes(ctx, oracle, n, S, nB):
    """Maximum sample count (64: one row's samples fill a wave in the straggler layout), unbalanced groups,
    row counts around the wave size, a single row."""
    d = synth.make(max(n, 2000), S)
    counts, nf = d["counts"][:n].copy(), d["nf"][:n].copy()
    if n == 1:
        counts[0] = np.maximum(counts[0], 3)  # a lone all-zero row has nothing to fit
    group = np.array([0] * (S - nB) + [1] * nB, dtype=np.int32)
    ref = oracle.nbglm_fit(counts, nf, group)
    got, sc = run_fit(ctx, dict(counts=counts, nf=nf), group)
    # a handful of rows cannot carry the parametric trend: both sides substitute the local regression (bit 16), or —
    # fewer than four usable rows — report that there is no trend at all (bit 1)
    assert (sc["status"] & 17) == (ref["status"] & 17), (sc["status"], ref["status"])
    if ref["status"] & 1:
        return
    nz = ref["allZero"] == 0
    assert np.array_equal(got["allZero"], ref["allZero"])
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6, equal_nan=True)
    explain_fit(f"edge shape {n} x {S} ({S - nB}v{nB})", oracle, counts, nf, group, got, sc)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 64, 193])
def test_fitDispGrid_list_lengths(ctx, oracle, m):
    """disp_grid_kernel (round 6) takes the rows whose line search did not converge THREE per wave and pass: fits whose gene-wise
    search lists exactly m such rows, m around that boundary and around a wave's and a launch's share — every listed row's two grid
    stages against the oracle's fitDispGrid (dispGeneEst to 1e-6 or refereed, same iteration count), beside 300 ordinary rows that
    carry the trend.  The rows are picked by the oracle: dispGeneIter = 100 and an estimate above 10 minDisp (DESeq2's refit rule)."""
    d = synth.make(120000, 8)
    ref_all = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], nthreads=min(16, os.cpu_count() or 1))
    grid_rows = np.flatnonzero((ref_all["dispGeneIter"] >= 100) & (ref_all["dispGeneEst"] > 1e-7) & (ref_all["allZero"] == 0))
    plain = np.flatnonzero((ref_all["dispGeneIter"] < 30) & (ref_all["dispGeneIter"] > 1) & (ref_all["allZero"] == 0))[:300]
    assert len(grid_rows) >= 193, len(grid_rows)
    rng = np.random.default_rng(m)
    rows = np.concatenate([grid_rows[:m], plain])
    rows = rows[rng.permutation(len(rows))]
    counts, nf = d["counts"][rows].copy(), d["nf"][rows].copy()
    ref = oracle.nbglm_fit(counts, nf, d["group"])
    got, sc = run_fit(ctx, dict(counts=counts, nf=nf), d["group"])
    is_grid = np.isin(rows, grid_rows[:m])
    # (xim — the one global the gene-wise search sees — differs between the sub-matrix and the full one in its 3rd digit: a listed
    # row may start elsewhere; what must hold is GPU = oracle on THIS matrix, row by row)
    assert (ref["dispGeneIter"][is_grid] >= 100).sum() >= max(1, m // 2), "the picked rows no longer reach the grid under this matrix's xim"
    assert np.array_equal(got["dispGeneIter"][is_grid], ref["dispGeneIter"][is_grid])
    assert (sc["status"] & 17) == (ref["status"] & 17)
    explain_fit(f"fitDispGrid list of {m} rows", oracle, counts, nf, d["group"], got, sc)


def _extreme_matrix():
    """4000 x 6 synthetic rows, 300 of them with counts up to 2^30 next to zeros and offsets over four decades."""
    rng = np.random.default_rng(17)
    n, S = 4000, 6
    d = synth.make(n, S)
    counts, nf = d["counts"].copy(), d["nf"].copy()
    big = rng.choice(n, 300, replace=False)
    counts[big] = rng.integers(2 ** 20, 2 ** 30, size=(300, S))
    counts[big[:100], 0] = 0
    nf[big[100:200]] *= np.exp(rng.normal(0, 2.0, size=(100, S)))
    nf /= np.exp(np.log(nf).mean(axis=1, keepdims=True))
    return counts, nf, d["group"], big


def test_fit_type_mean_and_trend_failure(ctx, oracle):
    """DESeq2's estimateDispersions(fitType = "mean") (opts.fitType = 1): one fitted dispersion for every row, the
    0.1 %-trimmed mean of the gene-wise estimates above 10 minDisp.  Also the way out when the parametric trend fails
    when the local substitute (test_fit_type_local_and_local_substitute) is not wanted: the failure is reported in the
    status bits, the refit with fitType = 1 is clean."""
    d = synth.make(30000, 8)
    got, sc = run_fit(ctx, d, d["group"], fitType=1)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], fitType=1)
    assert sc["trendCoef"][1] == 0.0 and ref["trendCoef"][1] == 0.0 and not (sc["status"] & 1)
    # the mean runs over the estimates above 10 minDisp = 1e-7: a row the two sides leave at different points of the
    # flat floor region (1e-8 here, 3e-7 there — both "zero") enters on one side only and moves it by 1/m = 3e-5
    assert np.isclose(sc["trendCoef"][0], ref["trendCoef"][0], rtol=3e-4)
    g = got["dispGeneEst"]
    x = np.sort(g[(ref["allZero"] == 0) & (g > 1e-7)])
    lo = int(np.floor(len(x) * 0.001))
    assert np.isclose(sc["trendCoef"][0], x[lo:len(x) - lo].mean(), rtol=1e-12)  # R's mean(x, trim = 0.001) of the GPU's own estimates
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], trendCoef=sc["trendCoef"])
    nz = ref["allZero"] == 0
    assert np.nanmax(got["dispFit"]) == np.nanmin(got["dispFit"]) == sc["trendCoef"][0]
    assert_rows_explained("fitType mean, 30000 x 8", oracle, d, d["group"], got, ref, ref["dispPriorVar"])
    # a matrix on which the parametric fit fails (300 rows with counts ~1e9): with DESeq2's substitution switched off the
    # failure is reported on both sides (status bit 1), and fitType = "mean" is a clean way out
    counts, nf, group, _ = _extreme_matrix()
    ctx.set_option("local_trend_substitute", 0)
    try:
        _, sc1 = run_fit(ctx, dict(counts=counts, nf=nf), group)
    finally:
        ctx.set_option("local_trend_substitute", 1)
    ref1 = oracle.nbglm_fit(counts, nf, group, noLocalSubstitute=1)
    assert (sc1["status"] & 1) and (ref1["status"] & 1) and not (sc1["status"] & 16)
    got2, sc2 = run_fit(ctx, dict(counts=counts, nf=nf), group, fitType=1)
    ref2 = oracle.nbglm_fit(counts, nf, group, fitType=1)
    assert not (sc2["status"] & 1) and not (ref2["status"] & 1) and np.isclose(sc2["trendCoef"][0], ref2["trendCoef"][0], rtol=1e-4)


def test_fit_type_local_and_local_substitute(ctx, oracle):
    """DESeq2's local-regression trend (localDispersionFit = locfit with its defaults) on the device: on request
    (fitType = 2), and as estimateDispersionsFit's substitute when the parametric fit fails.  The trend is a handful of
    vertices (bandwidth = an order statistic, value and slope from eight weighted sums); compared with the oracle's
    restatement through dispFit and everything downstream of it."""
    for n, S in ((30000, 8), (5000, 5)):
        d = synth.make(n, S)
        got, sc = run_fit(ctx, d, d["group"], fitType=2)
        ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], fitType=2)
        nz = ref["allZero"] == 0
        assert (sc["status"] & 16) and not (sc["status"] & 1) and np.all(np.isnan(sc["trendCoef"])) and (ref["status"] & 16)
        assert np.array_equal(np.isnan(got["dispFit"]), ~nz)
        # the oracle's own local fit on the GPU's gene-wise estimates reproduces the GPU's dispFit: the device arithmetic
        # (order statistic by radix select, two-stage sums) is right whatever a noise-decided row did to the inputs
        use = nz & (got["dispGeneEst"] > 1e-6)
        _, pred = oracle.local_dispersion_fit(ref["baseMean"][use], got["dispGeneEst"][use])
        assert np.allclose(got["dispFit"][nz], np.exp(pred(np.log(ref["baseMean"][nz]))), rtol=1e-9)
        # (the two FREE local trends differ where a noise-decided gene-wise row moved an order statistic of the fit: reported)
        rf = rel(got["dispFit"][nz], ref["dispFit"][nz])
        print(f"local trend {n} x {S}: free GPU vs free oracle dispFit max rel {rf.max():.2e}, rows beyond 1e-6: {int((rf > 1e-6).sum())}")
        assert np.isclose(sc["varLogDispEsts"], ref["varLogDispEsts"], rtol=1e-4) and np.isclose(sc["dispPriorVar"], ref["dispPriorVar"], rtol=1e-4)
        # everything downstream of the trend: the oracle under the GPU's fitted values, every row within the bounds or refereed
        explain_fit(f"local trend, {n} x {S}", oracle, d["counts"], d["nf"], d["group"], got, sc)
    # the substitution: same results as asking for the local fit, on both sides
    counts, nf, group, big = _extreme_matrix()
    a, sa = run_fit(ctx, dict(counts=counts, nf=nf), group)
    b, sb = run_fit(ctx, dict(counts=counts, nf=nf), group, fitType=2)
    assert (sa["status"] & 16) and not (sa["status"] & 1) and sa["status"] == sb["status"]
    for k in ("dispFit", "dispersion", "pvalue", "log2FoldChange"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    ref = oracle.nbglm_fit(counts, nf, group)
    assert (ref["status"] & 16) and not (ref["status"] & 1)
    nz = ref["allZero"] == 0
    # the substituted trend is the oracle's local fit of the GPU's own gene-wise estimates (the free oracle fit's differs where the
    # ~1e9 rows — noise-decided on both sides, test_fit_extreme_counts — moved an order statistic: reported)
    use = nz & (a["dispGeneEst"] > 1e-6)
    _, pred = oracle.local_dispersion_fit(ref["baseMean"][use], a["dispGeneEst"][use])
    assert np.allclose(a["dispFit"][nz], np.exp(pred(np.log(ref["baseMean"][nz]))), rtol=1e-9)
    rf = rel(a["dispFit"][nz], ref["dispFit"][nz])
    print(f"substituted local trend: free GPU vs free oracle dispFit max rel {rf.max():.2e}, rows beyond 1e-3: {int((rf > 1e-3).sum())} of {int(nz.sum())}")
    # a design ~1 fit (the theta grid's) takes the same path
    g0 = np.zeros(8, dtype=np.int32)
    d = synth.make(8000, 8)
    got, sc = run_fit(ctx, d, g0, fitType=2)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], g0, fitType=2)
    nz = ref["allZero"] == 0
    use = nz & (got["dispGeneEst"] > 1e-6)
    _, pred = oracle.local_dispersion_fit(ref["baseMean"][use], got["dispGeneEst"][use])
    assert np.allclose(got["dispFit"][nz], np.exp(pred(np.log(ref["baseMean"][nz]))), rtol=1e-9)
    explain_fit("local trend, design ~1, 8000 x 8", oracle, d["counts"], d["nf"], g0, got, sc)
    assert np.isclose(np.sum(got["deviance"][nz]), np.sum(ref["deviance"][nz]), rtol=1e-7)  # what the theta grid sums


def test_fit_extreme_counts(ctx, oracle):
    """Counts up to 2^30 next to zeros, offsets over four decades: no overflow in the integer sums, no lost rows,
    the same NA pattern as the oracle.

    For the 300 rows with counts ~1e9 the log-likelihood is ~1e10, so the oracle's (and R's) lgamma(y + 1/alpha) -
    lgamma(1/alpha) carries ~1e-5 of rounding noise while the MAP search stops on changes < 1e-6: its stopping
    point is noise-decided.  Checked with 50-digit arithmetic (mpmath) on the worst rows: the GPU's MAP value sits
    1e-9..3e-6 below the true maximum of the posterior, the oracle's 2e-6..2e-4.  Those rows are therefore held
    to 2 % on the dispersion and 0.01 log2 units on the fold change; every other row to the usual 1e-6."""
    n = 4000
    counts, nf, group, big = _extreme_matrix()
    # (the parametric trend fails on this matrix; here the coefficients reached at that point are compared, so DESeq2's
    # substitution of the local fit is switched off on both sides — with it: test_fit_type_local_and_local_substitute)
    ctx.set_option("local_trend_substitute", 0)
    try:
        got, sc = run_fit(ctx, dict(counts=counts, nf=nf), group)
    finally:
        ctx.set_option("local_trend_substitute", 1)
    ref = oracle.nbglm_fit(counts, nf, group, noLocalSubstitute=1)
    assert np.array_equal(got["allZero"], ref["allZero"])
    nz = ref["allZero"] == 0
    isbig = np.zeros(n, bool)
    isbig[big] = True
    check_close("baseMean", got["baseMean"], ref["baseMean"], nz, 1e-12)
    # gene-wise estimates: ordinary rows to 1e-6; the ~1e9 rows are noise-decided on BOTH double-precision sides, so
    # each side is compared with the binary128 arbiter instead
    # (ordinary rows: explain_fit below, every row)
    bigrows = np.nonzero(nz & isbig)[0]
    arb = oracle.arbitrate_disp(counts, nf, group, bigrows, ref)
    eg, eo = rel(got["dispGeneEst"][bigrows], arb), rel(ref["dispGeneEst"][bigrows], arb)
    print(f"counts ~1e9, gene-wise vs binary128 arbiter: GPU median {np.median(eg):.1e} max {eg.max():.1e}; oracle median {np.median(eo):.1e} max {eo.max():.1e}")
    assert np.median(eg) < 1e-4 and np.mean(eg < 2e-2) > 0.97
    # trend | gene-wise estimates, then everything downstream under the GPU's trend
    useg = nz & (got["dispGeneEst"] > 1e-6)
    cg, itg, rc = oracle.parametric_dispersion_fit(ref["baseMean"][useg], got["dispGeneEst"][useg])
    print("trend: oracle routine on the GPU's estimates", cg, itg, rc, "GPU", sc["trendCoef"], sc["trendOuterIter"], sc["status"], "oracle", ref["trendCoef"], ref["trendOuterIter"], ref["status"])
    # (this matrix makes DESeq2's parametric fit stop on its second pass — both sides report it and carry on with the
    # coefficients reached, which is what is compared)
    assert (rc != 0) == bool(sc["status"] & 1) == bool(ref["status"] & 1) and itg == sc["trendOuterIter"]
    assert np.allclose(cg, sc["trendCoef"], rtol=1e-9)
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-4)
    # ordinary rows: every one within the bounds or refereed, under the GPU's trend
    ref, _ = explain_fit("extreme-count matrix, the 3700 ordinary rows", oracle, counts, nf, group, got, sc, rows=~isbig)
    # the ~1e9 rows: MAP estimates against the binary128 arbiter, as the gene-wise ones above (both double-precision sides are
    # noise-decided there: the oracle's lgamma difference carries ~1e-5 of rounding noise, the search stops on 1e-6)
    arb_m = oracle.arbitrate_disp(counts, nf, group, bigrows, dict(dispGeneEst=got["dispGeneEst"], dispFit=got["dispFit"], dispPriorVar=sc["dispPriorVar"]), stage="map")
    em = rel(got["dispMAP"][bigrows], arb_m)
    print(f"counts ~1e9, MAP vs binary128 arbiter (under the GPU's own gene-wise estimates, trend and prior): GPU median {np.median(em):.1e} max {em.max():.1e}")
    assert np.median(em) < 1e-3 and np.mean(em < 2e-2) > 0.97
    # the IRLS stops on a relative deviance change of 1e-8 while the deviance itself (~250, the difference of two
    # ~1e10 sums) carries ~1e-5 of noise on both sides: fold changes agree to a few 1e-3 (absolute, log2 units)
    conv = nz & (ref["betaConv"] == 1) & (got["betaConv"] == 1)
    irls = conv & isbig & (ref["betaIter"] < 100) & (got["betaIter"] < 100)
    assert irls.sum() > 150 and np.max(np.abs(got["log2FoldChange"][irls] - ref["log2FoldChange"][irls])) < 1e-2
    assert np.all(np.isfinite(got["log2FoldChange"][nz])) and np.all(np.isfinite(got["lfcSE"][nz]))
    assert np.array_equal(np.isnan(got["pvalue"]), np.isnan(ref["pvalue"]))


def test_all_rows_zero_and_bad_inputs(ctx):
    import torch
    from chicdiff_amd import hip
    z = torch.zeros((4, 500), dtype=torch.int32, device=ctx.device)
    nf = torch.ones((4, 500), dtype=torch.float64, device=ctx.device)
    out, sc = ctx.nbglm_fit(z, nf, [0, 0, 1, 1])
    assert sc["status"] & 8 and sc["nAllZero"] == 500 and torch.isnan(out["pvalue"]).all()
    with pytest.raises(hip.ChicdiffHipError):
        ctx.nbglm_fit(z, nf, [0, 0, 0, 2])          # not a two-level design
    with pytest.raises(hip.ChicdiffHipError):
        ctx.nbglm_fit(z[:2].contiguous(), nf[:2].contiguous(), [0, 1])   # no residual degrees of freedom
    with pytest.raises(hip.ChicdiffHipError):
        ctx.nbglm_fit(z, nf, [1, 1, 1, 1])          # no sample in the reference level


def test_results_on_device(ctx, golden, oracle):
    """a9: independent filtering + BH on device reproduces the reference's padj column (24 863 rows: the 2 411 NA,
    quantile index 6, the BH values) and, on a synthetic 300 k table with ties and NA, the host restatement that the
    golden table pins; Cook's filter against the host restatement."""
    import torch
    import results_twin as results
    dev = lambda a, t=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(ctx.device)
    padj, info = ctx.independent_filtering(dev(golden["baseMean"]), dev(golden["pvalue"]))
    got, ref = padj.cpu().numpy(), golden["padj"]
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(ref).sum() == 2411
    assert info["index"] == 6 and 4.7967 < info["filterThreshold"] <= 4.7975
    ok = ~np.isnan(ref)
    assert np.allclose(got[ok], ref[ok], rtol=1e-13, atol=0)
    rng = np.random.default_rng(9)
    n = 300000
    bm = rng.lognormal(np.log(19), 1.4, n)
    bm[rng.uniform(size=n) < 0.02] = 0.0
    p = rng.uniform(size=n)
    p[rng.uniform(size=n) < 0.15] **= 6
    p[bm == 0] = np.nan
    p[rng.integers(0, n, 500)] = p[11]
    hp, hinfo = results.independent_filtering(bm, p)
    dp, dinfo = ctx.independent_filtering(dev(bm), dev(p))
    assert np.array_equal(dinfo["numRej"], hinfo["numRej"]) and dinfo["index"] == hinfo["index"]
    assert np.allclose(dinfo["theta"], hinfo["theta"], rtol=1e-15) and np.isclose(dinfo["filterThreshold"], hinfo["filterThreshold"], rtol=1e-15)
    assert np.allclose(dinfo["lowess"], hinfo["lowess"], rtol=1e-9)
    assert np.allclose(dp.cpu().numpy(), hp, rtol=1e-13, equal_nan=True)
    # Cook's filter
    S, m = 8, 20000
    d = synth.make(m, S)
    group = d["group"]
    mc = rng.gamma(1.0, 1.0, m)
    mc[rng.uniform(size=m) < 0.1] = np.nan
    am = rng.integers(0, S, m).astype(np.int32)
    pv = rng.uniform(size=m)
    ref_p, ref_n = results.cooks_filter(pv, mc, am, lambda idx: d["counts"][idx], group, cutoff=2.5)
    dpv = dev(pv)
    nout = ctx.cooks_filter(ctx.to_device(d["counts"], np.int32), group, dev(mc), dev(am, np.int32), dpv, 2.5)
    assert nout == ref_n > 0 and np.array_equal(dpv.cpu().numpy(), ref_p, equal_nan=True)


def test_independent_filtering_edge_cases(ctx):
    """The one-sort pipeline of results() on inputs that stress its ranks: fewer rows than a workgroup, a single row,
    no p-value at all, every p-value rejected (the p < alpha prefix is the whole table), none rejected, ties that
    straddle wave and workgroup boundaries, rows without baseMean — against the numpy restatement the golden
    table pins."""
    import torch
    import results_twin as results
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx.device)
    rng = np.random.default_rng(77)

    def check(bm, p, what):
        hp, hinfo = results.independent_filtering(bm, p)
        dp, dinfo = ctx.independent_filtering(dev(bm), dev(p))
        assert np.array_equal(dinfo["numRej"], hinfo["numRej"]), what
        assert dinfo["index"] == hinfo["index"], what
        assert np.allclose(dinfo["theta"], hinfo["theta"], rtol=1e-15, atol=1e-300), what
        assert np.isclose(dinfo["filterThreshold"], hinfo["filterThreshold"], rtol=1e-15, equal_nan=True), what
        assert np.allclose(dp.cpu().numpy(), hp, rtol=1e-13, equal_nan=True), what

    for n in (1, 2, 7, 63, 64, 65, 1023, 1024, 1025, 4097):
        bm = rng.lognormal(2.0, 1.0, n)
        p = rng.uniform(size=n) ** 3
        check(bm, p, f"n={n}")
    n = 30000
    bm = rng.lognormal(np.log(19), 1.4, n)
    check(bm, np.full(n, np.nan), "no p-value")
    check(bm, rng.uniform(0, 1e-9, n), "everything rejected")
    check(bm, rng.uniform(0.5, 1.0, n), "nothing rejected")
    check(bm, np.full(n, 1.0), "all p = 1")
    p = rng.uniform(size=n) ** 4
    p[:5000] = 1e-5                      # one tie group across many waves / workgroups of the p-order
    p[5000:9000] = 0.03
    check(bm, p, "long tie groups")
    bm0 = bm.copy()
    bm0[rng.uniform(size=n) < 0.4] = 0.0  # lower = mean(baseMean == 0) moves the theta grid
    check(bm0, p, "many zero baseMeans")
    bm1 = np.full(n, 7.0)                # every cutoff equal: every row passes every filter
    check(bm1, p, "constant baseMean")


def test_row_queue_kernels_at_the_edges_of_their_chunks(ctx, oracle):
    """The line searches and the IRLS hand rows to lanes in chunks of 64 schedule entries (the static deal in groups of 1-8 entries
    per wave, 64 lanes per wave): fits of 1 .. 200 rows — fewer rows than a wave has lanes, exactly one chunk, one row more,
    two chunks and a row — under a pinned trend and prior (so that a handful of rows cannot derail the global steps) against
    the oracle under the same pins: every row within 1e-6 or refereed, whatever the schedule option."""
    from chicdiff_amd import hip
    S = 8
    big = synth.make(4000, S)
    pins = dict(trendCoef=(0.06, 2.5), dispPriorVar=0.4)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 200):
        # take rows with counts: an all-zero row alone has nothing to fit
        keep = np.flatnonzero(big["counts"].sum(1) > 0)[:n]
        d = {"counts": np.ascontiguousarray(big["counts"][keep]), "nf": np.ascontiguousarray(big["nf"][keep]), "group": big["group"]}
        ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], **pins)
        for sched in (1, 0):
            ctx.set_option("line_search_schedule", sched)
            try:
                got, sc = run_fit(ctx, d, d["group"], **pins)
            finally:
                ctx.set_option("line_search_schedule", 1)
            assert np.array_equal(got["allZero"], ref["allZero"]) and not got["allZero"].any()
            for k in ("dispGeneEst", "dispMAP", "dispersion", "log2FoldChange", "pvalue"):
                assert not np.any(np.isnan(got[k])), (n, sched, k)
            assert_rows_explained(f"{n} rows, schedule {sched}", oracle, d, d["group"], got, ref, pins["dispPriorVar"])


def test_fit_fuzz_shapes_and_designs(ctx, oracle):
    """Random sample counts, group splits and row counts (both designs) against the oracle: every configuration must
    agree on the NA pattern and on every row (within the bounds, or refereed: explain_fit)."""
    rng = np.random.default_rng(2024)
    worst = 0
    for trial in range(14):
        S = int(rng.integers(3, 21))
        n = int(rng.integers(800, 4000))
        d = synth.make(n, S, start=int(rng.integers(0, 10 ** 6)))
        if trial % 4 == 3:
            group = np.zeros(S, dtype=np.int32)  # design ~1
        else:
            nB = int(rng.integers(1, S - 1))
            group = np.zeros(S, dtype=np.int32)
            group[rng.choice(S, nB, replace=False)] = 1
            if S - nB < 1 or S <= 2:
                continue
        ref = oracle.nbglm_fit(d["counts"], d["nf"], group)
        got, sc = run_fit(ctx, d, group)
        assert (sc["status"] & 17) == (ref["status"] & 17), (trial, S, n)
        if ref["status"] & 1:
            continue
        nz = ref["allZero"] == 0
        assert np.array_equal(got["allZero"], ref["allZero"])
        assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6, equal_nan=True), (trial, sc["trendCoef"], ref["trendCoef"])
        _, listed = explain_fit(f"fuzz trial {trial}: {n} x {S}, group {group.tolist()}", oracle, d["counts"], d["nf"], group, got, sc)
        worst = max(worst, int(listed.sum()))
        assert np.array_equal(np.isnan(got["pvalue"]), np.isnan(ref["pvalue"])), (trial, S, n)
        if not group.any():
            assert np.isclose(sc["sumDeviance"], ref["sumDeviance"], rtol=1e-6, equal_nan=True)
    print("fuzz: most refereed rows in one configuration:", worst)


@pytest.mark.parametrize("S,group", [(4, [0, 0, 1, 1]), (4, [0, 0, 0, 0]), (5, [0, 0, 1, 1, 1]), (3, [0, 0, 0])])
def test_prior_variance_by_simulation_matches_oracle(ctx, oracle, S, group):
    """Residual d.f. 1..3 (2v2, the reference's own test design, among them): the device histogram + the library's
    host-side simulation give the oracle's dispPriorVar, and with it the same dispersions and p-values."""
    from test_oracle import heterogeneous_counts
    counts, nf = heterogeneous_counts(6000, S, 1.3)
    group = np.asarray(group, dtype=np.int32)
    got, sc = run_fit(ctx, dict(counts=counts, nf=nf), group)
    ref = oracle.nbglm_fit(counts, nf, group)
    assert sc["status"] & 2 and ref["status"] & 2
    assert np.isclose(sc["varLogDispEsts"], ref["varLogDispEsts"], rtol=1e-7)
    assert sc["dispPriorVar"] == ref["dispPriorVar"] and sc["dispPriorVar"] > 0.3, (sc["dispPriorVar"], ref["dispPriorVar"])
    assert np.allclose(sc["trendCoef"], ref["trendCoef"], rtol=1e-6)
    ref_g, _ = explain_fit(f"prior variance by simulation, S = {S}, group {group.tolist()}", oracle, counts, nf, group, got, sc)
    assert ref_g["dispPriorVar"] == sc["dispPriorVar"]


def test_chinput_ingestion_on_device(ctx, oracle, tmp_path):
    """f2 end to end: a .chinput text file (format of chicdiff.R:828) -> host-thread parser ->
    bait filter + radix sort on the device -> count join, against the oracle fed the same rows (bit-exact)."""
    import time
    import torch
    from test_chinput import make_rows, write_chinput
    bait, oe, N = make_rows(400_000, seed=9)
    keep = np.sort(np.unique(np.stack([bait, oe], 1), axis=0, return_index=True)[1])
    bait, oe, N = bait[keep], oe[keep], N[keep]
    path = tmp_path / "big.chinput"
    write_chinput(path, bait, oe, N)
    rng = np.random.default_rng(5)
    ru_baits = np.unique(rng.choice(bait, 4000))
    flags = np.zeros(int(bait.max()) + 1, np.uint8)
    flags[ru_baits] = 1
    t0 = time.perf_counter()
    keys, vals, nrows = ctx.read_chinput(path, torch.as_tensor(flags).to(ctx.device))
    dt = time.perf_counter() - t0
    print(f"chinput: {os.path.getsize(path) / 1e6:.1f} MB, {nrows} rows read, {keys.numel()} kept, {dt * 1e3:.1f} ms")
    assert nrows == len(bait)
    rk, rv = oracle.count_table(bait, oe, N, flags)
    assert np.array_equal(keys.cpu().numpy(), rk) and np.array_equal(vals.cpu().numpy(), rv)
    pick = rng.choice(len(rk), 50_000)
    qb = (rk[pick] >> 32).astype(np.int32)
    qo = ((rk[pick] & 0xFFFFFFFF) + rng.integers(0, 2, len(pick))).astype(np.int32)  # half of them one fragment off
    order = np.lexsort((qo, qb))
    qb, qo = qb[order], qo[order]
    got = ctx.count_join(torch.as_tensor(qb).to(ctx.device), torch.as_tensor(qo).to(ctx.device), keys, vals).cpu().numpy()
    assert np.array_equal(got, oracle.count_join(qb, qo, rk, rv)) and (got > 0).sum() > 1000 and (got == 0).sum() > 1000


def test_pipeline_from_peaks_and_chinput_text_to_weighted_padj(ctx, oracle, golden, tmp_path):
    """Every row of SURVEY.md §8 in one chain, nothing but file names and small tables crossing the host boundary:
    peaks -> region universe (f4) -> per replicate: .chinput text -> key table (f2) -> count join (a1);
    Chicago tables -> Bmean / Tmean / FullMean per fragment (a3) -> window sums (a2) -> size factors, sc(theta),
    dispersions, Wald test (a4-a7, one call) -> Cook's cutoff, independent filtering, BH (a9) -> IHW application (f3);
    against the oracle run link by link on the same inputs (integers bit-exact, the rest to 1e-6)."""
    import torch
    from scipy import stats
    import results_twin as results
    from post_inputs import ihw_tables_from_golden, region_universe_case
    from test_chinput import write_chinput
    rng = np.random.default_rng(2026)
    S, RUexpand = 8, 5
    group = synth.groups(S)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(ctx.device)
    # f4: peaks on the reference's chr19 HindIII geometry -> RU rows in (regionID, otherEndID) order
    pb, po, chr_of = region_universe_case(seed=11, n=6000)
    nonempty = np.diff(oracle.region_universe(pb, po, RUexpand, chr_of)[0]) > 0   # a peak whose window has no fragment on the
    pb, po = pb[nonempty], po[nonempty]                                            # map / chromosome is not a region
    ru = ctx.region_universe(t(pb), t(po), RUexpand, t(chr_of))
    ptr_ref, rb_ref, rr_ref, ro_ref = oracle.region_universe(pb, po, RUexpand, chr_of)
    ru_bait, ru_oe = ru["baitID"].cpu().numpy(), ru["otherEndID"].cpu().numpy()
    assert np.array_equal(ru["region_ptr"].cpu().numpy(), ptr_ref) and np.array_equal(ru_bait, rb_ref) and np.array_equal(ru_oe, ro_ref)
    n, nru = len(pb), len(ru_bait)
    region_of = rr_ref.astype(np.int64) - 1
    # the experiment: one mean / dispersion / fold change per region, shared out over its fragments; a fragment pair
    # has ONE count per replicate however many windows hold it
    pair = (ru_bait.astype(np.int64) << 32) | ru_oe
    upair, first = np.unique(pair, return_index=True)
    mu = rng.lognormal(np.log(6.0), 1.0, n)[region_of[first]]
    alpha = (0.05 + 1.0 / mu) * rng.lognormal(0, 0.3, len(upair))
    lfc = np.where(rng.uniform(size=n) < 0.15, rng.normal(0, 1.5, n), 0.0)[region_of[first]]
    depth = rng.lognormal(0, 0.2, S)
    files, flags = [], np.zeros(int(chr_of.shape[0]), np.uint8)
    flags[np.unique(ru_bait)] = 1
    N_ref = np.zeros((nru, S), dtype=np.int32)
    fragN = torch.empty((S, nru), dtype=torch.int32, device=ctx.device)
    for s in range(S):
        m = mu * depth[s] * 2.0 ** (lfc * group[s])
        k = rng.negative_binomial(1.0 / alpha, 1.0 / (1.0 + alpha * m)).astype(np.int32)
        seen = k > 0                                              # chinput holds observed pairs only
        ob, oo = rng.integers(1, len(chr_of), 20000).astype(np.int64), rng.integers(1, len(chr_of), 20000).astype(np.int64)
        other = np.setdiff1d((ob << 32) | oo, upair)              # reads of pairs outside RU (incl. non-RU baits)
        kb = np.concatenate([upair[seen] >> 32, other >> 32]).astype(np.int32)
        ko = np.concatenate([upair[seen] & 0xFFFFFFFF, other & 0xFFFFFFFF]).astype(np.int32)
        kn = np.concatenate([k[seen], rng.integers(1, 30, len(other)).astype(np.int32)])
        shuffle = rng.permutation(len(kb))
        path = tmp_path / f"rep{s}.chinput"
        write_chinput(path, kb[shuffle], ko[shuffle], kn[shuffle])
        keys, vals, nrows = ctx.read_chinput(path, t(flags))       # f2
        fragN[s] = ctx.count_join(ru["baitID"], ru["otherEndID"], keys, vals)   # a1
        rk, rv = oracle.count_table(kb[shuffle], ko[shuffle], kn[shuffle], flags)
        assert np.array_equal(keys.cpu().numpy(), rk) and np.array_equal(vals.cpu().numpy(), rv)
        N_ref[:, s] = oracle.count_join(ru_bait, ru_oe, rk, rv)
        assert np.array_equal(N_ref[:, s], k[np.searchsorted(upair, pair)])
    assert np.array_equal(fragN.cpu().numpy().T, N_ref)
    # a3: Chicago tables (per replicate) on the same geometry
    rmap = np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chr19_HindIII_first3000.rmap"), dtype=str)
    nid = len(chr_of) - 1
    midsum = np.zeros(nid, dtype=np.int64)
    midsum[: len(rmap)] = rmap[:, 1].astype(np.int64) + rmap[:, 2].astype(np.int64)   # IDs renumbered 1..3000 in file order
    baits_all = np.unique(ru_bait)
    sj = np.full((S, nid), np.nan)
    sj[:, baits_all - 1] = np.exp(rng.normal(0, 0.3, (S, len(baits_all))))
    sj[:, baits_all[::23] - 1] = np.nan                         # baits Chicago filtered out: FullMean NA -> size-factor rows
    si = np.where(rng.random((S, nid)) < 0.8, np.exp(rng.normal(0, 0.3, (S, nid))), np.nan)
    ntblb, ntlb = 5, 6
    tblb = np.full((S, nid), -1, dtype=np.int32)
    tblb[:, baits_all - 1] = rng.integers(0, ntblb, (S, len(baits_all)))
    tlb = np.where(rng.random((S, nid)) < 0.85, rng.integers(0, ntlb, (S, nid)), -1).astype(np.int32)
    T = np.exp(rng.normal(-3, 0.5, (S, ntblb, ntlb)))
    distfun = np.zeros((S, 10))
    for s in range(S):
        fit = np.array([14.0 + 0.1 * s, -1.6, 0.05, -0.003])
        x = np.array([np.log(10000.0), np.log(1.5e6)])
        beta = fit[1] + 2 * fit[2] * x + 3 * fit[3] * x ** 2
        al = fit[0] + (fit[1] - beta) * x + fit[2] * x ** 2 + fit[3] * x ** 3
        distfun[s] = [*fit, al[0], beta[0], al[1], beta[1], x[0], x[1]]
    a3 = dict(bait=ru_bait, oe=ru_oe, id_min=1, midsum=midsum, sj=sj, si=si, tblb=tblb, tlb=tlb, T=T, distfun=distfun)
    _, _, fragFM = ctx.fragment_background(ru["baitID"], ru["otherEndID"], 1, t(midsum), t(sj), t(si), t(tblb), t(tlb), t(T), distfun)
    _, _, FM_frag_ref = oracle.fragment_background(**a3)
    assert np.allclose(fragFM.cpu().numpy(), FM_frag_ref, rtol=1e-13, equal_nan=True)
    # a2: window sums
    dN, dFM = ctx.window_sums(fragN, fragFM, ru["region_ptr"])
    N_w, FM_w = oracle.window_sums(N_ref, FM_frag_ref.T, ptr_ref)
    assert np.array_equal(dN.cpu().numpy().T, N_w)
    assert np.allclose(dFM.cpu().numpy().T, FM_w, rtol=1e-13, equal_nan=True) and np.isnan(FM_w).any()
    # a4-a7 in one call, then a9
    theta = 0.5
    out, sc = ctx.wald_test(dN, dFM, group, theta=theta, want=WANT + ["cooksArgmax"])
    got = {k: v.cpu().numpy().copy() for k, v in out.items()}   # (before the Cook's filter rewrites the p-values in place)
    sf = oracle.size_factors(N_w)
    assert np.allclose(sc["sizeFactors"], sf, rtol=1e-12)
    nf_dev = ctx.offsets(dFM, sc["sizeFactors"], theta).cpu().numpy().T   # the offsets the fused call formed (test_fused_wald_test_equals_composed_calls)
    assert np.allclose(nf_dev, oracle.offsets(FM_w, sf, theta), rtol=1e-12)
    # a4-a7: every row of the fit within the bounds of the oracle under the same trend, or refereed
    ref, listed = explain_fit("pipeline from peaks and chinput text", oracle, N_w, nf_dev, group, got, sc)
    # a9 on the device against the host restatement (pinned by the reference's golden table) fed the device's own numbers: exact
    cutoff = stats.f.ppf(0.99, 2, S - 2)
    ctx.cooks_filter(dN, group, out["maxCooks"], out["cooksArgmax"], out["pvalue"], cutoff)
    p_twin, _ = results.cooks_filter(got["pvalue"], got["maxCooks"], got["cooksArgmax"], lambda idx: N_w[idx], group, cutoff=cutoff)
    p_ref, _ = results.cooks_filter(ref["pvalue"], ref["maxCooks"], ref["cooksArgmax"], lambda idx: N_w[idx], group, cutoff=cutoff)
    d_padj, info = ctx.independent_filtering(out["baseMean"], out["pvalue"])
    got_p, got_padj = out["pvalue"].cpu().numpy(), d_padj.cpu().numpy()
    assert np.array_equal(got_p, p_twin, equal_nan=True)
    assert np.array_equal(np.isnan(got_p)[~listed], np.isnan(p_ref)[~listed])    # the oracle flags the same Cook's outliers
    padj_twin, info_twin = results.independent_filtering(got["baseMean"], got_p)
    padj_ref, info_ref = results.independent_filtering(ref["baseMean"], p_ref)
    assert info["index"] == info_twin["index"] and np.array_equal(np.isnan(got_padj), np.isnan(padj_twin))
    okp = ~np.isnan(padj_twin)
    assert np.allclose(got_padj[okp], padj_twin[okp], rtol=1e-12)
    # (against the oracle's own p-values the filter choice is a discrete decision on 50 rejection counts: reported)
    print(f"independent filtering: device index {info['index']}, on the oracle's p-values {info_ref['index']}")
    # f3: IHW application with the reference's trained weight table
    breaks, avWeights = ihw_tables_from_golden(golden)
    avDist = ((midsum[po.astype(np.int64).clip(1, nid) - 1] - midsum[pb.astype(np.int64) - 1]) / 2.0).astype(np.float64)
    avDist[avDist == 0] = 1.0
    w = ctx.ihw_apply(t(avDist), out["pvalue"], breaks, avWeights)
    g_ref, w_ref, wp_ref, wpadj_ref = oracle.ihw_apply(avDist, got_p, breaks, avWeights)
    assert np.array_equal(w["group"].cpu().numpy(), g_ref)
    assert np.allclose(w["weighted_padj"].cpu().numpy(), wpadj_ref, rtol=1e-12, equal_nan=True)
    # the same through the host mirrors a caller uses: getFullRegionDataHip() -> DESeq2Wrap() (r/R/*.R's tested twins)
    import pandas as pd
    from chicdiff_amd import post
    from chicdiff_amd.deseq2wrap import DESeq2Wrap
    RU_df = pd.DataFrame({"baitID": ru_bait, "regionID": rr_ref, "otherEndID": ru_oe}).sample(frac=1.0, random_state=3)  # any row order
    conds = ["ctrl" if g == 0 else "treat" for g in group]
    frd = post.getFullRegionDataHip(ctx, RU_df, [tmp_path / f"rep{s}.chinput" for s in range(S)], conds, a3)
    assert torch.equal(frd["fragN"], fragN) and torch.equal(frd["region_ptr"], ru["region_ptr"])
    assert torch.allclose(frd["fragFullMean"], fragFM, rtol=0, atol=0, equal_nan=True)
    rmapfile = tmp_path / "renumbered.rmap"
    with open(rmapfile, "w") as f:
        for k in range(len(rmap)):
            f.write(f"chr{max(int(chr_of[k + 1]), 0) + 1}\t{rmap[k, 1]}\t{rmap[k, 2]}\t{k + 1}\n")
    settings = {"norm": "combined", "theta": theta, "theta_grid": [0, 0.25, 0.5, 0.75, 1], "rmapfile": str(rmapfile),
                "saveAuxData": False, "outprefix": str(tmp_path / "x")}
    tab = DESeq2Wrap(settings, RU_df, frd, ctx=ctx)
    assert np.array_equal(tab["regionID"].to_numpy(), np.arange(1, n + 1)) and tab.attrs["theta"] == theta
    assert np.array_equal(tab["pvalue"].to_numpy(), got_p, equal_nan=True) and np.array_equal(tab["padj"].to_numpy(), got_padj, equal_nan=True)
    assert np.array_equal(tab["baitID"].to_numpy(), pb) and np.array_equal(tab["minOE"].to_numpy(), ru["minOE"].cpu().numpy())
    print(f"pipeline: {n} peaks, {nru} RU rows, {S} chinput files; filter index {info['index']} (oracle {info_ref['index']}), "
          f"padj < 0.05: {int(np.nansum(got_padj < 0.05))} (oracle {int(np.nansum(padj_ref < 0.05))})")


def test_full_size_C5_20M_x16_pipeline_properties_and_slice_parity(ctx, oracle):
    """BASELINE.json configs[4] on one GPU: 20 M interactions x 16 samples (8 v 8) through the composed path — size factors
    -> sc(theta) -> dispersions -> Wald test (one call) -> results(): Cook's cutoff + independent filtering + BH -> the
    application side of IHWcorrection.  At this size the oracle cannot run on every row, so: size-independent properties
    (size factors against torch's own medians, NA pattern, BH monotone in p, weights average 1, row permutation), and
    oracle parity on a 200 000-row slice with the three global scalars of the fit (trend, prior variance,
    varLogDispEsts) pinned to the whole fit's."""
    import torch
    from scipy import stats
    n, S, chunk = 20_000_000, 16, 2_000_000
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(20190123)
    group = synth.groups(S)
    gt = torch.as_tensor(group, device=dev, dtype=torch.float64)
    dk = torch.empty((S, n), dtype=torch.int32, device=dev)
    dfm = torch.empty((S, n), dtype=torch.float64, device=dev)
    sj = torch.exp(torch.randn(S, dtype=torch.float64, device=dev, generator=g) * 0.2)
    for lo in range(0, n, chunk):  # the generator of chicdiff_amd/synth.py, on the device
        m = min(chunk, n - lo)
        mu = torch.exp(torch.randn(m, dtype=torch.float64, device=dev, generator=g) * 1.4 + np.log(19.0))
        alpha = (0.05 + 2.0 / mu) * torch.exp(torch.randn(m, dtype=torch.float64, device=dev, generator=g) * 0.5)
        lfc = torch.where(torch.rand(m, dtype=torch.float64, device=dev, generator=g) < 0.10,
                          torch.randn(m, dtype=torch.float64, device=dev, generator=g), torch.zeros((), dtype=torch.float64, device=dev))
        r = sj[:, None] * torch.exp(torch.randn((S, m), dtype=torch.float64, device=dev, generator=g) * 0.25)
        r = r / torch.exp(torch.log(r).mean(0, keepdim=True))
        mean = mu[None, :] * torch.exp2(lfc[None, :] * gt[:, None]) * r
        lam = torch._standard_gamma((1.0 / alpha)[None, :].expand(S, m).contiguous(), generator=g) * (alpha[None, :] * mean)
        dk[:, lo:lo + m] = torch.poisson(lam, generator=g).to(torch.int32)
        dfm[:, lo:lo + m] = r * (mu[None, :] / S)  # region-level FullMean
        del mu, alpha, lfc, r, mean, lam
    want = ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue", "maxCooks", "cooksArgmax", "allZero", "betaConv", "betaIter",
            "dispGeneEst", "dispMAP", "dispFit", "dispOutlier", "dispIter"]
    out, sc = ctx.wald_test(dk, dfm, group, theta=0.5, want=want)
    print("C5 scalars", {k: sc[k] for k in ("trendCoef", "varLogDispEsts", "dispPriorVar", "nAllZero", "status")})
    assert not (sc["status"] & 1) and not (sc["status"] & 2)
    # size factors: median of ratios over the rows without a zero, by torch
    lk = torch.log(dk.to(torch.float64))
    okrow = torch.isfinite(lk).all(0)
    ratio = (lk - lk.mean(0, keepdim=True))[:, okrow]
    m = ratio.shape[1]
    srt = torch.sort(ratio, dim=1).values
    med = (srt[:, (m - 1) // 2] + srt[:, m // 2]) / 2
    assert np.allclose(sc["sizeFactors"], torch.exp(med).cpu().numpy(), rtol=1e-12)
    del lk, ratio, srt
    # results(): Cook's cutoff (8 v 8: applies) + independent filtering + BH
    p_raw = out["pvalue"].clone()
    nout = ctx.cooks_filter(dk, group, out["maxCooks"], out["cooksArgmax"], out["pvalue"], stats.f.ppf(0.99, 2, S - 2))
    padj, info = ctx.independent_filtering(out["baseMean"], out["pvalue"])
    p, q = out["pvalue"], padj
    az = out["allZero"] != 0
    assert int(az.sum()) == sc["nAllZero"] and bool(torch.isnan(p_raw)[az].all()) and not bool(torch.isnan(p_raw)[~az].any())
    assert int(torch.isnan(p).sum()) == int(az.sum()) + nout and 0 < nout < 0.02 * n
    assert bool(torch.isnan(q)[torch.isnan(p)].all())
    filtered = torch.isnan(q) & ~torch.isnan(p)
    assert bool((out["baseMean"][filtered] < info["filterThreshold"]).all()) and bool((out["baseMean"][~torch.isnan(q)] >= info["filterThreshold"]).all())
    ok = ~torch.isnan(q)
    order = torch.argsort(p[ok])
    qs = q[ok][order]
    assert bool((qs[1:] >= qs[:-1]).all()) and bool((q[ok] >= p[ok]).all()) and float(qs.max()) <= 1.0
    print(f"C5 results(): {nout} Cook's outliers, {int(filtered.sum())} rows filtered at baseMean < {info['filterThreshold']:.3f}, {int((q < 0.05).sum())} with padj < 0.05")
    # IHW application side (chicdiff.R:2038-2049) with a 5-group weight table
    av = torch.exp(torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 6 + 9)
    breaks = np.array([-np.inf, 10.5, 11.5, 12.5, 13.5, np.inf])
    ihw = ctx.ihw_apply(av, p, breaks, np.array([2.0, 1.5, 1.0, 0.6, 0.3]))
    w = ihw["weight"]
    assert abs(float(w.mean()) - 1.0) < 1e-9 and int(ihw["group"].min()) == 1 and int(ihw["group"].max()) == 5
    wp = ihw["weighted_pvalue"]
    okw = ~torch.isnan(wp)
    assert torch.equal(okw, ~torch.isnan(p)) and bool(torch.allclose(wp[okw], p[okw] / w[okw], rtol=1e-15))
    wq = ihw["weighted_padj"][okw][torch.argsort(wp[okw])]
    assert bool((wq[1:] >= wq[:-1]).all())
    # oracle parity on a slice, global scalars pinned
    lo, hi = 7_000_000, 7_200_000
    d_nf = ctx.offsets(dfm, sc["sizeFactors"], 0.5)
    cs, nfs = dk[:, lo:hi].T.cpu().numpy(), d_nf[:, lo:hi].T.cpu().numpy()
    # the fourth global scalar, xim = mean_j 1 / colMeans(nf)_j over the non-all-zero rows (it enters every start value)
    colmeans = (d_nf * (out["allZero"] == 0)).sum(1) / float(n - sc["nAllZero"])
    xim = float((1.0 / colmeans).mean())
    ref = oracle.nbglm_fit(cs, nfs, group, nthreads=min(16, os.cpu_count() or 1), trendCoef=sc["trendCoef"], dispPriorVar=sc["dispPriorVar"],
                           varLogDispEsts=sc["varLogDispEsts"], xim=xim)
    del d_nf, colmeans
    sl = {k: out[k][lo:hi].cpu().numpy() for k in want if k != "pvalue"}
    sl["pvalue"] = p_raw[lo:hi].cpu().numpy()
    # every row of the slice, optim-fallback rows included: within bounds or refereed (binary128 line search, IRLS trace)
    listed, wl = assert_rows_explained("C5 slice (200 000 of 20 M x 16), scalars pinned to the whole fit's", oracle, dict(counts=cs, nf=nfs), group, sl, ref,
                                       sc["dispPriorVar"])
    nz = (ref["allZero"] == 0) & ~wl
    check_close("baseMean(C5 slice)", sl["baseMean"], ref["baseMean"], nz, 1e-13)
    check_close("maxCooks(C5 slice)", sl["maxCooks"], ref["maxCooks"], nz & (ref["maxCooks"] > 1e-12), 1e-5)
    # row permutation
    del av, ihw, w, wp
    perm = torch.randperm(n, device=dev, generator=g)
    p1 = p_raw[perm].cpu().numpy()
    a1 = out["dispersion"][perm].cpu().numpy()
    dk2, dfm2 = dk[:, perm].contiguous(), dfm[:, perm].contiguous()
    del dk, dfm
    out3, sc3 = ctx.wald_test(dk2, dfm2, group, theta=0.5, want=["pvalue", "dispersion"])
    assert np.allclose(sc3["trendCoef"], sc["trendCoef"], rtol=1e-11) and np.array_equal(sc3["sizeFactors"], sc["sizeFactors"])
    p2 = out3["pvalue"].cpu().numpy()
    okp = ~np.isnan(p1)
    r = rel(p2[okp], p1[okp])
    ra = rel(out3["dispersion"].cpu().numpy()[okp], a1[okp])
    print("C5 permutation, free fits: dispersion max rel", ra.max(), "within 1e-9", np.mean(ra < 1e-9), "| pvalue max rel", r.max(), "within 1e-9", np.mean(r < 1e-9),
          "within 1e-6", np.mean(r < 1e-6), "| dispersion within 1e-8", np.mean(ra < 1e-8))
    # 20 M-term sums in another order, through ~20 IRLS passes, move the trend in its 11th-13th digit, and DESeq2's line search
    # answers that shift with a stopping decision flipped in ~1 row of 10 000 (tests/test_gpu_sharded.py prints the same for the
    # trend kernel run with another workgroup count): reported above.  What IS a property of the implementation: with the two
    # global scalars the order can touch pinned to the first fit's, permuting the rows permutes the results BIT FOR BIT — exact
    # medians, correctly rounded column sums, everything else row by row.
    assert np.array_equal(np.isnan(p1), np.isnan(p2)) and np.mean(ra < 1e-8) > 0.999, \
        "free fits of a permuted matrix: tolerance kept — 20 M-term trend sums in another order (see the comment above); pinned scalars: bit for bit, below"
    from chicdiff_amd import hip
    pin = hip.default_opts(trendCoef=sc["trendCoef"], dispPriorVar=sc["dispPriorVar"])
    out4, sc4 = ctx.wald_test(dk2, dfm2, group, theta=0.5, want=["pvalue", "dispersion"], opts=pin)
    assert np.array_equal(out4["pvalue"].cpu().numpy(), p1, equal_nan=True) and np.array_equal(out4["dispersion"].cpu().numpy(), a1, equal_nan=True)
    assert sc4["varLogDispEsts"] == sc["varLogDispEsts"]


def test_soak_changing_shapes_no_leak_and_run_to_run_identity(ctx):
    """Many fits of changing shape on one context (workspace regrowth both ways, theta-grid child contexts, host path):
    identical results whenever a shape comes back, and no device memory lost along the way."""
    import torch
    shapes = [(5000, 4), (120000, 8), (300, 6), (64, 16), (40000, 3), (1, 8), (30000, 33)]
    ref, free_mid = {}, None
    for it in range(42):
        n, S = shapes[it % len(shapes)]
        d = synth.make(n, S)
        g = np.zeros(S, np.int32) if S == 3 else d["group"]
        dk, dn = ctx.to_device(d["counts"], np.int32), ctx.to_device(d["nf"], np.float64)
        out, sc = ctx.nbglm_fit(dk, dn, g, want=["pvalue", "dispersion"])
        p = out["dispersion"].cpu().numpy()
        if (n, S) in ref:
            assert np.array_equal(ref[(n, S)], p, equal_nan=True), (n, S)
        ref[(n, S)] = p
        if it % 7 == 1:
            keep = d["counts"].sum(1) > 0
            fm = d["nf"] * (d["mu"][:, None] / S)
            dev = ctx.theta_grid(ctx.to_device(d["counts"][keep], np.int32), ctx.to_device(fm[keep], np.float64), np.ones(S), [0.0, 0.4, 1.0])
            assert np.all(np.isfinite(dev))
            r, _ = ctx.nbglm_fit_host(d["counts"], d["nf"], g, want=["dispersion"])
            assert np.array_equal(r["dispersion"], p, equal_nan=True)
        del dk, dn, out
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        if it == 20:
            free_mid = free
    assert free_mid - free < 64 * 2 ** 20, (free_mid, free)


@pytest.mark.parametrize("with_chinput", [True, False])
def test_chicdiffPipeline_mirror_from_peak_matrix_to_weighted_padj(ctx, oracle, golden, tmp_path, with_chinput):
    """chicdiffPipeline() (chicdiff.R:301-347) through the host mirrors with the device path behind every stage, given the
    reference's own settings list UNCHANGED except for its file entries (tests/golden/chr19_settings.json: device = "png",
    theta = NULL -> theta grid, 2 v 2): peak matrix -> region universe -> control universe -> getFullRegionData (both
    universes from one read of each Chicago table / chinput file; `with_chinput = False`: the Reduce(merge) branch of
    chicdiff.R:774-807) -> DESeq2Wrap x 2 -> IHWcorrection -> the 25-column result table.  Every block is compared with
    the oracle run on the same inputs."""
    import pandas as pd
    import torch
    from chicdiff_amd import pipeline, settings as st
    from pipeline_inputs import make_experiment, quantile_ihw, read_chicago_pickle
    settings, truth = make_experiment(tmp_path, npeaks=2500, with_chinput=with_chinput)
    assert settings["device"] == ["png"] and "hipDevice" not in settings and settings["theta"] is None
    s = st.asChicdiffSettings(settings)
    rng = np.random.default_rng(11)
    # stage by stage first (what chicdiffPipeline calls), each against the oracle
    RU = pipeline.getRegionUniverse(settings, ctx)
    rmap = pd.read_csv(s["rmapfile"], sep=r"\s+", header=None, quotechar='"')
    id_min, nid = truth["id_min"], truth["nid"]
    chr_of = np.full(id_min + nid, -1, dtype=np.int32)
    chr_of[rmap[3].to_numpy()] = 0
    ptr_ref, rb_ref, rr_ref, ro_ref = oracle.region_universe(truth["peak_bait"], truth["peak_oe"], 5, chr_of)
    assert np.array_equal(RU["region_ptr"].cpu().numpy(), ptr_ref) and np.array_equal(RU["csr_otherEndID"].cpu().numpy(), ro_ref)
    assert np.array_equal(RU["csr_baitID"].cpu().numpy(), rb_ref)
    RUc = pipeline.getControlRegionUniverse(settings, RU, ctx, rng=rng)
    nC = RUc["region_ptr"].numel() - 1
    assert 0.9 * len(truth["peak_bait"]) <= nC <= len(truth["peak_bait"]) and (np.diff(RUc["region_ptr"].cpu().numpy()) > 0).all()
    cb, co = RUc["csr_baitID"].cpu().numpy(), RUc["csr_otherEndID"].cpu().numpy()
    assert np.isin(cb, np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chr19_design.npz"))["bait_id"]).all()
    assert (np.abs(co.astype(np.int64) - cb) >= 2).all()                       # never the bait or its neighbours (.expandAvoidBait)
    frd = pipeline.getFullRegionData(settings, RU, RUc, ctx=ctx, read_chicago=read_chicago_pickle)
    assert len(frd) == 3 and not frd[0]["is_control"] and frd[1]["is_control"]
    assert list(frd[2].columns) == ["baitID", "otherEndID", "Nav", "Bav", "score", "oeID_mid", "condition"]   # countput, chicdiff.R:754-768
    S = 4
    flags = np.zeros(id_min + nid, np.uint8)
    flags[np.unique(np.concatenate([rb_ref, cb]))] = 1
    if with_chinput:
        tabs = [oracle.count_table(b, o, N, flags) for b, o, N in truth["chinput"]]
    else:
        tabs = [oracle.count_table(x["baitID"].to_numpy(), x["otherEndID"].to_numpy(), x["N"].to_numpy(), flags) for x in truth["xs"]]
    bg = pipeline.background_tables(truth["xs"], id_min, nid)
    N_ref = {}
    for blk, (ub, uo, uptr) in zip(frd[:2], [(rb_ref, ro_ref, ptr_ref), (cb, co, RUc["region_ptr"].cpu().numpy())]):
        if with_chinput:
            Nf = np.stack([oracle.count_join(ub, uo, k, v) for k, v in tabs], 1)
        else:
            Nf = oracle.count_join_inner(ub, uo, tabs)
            plain = np.stack([oracle.count_join(ub, uo, k, v) for k, v in tabs], 1)
            assert (plain != Nf).any() and ((Nf > 0).all(1) == (Nf > 0).any(1)).all()   # the inner merge drops pairs a replicate lacks
        assert np.array_equal(blk["fragN"].cpu().numpy().T, Nf) and Nf.sum() > 0
        _, _, FM = oracle.fragment_background(ub, uo, id_min, truth["midsum"], bg["sj"], bg["si"], bg["tblb"], bg["tlb"], bg["T"], bg["distfun"])
        assert np.allclose(blk["fragFullMean"].cpu().numpy(), FM, rtol=1e-13, equal_nan=True)
        av = oracle.region_avdist(ub, uo, uptr, id_min, truth["midsum"], np.zeros(nid, np.int32))
        assert np.array_equal(blk["avDist"].cpu().numpy(), av) and not np.isnan(av).any()
        N_ref[id(blk)] = (oracle.window_sums(Nf, FM.T, uptr), av)
    # the driver itself, same seed for the control draws
    out = pipeline.chicdiffPipeline(settings, ctx=ctx, read_chicago=read_chicago_pickle, ihw=quantile_ihw(), rng=np.random.default_rng(11))
    assert list(out.columns) == [str(c) for c in golden["__column_order__"]]    # the reference's own 25 columns, in its order
    theta = out.attrs["theta"]
    assert theta in s["theta_grid"]
    out_r = out.sort_values("regionID").reset_index(drop=True)
    n = len(truth["peak_bait"])
    assert np.array_equal(out_r["regionID"].to_numpy(), np.arange(1, n + 1)) and np.array_equal(out_r["baitID"].to_numpy(), truth["peak_bait"])
    (N_w, FM_w), av = N_ref[id(frd[0])]
    assert np.array_equal(out_r["avDist"].to_numpy(), av)
    group = np.array([0, 0, 1, 1])                                                # CD4 < Mono: alphabetical levels
    sf = oracle.size_factors(N_w)
    devs = [oracle.nbglm_fit(N_w, oracle.offsets(FM_w, sf, t), np.zeros(S, dtype=np.int32))["sumDeviance"] for t in s["theta_grid"]]
    assert s["theta_grid"][int(np.argmin(devs))] == theta
    # the table's fit = the library's fit on the block's matrices (the calls DESeq2Wrap makes; 2v2: no Cook's cutoff), bit for bit,
    # and that fit against the oracle under the same trend: every row within the bounds or refereed
    dN, dFM = ctx.window_sums(frd[0]["fragN"], frd[0]["fragFullMean"], frd[0]["region_ptr"])
    assert np.array_equal(dN.cpu().numpy().T, N_w)
    dnf = ctx.offsets(dFM, ctx.size_factors(dN), theta)
    fit, sc = ctx.nbglm_fit(dN, dnf, group, want=WANT)
    got = {k: v.cpu().numpy() for k, v in fit.items()}
    for col in ("baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue"):
        assert np.array_equal(out_r[col].to_numpy(), got[col], equal_nan=True), col
    nf_dev = dnf.cpu().numpy().T
    assert np.allclose(nf_dev, oracle.offsets(FM_w, sf, theta), rtol=1e-12, equal_nan=True)
    explain_fit(f"chicdiffPipeline mirror (with_chinput = {with_chinput})", oracle, N_w, nf_dev, group, got, sc)
    # IHW side: the control fit's covariate trains the stand-in ihw(); the application reproduces the oracle's columns
    ctl = pipeline.DESeq2Wrap(settings, RUc, frd[1], suffix="Control", theta=theta, ctx=ctx)
    df, w = quantile_ihw()(ctl["pvalue"].to_numpy(), np.abs(N_ref[id(frd[1])][1]), 0.05)
    look = pipeline.dist_lookup(df, w)
    from chicdiff_amd import post
    g_ref, w_ref, wp_ref, wpadj_ref = oracle.ihw_apply(av, out_r["pvalue"].to_numpy(), post.ihw_breaks(look["minLogDist"], look["maxLogDist"]), look["avWeights"])
    assert np.array_equal(out_r["group"].to_numpy(), np.where(g_ref == np.iinfo(np.int32).min, -1, g_ref))
    assert np.allclose(out_r["weighted_padj"].to_numpy(), wpadj_ref, rtol=1e-12, equal_nan=True)
    assert np.allclose(out_r["weight"].to_numpy(), w_ref, rtol=1e-13, equal_nan=True) and (np.diff(out["group"].to_numpy()) >= 0).all()
    assert os.path.exists(s["outprefix"] + "_results.csv") and os.path.exists(s["outprefix"] + "_countput.csv")
    print(f"pipeline mirror (chinput={with_chinput}): {n} regions, {nC} control regions, theta {theta}, "
          f"weighted padj < 0.05: {int((out['weighted_padj'] < 0.05).sum())}")


def test_region_avdist_on_device_reproduces_golden_avDist(ctx, oracle, golden, tmp_path):
    """chicdiff_hip_region_avdist_dev on the regions of the reference's own chr19 run and the reference's restriction map:
    the avDist column of the reference's result table (chicdiff.R:1965-1967), exactly, all 24 863 rows; trans rows give NA,
    fragments off the map are dropped, as the oracle has it.  Plus the header-only chinput file (an empty key table, N = 0)."""
    import torch
    from post_inputs import golden_regions_as_ru
    ru_bait, ru_oe, ptr, id_min, midsum = golden_regions_as_ru(golden)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(ctx.device)
    av = ctx.region_avdist(t(ru_bait), t(ru_oe), t(ptr), id_min, t(midsum)).cpu().numpy()
    assert np.array_equal(av, golden["avDist"])
    rng = np.random.default_rng(3)
    chr_codes = np.where(rng.random(len(midsum)) < 0.01, 1, 0).astype(np.int32)
    chr_codes[rng.integers(0, len(midsum), 40)] = -1
    av2 = ctx.region_avdist(t(ru_bait), t(ru_oe), t(ptr), id_min, t(midsum), t(chr_codes)).cpu().numpy()
    ref2 = oracle.region_avdist(ru_bait, ru_oe, ptr, id_min, midsum, chr_codes)
    assert np.array_equal(av2, ref2, equal_nan=True) and 0 < np.isnan(ref2).sum() < len(ref2)
    empty = tmp_path / "empty.chinput"
    empty.write_text("baitID\totherEndID\tN\totherEndLen\tdistSign\n")
    keys, vals, nrows = ctx.read_chinput(empty, None)
    assert nrows == 0 and keys.numel() == 0
    assert int(ctx.count_join(t(ru_bait[:1000]), t(ru_oe[:1000]), keys, vals).abs().sum()) == 0
