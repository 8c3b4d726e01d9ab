"""a9 — results() post-processing pinned by the reference's golden table: with the table's own
(baseMean, pvalue) the independent-filtering step must reproduce its padj column exactly: the
2 411 NA rows, the chosen quantile index (6 of 50) and the BH values."""
import numpy as np

import results_twin as results


def test_independent_filtering_reproduces_golden_padj(golden):
    padj, info = results.independent_filtering(golden["baseMean"], golden["pvalue"])
    ref = golden["padj"]
    assert np.array_equal(np.isnan(padj), np.isnan(ref)) and np.isnan(ref).sum() == 2411
    assert info["index"] == 6 and 4.7967 < info["filterThreshold"] <= 4.7975
    ok = ~np.isnan(ref)
    assert np.allclose(padj[ok], ref[ok], rtol=1e-14, atol=0)


def test_bh_matches_golden_weighted_padj(golden):
    assert np.allclose(results.bh_adjust(golden["weighted_pvalue"]), golden["weighted_padj"], rtol=1e-15)


def test_lowess_basic_properties():
    x = np.linspace(0, 1, 50)
    y = 3 * x + 1
    assert np.allclose(results.lowess(x, y, f=0.2), y, atol=1e-10)  # local linear fit is exact on a line
    rng = np.random.default_rng(0)
    yn = np.sin(4 * x) + rng.normal(0, 0.05, 50)
    fit = results.lowess(x, yn, f=0.2)
    assert np.sqrt(np.mean((fit - np.sin(4 * x)) ** 2)) < 0.05


def test_quantile_type7():
    x = np.array([1.0, 2.0, 4.0, 8.0])
    assert np.allclose(results.quantile7(x, [0, 0.25, 0.5, 1.0]), [1.0, 1.75, 3.0, 8.0])


def test_region_avdist_reproduces_golden_avDist(golden):
    """IHWcorrection's covariate (chicdiff.R:1965-1967): the oracle's mean(distSign) by region, fed the regions of the
    reference's own run and the reference's restriction map, reproduces the reference's avDist column exactly — all
    24 863 rows.  Pins the distSign definition of the long table (per-fragment round()ed midpoints, chicdiff.R:868-882;
    round((sum.oe - sum.bait) / 2) of :648 differs on 70 % of the rows) and the mean's arithmetic."""
    from oracle import oracle
    from post_inputs import golden_regions_as_ru
    ru_bait, ru_oe, ptr, id_min, midsum = golden_regions_as_ru(golden)
    av = oracle.region_avdist(ru_bait, ru_oe, ptr, id_min, midsum)
    assert np.array_equal(av, golden["avDist"])
    assert np.allclose(np.log(np.abs(av)), golden["avgLogDist"], rtol=4e-16, atol=0)  # out[, avgLogDist := log(abs(avDist))], chicdiff.R:2038 (the two libms differ by an ulp)
    other = np.array([np.mean(np.rint((midsum[ru_oe[a:b] - id_min] - midsum[ru_bait[a:b] - id_min]) / 2.0))
                      for a, b in zip(ptr[:2000], ptr[1:2001])])
    assert (other != golden["avDist"][:2000]).mean() > 0.5  # the Bmean-side distance formula is a different quantity
    # a trans row makes the region NA (mean() without na.rm); a fragment off the map is dropped by the merge: against a
    # literal per-region restatement
    chr_codes = np.zeros(len(midsum), dtype=np.int32)
    chr_codes[ru_oe[3] - id_min] = 1
    chr_codes[ru_oe[ptr[5]] - id_min] = -1
    av2 = oracle.region_avdist(ru_bait, ru_oe, ptr, id_min, midsum, chr_codes)
    mid = np.rint(0.5 * midsum)
    exp = np.empty(len(ptr) - 1)
    for i, (a, b) in enumerate(zip(ptr[:-1], ptr[1:])):
        cb, co = chr_codes[ru_bait[a:b] - id_min], chr_codes[ru_oe[a:b] - id_min]
        keep = (cb >= 0) & (co >= 0)
        dist = np.where(cb[keep] == co[keep], mid[ru_oe[a:b][keep] - id_min] - mid[ru_bait[a:b][keep] - id_min], np.nan)
        exp[i] = dist.mean() if keep.any() else np.nan
    assert np.array_equal(av2, exp, equal_nan=True) and 0 < np.isnan(exp).sum() < 200
    assert (exp[~np.isnan(exp)] != av[~np.isnan(exp)]).sum() > 0   # regions that lost the dropped fragment


def test_count_join_inner_is_reduce_merge():
    """No-chinput branch (chicdiff.R:774-807): Reduce(merge, tempForCounts) keeps a pair only when every replicate's
    Chicago table holds it — checked against a literal restatement with Python dicts."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    S, nru = 3, 4000
    ru_bait = rng.integers(1, 30, nru).astype(np.int32)
    ru_oe = rng.integers(1, 200, nru).astype(np.int32)
    tables, dicts = [], []
    for s in range(S):
        b, o = rng.integers(1, 30, 3000), rng.integers(1, 200, 3000)
        k = np.unique((b.astype(np.int64) << 32) | o)
        v = rng.integers(0, 50, len(k)).astype(np.int32)   # a Chicago table may hold N = 0 rows: presence, not N > 0, decides
        tables.append((k, v))
        dicts.append(dict(zip(k.tolist(), v.tolist())))
    got = oracle.count_join_inner(ru_bait, ru_oe, tables)
    merged = set(dicts[0])
    for d in dicts[1:]:
        merged &= set(d)                                    # merge(): inner join on (baitID, otherEndID)
    exp = np.zeros((nru, S), dtype=np.int32)
    for r, key in enumerate(((ru_bait.astype(np.int64) << 32) | ru_oe).tolist()):
        if key in merged:
            exp[r] = [d[key] for d in dicts]                # merge(x, temp, all.x = TRUE); N[is.na(N)] <- 0
    assert np.array_equal(got, exp) and (exp.sum(1) > 0).mean() > 0.05 and len(merged) < min(len(d) for d in dicts)
