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
