"""DESeq2's estimateDispersionsPriorVar for residual d.f. <= 3 draws from R's stream after set.seed(2) (SURVEY.md
Appendix A4), so the prior variance of the reference's own 2v2 design is deterministic.  These CPU tests pin the
restated R generators (oracle/r_rng.c and, independently written, chicdiff_amd/csrc/r_rng.h) to R outputs that are
common knowledge, the hist()/loess() restatements to independent numpy twins, and the product's host-side table to
the oracle's, bit for bit.  rgamma() has no memorised R output: distribution only (said so in oracle/README.md)."""
import ctypes as C

import numpy as np
import pytest
from scipy import special, stats

from oracle import oracle
from tests import np_twin

# set.seed(s); runif(3) / rnorm(3) / rexp(3) as R prints them (7-8 significant digits)
R_KNOWN = {
    ("runif", 1): [0.2655087, 0.3721239, 0.5728534],
    ("runif", 42): [0.9148060, 0.9370754, 0.2861395],
    ("runif", 123): [0.2875775, 0.7883051, 0.4089769],
    ("runif", 2): [0.1848823, 0.7023740, 0.5733263],
    ("rnorm", 1): [-0.6264538, 0.1836433, -0.8356286],
    ("rnorm", 42): [1.37095845, -0.56469817, 0.36312841],
    ("rnorm", 123): [-0.56047565, -0.23017749, 1.55870831],
    ("rnorm", 2): [-0.89691455, 0.18484918, 1.58784533],
    ("rexp", 1): [0.7551818, 1.1816428, 0.1457067],
    ("rexp", 123): [0.84345726, 0.57661027, 1.32905487],
    ("rexp", 42): [0.1983368, 0.6608953, 0.2834910],
}


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from chicdiff_amd import hip
    return hip.load_library()


def product_random(lib, kind, seed, n, a=0.0, b=0.0):
    out = np.empty(n)
    rc = lib.chicdiff_hip_selftest_r_random(kind, C.c_uint32(seed), C.c_double(a), C.c_double(b), C.c_int64(n),
                                            out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


@pytest.mark.parametrize("kind,seed", sorted(R_KNOWN))
def test_generators_reproduce_known_r_output(lib, kind, seed):
    want = np.array(R_KNOWN[(kind, seed)])
    got = oracle.r_random(kind, seed, 3)
    assert np.all(np.abs(got - want) <= 0.6 * 10.0 ** (np.floor(np.log10(np.abs(want))) - 6)), (got, want)  # printed digits
    assert np.array_equal(product_random(lib, ["runif", "rnorm", "rexp"].index(kind), seed, 3), got)


def test_product_and_oracle_streams_are_identical(lib):
    for k, name in enumerate(["runif", "rnorm", "rexp"]):
        assert np.array_equal(product_random(lib, k, 2, 200000), oracle.r_random(name, 2, 200000)), name
    for shape in (0.5, 1.0, 1.5, 4.0, 20.0):  # GS below 1, the three parameter ranges of GD above
        assert np.array_equal(product_random(lib, 3, 2, 200000, shape, 2.0), oracle.r_random("rgamma", 2, 200000, shape, 2.0))


def test_qnorm_as241_against_scipy_and_mpmath():
    p = np.concatenate([np.random.default_rng(0).random(100000), 10.0 ** -np.linspace(1, 300, 2000),
                        1 - 10.0 ** -np.linspace(1, 15, 300)])
    q = np.array([oracle.lib().oracle_r_qnorm(x) for x in p])
    ref = special.ndtri(p)
    assert np.max(np.abs(q - ref) / np.maximum(np.abs(ref), 1e-3)) < 5e-15
    import mpmath as mp
    mp.mp.dps = 40
    for x in (1e-300, 1e-100, 1e-20, 1.3e-11, 1e-5, 0.01, 0.074, 0.3, 0.5, 0.9, 1 - 1e-9):
        qq = oracle.lib().oracle_r_qnorm(x)
        err = abs(mp.ncdf(mp.mpf(qq)) - mp.mpf(x)) / mp.npdf(mp.mpf(qq))  # distance to the exact quantile
        assert float(err) <= 2e-15 * max(abs(qq), 1.0), x


@pytest.mark.parametrize("df", [1, 2, 3, 7.5, 40])
def test_rchisq_distribution(df):
    g = oracle.r_random("rgamma", 2, 400000, df / 2, 2.0)
    assert stats.kstest(g, "chi2", args=(df,)).pvalue > 1e-3
    assert abs(g.mean() - df) < 5 * np.sqrt(2 * df / len(g))
    assert stats.kstest(oracle.r_random("rexp", 7, 400000), "expon").pvalue > 1e-3
    assert stats.kstest(oracle.r_random("rnorm", 7, 400000), "norm").pvalue > 1e-3


def test_hist_fuzz_puts_values_on_a_break_where_r_does():
    """hist.default(right = TRUE) shifts the breaks up by 1e-7 * median(diff(breaks)) = 5e-8 (the first one down):
    a value on a break, or up to 5e-8 above it, is counted in the bin BELOW the break."""
    b = oracle.lib().oracle_prior_mc_bin
    b.argtypes = [C.c_double]
    assert b(0.0) == 19 and b(4e-8) == 19 and b(6e-8) == 20 and b(-1e-300) == 19
    assert b(-9.99999) == 0 and b(-9.5) == 0 and b(-9.5 + 6e-8) == 1 and b(9.5 + 6e-8) == 39 and b(9.9999) == 39
    assert b(-10.0) == -1 and b(10.0) == -1 and b(float("nan")) == -1
    x = np.random.default_rng(1).normal(0, 3, 100000)
    x[:2000] = np.round(x[:2000] * 2) / 2 + np.random.default_rng(2).choice([0, 3e-8, 7e-8, -3e-8], 2000)
    x = x[(x > -10) & (x < 10)]
    want = oracle.prior_mc_hist(x)
    got = np.bincount([b(v) for v in x], minlength=40)
    assert np.array_equal(got, want) and want.sum() == len(x)


def test_loess_restatement_matches_numpy_twin():
    x = np.arange(200) * (8 / 199)
    x[-1] = 8.0
    z = np.arange(1000) * (8 / 999)
    z[-1] = 8.0
    rng = np.random.default_rng(3)
    for y in ((x - 3) ** 2 * 0.01 + rng.normal(0, 0.003, 200), np.sin(x) + rng.normal(0, 0.1, 200), rng.normal(0, 1, 200)):
        want, verts = np_twin.loess_interpolate(x, y, z)
        assert len(verts) == 33  # 32 leaves of 6-7 grid values
        assert np.max(np.abs(oracle.loess_interp(x, y, z) - want)) < 1e-12 * max(1.0, np.abs(y).max())
    # a quadratic is reproduced exactly by local quadratic fits + cubic Hermite blending
    assert np.max(np.abs(oracle.loess_interp(x, 0.5 * x * x - x + 2, z) - (0.5 * z * z - z + 2))) < 1e-11


@pytest.mark.parametrize("df", [1, 2, 3])
def test_product_prior_variance_equals_oracle(lib, df):
    PD = C.POINTER(C.c_double)
    dens = np.empty((200, 40))
    assert lib.chicdiff_hip_selftest_prior_mc(df, None, dens.ctypes.data_as(PD), None) == 0
    tab = oracle.prior_mc_table(df)
    assert np.array_equal(dens, tab)
    assert np.allclose(tab.sum(1) * 0.5, 1.0)
    # first grid value: variance 0, i.e. log(chisq_df / df) alone — compare with its exact bin probabilities
    edges = np.arange(-20, 21) / 2.0
    pr = np.diff(stats.chi2.cdf(df * np.exp(edges), df))
    assert np.max(np.abs(tab[0] * 0.5 - pr)) < 5 * np.sqrt(0.25 / 1e4)
    rng = np.random.default_rng(df)
    for v in (0.3, 0.6, 1.0, 2.0, 4.0):
        res = np.log(rng.chisquare(df, 200000) / df) + rng.normal(0, np.sqrt(v), 200000)
        h = oracle.prior_mc_hist(res)
        pv = C.c_double()
        assert lib.chicdiff_hip_selftest_prior_mc(df, h.ctypes.data_as(PD), None, C.byref(pv)) == 0
        assert pv.value == oracle.prior_var_mc(res, df)
        assert abs(pv.value - v) < 0.12 + 0.03 * v
