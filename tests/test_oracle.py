"""Pin the CPU oracle (oracle/*.c): golden fixture relations, an independent numpy/scipy twin,
and first-principles properties (stationarity of the CR-APL and of the ridge score).

The reference has no tests and no input->output vectors for the DESeq2 boundary (SURVEY.md
§4, §8c) — "parity unpinned" — so these are what stands behind the oracle."""
import mpmath as mp
import numpy as np
import pytest
from scipy import special, stats

import np_twin
from chicdiff_amd import synth
from oracle import oracle


@pytest.fixture(scope="module")
def small():
    d = synth.make(4000, 8)
    r = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], want_mu=True)
    return d, r


# ---------------------------------------------------------------- golden fixture (reference data)
def test_fixture_pvalue_is_two_sided_normal(golden):
    p = oracle.pnorm_two_sided(golden["stat"])
    assert np.max(np.abs(p - golden["pvalue"]) / golden["pvalue"]) < 1e-14  # down to p = 2.45e-54


def test_fixture_stat_is_lfc_over_se(golden):
    assert np.array_equal(golden["log2FoldChange"] / golden["lfcSE"], golden["stat"])


def test_fixture_bh(golden):
    assert np.array_equal(oracle.bh_adjust(golden["weighted_pvalue"]), golden["weighted_padj"])
    ok = ~np.isnan(golden["padj"])
    got = oracle.bh_adjust(np.where(ok, golden["pvalue"], np.nan))
    assert np.array_equal(got[ok], golden["padj"][ok]) and ok.sum() == 24863 - 2411
    assert np.allclose(np_twin.bh(golden["weighted_pvalue"]), golden["weighted_padj"], rtol=1e-15)


def test_fixture_schema(golden):
    cols = list(golden["__column_order__"])
    assert cols[1:17] == ["baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj", "baitID", "maxOE",
                          "minOE", "regionID", "OEchr", "OEstart", "OEend", "baitchr", "baitstart", "baitend"]


# ---------------------------------------------------------------- special functions
def test_dnbinom_against_mpmath():
    mp.mp.dps = 50
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for _ in range(400):
        x, mu, size = float(rng.integers(0, 3000)), float(np.exp(rng.uniform(-2, 9))), float(np.exp(rng.uniform(-3, 19)))
        X, M, Z = mp.mpf(x), mp.mpf(mu), mp.mpf(size)
        ref = mp.loggamma(X + Z) - mp.loggamma(Z) - mp.loggamma(X + 1) + Z * mp.log(Z / (Z + M)) + X * mp.log(M / (Z + M))
        assert abs(L.oracle_dnbinom_mu_log(x, size, mu) - ref) <= 1e-10 * max(1, abs(ref))


def test_psi_functions():
    L = oracle.lib()
    xs = np.exp(np.random.default_rng(4).uniform(-11, 20, 2000))
    dg = np.array([L.oracle_digamma(float(x)) for x in xs])
    tg = np.array([L.oracle_trigamma(float(x)) for x in xs])
    assert np.max(np.abs(dg - special.digamma(xs)) / np.maximum(1, np.abs(special.digamma(xs)))) < 1e-14
    assert np.max(np.abs(tg - special.polygamma(1, xs)) / special.polygamma(1, xs)) < 1e-14


# ---------------------------------------------------------------- a5 size factors
def test_size_factors_match_twin():
    d = synth.make(5000, 8)
    assert np.allclose(oracle.size_factors(d["counts"]), np_twin.size_factors(d["counts"]), rtol=1e-14)
    k = d["counts"][:7].copy()  # odd and even survivor counts, zeros present
    assert np.allclose(oracle.size_factors(k), np_twin.size_factors(k), rtol=1e-14)


# ---------------------------------------------------------------- a6 dispersion objective
def _apl_mp(a, y, mu, g, prior):
    al = mp.exp(a)
    r = 1 / al
    ll = sum(mp.loggamma(mp.mpf(float(yy)) + r) - mp.loggamma(r) - yy * mp.log(mp.mpf(float(m)) + r)
             - r * mp.log(1 + mp.mpf(float(m)) * al) for yy, m in zip(y, mu))
    w = [1 / (1 / mp.mpf(float(m)) + al) for m in mu]
    wB = sum(x for x, gg in zip(w, g) if gg)
    wA = sum(x for x, gg in zip(w, g) if not gg)
    cr = -mp.log(wA * wB if any(g) else wA) / 2
    pr = 0 if prior is None else -(a - prior[0]) ** 2 / (2 * prior[1])
    return ll + cr + pr


@pytest.mark.parametrize("S,use_prior", [(4, False), (8, False), (8, True), (16, True)])
def test_log_posterior_matches_twin_and_gradient(S, use_prior):
    rng = np.random.default_rng(S)
    g = synth.groups(S)
    X = np_twin.design(g)
    for _ in range(50):
        mu = np.exp(rng.uniform(-0.5, 7, S))
        y = rng.poisson(mu).astype(float)
        la = rng.uniform(-12, 2)
        prior = (rng.uniform(-4, 0), rng.uniform(0.25, 2)) if use_prior else None
        kw = dict(prior_mean=prior[0], prior_sigmasq=prior[1], use_prior=True) if prior else {}
        got = oracle.log_posterior(la, y, mu, g, **kw)
        ref = np_twin.cr_apl(la, y, mu, X, prior)
        assert abs(got - ref) <= 1e-9 * max(1, abs(ref))
        with mp.workdps(40):
            fd = float(mp.diff(lambda t: _apl_mp(t, y.tolist(), mu.tolist(), g.tolist(), prior), mp.mpf(la)))
        d = oracle.log_posterior(la, y, mu, g, deriv=True, **kw)
        assert abs(d - fd) <= 1e-8 * max(1, abs(fd))


def test_intercept_only_objective():
    rng = np.random.default_rng(11)
    S = 4
    g = np.zeros(S, dtype=np.int32)
    mu = np.exp(rng.uniform(0, 5, S))
    y = rng.poisson(mu).astype(float)
    assert abs(oracle.log_posterior(-2.0, y, mu, g) - np_twin.cr_apl(-2.0, y, mu, np.ones((S, 1)))) < 1e-9


def test_gene_estimates_are_stationary(small):
    """Rows whose line search converged in the interior sit at d APL / d log(alpha) ~ 0."""
    d, r = small
    g = d["group"]
    it = r["dispGeneIter"]
    interior = (it > 1) & (it < 100) & (r["dispGeneEst"] > 1e-6) & (r["dispGeneEst"] < 9) & (r["allZero"] == 0)
    idx = np.nonzero(interior)[0][:300]
    grads = []
    for i in idx:
        y = d["counts"][i].astype(float)
        q = y / d["nf"][i]
        gm = np.where(g == 1, q[g == 1].mean(), q[g == 0].mean())
        mu = np.maximum(gm * d["nf"][i], 0.5)
        grads.append(oracle.log_posterior(np.log(r["dispGeneEst"][i]), y, mu, g, deriv=True))
    grads = np.abs(grads)
    # tolerance of the search is 1e-6 in the objective, so the gradient is small, not zero
    assert np.median(grads) < 2e-2 and np.quantile(grads, 0.99) < 0.5


# ---------------------------------------------------------------- A3 trend
def test_trend_matches_twin(small):
    d, r = small
    ok = (r["allZero"] == 0) & (r["dispGeneEst"] > 1e-6)
    coefs, it, rc = oracle.parametric_dispersion_fit(r["baseMean"][ok], r["dispGeneEst"][ok])
    ref, it_ref = np_twin.parametric_fit(r["baseMean"][ok], r["dispGeneEst"][ok])
    assert rc == 0 and it == it_ref
    assert np.allclose(coefs, ref, rtol=1e-9)
    assert np.allclose(coefs, r["trendCoef"], rtol=1e-15)
    res = np.log(r["dispGeneEst"][ok]) - np.log(coefs[0] + coefs[1] / r["baseMean"][ok])
    assert np.isclose(r["varLogDispEsts"], stats.median_abs_deviation(res, scale="normal") ** 2, rtol=1e-3)
    mad = 1.4826 * np.median(np.abs(res - np.median(res)))
    assert np.isclose(r["varLogDispEsts"], mad ** 2, rtol=1e-13)
    assert np.isclose(r["dispPriorVar"], max(mad ** 2 - special.polygamma(1, 3.0), 0.25), rtol=1e-13)


# ---------------------------------------------------------------- A4 MAP
def test_map_rules(small):
    d, r = small
    nz = r["allZero"] == 0
    out = r["dispOutlier"] == 1
    assert np.array_equal(r["dispersion"][nz & out], r["dispGeneEst"][nz & out])
    assert np.array_equal(r["dispersion"][nz & ~out], r["dispMAP"][nz & ~out])
    thr = np.log(r["dispFit"]) + 2 * np.sqrt(r["varLogDispEsts"])
    assert np.array_equal(out[nz], (np.log(r["dispGeneEst"]) > thr)[nz])
    assert np.all(np.isnan(r["dispersion"][~nz])) and np.all(np.isnan(r["pvalue"][~nz]))
    # the MAP estimate lies between the gene-wise estimate and the trend (shrinkage), up to search tolerance
    lo = np.minimum(r["dispGeneEst"], r["dispFit"]) * 0.98
    hi = np.maximum(r["dispGeneEst"], r["dispFit"]) * 1.02
    conv = nz & (r["dispIter"] < 100) & (r["dispGeneEst"] > 1e-7)
    assert np.mean((r["dispMAP"] >= lo)[conv] & (r["dispMAP"] <= hi)[conv]) > 0.995


# ---------------------------------------------------------------- A5 Wald
def test_wald_fit_properties(small):
    d, r = small
    g = d["group"]
    X = np_twin.design(g)
    lam = 1e-6 / np_twin.LN2 ** 2
    ok = np.nonzero((r["allZero"] == 0) & (r["betaConv"] == 1))[0][:400]
    for i in ok:
        y = d["counts"][i].astype(float)
        nf = d["nf"][i]
        a = r["dispersion"][i]
        b = np.array([r["beta0"][i], r["beta1"][i]]) * np_twin.LN2
        mu = nf * np.exp(X @ b)
        if mu.min() > 0.5:  # no minmu floor active: exact ridge-penalised MLE
            sc = np_twin.wald_score(b, y, nf, X, a, lam)
            scale = np.abs(X.T @ (y / (1 + a * mu))) + 1
            assert np.all(np.abs(sc) < 2e-4 * scale)  # IRLS stops on a 1e-8 relative deviance change
        cov = np_twin.wald_cov(b, nf, X, a, lam)
        assert np.isclose(r["lfcSE"][i], np.sqrt(cov[1, 1]) / np_twin.LN2, rtol=1e-9)
        assert np.isclose(r["deviance"][i], -2 * np_twin.nb_loglik(y, mu, a), rtol=1e-6, atol=1e-6)
        assert np.allclose(r["mu"][i], mu, rtol=1e-12)
    nz = r["allZero"] == 0
    assert np.allclose(r["stat"][nz], (r["log2FoldChange"] / r["lfcSE"])[nz], rtol=0, atol=0)
    assert np.allclose(r["pvalue"][nz], 2 * stats.norm.sf(np.abs(r["stat"][nz])), rtol=1e-12)


def test_constant_offsets_closed_form():
    """With nf == 1 and no floor, beta1 is the log2 ratio of group means (ridge is ~2e-6)."""
    rng = np.random.default_rng(5)
    n, S = 500, 8
    k = rng.negative_binomial(5, 5 / (5 + 200.0), (n, S)).astype(np.int32)
    g = synth.groups(S)
    r = oracle.nbglm_fit(k, np.ones((n, S)), g)
    ref = np.log2(k[:, g == 1].mean(1) / k[:, g == 0].mean(1))
    assert np.allclose(r["log2FoldChange"], ref, atol=1e-5)


def test_intercept_only_fit():
    d = synth.make(1500, 4)
    r = oracle.nbglm_fit(d["counts"], d["nf"], np.zeros(4, dtype=np.int32), dispPriorVar=0.7)
    nz = r["allZero"] == 0
    q = d["counts"] / d["nf"]
    assert np.allclose(r["beta0"][nz], np.log2(q.mean(1))[nz], rtol=1e-14)
    assert np.all(np.isnan(r["beta1"]))
    i = np.nonzero(nz)[0][0]
    mu = d["nf"][i] * 2 ** r["beta0"][i]
    assert np.isclose(r["deviance"][i], -2 * np_twin.nb_loglik(d["counts"][i], mu, r["dispersion"][i]), rtol=1e-7)
    assert np.isclose(np.nansum(r["deviance"]), np.nansum(r["deviance"][nz]))


# ---------------------------------------------------------------- a2 / a4 / a1
def test_window_sums_and_offsets():
    d = synth.make(300, 4, fragments=11)
    N, FM = oracle.window_sums(d["fragN"], d["fragFullMean"], d["region_ptr"])
    assert np.array_equal(N, d["counts"])  # multinomial split sums back to the region counts
    ref = d["fragFullMean"].reshape(300, 11, 4).sum(axis=1)
    assert np.allclose(FM, ref, rtol=1e-14, equal_nan=True) and np.isnan(FM).any()
    sf = oracle.size_factors(N)
    m3 = oracle.offsets(FM, sf)
    na = np.isnan(FM).any(axis=1)
    assert np.allclose(m3[na], sf[None, :]) and np.allclose(np.log(m3[~na]).mean(1), 0, atol=1e-13)
    for th in (0.0, 0.25, 1.0):
        sc = oracle.offsets(FM, sf, th)
        ref = m3 * (1 - th) + sf[None, :] * th
        ref = ref / np.exp(np.log(ref).mean(1, keepdims=True))
        assert np.allclose(sc, ref, rtol=1e-14)


def test_count_join():
    rng = np.random.default_rng(9)
    bait = rng.integers(1000, 1100, 5000).astype(np.int32)
    oe = rng.integers(0, 3000, 5000).astype(np.int32)
    keys = np.unique((bait.astype(np.int64) << 32) | oe)[::2]
    vals = rng.integers(1, 50, len(keys)).astype(np.int32)
    got = oracle.count_join(bait, oe, keys, vals)
    lut = dict(zip(keys.tolist(), vals.tolist()))
    ref = np.array([lut.get((int(b) << 32) | int(e), 0) for b, e in zip(bait, oe)], dtype=np.int32)
    assert np.array_equal(got, ref) and (got == 0).any() and (got > 0).any()


# ---------------------------------------------------------------- a3 fragment background
def _a3_inputs(seed=0, S=3):
    """Real restriction-fragment geometry (first 3000 chr19 HindIII fragments of the reference's rmap) +
    synthetic Chicago tables (s_j, s_i, tblb/tlb bins, Tmean table, refitted distance function)."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    rmap = np.loadtxt(os.path.join(here, "golden", "chr19_HindIII_first3000.rmap"), dtype=str)
    start, end, ids = rmap[:, 1].astype(np.int64), rmap[:, 2].astype(np.int64), rmap[:, 3].astype(np.int64)
    baits = np.loadtxt(os.path.join(here, "golden", "chr19_baitIDs_first3000.txt"), dtype=np.int64)
    rng = np.random.default_rng(seed)
    id_min, nid = int(ids[0]), len(ids)
    assert np.array_equal(ids, np.arange(id_min, id_min + nid))
    bait = np.repeat(baits[:150], 40)
    oe = bait + rng.integers(-60, 61, len(bait))
    keep = (np.abs(oe - bait) > 1) & (oe >= id_min) & (oe < id_min + nid)
    bait, oe = bait[keep].astype(np.int32), oe[keep].astype(np.int32)
    sj = np.full((S, nid), np.nan)
    sj[:, baits - id_min] = np.exp(rng.normal(0, 0.3, (S, len(baits))))
    sj[:, baits[::17] - id_min] = np.nan                       # baits filtered out by Chicago: s_j NA
    si = np.where(rng.random((S, nid)) < 0.8, np.exp(rng.normal(0, 0.3, (S, nid))), np.nan)
    ntblb, ntlb = 5, 6
    tblb = np.full((S, nid), -1, dtype=np.int32)
    tblb[:, baits - id_min] = rng.integers(0, ntblb, (S, len(baits)))
    tblb[np.isnan(sj)] = -1
    tlb = np.where(rng.random((S, nid)) < 0.85, rng.integers(0, ntlb, (S, nid)), -1).astype(np.int32)
    T = np.exp(rng.normal(-3, 0.5, (S, ntblb, ntlb)))
    T[:, 2, 3] = np.nan                                         # a (tblb, tlb) pair never observed
    distfun = np.zeros((S, 10))
    for s in range(S):
        fit = np.array([14.0 + 0.2 * s, -1.6, 0.05, -0.003])    # log f = cubic in log d
        omin, omax = np.log(10000.0), np.log(1.5e6)
        x = np.array([omin, omax])
        beta = fit[1] + 2 * fit[2] * x + 3 * fit[3] * x ** 2    # chicdiff.R:565-566
        alpha = fit[0] + (fit[1] - beta) * x + fit[2] * x ** 2 + fit[3] * x ** 3
        distfun[s] = [*fit, alpha[0], beta[0], alpha[1], beta[1], omin, omax]
    midsum = (start + end).astype(np.int64)
    return dict(bait=bait, oe=oe, id_min=id_min, midsum=midsum, sj=sj, si=si, tblb=tblb, tlb=tlb, T=T, distfun=distfun)


def test_fragment_background_matches_numpy_restatement():
    a = _a3_inputs()
    B, Tm, F = oracle.fragment_background(**a)
    b, o = a["bait"] - a["id_min"], a["oe"] - a["id_min"]
    dist = np.rint((a["midsum"][o] - a["midsum"][b]) / 2.0)
    for s in range(a["sj"].shape[0]):
        p = a["distfun"][s]
        ld = np.log(np.abs(dist))
        e = np.where(ld > p[9], p[6] + ld * p[7], np.where(ld < p[8], p[4] + ld * p[5], p[0] + p[1] * ld + p[2] * ld ** 2 + p[3] * ld ** 3))
        si = np.where(np.isnan(a["si"][s, o]), 1.0, a["si"][s, o])
        Bref = a["sj"][s, b] * si * np.exp(e)
        assert np.allclose(B[s], Bref, rtol=1e-13, equal_nan=True) and np.isnan(Bref).any()
        tb, tl = a["tblb"][s, b], a["tlb"][s, o]
        Tref = np.full(len(b), np.nan)
        both = (tb >= 0) & (tl >= 0)
        Tref[both] = a["T"][s][tb[both], tl[both]]
        only = (tb >= 0) & (tl < 0)
        Tref[only] = np.nanmin(a["T"][s], axis=1)[tb[only]]
        assert np.array_equal(Tm[s], Tref, equal_nan=True)
        assert np.allclose(F[s], Bref + Tref, rtol=1e-13, equal_nan=True)
    # the refitted distance function is C1 at both ends of the observed range (chicdiff.R:553, 565-569)
    p = a["distfun"][0]
    for edge in (p[8], p[9]):
        inside = p[0] + p[1] * edge + p[2] * edge ** 2 + p[3] * edge ** 3
        lin = (p[4] + edge * p[5]) if edge == p[8] else (p[6] + edge * p[7])
        assert abs(inside - lin) < 1e-9


def test_ihw_application_reproduces_golden_table(golden):
    """f3: group cut, weight renormalisation, weighted p and BH of chicdiff.R:2038-2049 against the reference's
    own result table (24 863 rows)."""
    from post_inputs import ihw_tables_from_golden
    breaks, w = ihw_tables_from_golden(golden)
    group, weight, wp, wpadj = oracle.ihw_apply(golden["avDist"], golden["pvalue"], breaks, w)
    assert np.array_equal(group, golden["group"])
    assert np.allclose(weight, golden["weight"], rtol=1e-14)
    assert np.allclose(wp, golden["weighted_pvalue"], rtol=1e-14)
    assert np.allclose(wpadj, golden["weighted_padj"], rtol=1e-14)
    # cut(): right-closed intervals, NA outside; one NA weight makes mean(avWeights), hence every weight, NA
    g2, w2, _, _ = oracle.ihw_apply(np.array([np.exp(breaks[1]), 0.5, 1e9]), np.array([0.1, 0.2, 0.3]), breaks, w)
    assert g2[0] == 1 and g2[1] == np.iinfo(np.int32).min and g2[2] == len(w) and np.all(np.isnan(w2))


@pytest.mark.parametrize("s", [5, 0, 1, 12])
def test_region_universe_matches_literal_restatement(s):
    """f4: the oracle's getRegionUniverse against a statement-by-statement twin on the reference's chr19 fragment IDs."""
    from post_inputs import region_universe_case, region_universe_literal
    bait, oe, chr_of = region_universe_case()
    ptr, rb, rr, ro = oracle.region_universe(bait, oe, s, chr_of)
    lit = region_universe_literal(bait, oe, s, chr_of)
    assert ptr[-1] == len(rb) == len(lit) > 0
    # RU.DT's order: stable sort by otherEndID, then by baitID, of the region-ordered rows
    o1 = np.argsort(ro, kind="stable")
    o2 = o1[np.argsort(rb[o1], kind="stable")]
    assert np.array_equal(np.stack([rb[o2], rr[o2], ro[o2]], axis=1), lit)
    assert np.array_equal(np.diff(ptr), np.bincount(rr, minlength=len(bait) + 1)[1:])
    with pytest.raises(ValueError):
        oracle.region_universe(np.array([5], np.int32), np.array([5], np.int32), s, chr_of)


@pytest.mark.parametrize("df", [1, 2, 3])
def test_prior_variance_by_simulation_recovers_a_known_variance(df):
    """A4 with residual d.f. <= 3: residuals drawn from the model DESeq2 simulates, log(chisq_df / df) + N(0, v),
    by an independent generator (numpy): the matched prior variance must come back as v (grid step 0.008, Monte
    Carlo noise of 1e4 draws per grid point), and never below DESeq2's floor of 0.25."""
    rng = np.random.default_rng(100 + df)
    for v in (0.6, 1.0, 2.0, 4.0):
        res = np.log(rng.chisquare(df, 300000) / df) + rng.normal(0, np.sqrt(v), 300000)
        got = oracle.prior_var_mc(res, df)
        assert abs(got - v) < 0.12 + 0.03 * v, (df, v, got)
    assert oracle.prior_var_mc(np.log(rng.chisquare(df, 100000) / df), df) == 0.25


def heterogeneous_counts(n, S, sdlog, seed=7):
    """NB counts whose true dispersions scatter widely around their trend (log sd `sdlog`): the synthetic benchmark
    generator's own scatter (0.5) sits at DESeq2's 0.25 floor of the prior variance."""
    rng = np.random.default_rng(seed)
    mu = rng.lognormal(np.log(60), 1.0, n)
    alpha = (0.05 + 2.0 / mu) * rng.lognormal(0, sdlog, n)
    nf = rng.lognormal(0, 0.2, (n, S))
    nf /= np.exp(np.log(nf).mean(axis=1, keepdims=True))
    lam = rng.gamma(1.0 / alpha[:, None], alpha[:, None] * mu[:, None] * nf)
    return rng.poisson(lam).astype(np.int32), nf


def test_fit_with_three_residual_df_uses_the_simulated_prior():
    counts, nf = heterogeneous_counts(6000, 4, 1.3)
    out = oracle.nbglm_fit(counts, nf, [0, 0, 1, 1])  # 2v2: m - p = 2
    assert out["status"] & 2
    closed = max(out["varLogDispEsts"] - special.polygamma(1, 1.0), 0.25)
    # the simulation-matched value: well above the floor for this scatter (true log-variance 1.69), and not the closed form
    assert 0.4 < out["dispPriorVar"] < 3.0 and not np.isclose(out["dispPriorVar"], closed, rtol=1e-3), (out["dispPriorVar"], closed)
    out1 = oracle.nbglm_fit(counts, nf, [0, 0, 0, 0])  # ~1 with 4 samples: d.f. 3
    assert out1["status"] & 2 and 0.4 < out1["dispPriorVar"] < 3.0
    d = synth.make(4000, 4)
    assert oracle.nbglm_fit(d["counts"], d["nf"], d["group"])["dispPriorVar"] == 0.25  # the benchmark generator sits at the floor


# ---------------------------------------------------------------- control flow against the numpy twins
@pytest.mark.parametrize("S", [4, 5, 8])
def test_gene_dispersion_search_matches_procedure_twin(S):
    """The whole gene-wise procedure (start values, Armijo search with its kappa schedule, the noIncrease and
    grid rules) re-run row by row with numpy lstsq / scipy gammaln+digamma on a general design matrix: same
    iteration counts, and estimates equal to well inside the search's own 1e-6 objective tolerance."""
    d = synth.make(1500, S, start=10000 * S)
    r = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
    X = np_twin.design(d["group"])
    rows = np.nonzero(r["allZero"] == 0)[0]
    xim = np.mean(1.0 / d["nf"][rows].mean(axis=0))  # estimateDispersionsGeneEst works on object[!allZero, ]
    # every kind of row: first rows, the rows that ran long or went to the grid, and the floor rows
    special_rows = rows[(r["dispGeneIter"][rows] >= 30) | (r["dispGeneIter"][rows] == 1) | (r["dispGeneEst"][rows] < 1e-7)]
    pick = np.unique(np.concatenate([rows[:250], special_rows[:150]]))
    it_same = n_cmp = 0
    for i in pick:
        a0, est, it = np_twin.gene_dispersion(d["counts"][i], d["nf"][i], X, xim)
        assert np.isclose(a0, r["dispInit"][i], rtol=1e-12), i
        if a0 <= 1e-8:
            # a search started on the floor: at alpha = 1e-8 the gradient is (digamma differences) * 1e16, all
            # cancellation noise in double precision, so the walk differs between any two implementations; where
            # it ends is what DESeq2 keeps
            assert np.isclose(np.log(est), np.log(r["dispGeneEst"][i]), atol=5e-2), (i, est, r["dispGeneEst"][i])
            continue
        n_cmp += 1
        it_same += it == r["dispGeneIter"][i]
        if it == r["dispGeneIter"][i]:
            assert np.isclose(est, r["dispGeneEst"][i], rtol=1e-7), (i, est, r["dispGeneEst"][i])
        else:  # a comparison decided by rounding: both answers sit on the same flat top
            assert np.isclose(np.log(est), np.log(r["dispGeneEst"][i]), atol=5e-2), (i, est, r["dispGeneEst"][i])
    assert n_cmp > 80 and it_same >= n_cmp - 2, (it_same, n_cmp)


@pytest.mark.parametrize("S", [4, 7, 8])
def test_wald_irls_matches_qr_twin(S):
    """fitBeta as DESeq2 writes it (QR of the ridge-augmented weighted design) against the oracle's closed-form
    normal equations: same iteration count, coefficients, standard errors, deviance, fitted means."""
    d = synth.make(1200, S, start=7000 * S)
    r = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], want_mu=True)
    X = np_twin.design(d["group"])
    lam = np.array([1e-6, 1e-6]) / np_twin.LN2 ** 2
    rows = np.nonzero((r["allZero"] == 0) & (r["betaConv"] == 1) & (r["betaIter"] < 100))[0]
    pick = np.unique(np.concatenate([rows[:200], rows[np.argsort(-r["betaIter"][rows])[:60]]]))
    for i in pick:
        y = d["counts"][i].astype(float)
        nf = d["nf"][i]
        start = np.linalg.lstsq(X, np.log(y / nf + 0.1), rcond=None)[0]
        b, it, dev, var, mu = np_twin.fit_beta(y, nf, X, r["dispersion"][i], lam, start)
        assert it == r["betaIter"][i], (i, it, r["betaIter"][i])
        assert np.allclose(b / np_twin.LN2, [r["beta0"][i], r["beta1"][i]], rtol=1e-8, atol=1e-10), i
        assert np.allclose(np.sqrt(var) / np_twin.LN2, [r["se0"][i], r["se1"][i]], rtol=1e-9), i
        unfloored = nf * np.exp(X @ b)
        assert np.allclose(r["mu"][i], unfloored, rtol=1e-8)
        assert np.isclose(r["deviance"][i], -2 * np_twin.nb_loglik(y, unfloored, r["dispersion"][i]), rtol=1e-8, atol=1e-8)


def test_map_search_matches_procedure_twin():
    """The MAP stage: same search with the log-normal prior centred on the trend, started from the gene-wise
    estimate (or the trend when that is below a tenth of it)."""
    S = 8
    d = synth.make(1500, S, start=77000)
    r = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], want_mu=False)
    X = np_twin.design(d["group"])
    rows = np.nonzero(r["allZero"] == 0)[0]
    pick = np.unique(np.concatenate([rows[:150], rows[np.argsort(-r["dispIter"][rows])[:50]]]))
    same = 0
    for i in pick:
        y = d["counts"][i].astype(float)
        q = y / d["nf"][i]
        mu = np.maximum(X @ np.linalg.lstsq(X, q, rcond=None)[0] * d["nf"][i], 0.5)
        dg, df = r["dispGeneEst"][i], r["dispFit"][i]
        start = dg if dg > 0.1 * df else df
        prior = (np.log(df), r["dispPriorVar"])
        a, it, first, last = np_twin.fit_disp(y, mu, X, np.log(start), prior, np.log(1e-8 / 10))
        est = min(np.exp(a), 10.0)
        if not (it < 100):
            est = np_twin.fit_disp_grid(y, mu, X, S, prior)
        est = min(max(est, 1e-8), 10.0)
        same += it == r["dispIter"][i]
        assert np.isclose(np.log(est), np.log(r["dispMAP"][i]), atol=1e-6 if it == r["dispIter"][i] else 5e-2), i
    assert same >= len(pick) - 2


# ---------------------------------------------------------------- local-regression trend (DESeq2 fitType = "local")
def test_local_dispersion_fit_matches_twin():
    """oracle/locfit_oracle.c against the numpy restatement (sorted-window nearest neighbours, lstsq local quadratic):
    same tree, same bandwidths, values and slopes at the vertices, same predictions inside and outside the data range."""
    rng = np.random.default_rng(5)
    for n in (40, 700, 6000):
        means = rng.lognormal(np.log(19), 1.4, n)
        disps = (0.05 + 2 / means) * rng.lognormal(0, 0.5, n)
        v, pred = oracle.local_dispersion_fit(means, disps)
        tv, tpred = np_twin.locfit_1d(np.log(means), np.log(disps), means)
        assert len(v["x"]) == len(tv["x"]) >= 3 and np.array_equal(v["x"], tv["x"]) and np.array_equal(v["h"], tv["h"])
        assert np.allclose(v["f"], tv["f"], rtol=0, atol=1e-10) and np.allclose(v["d"], tv["d"], rtol=0, atol=1e-10)
        z = np.log(rng.lognormal(np.log(19), 2.2, 500))
        assert (z < v["x"][0]).any() and (z > v["x"][-1]).any()  # linear continuation beyond the data range
        assert np.allclose(pred(z), tpred(z), rtol=0, atol=1e-10)
        # every cell is at most 0.8 bandwidths wide (the rule that grew the tree), and the curve is C1 at the vertices
        assert np.all(np.diff(v["x"]) <= 0.8 * np.minimum(v["h"][:-1], v["h"][1:]) * (1 + 1e-12))
        eps = 1e-6
        inner = v["x"][1:-1]
        assert np.allclose((pred(inner + eps) - pred(inner - eps)) / (2 * eps), v["d"][1:-1], atol=1e-5)
    # the fit recovers a smooth trend
    means = rng.lognormal(np.log(19), 1.4, 20000)
    _, pred = oracle.local_dispersion_fit(means, (0.05 + 2 / means) * rng.lognormal(0, 0.3, 20000))
    zz = np.linspace(np.log(3), np.log(300), 9)
    assert np.max(np.abs(pred(zz) - np.log(0.05 + 2 / np.exp(zz)))) < 0.05


def test_fit_type_local_and_substitution_for_a_failed_parametric_fit():
    """fitType = 2 uses the local regression; fitType = 0 substitutes it when the parametric fit fails (as DESeq2 does),
    unless told not to; downstream quantities are the ones the parametric path would compute from that dispFit."""
    d = synth.make(3000, 8)
    r = oracle.nbglm_fit(d["counts"], d["nf"], d["group"], fitType=2)
    assert r["status"] & 16 and not r["status"] & 1 and np.all(np.isnan(r["trendCoef"]))
    nz = r["allZero"] == 0
    use = nz & (r["dispGeneEst"] > 1e-6)
    _, pred = oracle.local_dispersion_fit(r["baseMean"][use], r["dispGeneEst"][use])
    assert np.allclose(r["dispFit"][nz], np.exp(pred(np.log(r["baseMean"][nz]))), rtol=1e-12)
    res = np.log(r["dispGeneEst"][use]) - np.log(r["dispFit"][use])
    assert np.isclose(r["varLogDispEsts"], (1.4826 * np.median(np.abs(res - np.median(res)))) ** 2, rtol=1e-12)
    p = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
    # the local fit averages log dispersions, the Gamma GLM dispersions: with the spread of gene-wise estimates from eight
    # samples (sd of the log ~ 0.9) the two trends differ by about sigma^2 / 2, the same everywhere along the curve
    lr = np.log(p["dispFit"] / r["dispFit"])[nz]
    assert not p["status"] & 16 and 0.1 < np.median(lr) < 0.8 and np.std(lr) < 0.3
    # a matrix whose parametric fit fails: substituted by default, reported when the substitution is switched off
    rng = np.random.default_rng(17)
    n, S = 4000, 6
    dd = synth.make(n, S)
    counts, nf = dd["counts"].copy(), dd["nf"].copy()
    big = rng.choice(n, 300, replace=False)
    counts[big] = rng.integers(2 ** 20, 2 ** 30, size=(300, S))
    counts[big[:100], 0] = 0
    nf[big[100:200]] *= np.exp(rng.normal(0, 2.0, size=(100, S)))
    nf /= np.exp(np.log(nf).mean(axis=1, keepdims=True))
    a = oracle.nbglm_fit(counts, nf, dd["group"])
    b = oracle.nbglm_fit(counts, nf, dd["group"], noLocalSubstitute=1)
    c = oracle.nbglm_fit(counts, nf, dd["group"], fitType=2)
    assert a["status"] & 16 and not a["status"] & 1 and b["status"] & 1 and not b["status"] & 16
    assert np.array_equal(a["dispFit"], c["dispFit"], equal_nan=True) and np.array_equal(a["pvalue"], c["pvalue"], equal_nan=True)
