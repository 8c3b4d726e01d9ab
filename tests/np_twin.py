"""Independent numpy/scipy restatement of the per-row objectives.

A second, vectorised derivation used only to cross-check oracle/*.c: it shares no code with
the oracle (scipy.special gammaln/digamma, scipy.stats nbinom/norm, numpy lstsq)."""
import numpy as np
from scipy import special, stats

LN2 = np.log(2.0)


def design(group):
    g = np.asarray(group)
    return np.column_stack([np.ones(len(g)), g]) if g.any() else np.ones((len(g), 1))


def cr_apl(log_alpha, y, mu, X, prior=None):
    """Cox-Reid adjusted profile log-likelihood of one row in a = log(alpha) (A2.6)."""
    a = np.exp(log_alpha)
    r = 1.0 / a
    ll = np.sum(special.gammaln(y + r) - special.gammaln(r) - y * np.log(mu + r) - r * np.log1p(mu * a))
    W = np.diag(1.0 / (1.0 / mu + a))
    cr = -0.5 * np.linalg.slogdet(X.T @ W @ X)[1]
    pr = 0.0 if prior is None else -0.5 * (log_alpha - prior[0]) ** 2 / prior[1]
    return ll + cr + pr


def nb_loglik(y, mu, alpha):
    size = 1.0 / alpha
    return stats.nbinom.logpmf(y, size, size / (size + mu)).sum()


def wald_score(beta_nat, y, nf, X, alpha, lam):
    """Gradient of the ridge-penalised NB log-likelihood in natural-log beta (must vanish at the fit)."""
    mu = nf * np.exp(X @ beta_nat)
    return X.T @ ((y - mu) / (1.0 + alpha * mu)) - lam * beta_nat


def wald_cov(beta_nat, nf, X, alpha, lam, minmu=0.5):
    mu = np.maximum(nf * np.exp(X @ beta_nat), minmu)
    w = mu / (1.0 + alpha * mu)
    A = X.T @ (w[:, None] * X)
    Mi = np.linalg.inv(A + lam * np.eye(X.shape[1]))
    return Mi @ A @ Mi


def gamma_identity_glm(x, y, start, eps=1e-8, maxit=25):
    """R glm.fit for Gamma(link='identity'), y ~ 1 + x, via weighted least squares (lstsq/QR)."""
    Xd = np.column_stack([np.ones_like(x), x])
    b = np.array(start, float)
    mu = Xd @ b
    dev = lambda m: np.sum(-2.0 * (np.log(y / m) - (y - m) / m))
    devold = dev(mu)
    conv = False
    for _ in range(maxit):
        w = 1.0 / mu
        b = np.linalg.lstsq(Xd * w[:, None], y * w, rcond=None)[0]
        mu = Xd @ b
        d = dev(mu)
        if abs(d - devold) / (abs(d) + 0.1) < eps:
            conv = True
            break
        devold = d
    return b, conv


def parametric_fit(means, disps):
    coefs = np.array([0.1, 1.0])
    it = 0
    while True:
        r = disps / (coefs[0] + coefs[1] / means)
        good = (r > 1e-4) & (r < 15)
        new, conv = gamma_identity_glm(1.0 / means[good], disps[good], coefs)
        old, coefs = coefs, new
        if not np.all(coefs > 0):
            raise RuntimeError("parametric dispersion fit failed")
        if np.sum(np.log(coefs / old) ** 2) < 1e-6 and conv:
            return coefs, it
        it += 1
        if it > 10:
            raise RuntimeError("dispersion fit did not converge")


def size_factors(counts):
    with np.errstate(divide="ignore"):
        lc = np.log(counts.astype(float))
    lg = lc.mean(axis=1)
    sf = []
    for j in range(counts.shape[1]):
        sel = np.isfinite(lg) & (counts[:, j] > 0)
        sf.append(np.exp(np.median(lc[sel, j] - lg[sel])))
    return np.array(sf)


def bh(p):
    p = np.asarray(p, float)
    out = np.full_like(p, np.nan)
    ok = ~np.isnan(p)
    q = p[ok]
    n = len(q)
    o = np.argsort(-q, kind="stable")
    v = np.minimum.accumulate(n / np.arange(n, 0, -1) * q[o])
    res = np.empty(n)
    res[o] = np.minimum(1.0, v)
    out[ok] = res
    return out


def loess_interpolate(x, y, z, span=0.2, cell=0.2):
    """R's loess(y ~ x, span, degree = 2, family = "gaussian", surface = "interpolate") predicted at z, for sorted
    distinct 1-d x: k-d tree vertices (cells cut between their two middle points until <= floor(n*span*cell) points,
    box widened by 0.5 %), local quadratic tricube fits over the q = floor(n*span + 1e-5) nearest at the vertices
    (numpy lstsq), cubic Hermite blending in between (scipy CubicHermiteSpline)."""
    from scipy.interpolate import CubicHermiteSpline
    x, y = np.asarray(x, float), np.asarray(y, float)
    n = len(x)
    q, fc = int(np.floor(n * span + 1e-5)), int(np.floor(n * (span * cell)))
    margin = 0.005 * max(x[-1] - x[0], 1e-10 * max(abs(x[0]), abs(x[-1])) + 1e-30)
    verts = [x[0] - margin, x[-1] + margin]

    def split(lo, hi):  # 1-based inclusive
        if hi - lo + 1 <= fc:
            return
        m = (lo + hi) // 2
        verts.append((x[m - 1] + x[m]) / 2)
        split(lo, m)
        split(m + 1, hi)

    split(1, n)
    v = np.sort(verts)
    val, slope = [], []
    for s in v:
        d = x - s
        idx = np.argsort(np.abs(d), kind="stable")[:q]
        h = np.abs(d[idx]).max()
        sw = np.sqrt((1 - (np.abs(d[idx]) / h) ** 3) ** 3)
        X = np.stack([np.ones(q), d[idx], d[idx] ** 2], 1)
        b = np.linalg.lstsq(X * sw[:, None], y[idx] * sw, rcond=None)[0]
        val.append(b[0])
        slope.append(b[1])
    return CubicHermiteSpline(v, val, slope)(np.asarray(z, float)), v
