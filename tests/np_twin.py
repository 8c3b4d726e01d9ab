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


# ------------------------------------------------------------------------------------------------
# Control-flow twins: the whole per-row procedures (not only their objectives), written against a
# general design matrix with numpy's QR / lstsq and scipy's special functions, so that they share
# neither code nor algebraic shortcuts (closed-form 2x2 normal equations, group means) with oracle/*.c.
def dcr_apl(log_alpha, y, mu, X, prior=None):
    """d/d log(alpha) of cr_apl: the analytic form DESeq2's line search uses (A2.6)."""
    a = np.exp(log_alpha)
    r = 1.0 / a
    ll = r * r * np.sum(special.digamma(r) + np.log1p(mu * a) - mu * a / (1.0 + mu * a)
                        - special.digamma(y + r) + y / (mu + r))
    w = 1.0 / (1.0 / mu + a)
    B = X.T @ (w[:, None] * X)
    dB = X.T @ ((-w * w)[:, None] * X)
    cr = -0.5 * np.trace(np.linalg.solve(B, dB))
    pr = 0.0 if prior is None else -(log_alpha - prior[0]) / prior[1]
    return (ll + cr) * a + pr


def fit_disp(y, mu, X, log_alpha0, prior=None, min_log_alpha=np.log(1e-9), kappa0=1.0, tol=1e-6, maxit=100):
    """Backtracking gradient ascent in log(alpha); returns (log_alpha, iterations, initial lp, last lp)."""
    eps = 1.0e-4
    a = log_alpha0
    lp = cr_apl(a, y, mu, X, prior)
    dlp = dcr_apl(a, y, mu, X, prior)
    first = lp
    kappa, it, acc = kappa0, 0, 0
    for _ in range(maxit):
        it += 1
        prop = a + kappa * dlp
        if prop < -30.0:
            kappa = (-30.0 - a) / dlp
        if prop > 10.0:
            kappa = (10.0 - a) / dlp
        if -cr_apl(a + kappa * dlp, y, mu, X, prior) <= -lp - kappa * eps * dlp * dlp:
            acc += 1
            a = a + kappa * dlp
            new = cr_apl(a, y, mu, X, prior)
            change = new - lp
            if change < tol:
                lp = new
                break
            if a < min_log_alpha:  # lp keeps its previous value on this exit
                break
            lp = new
            dlp = dcr_apl(a, y, mu, X, prior)
            kappa = min(kappa * 1.1, kappa0)
            if acc % 5 == 0:
                kappa /= 2.0
        else:
            kappa /= 2.0
    return a, it, first, lp


def fit_disp_grid(y, mu, X, S, prior=None):
    grid = np.linspace(np.log(1e-8), np.log(max(10.0, S)), 20)
    v = [cr_apl(t, y, mu, X, prior) for t in grid]
    a_hat = grid[int(np.argmax(v))]
    delta = grid[1] - grid[0]
    fine = np.linspace(a_hat - delta, a_hat + delta, 20)
    v = [cr_apl(t, y, mu, X, prior) for t in fine]
    return np.exp(fine[int(np.argmax(v))])


def gene_dispersion(counts_row, nf_row, X, xim, min_disp=1e-8, minmu=0.5, maxit=100):
    """estimateDispersionsGeneEst for one row: rough / moments start, linear-model mu, line search, the
    noIncrease and grid rules.  Returns (dispInit, dispGeneEst, iterations)."""
    S, p = X.shape
    y = counts_row.astype(float)
    q = y / nf_row
    max_disp = max(10.0, S)
    fitted = X @ np.linalg.lstsq(X, q, rcond=None)[0]       # hat-matrix fit of the normalised counts
    m = np.maximum(fitted, 1.0)
    rough = max(np.sum(((q - m) ** 2 - m) / m ** 2) / (S - p), 0.0)
    bm, bv = q.mean(), q.var(ddof=1)
    a0 = min(max(min_disp, min(rough, (bv - xim * bm) / bm ** 2)), max_disp)
    mu = np.maximum(fitted * nf_row, minmu)
    a, it, first, last = fit_disp(y, mu, X, np.log(a0), None, np.log(min_disp / 10), maxit=maxit)
    d = min(np.exp(a), max_disp)
    if last < first + abs(first) / 1e6:
        d = a0
    if not (it < maxit and it != 1) and d > min_disp * 10:
        d = fit_disp_grid(y, mu, X, S)
    return a0, min(max(d, min_disp), max_disp), it


def fit_beta(y, nf, X, alpha, lam, beta0, minmu=0.5, tol=1e-8, maxit=100, large=30.0):
    """fitBeta's IRLS the way DESeq2 writes it: QR of the ridge-augmented weighted design, natural-log beta.
    Returns (beta, iterations, deviance at the last iterate, var(beta) sandwich diagonal, mu)."""
    n, p = X.shape
    beta = np.array(beta0, float)
    mu = np.maximum(nf * np.exp(X @ beta), minmu)
    ridge = np.diag(np.full(p, lam) if np.isscalar(lam) else lam)
    dev = dev_old = 0.0
    it = 0
    size = 1.0 / alpha
    for t in range(maxit):
        it += 1
        w = mu / (1.0 + alpha * mu)
        A = np.vstack([X * np.sqrt(w)[:, None], np.sqrt(ridge)])
        Q, R = np.linalg.qr(A)
        z = np.log(mu / nf) + (y - mu) / mu
        rhs = np.concatenate([z * np.sqrt(w), np.zeros(p)])
        beta = np.linalg.solve(R, Q.T @ rhs)
        if np.any(np.abs(beta) > large):
            it = maxit
            break
        mu = np.maximum(nf * np.exp(X @ beta), minmu)
        dev = -2.0 * stats.nbinom.logpmf(y, size, size / (size + mu)).sum()
        conv = abs(dev - dev_old) / (abs(dev) + 0.1)
        if np.isnan(conv):
            it = maxit
            break
        if t > 0 and conv < tol:
            break
        dev_old = dev
    w = mu / (1.0 + alpha * mu)
    xtwx = X.T @ (w[:, None] * X)
    inv = np.linalg.inv(xtwx + ridge)
    return beta, it, dev, np.diag(inv @ xtwx @ inv), mu


# ------------------------------------------------------------------------------------------------
# locfit(y ~ x, weights = w) with locfit's defaults, one dimension — an independent restatement (sorted windows for
# the nearest neighbours, numpy lstsq for the local quadratic) of what oracle/locfit_oracle.c does
def locfit_1d(x, y, w, alpha=0.7, cut=0.8, deg=2):
    x, y, w = (np.asarray(a, float) for a in (x, y, w))
    n = len(x)
    k = int(n * alpha)

    def vertex(xv):
        dist = np.sort(np.abs(x - xv))
        h = dist[k - 1]                                   # the k-th smallest distance
        u = np.abs(x - xv) / h
        use = u < 1
        ww = w[use] * (1 - u[use] ** 3) ** 3
        dx = x[use] - xv
        B = np.column_stack([np.ones(use.sum()), dx, dx ** 2 / 2])[:, : deg + 1]
        sw = np.sqrt(ww)
        beta = np.linalg.lstsq(B * sw[:, None], y[use] * sw, rcond=None)[0]
        return h, beta[0], beta[1]

    verts = {}
    for xv in (x.min(), x.max()):
        verts[xv] = vertex(xv)

    def grow(l, r):
        hmin = min(verts[l][0], verts[r][0])
        if (r - l) / hmin > cut:
            m = (l + r) / 2
            verts[m] = vertex(m)
            grow(l, m)
            grow(m, r)

    grow(x.min(), x.max())
    vx = np.array(sorted(verts))
    vh, vf, vd = (np.array([verts[v][q] for v in vx]) for q in range(3))

    def predict(z):
        z = np.atleast_1d(np.asarray(z, float))
        j = np.clip(np.searchsorted(vx, z, side="right") - 1, 0, len(vx) - 2)
        width = vx[j + 1] - vx[j]
        t = (z - vx[j]) / width
        inside = (t >= 0) & (t <= 1)
        p1 = np.where(inside, t * t * (3 - 2 * t), (t > 1).astype(float))
        p0 = 1 - p1
        p2 = np.where(inside, t * (1 - t) ** 2, np.where(t < 0, t, 0.0))
        p3 = np.where(inside, t * t * (t - 1), np.where(t > 1, t - 1, 0.0))
        return p0 * vf[j] + p1 * vf[j + 1] + (p2 * vd[j] + p3 * vd[j + 1]) * width

    return dict(x=vx, h=vh, f=vf, d=vd), predict
