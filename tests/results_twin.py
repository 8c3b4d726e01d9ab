"""Row a9 of SURVEY.md §8: what DESeq2's ``results()`` does after the Wald test, as Chicdiff calls
it with no arguments (chicdiff.R:1721/1730/1739):

  * Cook's-distance outlier flagging: p <- NA where maxCooks > qf(.99, p, m - p), unless (two-level
    single-factor design) at least 3 counts in the row exceed the count of the max-Cook's sample;
    only when some group has >= 3 replicates;
  * independent filtering on baseMean (theta = seq(mean(filter == 0), .95, length 50), BH at
    alpha = 0.1 per quantile cutoff, ``lowess(numRej ~ theta, f = 1/5)``, first theta whose
    rejections exceed max(fit) - RMS residual);
  * ``padj`` = BH over the survivors, NA for the filtered rows.

TEST INFRASTRUCTURE (a numpy twin the device entry points chicdiff_hip_cooks_filter_dev /
chicdiff_hip_independent_filtering_dev are checked against), not product code: O(n log n) once + O(n) per
quantile; pinned by the reference's golden table (tests/test_results_postprocessing.py: 24 863 real
(baseMean, pvalue, padj) triples).
"""
from __future__ import annotations

import numpy as np


def lowess(x, y, f=2.0 / 3.0, nsteps=3, delta=None):
    """Cleveland's LOWESS as in R ``stats::lowess`` (clowess.c): x ascending; returns fitted y."""
    x = np.asarray(x, float)
    y = np.asarray(y, float)
    n = len(x)
    if delta is None:
        delta = 0.01 * (x[-1] - x[0])
    ys = np.zeros(n)
    if n < 2:
        ys[:] = y
        return ys
    ns = max(2, min(n, int(f * n + 1e-7)))
    rw = np.ones(n)
    res = np.zeros(n)

    def lowest(xs, nleft, nright, userw):
        rng = x[-1] - x[0]
        h = max(xs - x[nleft], x[nright] - xs)
        h9, h1 = 0.999 * h, 0.001 * h
        w = np.zeros(n)
        a = 0.0
        j = nleft
        nrt = nright
        while j < n:
            r = abs(x[j] - xs)
            if r <= h9:
                w[j] = 1.0 if r <= h1 else (1.0 - (r / h) ** 3) ** 3
                if userw:
                    w[j] *= rw[j]
                a += w[j]
            elif x[j] > xs:
                break
            j += 1
        nrt = j - 1
        if a <= 0:
            return None
        w[nleft:nrt + 1] /= a
        if h > 0:
            a = float(np.dot(w[nleft:nrt + 1], x[nleft:nrt + 1]))
            b = xs - a
            c = float(np.dot(w[nleft:nrt + 1], (x[nleft:nrt + 1] - a) ** 2))
            if np.sqrt(c) > 0.001 * rng:
                b /= c
                w[nleft:nrt + 1] *= (b * (x[nleft:nrt + 1] - a) + 1.0)
        return float(np.dot(w[nleft:nrt + 1], y[nleft:nrt + 1]))

    for it in range(nsteps + 1):
        nleft, nright = 0, ns - 1
        last = -1
        i = 0
        while True:
            if nright < n - 1:
                d1 = x[i] - x[nleft]
                d2 = x[nright + 1] - x[i]
                if d1 > d2:
                    nleft += 1
                    nright += 1
                    continue
            v = lowest(x[i], nleft, nright, it > 0)
            ys[i] = y[i] if v is None else v
            if last < i - 1:
                denom = x[i] - x[last]
                for j in range(last + 1, i):
                    alpha = (x[j] - x[last]) / denom
                    ys[j] = alpha * ys[i] + (1.0 - alpha) * ys[last]
            last = i
            cut = x[last] + delta
            i = last + 1
            while i < n:
                if x[i] > cut:
                    break
                if x[i] == x[last]:
                    ys[i] = ys[last]
                    last = i
                i += 1
            i = max(last + 1, i - 1)
            if last >= n - 1:
                break
        res = y - ys
        if it >= nsteps:
            break
        sc = np.sum(np.abs(res)) / n
        rw = np.abs(res)
        srt = np.sort(rw)
        m1 = n // 2
        cmad = 3.0 * (srt[m1] + srt[n - m1 - 1])
        if cmad < 1e-7 * sc:
            break
        c9, c1 = 0.999 * cmad, 0.001 * cmad
        r = np.abs(res)
        rw = np.where(r <= c1, 1.0, np.where(r <= c9, (1.0 - (r / cmad) ** 2) ** 2, 0.0))
    return ys


def bh_adjust(p):
    """p.adjust(p, "BH"); NaN (NA) entries stay NaN and do not count in n."""
    p = np.asarray(p, float)
    out = np.full(p.shape, np.nan)
    ok = ~np.isnan(p)
    q = p[ok]
    m = len(q)
    if m:
        o = np.argsort(-q, kind="stable")
        v = np.minimum.accumulate(m / np.arange(m, 0, -1) * q[o])
        r = np.empty(m)
        r[o] = np.minimum(1.0, v)
        out[ok] = r
    return out


def quantile7(x, probs):
    """R quantile(type = 7)."""
    xs = np.sort(np.asarray(x, float))
    n = len(xs)
    h = (n - 1) * np.asarray(probs, float)
    lo = np.floor(h).astype(int)
    hi = np.minimum(lo + 1, n - 1)
    f = h - lo
    return np.where((f > 0) & (xs[hi] != xs[lo]), (1.0 - f) * xs[lo] + f * xs[hi], xs[lo])   # qs[i] <- (1 - h) * qs[i] + h * x[hi[i]]


def cooks_filter(pvalue, maxCooks, cooksArgmax, counts_of_rows, group, cutoff=None):
    """p <- NA for Cook's outliers (DESeq2 results(), default cooksCutoff = qf(.99, p, m - p)).

    maxCooks / cooksArgmax come from the fit (chicdiff_nbglm_out); `counts_of_rows(idx)` returns the
    raw counts (len(idx), S) of the flagged rows (so the host never pulls the whole matrix).
    Two-level single-factor design: the p-value is kept when >= 3 counts of the row are larger than
    the count of the max-Cook's sample (the outlier is a low count)."""
    from scipy import stats

    pvalue = np.array(pvalue, float)
    group = np.asarray(group)
    m, p = len(group), (2 if group.any() else 1)
    sizes = [(group == 0).sum(), (group == 1).sum()]
    if m <= p or max(sizes) < 3:
        return pvalue, 0
    if cutoff is None:
        cutoff = stats.f.ppf(0.99, p, m - p)
    outlier = np.nan_to_num(np.asarray(maxCooks, float), nan=-np.inf) > cutoff
    if p == 2 and outlier.any():
        idx = np.nonzero(outlier)[0]
        k = np.asarray(counts_of_rows(idx))
        out_count = k[np.arange(len(idx)), np.asarray(cooksArgmax)[idx]]
        keep = (k > out_count[:, None]).sum(1) >= 3
        outlier[idx[keep]] = False
    pvalue[outlier] = np.nan
    return pvalue, int(outlier.sum())


def independent_filtering(baseMean, pvalue, alpha=0.1):
    """DESeq2 pvalueAdjustment(independentFiltering=TRUE): returns (padj, info dict)."""
    baseMean = np.asarray(baseMean, float)
    pvalue = np.asarray(pvalue, float)
    n = len(pvalue)
    lower = float(np.mean(baseMean == 0))
    upper = 0.95 if lower < 0.95 else 1.0
    theta = np.linspace(lower, upper, 50)
    cutoffs = quantile7(baseMean, theta)
    # numRej per cutoff: BH rejections at level alpha among rows with filter >= cutoff
    order = np.argsort(pvalue, kind="stable")  # NaN last
    p_sorted = pvalue[order]
    bm_sorted = baseMean[order]
    valid = ~np.isnan(p_sorted)
    numRej = np.zeros(50)
    for t, c in enumerate(cutoffs):
        use = (bm_sorted >= c) & valid
        m = int(use.sum())
        if m == 0:
            continue
        ps = p_sorted[use]
        ok = ps <= alpha * np.arange(1, m + 1) / m
        # BH rejects every hypothesis up to the largest k with p_(k) <= alpha k / m  (padj < alpha is strict
        # in DESeq2: colSums(filtPadj < alpha)); count padj < alpha exactly:
        padj_sorted = np.minimum.accumulate((m / np.arange(m, 0, -1) * ps[::-1]))[::-1]
        numRej[t] = np.sum(np.minimum(1.0, padj_sorted) < alpha)
        del ok
    fit = lowess(theta, numRej, f=1.0 / 5.0)
    if numRej.max() <= 10:
        j = 0
    else:
        residual = np.zeros(1) if np.all(numRej == 0) else numRej[numRej > 0] - fit[numRej > 0]
        thresh = fit.max() - np.sqrt(np.mean(residual ** 2))
        above = np.nonzero(numRej > thresh)[0]
        j = int(above[0]) if len(above) else 0
    use = baseMean >= cutoffs[j]
    padj = np.full(n, np.nan)
    padj[use] = bh_adjust(pvalue[use])
    return padj, {"filterThreshold": float(cutoffs[j]), "filterTheta": float(theta[j]), "index": j + 1,
                  "numRej": numRej, "theta": theta, "lowess": fit}
