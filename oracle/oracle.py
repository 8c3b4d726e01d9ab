"""ctypes front-end of the parity oracle — TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module (it is the *checker*; never the thing shipped or measured as the
product).  ``chicdiff_amd`` never imports it.  PARITY UNPINNED at the DESeq2 boundary —
see ``oracle/README.md``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

ST_TREND_FAILED, ST_PRIORVAR_MC, ST_BETA_NONCONV, ST_ALLZERO_ROWS = 1, 2, 4, 8


def build(force: bool = False) -> str:
    """Compile oracle/*.c with gcc (``make -C oracle``).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(_HERE) if f.endswith((".c", ".h"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class Opts(C.Structure):
    _fields_ = [("minDisp", C.c_double), ("dispTol", C.c_double), ("kappa0", C.c_double),
                ("maxit", C.c_int32), ("betaMaxit", C.c_int32), ("betaTol", C.c_double),
                ("minmu", C.c_double), ("outlierSD", C.c_double), ("dispPriorVar", C.c_double),
                ("nthreads", C.c_int32), ("_pad", C.c_int32), ("trendCoef", C.c_double * 2), ("fitType", C.c_int32),
                ("noLocalSubstitute", C.c_int32), ("varLogDispEsts", C.c_double), ("dispFitIn", C.c_void_p), ("xim", C.c_double)]


_PD, _PI = C.POINTER(C.c_double), C.POINTER(C.c_int32)
_OUT_D = ["baseMean", "baseVar"]
_OUT_FIELDS = [("baseMean", _PD), ("baseVar", _PD), ("allZero", _PI), ("dispInit", _PD),
               ("dispGeneEst", _PD), ("dispGeneIter", _PI), ("dispFit", _PD), ("dispMAP", _PD),
               ("dispersion", _PD), ("dispIter", _PI), ("dispOutlier", _PI), ("beta0", _PD),
               ("beta1", _PD), ("se0", _PD), ("se1", _PD), ("stat", _PD), ("pvalue", _PD),
               ("deviance", _PD), ("betaConv", _PI), ("betaIter", _PI), ("maxCooks", _PD), ("cooksArgmax", _PI), ("mu", _PD)]


class Out(C.Structure):
    _fields_ = _OUT_FIELDS + [("trendCoef", C.c_double * 2), ("varLogDispEsts", C.c_double),
                              ("dispPriorVar", C.c_double), ("sumDeviance", C.c_double),
                              ("trendOuterIter", C.c_int32), ("status", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        for name in ("oracle_dnbinom_mu_log",):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double] * 3
        for name in ("oracle_pnorm_two_sided", "oracle_pnorm", "oracle_digamma", "oracle_trigamma",
                     "oracle_lgamma", "oracle_stirlerr"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        L.oracle_bd0.restype = C.c_double
        L.oracle_bd0.argtypes = [C.c_double, C.c_double]
        for name in ("oracle_log_posterior", "oracle_dlog_posterior"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double, _PD, _PD, _PI, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32]
        L.oracle_nbglm_default_opts.argtypes = [C.POINTER(Opts)]
        L.oracle_size_factors.argtypes = [_PI, C.c_int64, C.c_int32, _PD]
        L.oracle_nbglm_fit.argtypes = [_PI, _PD, C.c_int64, C.c_int32, _PI, C.POINTER(Opts), C.POINTER(Out)]
        L.oracle_parametric_dispersion_fit.argtypes = [_PD, _PD, C.c_int64, _PD, _PI]
        L.oracle_median.restype = C.c_double
        L.oracle_median.argtypes = [_PD, C.c_int64]
        L.oracle_window_sums.argtypes = [_PI, _PD, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.c_int64, _PI, _PD]
        L.oracle_offsets.argtypes = [_PD, _PD, C.c_int64, C.c_int32, C.c_double, _PD]
        L.oracle_count_join.argtypes = [_PI, _PI, C.c_int64, C.POINTER(C.c_int64), _PI, C.c_int64, _PI]
        L.oracle_bh_adjust.argtypes = [_PD, C.c_int64, _PD]
        P64 = C.POINTER(C.c_int64)
        L.oracle_fragment_background.argtypes = [_PI, _PI, C.c_int64, C.c_int32, C.c_int32, P64, C.c_int32, _PD, _PD, _PI, _PI,
                                                 _PD, C.c_int32, C.c_int32, _PD, _PD, _PD, _PD]
        L.oracle_arbitrate_disp.argtypes = [_PI, _PD, C.c_int64, C.c_int32, _PI, C.POINTER(C.c_int64), C.c_int64, C.c_int32, _PD, _PD, _PD,
                                            C.c_double, C.POINTER(Opts), _PD]
        L.oracle_prior_var_mc.restype = C.c_double
        L.oracle_prior_var_mc.argtypes = [_PD, C.c_int32]
        L.oracle_prior_mc_bin.argtypes = [C.c_double]
        L.oracle_prior_mc_table.argtypes = [C.c_int32, _PD]
        L.oracle_loess_interp.argtypes = [_PD, _PD, C.c_int32, C.c_double, C.c_double, _PD, C.c_int32, _PD]
        for name in ("oracle_r_runif", "oracle_r_rnorm", "oracle_r_rexp"):
            getattr(L, name).argtypes = [C.c_uint32, C.c_int64, _PD]
            getattr(L, name).restype = None
        L.oracle_r_rgamma_vec.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int64, _PD]
        L.oracle_r_rgamma_vec.restype = None
        L.oracle_r_qnorm.restype = C.c_double
        L.oracle_r_qnorm.argtypes = [C.c_double]
        L.oracle_ihw_apply.argtypes = [_PD, _PD, C.c_int64, _PD, _PD, C.c_int32, _PI, _PD, _PD, _PD]
        L.oracle_region_universe.restype = C.c_int64
        L.oracle_region_avdist.argtypes = [_PI, _PI, C.POINTER(C.c_int64), C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _PI, _PD]
        L.oracle_count_join_inner.argtypes = [_PI, _PI, C.c_int64, C.c_int32, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(_PI),
                                              C.POINTER(C.c_int64), _PI]
        L.oracle_irls_trace.argtypes = [_PI, _PD, C.c_int64, C.c_int32, _PI, C.c_int64, C.c_double, C.c_int32, _PD, _PD, _PD]
        L.oracle_region_universe.argtypes = [_PI, _PI, C.c_int64, C.c_int32, _PI, C.c_int32, P64, _PI, _PI, _PI]
        _lib = L
    return _lib


def _pd(a):
    return a.ctypes.data_as(_PD)


def _pi(a):
    return a.ctypes.data_as(_PI)


def _cm(a, dtype):
    """n x S array -> contiguous column-major (sample-major) buffer of `dtype`."""
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().oracle_nbglm_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "trendCoef":
            o.trendCoef[0], o.trendCoef[1] = float(v[0]), float(v[1])
        elif k == "dispFitIn":  # per-row fitted dispersions (kept alive on the struct)
            o._dispFitIn = np.ascontiguousarray(v, dtype=np.float64)
            o.dispFitIn = o._dispFitIn.ctypes.data
        else:
            setattr(o, k, v)
    return o


def arbitrate_disp(counts, nf, group, rows, fit, stage="gene") -> np.ndarray:
    """binary128 re-run of the dispersion line search for `rows` (the referee for rows on which the double-precision
    restatement and the GPU disagree).  fit: the dict nbglm_fit returned (dispInit; for stage "map" also dispGeneEst,
    dispFit, dispPriorVar — or any dict with those entries, e.g. with arbitrated gene-wise values substituted)."""
    k = _cm(counts, np.int32)
    f = _cm(nf, np.float64)
    n, S = k.shape
    g = np.ascontiguousarray(group, dtype=np.int32)
    r = np.ascontiguousarray(rows, dtype=np.int64)
    col = lambda name: np.ascontiguousarray(fit[name], dtype=np.float64)
    out = np.empty(len(r))
    if stage == "gene":
        di = col("dispInit")
        rc = lib().oracle_arbitrate_disp(_pi(k), _pd(f), n, S, _pi(g), r.ctypes.data_as(C.POINTER(C.c_int64)), len(r), 0, _pd(di), None, None,
                                         0.0, None, _pd(out))
    else:
        dg, df = col("dispGeneEst"), col("dispFit")
        rc = lib().oracle_arbitrate_disp(_pi(k), _pd(f), n, S, _pi(g), r.ctypes.data_as(C.POINTER(C.c_int64)), len(r), 1, None, _pd(dg), _pd(df),
                                         float(fit["dispPriorVar"]), None, _pd(out))
    if rc:
        raise RuntimeError(f"oracle_arbitrate_disp rc={rc}")
    return out


def irls_trace(counts, nf, group, row, alpha, steps=12):
    """One row's IRLS iterates without the stopping rule: (lfc on the log2 scale per step, conv_test per step)."""
    k = _cm(counts, np.int32)
    f = _cm(nf, np.float64)
    n, S = k.shape
    g = np.ascontiguousarray(group, dtype=np.int32)
    b0, b1, cv = np.empty(steps), np.empty(steps), np.empty(steps)
    rc = lib().oracle_irls_trace(_pi(k), _pd(f), n, S, _pi(g), int(row), float(alpha), steps, _pd(b0), _pd(b1), _pd(cv))
    if rc:
        raise RuntimeError(f"oracle_irls_trace rc={rc}")
    return b1 / np.log(2.0), cv


def size_factors(counts) -> np.ndarray:
    k = _cm(counts, np.int32)
    n, S = k.shape
    sf = np.empty(S)
    rc = lib().oracle_size_factors(_pi(k), n, S, _pd(sf))
    if rc:
        raise RuntimeError(f"oracle_size_factors rc={rc}")
    return sf


def nbglm_fit(counts, nf, group, want_mu: bool = False, **optkw) -> dict:
    """estimateDispersions + nbinomWaldTest.  counts/nf: (n, S); group: (S,) of 0/1 (all 0 = ~1)."""
    k = _cm(counts, np.int32)
    f = _cm(nf, np.float64)
    n, S = k.shape
    assert f.shape == (n, S)
    g = np.ascontiguousarray(group, dtype=np.int32)
    o = default_opts(**optkw)
    out = Out()
    keep = {}
    for name, typ in _OUT_FIELDS:
        if name == "mu":
            if not want_mu:
                continue
            arr = np.empty((n, S), order="F")
        else:
            arr = np.empty(n, dtype=np.float64 if typ is _PD else np.int32)
        keep[name] = arr
        setattr(out, name, arr.ctypes.data_as(typ))
    rc = lib().oracle_nbglm_fit(_pi(k), _pd(f), n, S, _pi(g), C.byref(o), C.byref(out))
    if rc:
        raise RuntimeError(f"oracle_nbglm_fit rc={rc}")
    keep.update(trendCoef=np.array(out.trendCoef[:]), varLogDispEsts=out.varLogDispEsts,
                dispPriorVar=out.dispPriorVar, sumDeviance=out.sumDeviance,
                trendOuterIter=out.trendOuterIter, status=out.status)
    keep["log2FoldChange"] = keep["beta1"]
    keep["lfcSE"] = keep["se1"]
    return keep


def log_posterior(log_alpha, y, mu, group, prior_mean=0.0, prior_sigmasq=1.0, use_prior=False, deriv=False):
    y = np.ascontiguousarray(y, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    g = np.ascontiguousarray(group, dtype=np.int32)
    p = 2 if g.any() else 1
    fn = lib().oracle_dlog_posterior if deriv else lib().oracle_log_posterior
    return fn(log_alpha, _pd(y), _pd(mu), _pi(g), len(y), p, prior_mean, prior_sigmasq, int(use_prior))


def parametric_dispersion_fit(means, disps):
    m = np.ascontiguousarray(means, dtype=np.float64)
    d = np.ascontiguousarray(disps, dtype=np.float64)
    coefs = np.empty(2)
    it = C.c_int32(0)
    rc = lib().oracle_parametric_dispersion_fit(_pd(m), _pd(d), len(m), _pd(coefs), C.byref(it))
    return coefs, it.value, rc


def window_sums(fragN, fragFullMean, region_ptr):
    """fragN/fragFullMean: (nfrag, S); region_ptr: (n+1,) -> (N (n,S) int32, FullMean (n,S))."""
    rp = np.ascontiguousarray(region_ptr, dtype=np.int64)
    n = len(rp) - 1
    fn = _cm(fragN, np.int32) if fragN is not None else None
    ff = _cm(fragFullMean, np.float64) if fragFullMean is not None else None
    nfrag, S = (fn if fn is not None else ff).shape
    N = np.zeros((n, S), dtype=np.int32, order="F")
    FM = np.zeros((n, S), dtype=np.float64, order="F")
    rc = lib().oracle_window_sums(_pi(fn) if fn is not None else None, _pd(ff) if ff is not None else None,
                                  nfrag, S, rp.ctypes.data_as(C.POINTER(C.c_int64)), n, _pi(N), _pd(FM))
    if rc:
        raise RuntimeError(f"oracle_window_sums rc={rc}")
    return N, FM


def offsets(FullMean, sizeFactors, theta=None):
    """theta=None -> normFactorsM3 (norm='fullmean'); else the theta-mixed, renormalised sc."""
    fm = _cm(FullMean, np.float64)
    n, S = fm.shape
    sf = np.ascontiguousarray(sizeFactors, dtype=np.float64)
    out = np.empty((n, S), order="F")
    lib().oracle_offsets(_pd(fm), _pd(sf), n, S, float("nan") if theta is None else float(theta), _pd(out))
    return out


def count_join(ru_bait, ru_oe, keys, vals):
    b = np.ascontiguousarray(ru_bait, dtype=np.int32)
    e = np.ascontiguousarray(ru_oe, dtype=np.int32)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    v = np.ascontiguousarray(vals, dtype=np.int32)
    out = np.empty(len(b), dtype=np.int32)
    lib().oracle_count_join(_pi(b), _pi(e), len(b), k.ctypes.data_as(C.POINTER(C.c_int64)), _pi(v), len(k), _pi(out))
    return out


def bh_adjust(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(p)
    lib().oracle_bh_adjust(_pd(p), len(p), _pd(out))
    return out


def pnorm_two_sided(z):
    f = lib().oracle_pnorm_two_sided
    return np.array([f(float(x)) for x in np.atleast_1d(z)])


def fragment_background(bait, oe, id_min, midsum, sj, si, tblb, tlb, T, distfun):
    """sj/si/tblb/tlb: (S, nid); T: (S, ntblb, ntlb); distfun: (S, 10).  Returns Bmean, Tmean, FullMean (S, nru)."""
    b = np.ascontiguousarray(bait, dtype=np.int32)
    o = np.ascontiguousarray(oe, dtype=np.int32)
    ms = np.ascontiguousarray(midsum, dtype=np.int64)
    sj = np.ascontiguousarray(sj, dtype=np.float64)
    si = np.ascontiguousarray(si, dtype=np.float64)
    tb = np.ascontiguousarray(tblb, dtype=np.int32)
    tl = np.ascontiguousarray(tlb, dtype=np.int32)
    T = np.ascontiguousarray(T, dtype=np.float64)
    df = np.ascontiguousarray(distfun, dtype=np.float64)
    S, nid = sj.shape
    nru = len(b)
    B, Tm, F = (np.empty((S, nru)) for _ in range(3))
    rc = lib().oracle_fragment_background(_pi(b), _pi(o), nru, int(id_min), nid, ms.ctypes.data_as(C.POINTER(C.c_int64)), S,
                                          _pd(sj), _pd(si), _pi(tb), _pi(tl), _pd(T), T.shape[1], T.shape[2], _pd(df),
                                          _pd(B), _pd(Tm), _pd(F))
    if rc:
        raise RuntimeError(f"oracle_fragment_background rc={rc}")
    return B, Tm, F


def ihw_apply(avDist, pvalue, breaks, avWeights):
    """chicdiff.R:2038-2049.  Returns group (1-based, INT32_MIN = NA), weight, weighted_pvalue, weighted_padj."""
    d = np.ascontiguousarray(avDist, dtype=np.float64)
    p = np.ascontiguousarray(pvalue, dtype=np.float64)
    b = np.ascontiguousarray(breaks, dtype=np.float64)
    w = np.ascontiguousarray(avWeights, dtype=np.float64)
    n = len(d)
    group = np.empty(n, dtype=np.int32)
    weight, wp, wpadj = (np.empty(n) for _ in range(3))
    rc = lib().oracle_ihw_apply(_pd(d), _pd(p), n, _pd(b), _pd(w), len(w), _pi(group), _pd(weight), _pd(wp), _pd(wpadj))
    if rc:
        raise RuntimeError(f"oracle_ihw_apply rc={rc}")
    return group, weight, wp, wpadj


def region_universe(bait, oe, RUexpand, chr_of):
    """chicdiff.R:353-426.  chr_of[0..maxfrag] (-1 = ID not on the map).  Returns region_ptr (n+1) and the RU rows
    (baitID, regionID, otherEndID) in (regionID, otherEndID) order."""
    b = np.ascontiguousarray(bait, dtype=np.int32)
    o = np.ascontiguousarray(oe, dtype=np.int32)
    c = np.ascontiguousarray(chr_of, dtype=np.int32)
    n, maxfrag = len(b), len(c) - 1
    ptr = np.empty(n + 1, dtype=np.int64)
    P64 = C.POINTER(C.c_int64)
    total = lib().oracle_region_universe(_pi(b), _pi(o), n, int(RUexpand), _pi(c), maxfrag, ptr.ctypes.data_as(P64), None, None, None)
    if total < 0:
        raise ValueError("Invalid parameters (baitID == oeID)")
    rb, rr, ro = (np.empty(total, dtype=np.int32) for _ in range(3))
    lib().oracle_region_universe(_pi(b), _pi(o), n, int(RUexpand), _pi(c), maxfrag, ptr.ctypes.data_as(P64), _pi(rb), _pi(rr), _pi(ro))
    return ptr, rb, rr, ro


def region_avdist(ru_bait, ru_oe, region_ptr, id_min, midsum, chr_codes=None):
    """IHWcorrection's avDist = mean(distSign) by regionID (chicdiff.R:1965-1967, :868-882) over CSR-ordered RU rows."""
    b = np.ascontiguousarray(ru_bait, dtype=np.int32)
    o = np.ascontiguousarray(ru_oe, dtype=np.int32)
    ptr = np.ascontiguousarray(region_ptr, dtype=np.int64)
    ms = np.ascontiguousarray(midsum, dtype=np.int64)
    ch = None if chr_codes is None else np.ascontiguousarray(chr_codes, dtype=np.int32)
    n = len(ptr) - 1
    out = np.empty(n)
    P64 = C.POINTER(C.c_int64)
    lib().oracle_region_avdist(_pi(b), _pi(o), ptr.ctypes.data_as(P64), n, int(id_min), len(ms), ms.ctypes.data_as(P64),
                               None if ch is None else _pi(ch), _pd(out))
    return out


def count_join_inner(ru_bait, ru_oe, tables):
    """No-chinput branch (chicdiff.R:774-807): ``tables`` = [(keys, vals)] per replicate; returns N (nru, S)."""
    b = np.ascontiguousarray(ru_bait, dtype=np.int32)
    o = np.ascontiguousarray(ru_oe, dtype=np.int32)
    S, nru = len(tables), len(b)
    ks = [np.ascontiguousarray(k, dtype=np.int64) for k, _ in tables]
    vs = [np.ascontiguousarray(v, dtype=np.int32) for _, v in tables]
    P64 = C.POINTER(C.c_int64)
    kp = (P64 * S)(*[k.ctypes.data_as(P64) for k in ks])
    vp = (_PI * S)(*[_pi(v) for v in vs])
    nk = (C.c_int64 * S)(*[len(k) for k in ks])
    out = np.empty((S, nru), dtype=np.int32)
    lib().oracle_count_join_inner(_pi(b), _pi(o), nru, S, kp, vp, nk, _pi(out))
    return out.T.copy()


def count_table(bait, oe, N, bait_in_RU=None):
    """f2: chicdiff.R:826-831 and :849 on the parsed chinput columns: keep rows whose bait is an RU bait
    (x[J(baits)]), key by (baitID, otherEndID).  Returns keys (baitID << 32 | otherEndID, ascending) and N."""
    b = np.asarray(bait, dtype=np.int64)
    o = np.asarray(oe, dtype=np.int64)
    v = np.asarray(N, dtype=np.int32)
    if bait_in_RU is not None:
        flags = np.asarray(bait_in_RU).astype(bool)
        keep = (b < len(flags)) & flags[np.minimum(b, len(flags) - 1)]
        b, o, v = b[keep], o[keep], v[keep]
    keys = (b << 32) | o
    order = np.argsort(keys, kind="stable")
    return keys[order], v[order]


def prior_var_mc(residuals, df):
    """Simulation-matched dispPriorVar for residual d.f. <= 3 from the log dispersion residuals."""
    r = np.asarray(residuals, dtype=np.float64)
    return lib().oracle_prior_var_mc(_pd(prior_mc_hist(r)), int(df))


def prior_mc_hist(residuals) -> np.ndarray:
    """hist(x[x > -10 & x < 10], breaks = -20:20/2)$counts with hist.default's 1e-7 fuzz on the breaks."""
    r = np.asarray(residuals, dtype=np.float64)
    r = r[(r > -10) & (r < 10)]
    fb = np.arange(-20, 21) / 2.0 + 5e-8
    fb[0] = -10.0 - 5e-8
    b = np.searchsorted(fb, r, side="left") - 1  # fb[b] < r <= fb[b+1]
    return np.bincount(np.clip(b, 0, 39), minlength=40).astype(np.float64)


def prior_mc_table(df) -> np.ndarray:
    """The 200 x 40 simulated densities DESeq2's set.seed(2) stream gives for residual d.f. `df`."""
    out = np.empty((200, 40))
    if lib().oracle_prior_mc_table(int(df), _pd(out)):
        raise ValueError("df must be 1..3")
    return out


def loess_interp(x, y, z, span=0.2, cell=0.2) -> np.ndarray:
    x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
    out = np.empty(len(z))
    if lib().oracle_loess_interp(_pd(x), _pd(y), len(x), span, cell, _pd(z), len(z), _pd(out)) < 0:
        raise ValueError("loess_interp: unsupported size")
    return out


def r_random(kind, seed, n, *args) -> np.ndarray:
    """set.seed(seed); runif(n) / rnorm(n) / rexp(n) / rgamma(n, shape, scale = s) from the restated R generators."""
    out = np.empty(n)
    if kind == "rgamma":
        lib().oracle_r_rgamma_vec(seed, float(args[0]), float(args[1]), n, _pd(out))
    else:
        getattr(lib(), {"runif": "oracle_r_runif", "rnorm": "oracle_r_rnorm", "rexp": "oracle_r_rexp"}[kind])(seed, n, _pd(out))
    return out


class LocFit(C.Structure):
    _fields_ = [("nv", C.c_int32), ("_pad", C.c_int32), ("x", C.c_double * 100), ("h", C.c_double * 100), ("f", C.c_double * 100),
                ("d", C.c_double * 100)]


def local_dispersion_fit(means, disps):
    """DESeq2 localDispersionFit (locfit defaults).  Returns (vertices dict, predict(log means) -> log dispersion)."""
    m = np.ascontiguousarray(means, dtype=np.float64)
    d = np.ascontiguousarray(disps, dtype=np.float64)
    fit = LocFit()
    L = lib()
    L.oracle_locfit_eval.restype = C.c_double
    L.oracle_locfit_eval.argtypes = [C.POINTER(LocFit), C.c_double]
    rc = L.oracle_local_dispersion_fit(_pd(m), _pd(d), C.c_int64(len(m)), C.byref(fit))
    if rc:
        raise RuntimeError(f"oracle_local_dispersion_fit rc={rc}")
    nv = fit.nv
    vert = dict(x=np.array(fit.x[:nv]), h=np.array(fit.h[:nv]), f=np.array(fit.f[:nv]), d=np.array(fit.d[:nv]))
    return vert, lambda lx: np.array([L.oracle_locfit_eval(C.byref(fit), float(v)) for v in np.atleast_1d(lx)])
