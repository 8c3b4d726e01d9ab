"""The `.Call` shim EXECUTED.

r/src/chicdiff_hip_shim.c had never run: R is absent from the authoring image and from the GPU box.  tests/r_stub/rstub.c is a
small FUNCTIONAL stand-in for the slice of R's C API the shim uses (vectors, lists with names, PROTECT as a counted stack,
external pointers with tag / protected slot, finalizers as a list the test runs in the order it likes, Rf_error as a longjmp to
the harness); tests/r_stub/build_harness.sh links the REAL shim against it and the REAL libchicdiff_hip.so.  Here every one of
the 23 registered routines is driven with the inputs the ctypes tests use: results equal to the ctypes path (chicdiff_amd/hip.py)
bit for bit, PROTECT depth unchanged after every call, device allocations balanced — also after an Rf_error forced at every
allocation site of every routine and at the shim's own argument checks (wrong length, NA count, S = 65) — and finalizers run
in registration order, reversed, contexts first and contexts last.

What this shows: what the first `R CMD SHLIB` + `.Call` would have shown (a crash, a wrong pointer or layout, an unbalanced
PROTECT, a leak on an error path).  What it does NOT show: anything about R itself — the stand-in is written from "Writing R
Extensions", nothing is garbage-collected, so a MISSING PROTECT stays invisible (only a wrong count is seen).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "harness", "librshim_harness.so")
SOURCES = [os.path.join(ROOT, "r", "src", "chicdiff_hip_shim.c"), os.path.join(ROOT, "tests", "r_stub", "rstub.c"),
           os.path.join(ROOT, "tests", "r_stub", "rstub_count.c"), os.path.join(ROOT, "tests", "r_stub", "Rinternals.h"),
           os.path.join(ROOT, "include", "chicdiff_hip.h"), os.path.join(ROOT, "chicdiff_amd", "lib", "libchicdiff_hip.so")]

INTSXP, REALSXP, STRSXP, VECSXP, EXTPTRSXP, NILSXP = 13, 14, 16, 19, 22, 0
SEXP = C.c_void_p


def build_harness():
    stale = not os.path.exists(HARNESS) or any(os.path.getmtime(s) > os.path.getmtime(HARNESS) for s in SOURCES if os.path.exists(s))
    if stale:
        r = subprocess.run(["sh", os.path.join(ROOT, "tests", "r_stub", "build_harness.sh")], capture_output=True, text=True,
                           env=dict(os.environ, CHICDIFF_HARNESS_WERROR="1"))
        if r.returncode != 0:
            pytest.fail("the R-stub harness did not build:\n" + r.stderr, pytrace=False)
    return HARNESS


class RStub:
    """ctypes face of the harness: builds arguments, makes the .Call, checks the PROTECT stack after every call."""

    def __init__(self):
        from chicdiff_amd import hip
        hip.load_library()  # (the very library the harness is linked against: loaded once, by its real path)
        self.L = L = C.CDLL(build_harness())
        for name, res, args in [
            ("rstub_load", C.c_int, []), ("rstub_routine_name", C.c_char_p, [C.c_int]), ("rstub_routine_nargs", C.c_int, [C.c_int]),
            ("rstub_last_error", C.c_char_p, []), ("rstub_fault", C.c_char_p, []), ("rstub_clear_fault", None, []),
            ("rstub_protect_depth", C.c_int, []), ("rstub_protect_high_water", C.c_int, []), ("rstub_call", SEXP, [C.c_char_p, C.c_int, C.POINTER(SEXP)]),
            ("rstub_run_finalizers", C.c_int, [C.c_int]), ("rstub_pending_finalizers", C.c_int, []), ("rstub_free_all", None, []),
            ("rstub_nil", SEXP, []), ("rstub_mk_int", SEXP, [C.c_void_p, C.c_ssize_t]), ("rstub_mk_real", SEXP, [C.c_void_p, C.c_ssize_t]),
            ("rstub_mk_string", SEXP, [C.c_char_p]), ("rstub_mk_list", SEXP, [C.c_ssize_t]), ("rstub_set_elt", None, [SEXP, C.c_ssize_t, SEXP]),
            ("rstub_set_dim", None, [SEXP, C.c_void_p, C.c_int]), ("rstub_na_real", C.c_double, []), ("rstub_typeof", C.c_int, [SEXP]),
            ("rstub_length", C.c_ssize_t, [SEXP]), ("rstub_data", C.c_void_p, [SEXP]), ("rstub_elt", SEXP, [SEXP, C.c_ssize_t]),
            ("rstub_name", C.c_char_p, [SEXP, C.c_ssize_t]), ("rstub_extptr_addr", C.c_void_p, [SEXP]),
            ("rstub_mallocs", C.c_long, []), ("rstub_frees", C.c_long, []), ("rstub_fail_malloc_in", None, [C.c_long]), ("rstub_fail_pending", C.c_long, []),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.nroutines = L.rstub_load()
        self.nil = L.rstub_nil()
        self.NA_real = L.rstub_na_real()
        self.calls = 0

    # ---- arguments ----
    def int(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32).ravel()
        return self.L.rstub_mk_int(a.ctypes.data, a.size)

    def real(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        return self.L.rstub_mk_real(a.ctypes.data, a.size)

    def mat(self, a):
        """an R matrix n x S from an (n, S) array: column-major storage"""
        a = np.asarray(a)
        flat = np.asfortranarray(a).ravel(order="F")
        return self.int(flat) if a.dtype.kind == "i" else self.real(flat)

    def string(self, s):
        return self.L.rstub_mk_string(os.fsencode(s))

    def list(self, items):
        l = self.L.rstub_mk_list(len(items))
        for i, v in enumerate(items):
            self.L.rstub_set_elt(l, i, v)
        return l

    # ---- the call ----
    def call(self, name, *args, expect_error=None):
        arr = (SEXP * max(len(args), 1))(*args)
        depth = self.L.rstub_protect_depth()
        out = self.L.rstub_call(name.encode(), len(args), arr)
        self.calls += 1
        fault = self.L.rstub_fault()
        assert fault is None, (name, fault.decode())
        assert self.L.rstub_protect_depth() == depth, (name, "PROTECT depth moved")
        if expect_error is not None:
            assert not out, f"{name}: expected an error matching {expect_error!r}, the call returned"
            msg = self.L.rstub_last_error().decode()
            assert expect_error in msg, (name, msg)
            return msg
        assert out, (name, self.L.rstub_last_error().decode())
        return out

    # ---- results ----
    def value(self, s):
        """R value -> Python: vectors as numpy copies, lists as dict (names) or list, external pointers as themselves"""
        t = self.L.rstub_typeof(s)
        n = self.L.rstub_length(s)
        if t == NILSXP:
            return None
        if t in (INTSXP, REALSXP):
            ct = C.c_int32 if t == INTSXP else C.c_double
            return np.ctypeslib.as_array(C.cast(self.L.rstub_data(s), C.POINTER(ct)), shape=(n,)).copy() if n else np.empty(0, dtype=ct)
        if t == VECSXP:
            names = [self.L.rstub_name(s, i) for i in range(n)]
            vals = [self.value(self.L.rstub_elt(s, i)) for i in range(n)]
            return {k.decode(): v for k, v in zip(names, vals)} if all(names) and n else vals
        if t == EXTPTRSXP:
            return s
        raise AssertionError(f"unexpected SEXP type {t}")

    def download(self, buf):
        return self.value(self.call("chicdiff_hip_download", buf))

    def outstanding(self):
        return self.L.rstub_mallocs() - self.L.rstub_frees()


def test_harness_loads_and_registers_every_routine_with_dynamic_lookup_off():
    """CPU: the harness links (shim + stand-in + the real library), R_init_chicdiffhip registers 23 .Call routines whose arities
    are the ones the R sources use (tests/test_r_shim.py parses those), R_useDynamicSymbols(FALSE) was called, and without a GPU
    chicdiff_hip_open ends in an Rf_error that leaves the PROTECT stack where it was."""
    import re
    r = RStub()
    assert r.nroutines == 23, r.nroutines  # (negative: dynamic lookup left on)
    src = open(SOURCES[0]).read()
    reg = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(chicdiff_hip_\w+)", \(DL_FUNC\)&\1, (\d+)\}', src)}
    got = {r.L.rstub_routine_name(i).decode(): r.L.rstub_routine_nargs(i) for i in range(r.nroutines)}
    assert got == reg
    r.call("chicdiff_hip_nope", r.nil, expect_error="not in the registered routines")
    r.call("chicdiff_hip_open", r.nil, r.nil, expect_error="Incorrect number of arguments")
    import torch
    if not torch.cuda.is_available():
        r.call("chicdiff_hip_open", r.int([0]), expect_error="chicdiff_hip")
    # arguments of the wrong kind are refused by the shim's own checks before anything touches a device
    r.call("chicdiff_hip_download", r.int([1]), expect_error="not a device vector")
    r.call("chicdiff_hip_upload", r.int([0]), r.int([1]), expect_error="not a context")
    assert r.value(r.call("chicdiff_hip_release", r.int([1]))) is None and r.value(r.call("chicdiff_hip_close", r.real([1.0]))) is None
    r.L.rstub_free_all()


@pytest.fixture(scope="module")
def rs():
    r = RStub()
    yield r
    r.L.rstub_run_finalizers(0)
    r.L.rstub_free_all()


@pytest.fixture(scope="module")
def hctx():
    from chicdiff_amd import hip
    c = hip.HipContext(0)
    yield c
    c.close()


def _t(hctx, a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(hctx.device)


@pytest.mark.gpu
def test_every_call_routine_matches_the_ctypes_path_bit_for_bit(rs, hctx, tmp_path):
    import torch
    from chicdiff_amd import synth
    r = rs
    base_out = r.outstanding()
    ctx = r.call("chicdiff_hip_open", r.int([0]))
    used = {"chicdiff_hip_open"}

    def call(name, *a, **kw):
        used.add(name)
        return r.call(name, *a, **kw)

    # ---- device vectors: upload / alloc / download / release ----------------------------------------------------------
    xi, xd = np.arange(-5, 1000, dtype=np.int32), np.linspace(-1, 1, 777)
    bi, bd = call("chicdiff_hip_upload", ctx, r.int(xi)), call("chicdiff_hip_upload", ctx, r.real(xd))
    assert np.array_equal(r.value(call("chicdiff_hip_download", bi)), xi) and np.array_equal(r.value(call("chicdiff_hip_download", bd)), xd)
    ba = call("chicdiff_hip_alloc", ctx, r.string("double"), r.real([10]))
    assert r.value(call("chicdiff_hip_download", ba)).shape == (10,)
    call("chicdiff_hip_alloc", ctx, r.string("float"), r.real([10]), expect_error="bad arguments")
    for b in (bi, bd, ba):
        call("chicdiff_hip_release", b)
    call("chicdiff_hip_download", bi, expect_error="released")
    call("chicdiff_hip_release", bi)  # twice: harmless
    assert r.outstanding() == base_out

    # ---- a2 window sums, a5 size factors, a4 offsets, a8 theta grid --------------------------------------------------------
    n, S, F = 20000, 4, 11
    d = synth.make(n, S, fragments=F)
    ptr = np.arange(0, (n + 1) * F, F, dtype=np.float64)
    fragN, fragFM = d["fragN"].astype(np.int32), d["fragFullMean"]          # (n * F, S)
    ws = r.value(call("chicdiff_hip_window_sums", ctx, r.mat(fragN), r.mat(fragFM), r.real(ptr), r.int([S])))
    N_ref, FM_ref = hctx.window_sums(_t(hctx, fragN.T, np.int32), _t(hctx, fragFM.T, np.float64), _t(hctx, ptr, np.int64))
    dN, dFM = ws["N"], ws["FullMean"]
    assert np.array_equal(r.download(dN), N_ref.cpu().numpy().ravel()) and np.array_equal(r.download(dFM), FM_ref.cpu().numpy().ravel(), equal_nan=True)
    assert np.array_equal(N_ref.cpu().numpy().T, d["counts"])
    ws2 = r.value(call("chicdiff_hip_window_sums", ctx, r.mat(fragN), r.nil, r.real(ptr), r.int([S])))
    assert ws2["FullMean"] is None and np.array_equal(r.download(ws2["N"]), N_ref.cpu().numpy().ravel())
    call("chicdiff_hip_release", ws2["N"])
    bad_ptr = ptr.copy(); bad_ptr[5] = bad_ptr[4] - 1
    call("chicdiff_hip_window_sums", ctx, r.mat(fragN), r.nil, r.real(bad_ptr), r.int([S]), expect_error="ascending")
    call("chicdiff_hip_window_sums", ctx, r.mat(fragN[:-1]), r.nil, r.real(ptr), r.int([S]), expect_error="wrong type or length")

    nS = (r.real([n]), r.int([S]))
    sf = r.value(call("chicdiff_hip_size_factors", ctx, dN, *nS))
    sf_ref = hctx.size_factors(N_ref)
    assert np.array_equal(sf, sf_ref)
    assert np.array_equal(r.value(call("chicdiff_hip_size_factors", ctx, r.mat(d["counts"].astype(np.int32)), *nS)), sf_ref)  # host matrix: uploaded into a temporary

    for theta in (0.25, r.NA_real):
        off = call("chicdiff_hip_offsets", ctx, dFM, r.real(sf), r.real([theta]), *nS)
        ref = hctx.offsets(FM_ref, sf_ref, None if np.isnan(theta) else theta)
        assert np.array_equal(r.download(off), ref.cpu().numpy().ravel(), equal_nan=True)
        call("chicdiff_hip_release", off)
    off = call("chicdiff_hip_offsets", ctx, r.nil, r.real(sf), r.real([0.5]), *nS)      # norm = "standard": the size factors per column
    assert np.array_equal(r.download(off).reshape(S, n), np.repeat(sf_ref[:, None], n, 1))
    call("chicdiff_hip_release", off)
    call("chicdiff_hip_offsets", ctx, dFM, r.real(sf[:-1]), r.real([0.5]), *nS, expect_error="one size factor per sample")

    grid = [0.0, 0.5, 1.0]
    tg = r.value(call("chicdiff_hip_theta_grid", ctx, dN, dFM, r.real(sf), r.real(grid), *nS))
    assert np.array_equal(tg, hctx.theta_grid(N_ref, FM_ref, sf_ref, grid), equal_nan=True)

    # ---- the fused call and the fit on given normalisation factors, with results() --------------------------------------------------
    n8, S8 = 30000, 8
    d8 = synth.make(n8, S8)
    k8 = d8["counts"].astype(np.int32)
    fm8 = d8["nf"] * (d8["mu"][:, None] / S8)
    g8 = d8["group"].astype(np.int32)
    from scipy import stats
    cutoff = float(stats.f.ppf(0.99, 2, S8 - 2))
    n8S = (r.real([n8]), r.int([S8]))
    wt = r.value(call("chicdiff_hip_wald_test", ctx, r.mat(k8), r.mat(fm8), r.int(g8), r.real([0.5]), r.real([r.NA_real]), r.int([0]),
                      r.real([cutoff]), r.real([0.1]), *n8S))
    dk, dfm = _t(hctx, k8.T, np.int32), _t(hctx, fm8.T, np.float64)
    want = ["baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "dispGeneEst", "dispFit", "dispersion", "deviance", "maxCooks", "betaConv",
            "dispOutlier", "cooksArgmax"]
    ref, sc = hctx.wald_test(dk, dfm, g8, theta=0.5, want=want)
    p_ref = ref["pvalue"].clone()
    n_out = hctx.cooks_filter(dk, g8, ref["maxCooks"], ref["cooksArgmax"], p_ref, cutoff)
    padj_ref, info = hctx.independent_filtering(ref["baseMean"], p_ref, 0.1)
    for k in ("baseMean", "log2FoldChange", "lfcSE", "stat", "dispGeneEst", "dispFit", "dispersion", "deviance", "maxCooks", "betaConv", "dispOutlier"):
        assert np.array_equal(wt[k], ref[k].cpu().numpy(), equal_nan=True), k
    assert np.array_equal(wt["pvalue"], p_ref.cpu().numpy(), equal_nan=True) and np.array_equal(wt["padj"], padj_ref.cpu().numpy(), equal_nan=True)
    assert wt["nCooksOutliers"][0] == n_out and wt["filterThreshold"][0] == info["filterThreshold"] and wt["filterTheta"][0] == info["filterTheta"]
    assert np.array_equal(wt["sizeFactors"], sc["sizeFactors"]) and np.array_equal(wt["trendCoef"], sc["trendCoef"])
    assert wt["dispPriorVar"][0] == sc["dispPriorVar"] and wt["varLogDispEsts"][0] == sc["varLogDispEsts"] and wt["status"][0] == sc["status"]
    assert (np.isnan(wt["sumDeviance"][0]) and np.isnan(sc["sumDeviance"])) or wt["sumDeviance"][0] == sc["sumDeviance"]
    assert list(wt) == ["baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj", "dispGeneEst", "dispFit", "dispersion", "deviance", "maxCooks",
                        "betaConv", "dispOutlier", "sizeFactors", "trendCoef", "varLogDispEsts", "dispPriorVar", "sumDeviance", "status",
                        "nCooksOutliers", "filterThreshold", "filterTheta"]

    nf8 = hctx.offsets(dfm, sc["sizeFactors"], 0.5)
    ft = r.value(call("chicdiff_hip_fit", ctx, r.mat(k8), r.mat(nf8.cpu().numpy().T), r.int(g8), r.real([r.NA_real]), r.int([0]), r.real([r.NA_real]),
                      r.real([0.1]), *n8S))
    ref2, sc2 = hctx.nbglm_fit(dk, nf8, g8, want=want)
    padj2, _ = hctx.independent_filtering(ref2["baseMean"], ref2["pvalue"], 0.1)
    for k in ("baseMean", "log2FoldChange", "pvalue", "dispersion", "maxCooks"):
        assert np.array_equal(ft[k], ref2[k].cpu().numpy(), equal_nan=True), k
    assert np.array_equal(ft["padj"], padj2.cpu().numpy(), equal_nan=True) and ft["nCooksOutliers"][0] == 0 and np.isnan(ft["sizeFactors"]).all()
    assert np.array_equal(ft["log2FoldChange"], wt["log2FoldChange"], equal_nan=True)   # the fused call is the composed calls

    # ---- a9 alone, f3 ------------------------------------------------------------------------------------------------------------------
    bm, pv = wt["baseMean"], wt["pvalue"]
    pa = r.value(call("chicdiff_hip_padj", ctx, r.real(bm), r.real(pv), r.real([0.1])))
    assert np.array_equal(pa["padj"], padj_ref.cpu().numpy(), equal_nan=True) and np.array_equal(pa["numRej"], info["numRej"])
    assert pa["filterThreshold"][0] == info["filterThreshold"]
    call("chicdiff_hip_padj", ctx, r.real(bm), r.real(pv[:-1]), r.real([0.1]), expect_error="bad arguments")
    rng = np.random.default_rng(5)
    av = np.exp(rng.uniform(8, 14, n8)) * rng.choice([-1.0, 1.0], n8)
    breaks = np.array([7.0, 9.0, 10.5, 12.0, 15.0])
    wts = np.array([1.7, 1.2, 0.8, 0.3])
    ih = r.value(call("chicdiff_hip_ihw_apply", ctx, r.real(av), r.real(pv), r.real(breaks), r.real(wts)))
    ref_ih = hctx.ihw_apply(_t(hctx, av, np.float64), _t(hctx, pv, np.float64), breaks, wts)
    for k in ("group", "weight", "weighted_pvalue", "weighted_padj"):
        assert np.array_equal(ih[k], ref_ih[k].cpu().numpy(), equal_nan=True), k
    call("chicdiff_hip_ihw_apply", ctx, r.real(av), r.real(pv), r.real(breaks), r.real(wts[:-1]), expect_error="bad arguments")

    # ---- f4 region universe, avDist -------------------------------------------------------------------------------------------------------
    from post_inputs import region_universe_case
    bait, oe, chr_of = region_universe_case(n=3000)
    ru = r.value(call("chicdiff_hip_region_universe", ctx, r.int(bait), r.int(oe), r.int([5]), r.int(chr_of)))
    ru_ref = hctx.region_universe(_t(hctx, bait, np.int32), _t(hctx, oe, np.int32), 5, _t(hctx, chr_of, np.int32))
    for k in ("baitID", "regionID", "otherEndID"):
        assert np.array_equal(ru[k], ru_ref[k].cpu().numpy()), k
    nid = len(chr_of)
    midsum = (np.arange(nid, dtype=np.float64) * 4000.0 + 1500.0)
    rptr = ru_ref["region_ptr"].cpu().numpy().astype(np.float64)
    avd = call("chicdiff_hip_region_avdist", ctx, r.int(ru["baitID"]), r.int(ru["otherEndID"]), r.real(rptr), r.int([0]), r.real(midsum), r.int(chr_of))
    avd_ref = hctx.region_avdist(ru_ref["baitID"], ru_ref["otherEndID"], ru_ref["region_ptr"], 0, _t(hctx, midsum, np.int64), _t(hctx, chr_of, np.int32))
    assert np.array_equal(r.download(avd), avd_ref.cpu().numpy(), equal_nan=True)
    avd2 = call("chicdiff_hip_region_avdist", ctx, r.int(ru["baitID"]), r.int(ru["otherEndID"]), r.real(rptr), r.int([0]), r.real(midsum), r.nil)
    avd2_ref = hctx.region_avdist(ru_ref["baitID"], ru_ref["otherEndID"], ru_ref["region_ptr"], 0, _t(hctx, midsum, np.int64), None)
    assert np.array_equal(r.download(avd2), avd2_ref.cpu().numpy(), equal_nan=True)
    call("chicdiff_hip_release", avd); call("chicdiff_hip_release", avd2)

    # ---- f2 + a1: chinput file -> key table -> count join; the branch without chinput files ----------------------------------------------------
    from test_chinput import write_chinput
    rngc = np.random.default_rng(21)
    nrows = 60000
    pairs = rngc.choice(3000 * 3000, nrows, replace=False)
    cb, co = (pairs // 3000 + 1).astype(np.int32), (pairs % 3000 + 1).astype(np.int32)
    cN = rngc.integers(1, 200, nrows).astype(np.int32)
    path = str(tmp_path / "rep1.chinput")
    write_chinput(path, cb, co, cN)
    baits = np.sort(rngc.choice(np.arange(1, 3001), 900, replace=False)).astype(np.int32)
    flags_np = np.zeros(int(baits.max()) + 1, dtype=np.uint8); flags_np[baits] = 1
    dflags = _t(hctx, flags_np, np.uint8)
    flags = call("chicdiff_hip_bait_flags", ctx, r.int(baits))
    for arg in (flags, r.int(baits), r.nil):
        tab = call("chicdiff_hip_chinput_table", ctx, r.string(path), arg)
        tv = r.value(tab)
        keys_ref, vals_ref, nrows_ref = hctx.read_chinput(path, None if arg == r.nil else dflags)
        nk = int(tv["nkeys"][0])
        assert nk == keys_ref.numel() and int(tv["nrows"][0]) == nrows_ref
        assert np.array_equal(r.download(tv["keys"]).view(np.int64)[:nk], keys_ref.cpu().numpy()) and np.array_equal(r.download(tv["vals"])[:nk], vals_ref.cpu().numpy())
        if arg == r.nil:
            continue
        qb, qo = rngc.integers(1, 3001, 20000).astype(np.int32), rngc.integers(1, 3001, 20000).astype(np.int32)
        order = np.lexsort((qo, qb)); qb, qo = qb[order], qo[order]
        cj = call("chicdiff_hip_count_join", ctx, r.int(qb), r.int(qo), tab, r.nil, r.real([0]))
        cj_ref = hctx.count_join(_t(hctx, qb, np.int32), _t(hctx, qo, np.int32), keys_ref, vals_ref)
        assert np.array_equal(r.download(cj), cj_ref.cpu().numpy()) and (cj_ref > 0).sum() > 10
        outm = call("chicdiff_hip_alloc", ctx, r.string("integer"), r.real([2 * len(qb)]))   # in place, second column of an nru x 2 matrix
        same = call("chicdiff_hip_count_join", ctx, r.int(qb), r.int(qo), tab, outm, r.real([1]))
        assert same == outm and np.array_equal(r.download(outm)[len(qb):], cj_ref.cpu().numpy())
        call("chicdiff_hip_count_join", ctx, r.int(qb), r.int(qo), tab, outm, r.real([2]), expect_error="column outside")
        call("chicdiff_hip_release", cj); call("chicdiff_hip_release", outm)
    call("chicdiff_hip_chinput_table", ctx, r.string(str(tmp_path / "absent.chinput")), r.nil, expect_error="chicdiff_hip_chinput_table")
    call("chicdiff_hip_bait_flags", ctx, r.int([3, -1]), expect_error="NA or negative bait ID")
    tabs, tabs_ref = [], []
    for s in range(3):
        sel = rngc.uniform(size=nrows) < 0.8
        t = call("chicdiff_hip_count_table", ctx, r.int(cb[sel]), r.int(co[sel]), r.int(cN[sel] + s), flags)
        kr, vr = hctx.count_table(_t(hctx, cb[sel], np.int32), _t(hctx, co[sel], np.int32), _t(hctx, cN[sel] + s, np.int32), dflags)
        tv = r.value(t)
        nk = int(tv["nkeys"][0])
        assert nk == kr.numel() and np.array_equal(r.download(tv["keys"]).view(np.int64)[:nk], kr.cpu().numpy()) and np.array_equal(r.download(tv["vals"])[:nk], vr.cpu().numpy())
        tabs.append(t); tabs_ref.append((kr, vr))
    ji = call("chicdiff_hip_count_join_inner", ctx, r.int(qb), r.int(qo), r.list(tabs))
    ji_ref = hctx.count_join_inner(_t(hctx, qb, np.int32), _t(hctx, qo, np.int32), tabs_ref)
    assert np.array_equal(r.download(ji), ji_ref.cpu().numpy().ravel()) and (ji_ref > 0).sum() > 10
    call("chicdiff_hip_count_join_inner", ctx, r.int(qb), r.int(qo), r.list([r.int([1])]), expect_error="does not come from chicdiff_hip_count_table")
    # the chinput branch for all replicates in one pass: S left joins from one read of the RU rows (same table lists)
    jm = call("chicdiff_hip_count_join_multi", ctx, r.int(qb), r.int(qo), r.list(tabs))
    jm_ref = hctx.count_join_multi(_t(hctx, qb, np.int32), _t(hctx, qo, np.int32), tabs_ref)
    assert np.array_equal(r.download(jm), jm_ref.cpu().numpy().ravel()) and (jm_ref > 0).sum() > (ji_ref > 0).sum()
    for s_, (kr, vr) in enumerate(tabs_ref):
        assert np.array_equal(jm_ref[s_].cpu().numpy(), hctx.count_join(_t(hctx, qb, np.int32), _t(hctx, qo, np.int32), kr, vr).cpu().numpy())
    call("chicdiff_hip_count_join_multi", ctx, r.int(qb), r.int(qo), r.list([]), expect_error="a list of 1..64 key tables expected")
    call("chicdiff_hip_release", jm)

    # ---- a3 fragment background ------------------------------------------------------------------------------------------------------------------
    from test_oracle import _a3_inputs
    a = _a3_inputs(seed=3, S=4)
    Sb, nidb = a["sj"].shape
    T = a["T"]                                               # (S, ntblb, ntlb), tlb fastest
    Tr = r.real(T.ravel())                                   # = R's ntlb x ntblb x S array, column-major
    r.L.rstub_set_dim(Tr, np.array([T.shape[2], T.shape[1], T.shape[0]], dtype=np.int32).ctypes.data, 3)
    na_int = np.int32(-2 ** 31)
    tblb_r = np.where(a["tblb"] < 0, na_int, a["tblb"] + 1).astype(np.int32)   # R: 1-based bin codes, NA_integer_ = absent
    tlb_r = np.where(a["tlb"] < 0, na_int, a["tlb"] + 1).astype(np.int32)
    fb = r.value(call("chicdiff_hip_fragment_background", ctx, r.int(a["bait"]), r.int(a["oe"]), r.int([a["id_min"]]), r.real(a["midsum"].astype(np.float64)),
                      r.real(a["sj"].ravel()), r.real(a["si"].ravel()), r.int(tblb_r.ravel()), r.int(tlb_r.ravel()), Tr,
                      r.real(np.asarray(a["distfun"], dtype=np.float64).ravel()), r.int([Sb])))
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(hctx.device)
    B, Tm, Fm = hctx.fragment_background(t(a["bait"]), t(a["oe"]), a["id_min"], t(a["midsum"]), t(a["sj"]), t(a["si"]), t(a["tblb"]), t(a["tlb"]), t(a["T"]), a["distfun"])
    for k, refv in (("Bmean", B), ("Tmean", Tm), ("FullMean", Fm)):
        assert np.array_equal(r.download(fb[k]), refv.cpu().numpy().ravel(), equal_nan=True), k
    bad = tlb_r.copy(); bad.flat[0] = 99
    call("chicdiff_hip_fragment_background", ctx, r.int(a["bait"]), r.int(a["oe"]), r.int([a["id_min"]]), r.real(a["midsum"].astype(np.float64)),
         r.real(a["sj"].ravel()), r.real(a["si"].ravel()), r.int(tblb_r.ravel()), r.int(bad.ravel()), Tr,
         r.real(np.asarray(a["distfun"], dtype=np.float64).ravel()), r.int([Sb]), expect_error="bin code outside")

    # ---- every registered routine has run; nothing leaks once the R side lets go ------------------------------------------------------------------
    call("chicdiff_hip_close", ctx)
    used.add("chicdiff_hip_release"); used.add("chicdiff_hip_download")
    registered = {r.L.rstub_routine_name(i).decode() for i in range(r.nroutines)}
    assert used == registered, registered - used
    call("chicdiff_hip_upload", ctx, r.int([1]), expect_error="the context has been closed")
    r.L.rstub_run_finalizers(1)
    assert r.L.rstub_pending_finalizers() == 0 and r.L.rstub_fault() is None
    print(f"{r.calls} .Call invocations, PROTECT high-water mark {r.L.rstub_protect_high_water()}, device allocations {r.L.rstub_mallocs()}, freed by the shim {r.L.rstub_frees()}")


def _fit_args(r, ctx, n=3000, S=8, counts=None, group=None):
    from chicdiff_amd import synth
    d = synth.make(n, S)
    k = d["counts"].astype(np.int32) if counts is None else counts
    fm = d["nf"] * (d["mu"][:, None] / S)
    g = d["group"].astype(np.int32) if group is None else group
    return (ctx, r.mat(k), r.mat(fm), r.int(g), r.real([0.5]), r.real([r.NA_real]), r.int([0]), r.real([r.NA_real]), r.real([0.1]), r.real([n]), r.int([S]))


@pytest.mark.gpu
@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_forced_errors_leak_nothing_whatever_the_finalizer_order(rs, hctx, order):
    """An Rf_error() is a longjmp out of the shim: whatever device memory the routine held must be owned by an external pointer
    whose finalizer releases it.  Errors are forced (i) by the shim's own checks — wrong length, an NA count, S = 65 — and (ii) at
    EVERY allocation site of four routines, by making the k-th chicdiff_hip_malloc fail for k = 0, 1, ... until the call goes
    through.  Then the finalizers run — in registration order, reversed, contexts first (a buffer's finalizer finds its context
    gone and has nothing to free: the library released the context's outstanding vectors with it), contexts last — and the
    device's free memory is back where it started."""
    import torch
    from chicdiff_amd import synth
    from post_inputs import region_universe_case
    r = rs
    n, S, F = 500, 4, 11
    d = synth.make(n, S, fragments=F)
    ptr = np.arange(0, (n + 1) * F, F, dtype=np.float64)
    bait, oe, chr_of = region_universe_case(n=300)
    k_na = synth.make(3000, 8)["counts"].astype(np.int32)
    k_na[17, 3] = -2 ** 31  # NA_integer_

    def one_round():
        ctx = r.call("chicdiff_hip_open", r.int([0]))
        m0 = r.outstanding()
        # (i) the shim's own checks and the library's verdicts
        a = _fit_args(r, ctx)
        r.call("chicdiff_hip_wald_test", *a[:9], r.real([3001]), a[10], expect_error="wrong type or length")
        r.call("chicdiff_hip_wald_test", *_fit_args(r, ctx, counts=k_na), expect_error="chicdiff_hip_wald_test")
        r.call("chicdiff_hip_wald_test", *_fit_args(r, ctx, group=np.zeros(7, dtype=np.int32)), expect_error="one 0/1 entry per sample")
        big = np.ones((10, 65), dtype=np.int32)
        r.call("chicdiff_hip_wald_test", ctx, r.mat(big), r.nil, r.int(np.arange(65) % 2), r.real([0.5]), r.real([r.NA_real]), r.int([0]),
               r.real([r.NA_real]), r.real([0.1]), r.real([10]), r.int([65]), expect_error="at most 64 samples")
        r.call("chicdiff_hip_size_factors", ctx, r.real([1.0, 2.0]), r.real([1]), r.int([2]), expect_error="wrong type or length")
        # (ii) every allocation site of four routines
        cases = {
            "chicdiff_hip_wald_test": lambda: _fit_args(r, ctx, n=500, S=4),
            "chicdiff_hip_window_sums": lambda: (ctx, r.mat(d["fragN"].astype(np.int32)), r.mat(d["fragFullMean"]), r.real(ptr), r.int([S])),
            "chicdiff_hip_region_universe": lambda: (ctx, r.int(bait), r.int(oe), r.int([5]), r.int(chr_of)),
            "chicdiff_hip_ihw_apply": lambda: (ctx, r.real(np.full(100, 1e4)), r.real(np.linspace(0.001, 1, 100)), r.real([5.0, 9.0, 12.0]), r.real([1.5, 0.5])),
        }
        sites = {}
        for name, mk in cases.items():
            kth = 0
            while True:
                r.L.rstub_fail_malloc_in(kth)
                arr = mk()
                out = r.L.rstub_call(name.encode(), len(arr), (SEXP * len(arr))(*arr))
                assert r.L.rstub_fault() is None, (name, kth, r.L.rstub_fault())
                assert r.L.rstub_protect_depth() == 0, (name, kth)
                if r.L.rstub_fail_pending() >= 0:   # the failure was never reached: the call needed fewer allocations and went through
                    r.L.rstub_fail_malloc_in(-1)
                    assert out, (name, kth, r.L.rstub_last_error())
                    break
                assert not out and "chicdiff_hip" in r.L.rstub_last_error().decode(), (name, kth)
                kth += 1
                assert kth < 64
            sites[name] = kth
        assert sites["chicdiff_hip_wald_test"] >= 16 and min(sites.values()) >= 4, sites  # (2 inputs + 11 + 3 result columns)
        # what is still allocated belongs to external pointers nobody released: results of the calls that went through, and the
        # temporaries the longjmps left behind.  Now the finalizers, in this run's order.
        left = r.outstanding() - m0
        from chicdiff_amd import hip
        assert left > 0 and hip.load_library().chicdiff_hip_outstanding_allocations(C.c_void_p(r.L.rstub_extptr_addr(ctx))) == left  # the library's own list agrees
        ran = r.L.rstub_run_finalizers(order)
        assert ran > 0 and r.L.rstub_pending_finalizers() == 0 and r.L.rstub_fault() is None
        if order in (0, 2):   # the context went first (order 0: it was registered first): the library released its outstanding vectors with it,
            assert r.outstanding() - m0 == left      # ... the buffers' finalizers found it gone and freed nothing themselves
        else:                 # buffers first: every one released through the shim's chicdiff_hip_free, then the context
            assert r.outstanding() == m0, (r.outstanding(), m0)
        r.L.rstub_free_all()
        return sites, left

    r.L.rstub_run_finalizers(0)
    one_round()   # warm-up: code objects loaded, the ctypes context's workspace in its final state
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    sites, left = one_round()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    # free device memory before / after the second round: a coarse look only (hipMalloc hands out granules, and the runtime keeps
    # per-stream resources of its own across hipStreamDestroy — tools/ctx_memory.py); what the shim and the library owe is settled
    # exactly by the two counts above
    assert free1 >= free0 - (64 << 20), (free0, free1)
    print(f"order {order}: allocation sites exercised {sites}; {left} vectors outstanding before the finalizers; free device memory {free0 >> 20} -> {free1 >> 20} MiB")
