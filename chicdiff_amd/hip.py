"""ctypes binding of the C ABI in ``include/chicdiff_hip.h`` (``chicdiff_amd/lib/libchicdiff_hip.so``).

This is the product path: there is NO CPU fallback.  If the shared library has not been
built, or no MI355X is visible, construction raises — it never routes to ``oracle/``.
PyTorch is used only as plumbing: device buffers (``torch.Tensor``), the current HIP stream,
and ``torch.distributed`` for the sum-all-reduce hook.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .dist import ALLGATHER_FN, ALLREDUCE_FN

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libchicdiff_hip.so")

ST_TREND_FAILED, ST_PRIORVAR_MC, ST_BETA_NONCONV, ST_ALLZERO_ROWS, ST_TREND_LOCAL = 1, 2, 4, 8, 16


# every symbol include/chicdiff_hip.h declares (tests check the library exports each)
EXPORTS = [
    "chicdiff_hip_create", "chicdiff_hip_destroy", "chicdiff_hip_last_error", "chicdiff_hip_set_stream",
    "chicdiff_hip_set_allreduce", "chicdiff_hip_set_allgather", "chicdiff_hip_last_refits", "chicdiff_hip_set_option", "chicdiff_hip_default_opts", "chicdiff_hip_size_factors_dev",
    "chicdiff_hip_offsets_dev", "chicdiff_hip_window_sums_dev", "chicdiff_hip_count_join_dev",
    "chicdiff_hip_fragment_background_dev", "chicdiff_hip_bh_adjust_dev", "chicdiff_hip_ihw_apply_dev",
    "chicdiff_hip_region_universe_count_dev", "chicdiff_hip_region_universe_fill_dev", "chicdiff_hip_region_universe_dev", "chicdiff_hip_count_table_dev",
    "chicdiff_hip_chinput_read", "chicdiff_hip_chinput_table_dev", "chicdiff_hip_region_avdist_dev",
    "chicdiff_hip_count_join_inner_dev", "chicdiff_hip_count_join_multi_dev",
    "chicdiff_hip_malloc", "chicdiff_hip_free", "chicdiff_hip_outstanding_allocations", "chicdiff_hip_memcpy_h2d", "chicdiff_hip_memcpy_d2h",
    "chicdiff_hip_rccl_unique_id", "chicdiff_hip_rccl_init", "chicdiff_hip_cooks_filter_dev",
    "chicdiff_hip_independent_filtering_dev",
    "chicdiff_hip_nbglm_fit_dev", "chicdiff_hip_nbglm_fit", "chicdiff_hip_wald_test_dev", "chicdiff_hip_theta_grid_dev",
    "chicdiff_hip_wald_pvalues_dev", "chicdiff_hip_selftest_math_dev", "chicdiff_hip_selftest_r_random",
    "chicdiff_hip_selftest_prior_mc", "chicdiff_hip_selftest_chinput", "chicdiff_hip_kernel_times", "chicdiff_hip_enable_timing",
]


class Opts(C.Structure):
    _fields_ = [("minDisp", C.c_double), ("dispTol", C.c_double), ("kappa0", C.c_double),
                ("maxit", C.c_int32), ("betaMaxit", C.c_int32), ("betaTol", C.c_double),
                ("minmu", C.c_double), ("outlierSD", C.c_double), ("dispPriorVar", C.c_double),
                ("trendCoef", C.c_double * 2), ("fitType", C.c_int32), ("_pad", C.c_int32)]


OUT_DOUBLE = ["baseMean", "baseVar", "dispGeneEst", "dispFit", "dispMAP", "dispersion", "log2FoldChange",
              "lfcSE", "stat", "pvalue", "intercept", "interceptSE", "deviance", "maxCooks"]
OUT_INT = ["dispGeneIter", "dispIter", "dispOutlier", "betaConv", "betaIter", "allZero", "cooksArgmax"]


class Out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in OUT_DOUBLE + OUT_INT]


class Scalars(C.Structure):
    _fields_ = [("trendCoef", C.c_double * 2), ("varLogDispEsts", C.c_double), ("dispPriorVar", C.c_double),
                ("sumDeviance", C.c_double), ("nAllZero", C.c_int64), ("trendOuterIter", C.c_int32),
                ("status", C.c_int32)]


class ResultsInfo(C.Structure):
    _fields_ = [("filterThreshold", C.c_double), ("filterTheta", C.c_double), ("alpha", C.c_double), ("index", C.c_int32),
                ("_pad", C.c_int32), ("theta", C.c_double * 50), ("numRej", C.c_double * 50), ("lowess", C.c_double * 50)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_int32), ("_pad", C.c_int32), ("bytes", C.c_double)]


class ChicdiffHipError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """Load the HIP library; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own ROCm runtime (libamdhip64 & co.), and whichever copy is loaded first serves the
    # whole process.  Loading this library before torch would bind everybody to /opt/rocm's copy, which torch's
    # build does not match ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    path = os.environ.get("CHICDIFF_HIP_LIB", LIB_PATH)  # (override: A/B runs of two builds)
    if not os.path.exists(path):
        raise ChicdiffHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    L = C.CDLL(path)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    L.chicdiff_hip_create.argtypes = [C.POINTER(vp), i32]
    L.chicdiff_hip_destroy.argtypes = [vp]
    L.chicdiff_hip_destroy.restype = None
    L.chicdiff_hip_last_error.argtypes = [vp]
    L.chicdiff_hip_last_error.restype = C.c_char_p
    L.chicdiff_hip_set_stream.argtypes = [vp, vp]
    L.chicdiff_hip_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, i32, i32]
    L.chicdiff_hip_set_allgather.argtypes = [vp, ALLGATHER_FN, vp]
    L.chicdiff_hip_last_refits.argtypes = [vp]
    L.chicdiff_hip_last_refits.restype = i32
    L.chicdiff_hip_set_option.argtypes = [vp, C.c_char_p, i64]
    L.chicdiff_hip_default_opts.argtypes = [C.POINTER(Opts)]
    L.chicdiff_hip_default_opts.restype = None
    L.chicdiff_hip_size_factors_dev.argtypes = [vp, vp, i64, i32, C.POINTER(dbl)]
    L.chicdiff_hip_offsets_dev.argtypes = [vp, vp, C.POINTER(dbl), i64, i32, dbl, vp]
    L.chicdiff_hip_window_sums_dev.argtypes = [vp, vp, vp, i64, i32, vp, i64, vp, vp]
    L.chicdiff_hip_count_join_dev.argtypes = [vp, vp, vp, i64, vp, vp, i64, vp]
    L.chicdiff_hip_fragment_background_dev.argtypes = [vp, vp, vp, i64, i32, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32,
                                                       C.POINTER(dbl), vp, vp, vp]
    L.chicdiff_hip_cooks_filter_dev.argtypes = [vp, vp, i64, i32, C.POINTER(i32), vp, vp, dbl, vp, C.POINTER(i64)]
    L.chicdiff_hip_independent_filtering_dev.argtypes = [vp, vp, vp, i64, dbl, vp, C.POINTER(ResultsInfo)]
    L.chicdiff_hip_rccl_unique_id.argtypes = [vp, C.c_char_p, vp]
    L.chicdiff_hip_rccl_init.argtypes = [vp, C.c_char_p, vp, i32, i32]
    L.chicdiff_hip_malloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.chicdiff_hip_free.argtypes = [vp, vp]
    L.chicdiff_hip_outstanding_allocations.argtypes = [vp]
    L.chicdiff_hip_outstanding_allocations.restype = i64
    L.chicdiff_hip_memcpy_h2d.argtypes = [vp, vp, vp, C.c_uint64]
    L.chicdiff_hip_memcpy_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    L.chicdiff_hip_count_table_dev.argtypes = [vp, vp, vp, vp, i64, vp, i32, vp, vp, C.POINTER(i64)]
    L.chicdiff_hip_chinput_read.argtypes = [vp, C.c_char_p, i32, C.POINTER(i64)]
    L.chicdiff_hip_chinput_table_dev.argtypes = [vp, vp, i32, vp, vp, C.POINTER(i64)]
    L.chicdiff_hip_bh_adjust_dev.argtypes = [vp, vp, i64, vp]
    L.chicdiff_hip_region_avdist_dev.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp]
    L.chicdiff_hip_count_join_inner_dev.argtypes = [vp, vp, vp, i64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), vp]
    L.chicdiff_hip_count_join_multi_dev.argtypes = [vp, vp, vp, i64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), vp]
    L.chicdiff_hip_ihw_apply_dev.argtypes = [vp, vp, vp, i64, C.POINTER(dbl), C.POINTER(dbl), i32, vp, vp, vp, vp]
    L.chicdiff_hip_region_universe_count_dev.argtypes = [vp, vp, vp, i64, i32, vp, i32, vp, vp, vp, C.POINTER(i64)]
    L.chicdiff_hip_region_universe_fill_dev.argtypes = [vp, vp, vp, i64, i32, vp, i32, vp, vp, vp, vp]
    L.chicdiff_hip_region_universe_dev.argtypes = [vp, vp, vp, i64, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.chicdiff_hip_nbglm_fit_dev.argtypes = [vp, vp, vp, i64, i32, C.POINTER(i32), C.POINTER(Opts), C.POINTER(Out),
                                             C.POINTER(Scalars)]
    L.chicdiff_hip_nbglm_fit.argtypes = L.chicdiff_hip_nbglm_fit_dev.argtypes
    L.chicdiff_hip_wald_test_dev.argtypes = [vp, vp, vp, i64, i32, C.POINTER(i32), dbl, C.POINTER(Opts), C.POINTER(Out),
                                             C.POINTER(Scalars), C.POINTER(dbl)]
    L.chicdiff_hip_theta_grid_dev.argtypes = [vp, vp, vp, C.POINTER(dbl), i64, i32, C.POINTER(dbl), i32,
                                              C.POINTER(Opts), C.POINTER(dbl)]
    L.chicdiff_hip_wald_pvalues_dev.argtypes = [vp, vp, i64, vp]
    L.chicdiff_hip_selftest_math_dev.argtypes = [vp, i32, vp, i64, vp]
    L.chicdiff_hip_kernel_times.argtypes = [vp, C.POINTER(KernelTime), i32]
    L.chicdiff_hip_kernel_times.restype = i32
    L.chicdiff_hip_enable_timing.argtypes = [vp, i32]
    _lib = L
    return L


def default_opts(**kw) -> Opts:
    o = Opts()
    load_library().chicdiff_hip_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "trendCoef":
            o.trendCoef[0], o.trendCoef[1] = float(v[0]), float(v[1])
        else:
            setattr(o, k, v)
    return o


def _scalars_dict(s: Scalars) -> dict:
    return dict(trendCoef=np.array(s.trendCoef[:]), varLogDispEsts=s.varLogDispEsts, dispPriorVar=s.dispPriorVar,
                sumDeviance=s.sumDeviance, nAllZero=s.nAllZero, trendOuterIter=s.trendOuterIter, status=s.status)


class HipContext:
    """One context per process/GPU.  Device buffers are torch tensors on ``cuda:<device>``
    (torch's name for a ROCm device), laid out sample-major: a tensor of shape (S, n),
    contiguous, is the C ABI's column-major n x S matrix."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch

        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise ChicdiffHipError("no MI355X visible to PyTorch-ROCm: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        rc = self.lib.chicdiff_hip_create(C.byref(h), device)
        if rc:
            raise ChicdiffHipError(self.lib.chicdiff_hip_last_error(None).decode())
        self.h = h
        self._cb = None
        self._comm_tensors = {}
        if use_torch_stream:
            self.use_stream(torch.cuda.current_stream(self.device))

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.chicdiff_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise ChicdiffHipError(f"[{rc}] " + self.lib.chicdiff_hip_last_error(self.h).decode())

    def use_stream(self, stream):
        self._check(self.lib.chicdiff_hip_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def set_option(self, name: str, value: int):
        """Tuning / test options of include/chicdiff_hip.h (results never depend on them)."""
        self._check(self.lib.chicdiff_hip_set_option(self.h, name.encode(), int(value)))

    def enable_timing(self, on=True):
        self.lib.chicdiff_hip_enable_timing(self.h, int(on))

    def kernel_times(self) -> dict:
        buf = (KernelTime * 32)()
        k = self.lib.chicdiff_hip_kernel_times(self.h, buf, 32)
        return {buf[i].name.decode(): (buf[i].ms, buf[i].launches) for i in range(min(k, 32))}

    def collective_stats(self) -> dict:
        """Collectives of the last call (timing mode 1): {"allreduce" / "allgather": (count, ms on the stream, bytes handed over)}."""
        buf = (KernelTime * 32)()
        k = self.lib.chicdiff_hip_kernel_times(self.h, buf, 32)
        return {buf[i].name.decode(): (buf[i].launches, buf[i].ms, buf[i].bytes) for i in range(min(k, 32))
                if buf[i].name in (b"allreduce", b"allgather")}

    def last_refits(self) -> int:
        """Refits the last call went through (select overflow / barrier timeout / local substitute); the same on every rank."""
        return int(self.lib.chicdiff_hip_last_refits(self.h))

    def set_process_group(self, group=None, memory="device", allgather=True):
        """Route the library's collectives through torch.distributed (backend nccl = RCCL): the sum-all-reduce hook and,
        unless ``allgather`` is False, the all-gather hook for the rows of the dispersion trend."""
        from .dist import AllReduceHook

        self._hook = AllReduceHook(group, memory=memory, device=self.device)
        self._sharded = True  # (from here on the context's fits are pieces of a sharded fit: dist.theta_grid_replicas refuses it)
        self._check(self.lib.chicdiff_hip_set_allreduce(self.h, self._hook.fn, None, self._hook.world, self._hook.rank))
        if allgather:
            self._check(self.lib.chicdiff_hip_set_allgather(self.h, self._hook.gather_fn, None))

    def init_rccl(self, group=None, librccl_path=None):
        """Direct RCCL: the library makes its own communicator over the ranks of ``group`` and issues
        ncclAllReduce itself (no Python callback per collective).  torch.distributed only carries the
        128-byte unique id.  Uses the librccl torch itself loaded unless ``librccl_path`` says otherwise."""
        import os

        import torch.distributed as dist

        torch = self.torch
        if librccl_path is None:
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            librccl_path = cand if os.path.exists(cand) else "librccl.so"
        path = librccl_path.encode()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = self.device if dist.get_backend(group) == "nccl" else "cpu"

        def all_ok(ok: bool) -> bool:  # every rank must take the same branch, or the collectives that follow hang
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return bool(flag.item())

        # phase 1, local: can this rank open librccl at all?  (every rank makes an id; only rank 0's is used)
        ident = (C.c_char * 128)()
        rc = self.lib.chicdiff_hip_rccl_unique_id(self.h, path, ident)
        msg = self.lib.chicdiff_hip_last_error(self.h).decode() if rc else ""
        if not all_ok(rc == 0):
            raise ChicdiffHipError(f"librccl not usable on every rank ({msg or 'another rank failed'})")
        # phase 2, collective: share rank 0's id, create the communicator
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        buf = (C.c_char * 128).from_buffer_copy(box[0])
        rc = self.lib.chicdiff_hip_rccl_init(self.h, path, buf, world, rank)
        msg = self.lib.chicdiff_hip_last_error(self.h).decode() if rc else ""
        if not all_ok(rc == 0):
            self.lib.chicdiff_hip_set_allreduce(self.h, C.cast(None, ALLREDUCE_FN), None, 1, 0)
            raise ChicdiffHipError(f"ncclCommInitRank did not succeed on every rank ({msg or 'another rank failed'})")
        self._hook = None
        self._sharded = True

    def to_device(self, a, dtype):
        """(n, S) host array -> (S, n) contiguous device tensor (sample-major)."""
        t = self.torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=dtype).T))
        return t.to(self.device)

    # -- a5 ---------------------------------------------------------------------------------
    def size_factors(self, d_counts) -> np.ndarray:
        S, n = d_counts.shape
        sf = (C.c_double * S)()
        self._check(self.lib.chicdiff_hip_size_factors_dev(self.h, d_counts.data_ptr(), n, S, sf))
        return np.array(sf[:])

    # -- a4 ---------------------------------------------------------------------------------
    def offsets(self, d_fullmean, size_factors, theta=None, out=None):
        S, n = d_fullmean.shape
        sf = (C.c_double * S)(*[float(x) for x in size_factors])
        if out is None:
            out = self.torch.empty_like(d_fullmean)
        th = float("nan") if theta is None else float(theta)
        self._check(self.lib.chicdiff_hip_offsets_dev(self.h, d_fullmean.data_ptr(), sf, n, S, th, out.data_ptr()))
        return out

    # -- a2 ---------------------------------------------------------------------------------
    def window_sums(self, d_fragN, d_fragFM, d_region_ptr):
        torch = self.torch
        ref = d_fragN if d_fragN is not None else d_fragFM
        S, nfrag = ref.shape
        n = d_region_ptr.numel() - 1
        N = torch.empty((S, n), dtype=torch.int32, device=self.device) if d_fragN is not None else None
        FM = torch.empty((S, n), dtype=torch.float64, device=self.device) if d_fragFM is not None else None
        self._check(self.lib.chicdiff_hip_window_sums_dev(
            self.h, d_fragN.data_ptr() if d_fragN is not None else None,
            d_fragFM.data_ptr() if d_fragFM is not None else None, nfrag, S, d_region_ptr.data_ptr(), n,
            N.data_ptr() if N is not None else None, FM.data_ptr() if FM is not None else None))
        return N, FM

    # -- a1 ---------------------------------------------------------------------------------
    def count_join(self, d_bait, d_oe, d_keys, d_vals):
        out = self.torch.empty_like(d_bait)
        self._check(self.lib.chicdiff_hip_count_join_dev(self.h, d_bait.data_ptr(), d_oe.data_ptr(), d_bait.numel(),
                                                         d_keys.data_ptr(), d_vals.data_ptr(), d_keys.numel(),
                                                         out.data_ptr()))
        return out

    def count_join_multi(self, d_bait, d_oe, tables, out=None):
        """The chinput branch for all replicates at once (chicdiff.R:843-858, the loop over the replicates): ``tables`` =
        [(keys, vals)] per replicate; N (S, nru), row s = ``count_join`` with table s, from ONE read of the RU rows."""
        S, nru = len(tables), d_bait.numel()
        if out is None:
            out = self.torch.empty((S, nru), dtype=self.torch.int32, device=self.device)
        assert out.shape == (S, nru) and out.dtype == self.torch.int32 and out.is_contiguous() and S >= 1
        kp = (C.c_void_p * S)(*[k.data_ptr() for k, _ in tables])
        vp_ = (C.c_void_p * S)(*[v.data_ptr() for _, v in tables])
        nk = (C.c_int64 * S)(*[k.numel() for k, _ in tables])
        self._check(self.lib.chicdiff_hip_count_join_multi_dev(self.h, d_bait.data_ptr(), d_oe.data_ptr(), nru, S, kp, vp_, nk,
                                                               out.data_ptr()))
        return out

    def count_join_inner(self, d_bait, d_oe, tables):
        """No-chinput branch (chicdiff.R:774-807): ``tables`` = [(keys, vals)] per replicate (device tensors as
        ``count_table`` returns them); N (S, nru) with the reference's Reduce(merge) semantics — a pair keeps its
        counts only when every replicate's table holds it."""
        S, nru = len(tables), d_bait.numel()
        out = self.torch.empty((S, nru), dtype=self.torch.int32, device=self.device)
        kp = (C.c_void_p * S)(*[k.data_ptr() for k, _ in tables])
        vp_ = (C.c_void_p * S)(*[v.data_ptr() for _, v in tables])
        nk = (C.c_int64 * S)(*[k.numel() for k, _ in tables])
        self._check(self.lib.chicdiff_hip_count_join_inner_dev(self.h, d_bait.data_ptr(), d_oe.data_ptr(), nru, S, kp, vp_, nk,
                                                               out.data_ptr()))
        return out

    def region_avdist(self, d_bait, d_oe, d_region_ptr, id_min, d_midsum, d_chr=None):
        """avDist = mean(distSign) by regionID (chicdiff.R:1965-1967, :868-882) for CSR-ordered RU rows: the covariate
        IHWcorrection() takes from the long table."""
        n = d_region_ptr.numel() - 1
        out = self.torch.empty(n, dtype=self.torch.float64, device=self.device)
        self._check(self.lib.chicdiff_hip_region_avdist_dev(self.h, d_bait.data_ptr(), d_oe.data_ptr(), d_bait.numel(),
                                                            d_region_ptr.data_ptr(), n, int(id_min), d_midsum.numel(),
                                                            d_midsum.data_ptr(), d_chr.data_ptr() if d_chr is not None else None,
                                                            out.data_ptr()))
        return out

    # -- a3 ---------------------------------------------------------------------------------
    def fragment_background(self, d_bait, d_oe, id_min, d_midsum, d_sj, d_si, d_tblb, d_tlb, d_T, distfun, only_fullmean=False):
        """Bmean, Tmean, FullMean (S, nru) for RU rows (d_bait, d_oe); tables as in the header.  ``only_fullmean``: the first two
        come back as None and are never written (FullMean is the one column DESeq2Wrap reads, chicdiff.R:896, :1543)."""
        torch = self.torch
        S, nid = d_sj.shape
        nru = d_bait.numel()
        df = np.ascontiguousarray(distfun, dtype=np.float64)
        assert df.shape == (S, 10) and d_T.shape[0] == S
        outs = [None if (only_fullmean and k < 2) else torch.empty((S, nru), dtype=torch.float64, device=self.device) for k in range(3)]
        ptr = lambda t: t.data_ptr() if t is not None else None
        self._check(self.lib.chicdiff_hip_fragment_background_dev(
            self.h, d_bait.data_ptr(), d_oe.data_ptr(), nru, int(id_min), nid, d_midsum.data_ptr(), S, d_sj.data_ptr(),
            d_si.data_ptr(), d_tblb.data_ptr(), d_tlb.data_ptr(), d_T.data_ptr(), d_T.shape[1], d_T.shape[2],
            df.ctypes.data_as(C.POINTER(C.c_double)), ptr(outs[0]), ptr(outs[1]), ptr(outs[2])))
        return outs

    # -- f2: chinput columns -> key table of the count join -------------------------------------
    def count_table(self, d_bait, d_oe, d_N, d_bait_in_RU=None):
        """(keys, vals) of ``count_join`` from unsorted chinput columns; rows whose bait is not flagged in
        ``d_bait_in_RU`` (uint8 per ID) are dropped (chicdiff.R:828-831, :849)."""
        torch = self.torch
        n = d_bait.numel()
        keys = torch.empty(n, dtype=torch.int64, device=self.device)
        vals = torch.empty(n, dtype=torch.int32, device=self.device)
        nk = C.c_int64(0)
        self._check(self.lib.chicdiff_hip_count_table_dev(
            self.h, d_bait.data_ptr(), d_oe.data_ptr(), d_N.data_ptr(), n,
            d_bait_in_RU.data_ptr() if d_bait_in_RU is not None else None,
            d_bait_in_RU.numel() - 1 if d_bait_in_RU is not None else 0, keys.data_ptr(), vals.data_ptr(), C.byref(nk)))
        return keys[: nk.value], vals[: nk.value]

    def read_chinput(self, path, d_bait_in_RU=None, nthreads=0):
        """fread(chinput)[, c("baitID", "otherEndID", "N")] restricted to the RU baits -> (keys, vals) of ``count_join``
        (chicdiff.R:828-831, :849): text parsed by host threads, bait filter + sort on the device."""
        torch = self.torch
        nrows = C.c_int64(0)
        self._check(self.lib.chicdiff_hip_chinput_read(self.h, os.fsencode(path), int(nthreads), C.byref(nrows)))
        keys = torch.empty(nrows.value, dtype=torch.int64, device=self.device)
        vals = torch.empty(nrows.value, dtype=torch.int32, device=self.device)
        nk = C.c_int64(0)
        self._check(self.lib.chicdiff_hip_chinput_table_dev(
            self.h, d_bait_in_RU.data_ptr() if d_bait_in_RU is not None else None,
            d_bait_in_RU.numel() - 1 if d_bait_in_RU is not None else 0, keys.data_ptr(), vals.data_ptr(), C.byref(nk)))
        return keys[: nk.value], vals[: nk.value], nrows.value

    # -- a9: results() ---------------------------------------------------------------------------
    def cooks_filter(self, d_counts, group, d_maxCooks, d_cooksArgmax, d_pvalue, cutoff):
        """p <- NA for Cook's outliers, in place on ``d_pvalue``; returns the number of rows set to NA."""
        S, n = d_counts.shape
        g = (C.c_int32 * S)(*[int(x) for x in group])
        nout = C.c_int64(0)
        self._check(self.lib.chicdiff_hip_cooks_filter_dev(self.h, d_counts.data_ptr(), n, S, g, d_maxCooks.data_ptr(),
                                                           d_cooksArgmax.data_ptr(), float(cutoff), d_pvalue.data_ptr(), C.byref(nout)))
        return nout.value

    def independent_filtering(self, d_baseMean, d_pvalue, alpha=0.1):
        """DESeq2 pvalueAdjustment(independentFiltering = TRUE): returns (padj device tensor, info dict)."""
        torch = self.torch
        n = d_pvalue.numel()
        padj = torch.empty(n, dtype=torch.float64, device=self.device)
        info = ResultsInfo()
        self._check(self.lib.chicdiff_hip_independent_filtering_dev(self.h, d_baseMean.data_ptr(), d_pvalue.data_ptr(), n, float(alpha),
                                                                    padj.data_ptr(), C.byref(info)))
        return padj, dict(filterThreshold=info.filterThreshold, filterTheta=info.filterTheta, index=info.index,
                          theta=np.array(info.theta[:]), numRej=np.array(info.numRej[:]), lowess=np.array(info.lowess[:]))

    # -- f1 / f3: BH and the IHW application side ----------------------------------------------
    def bh_adjust(self, d_p):
        """p.adjust(p, "BH") on a device vector (NaN = NA)."""
        torch = self.torch
        assert d_p.dtype == torch.float64 and d_p.is_contiguous()
        out = torch.empty_like(d_p)
        self._check(self.lib.chicdiff_hip_bh_adjust_dev(self.h, d_p.data_ptr(), d_p.numel(), out.data_ptr()))
        return out

    def ihw_apply(self, d_avDist, d_pvalue, breaks, avWeights):
        """chicdiff.R:2038-2049: returns dict(group, weight, weighted_pvalue, weighted_padj) of device tensors."""
        torch = self.torch
        n = d_avDist.numel()
        b = np.ascontiguousarray(breaks, dtype=np.float64)
        w = np.ascontiguousarray(avWeights, dtype=np.float64)
        assert len(b) == len(w) + 1 and d_pvalue.numel() == n
        group = torch.empty(n, dtype=torch.int32, device=self.device)
        weight, wp, wpadj = (torch.empty(n, dtype=torch.float64, device=self.device) for _ in range(3))
        P = C.POINTER(C.c_double)
        self._check(self.lib.chicdiff_hip_ihw_apply_dev(self.h, d_avDist.data_ptr(), d_pvalue.data_ptr(), n, b.ctypes.data_as(P),
                                                        w.ctypes.data_as(P), len(w), group.data_ptr(), weight.data_ptr(),
                                                        wp.data_ptr(), wpadj.data_ptr()))
        return dict(group=group, weight=weight, weighted_pvalue=wp, weighted_padj=wpadj)

    # -- f4: region universe ---------------------------------------------------------------------
    def region_universe(self, d_bait, d_oe, RUexpand, d_chr_of):
        """getRegionUniverse window mode (chicdiff.R:353-426).  d_chr_of: int32 (maxfrag + 1,), -1 = not on the map.
        Returns dict(region_ptr, minOE, maxOE, baitID, regionID, otherEndID); the RU rows are in
        (regionID, otherEndID) order."""
        torch = self.torch
        n = d_bait.numel()
        maxfrag = d_chr_of.numel() - 1
        ptr = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        mn, mx = (torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(2))
        total = C.c_int64(0)
        cap = n * max(2 * int(RUexpand) + 1, 2)   # the upper bound (two for RUexpand = 0: R's descending a:b beside a bait): scan and fill in one call (round 5)
        rb, rr, ro = (torch.empty(max(cap, 1), dtype=torch.int32, device=self.device) for _ in range(3))
        self._check(self.lib.chicdiff_hip_region_universe_dev(self.h, d_bait.data_ptr(), d_oe.data_ptr(), n, int(RUexpand), d_chr_of.data_ptr(),
                                                              maxfrag, ptr.data_ptr(), mn.data_ptr(), mx.data_ptr(), rb.data_ptr(), rr.data_ptr(),
                                                              ro.data_ptr(), cap, C.byref(total)))
        self.last_region_universe_ms = self.kernel_times().get("region_universe", (0.0, 0))[0]
        rb, rr, ro = rb[: total.value], rr[: total.value], ro[: total.value]
        if 2 * total.value < cap:  # (a view would pin the whole upper-bound allocation for as long as the universe lives)
            rb, rr, ro = rb.clone(), rr.clone(), ro.clone()
        return dict(region_ptr=ptr, minOE=mn, maxOE=mx, baitID=rb, regionID=rr, otherEndID=ro)

    # -- a6 + a7 ----------------------------------------------------------------------------
    def nbglm_fit(self, d_counts, d_nf, group, want=None, opts: Opts | None = None, outputs: dict | None = None):
        """estimateDispersions + nbinomWaldTest on device-resident (S, n) tensors.

        Returns (outputs, scalars): ``outputs`` maps the requested column names to device
        tensors of length n (pass ``outputs`` to reuse buffers)."""
        torch = self.torch
        S, n = d_counts.shape
        assert d_nf.shape == (S, n) and d_counts.dtype == torch.int32 and d_nf.dtype == torch.float64
        assert d_counts.is_contiguous() and d_nf.is_contiguous()
        want = list(want) if want is not None else ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"]
        out = Out()
        bufs = outputs if outputs is not None else {}
        for k in want:
            if k not in bufs:
                bufs[k] = torch.empty(n, dtype=torch.float64 if k in OUT_DOUBLE else torch.int32, device=self.device)
            setattr(out, k, bufs[k].data_ptr())
        g = (C.c_int32 * S)(*[int(x) for x in group])
        sc = Scalars()
        self._check(self.lib.chicdiff_hip_nbglm_fit_dev(self.h, d_counts.data_ptr(), d_nf.data_ptr(), n, S, g,
                                                        C.byref(opts) if opts is not None else None, C.byref(out),
                                                        C.byref(sc)))
        return bufs, _scalars_dict(sc)

    def wald_test(self, d_counts, d_fullmean, group, theta=None, want=None, opts: Opts | None = None,
                  outputs: dict | None = None):
        """size factors -> offsets(theta) -> dispersions -> Wald test in one enqueue (device-resident)."""
        torch = self.torch
        S, n = d_counts.shape
        assert d_fullmean.shape == (S, n) and d_counts.is_contiguous() and d_fullmean.is_contiguous()
        want = list(want) if want is not None else ["baseMean", "dispersion", "log2FoldChange", "lfcSE", "stat", "pvalue"]
        out = Out()
        bufs = outputs if outputs is not None else {}
        for k in want:
            if k not in bufs:
                bufs[k] = torch.empty(n, dtype=torch.float64 if k in OUT_DOUBLE else torch.int32, device=self.device)
            setattr(out, k, bufs[k].data_ptr())
        g = (C.c_int32 * S)(*[int(x) for x in group])
        sc = Scalars()
        sf = (C.c_double * S)()
        self._check(self.lib.chicdiff_hip_wald_test_dev(
            self.h, d_counts.data_ptr(), d_fullmean.data_ptr(), n, S, g, float("nan") if theta is None else float(theta),
            C.byref(opts) if opts is not None else None, C.byref(out), C.byref(sc), sf))
        res = _scalars_dict(sc)
        res["sizeFactors"] = np.array(sf[:])
        return bufs, res

    def nbglm_fit_host(self, counts, nf, group, want=None, opts: Opts | None = None):
        """Host-buffer entry point (what the R .Call shim uses): numpy (n, S) in, numpy out."""
        k = np.asfortranarray(np.asarray(counts, dtype=np.int32))
        f = np.asfortranarray(np.asarray(nf, dtype=np.float64))
        n, S = k.shape
        want = list(want) if want is not None else OUT_DOUBLE + OUT_INT
        out = Out()
        res = {}
        for name in want:
            res[name] = np.empty(n, dtype=np.float64 if name in OUT_DOUBLE else np.int32)
            setattr(out, name, res[name].ctypes.data)
        g = (C.c_int32 * S)(*[int(x) for x in group])
        sc = Scalars()
        self._check(self.lib.chicdiff_hip_nbglm_fit(self.h, k.ctypes.data, f.ctypes.data, n, S, g,
                                                    C.byref(opts) if opts is not None else None, C.byref(out), C.byref(sc)))
        return res, _scalars_dict(sc)

    # -- a8 ---------------------------------------------------------------------------------
    def theta_grid(self, d_counts, d_fullmean, size_factors, thetas, opts: Opts | None = None) -> np.ndarray:
        S, n = d_counts.shape
        sf = (C.c_double * S)(*[float(x) for x in size_factors])
        th = (C.c_double * len(thetas))(*[float(x) for x in thetas])
        dev = (C.c_double * len(thetas))()
        self._check(self.lib.chicdiff_hip_theta_grid_dev(self.h, d_counts.data_ptr(), d_fullmean.data_ptr(), sf, n, S, th,
                                                         len(thetas), C.byref(opts) if opts is not None else None, dev))
        return np.array(dev[:])

    def selftest_math(self, op: int, d_x):
        out = self.torch.empty_like(d_x)
        self._check(self.lib.chicdiff_hip_selftest_math_dev(self.h, op, d_x.data_ptr(), d_x.numel(), out.data_ptr()))
        return out

    def wald_pvalues(self, d_stat):
        out = self.torch.empty_like(d_stat)
        self._check(self.lib.chicdiff_hip_wald_pvalues_dev(self.h, d_stat.data_ptr(), d_stat.numel(), out.data_ptr()))
        return out
