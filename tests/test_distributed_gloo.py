"""The row-sharded (N > 1) path on CPU: world_size 2 over gloo.

The product's driver (chicdiff_amd/csrc/fit_driver.h) and state machines (fit_state.h) are
compiled into a test-only CPU backend (tests/harness/shard_harness.cpp); each rank holds a row
shard, every global sum goes through chicdiff_amd.dist.AllReduceHook (the same callback type the
HIP library calls, here on host memory), and the result must equal the single-rank oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS_SRC = os.path.join(ROOT, "tests", "harness", "shard_harness.cpp")
HARNESS_SO = os.path.join(ROOT, "tests", "harness", "libshard_harness.so")


def build_harness():
    if not os.path.exists(HARNESS_SO) or os.path.getmtime(HARNESS_SO) < max(
            os.path.getmtime(HARNESS_SRC), os.path.getmtime(os.path.join(ROOT, "chicdiff_amd", "csrc", "fit_state.h")),
            os.path.getmtime(os.path.join(ROOT, "chicdiff_amd", "csrc", "fit_driver.h"))):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", HARNESS_SO, HARNESS_SRC], check=True)
    from chicdiff_amd.dist import ALLREDUCE_FN
    L = C.CDLL(HARNESS_SO)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.harness_trend_mad.argtypes = [pd, pd, pi, C.c_int64, C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                    C.c_int32, ALLREDUCE_FN, C.c_void_p, pd]
    L.harness_size_factors.argtypes = [pi, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, ALLREDUCE_FN, C.c_void_p, pd]
    return L


def _worker(rank, world, port, n, S, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from chicdiff_amd import synth
    from chicdiff_amd.dist import AllReduceHook, shard_bounds
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = build_harness()
        d = synth.make(n, S)
        ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])  # per-row inputs of the global steps + expected scalars
        lo, hi = shard_bounds(n, world, rank)
        hook = AllReduceHook(memory="host")
        f = lambda a, t: np.ascontiguousarray(a[lo:hi], dtype=t)
        bm, dg, az = f(ref["baseMean"], np.float64), f(np.nan_to_num(ref["dispGeneEst"]), np.float64), f(ref["allZero"], np.int32)
        out = np.zeros(8)
        pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        rc = L.harness_trend_mad(bm.ctypes.data_as(pd), dg.ctypes.data_as(pd), az.ctypes.data_as(pi), hi - lo, 1e-8, S, 2,
                                 float("nan"), world, rank, 0, hook.fn, None, out.ctypes.data_as(pd))
        assert rc == 0 and hook.error is None, (rc, hook.error)
        k = np.asfortranarray(d["counts"][lo:hi].astype(np.int32))
        sf = np.zeros(S)
        calls0 = hook.calls
        rc = L.harness_size_factors(k.ctypes.data_as(pi), hi - lo, S, world, rank, 0, hook.fn, None, sf.ctypes.data_as(pd))
        assert rc == 0 and hook.error is None, (rc, hook.error)
        assert hook.calls - calls0 == 4  # two histogram rounds, the count rows, the gathered candidates
        # the same medians through all six histogram rounds (what an overflowing candidate list falls back to)
        out6, sf6 = np.zeros(8), np.zeros(S)
        assert L.harness_trend_mad(bm.ctypes.data_as(pd), dg.ctypes.data_as(pd), az.ctypes.data_as(pi), hi - lo, 1e-8, S, 2,
                                   float("nan"), world, rank, 1, hook.fn, None, out6.ctypes.data_as(pd)) == 0
        calls0 = hook.calls
        assert L.harness_size_factors(k.ctypes.data_as(pi), hi - lo, S, world, rank, 1, hook.fn, None, sf6.ctypes.data_as(pd)) == 0
        assert hook.calls - calls0 == 6 and hook.error is None
        assert np.array_equal(out6, out) and np.array_equal(sf6, sf)
        q.put((rank, out.tolist(), sf.tolist(), hook.calls, ref["trendCoef"].tolist(), ref["varLogDispEsts"],
               ref["dispPriorVar"], ref["trendOuterIter"], oracle.size_factors(d["counts"]).tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,S", [(6001, 8), (3000, 4)])
def test_sharded_global_steps_match_single_rank_oracle(n, S):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:  # a free port chosen by the OS (a fixed one can collide with a stale rendezvous)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, sf, calls, coefs, varlog, prior, outer, sf_ref in res:
        assert calls > 20  # trend passes + 3 selects x 7 all-reduces really went through gloo
        assert np.allclose(out[:2], coefs, rtol=1e-9), (out, coefs)
        assert np.isclose(out[2], varlog, rtol=1e-9) and np.isclose(out[3], prior, rtol=1e-9)
        assert int(out[4]) == outer and out[5] == 0
        assert np.allclose(sf, sf_ref, rtol=1e-13)
    # both ranks end with bit-identical scalars (they consumed the same all-reduced sums)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]


def test_single_rank_harness_matches_oracle():
    """world_size 1: no callback involved; the driver + state machines alone reproduce the oracle."""
    sys.path.insert(0, ROOT)
    from chicdiff_amd import synth
    from chicdiff_amd.dist import ALLREDUCE_FN, shard_bounds
    from oracle import oracle
    assert [shard_bounds(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    L = build_harness()
    d = synth.make(5000, 6)
    ref = oracle.nbglm_fit(d["counts"], d["nf"], d["group"])
    out = np.zeros(8)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    bm = np.ascontiguousarray(ref["baseMean"])
    dg = np.ascontiguousarray(np.nan_to_num(ref["dispGeneEst"]))
    az = np.ascontiguousarray(ref["allZero"], dtype=np.int32)
    null_cb = C.cast(None, ALLREDUCE_FN)
    rc = L.harness_trend_mad(bm.ctypes.data_as(pd), dg.ctypes.data_as(pd), az.ctypes.data_as(pi), len(bm), 1e-8, 6, 2,
                             float("nan"), 1, 0, 0, null_cb, None, out.ctypes.data_as(pd))
    assert rc == 0
    assert np.allclose(out[:2], ref["trendCoef"], rtol=1e-10)
    assert np.isclose(out[2], ref["varLogDispEsts"], rtol=1e-10) and np.isclose(out[3], ref["dispPriorVar"], rtol=1e-10)
    k = np.asfortranarray(d["counts"].astype(np.int32))
    sf = np.zeros(6)
    assert L.harness_size_factors(k.ctypes.data_as(pi), len(k), 6, 1, 0, 0, null_cb, None, sf.ctypes.data_as(pd)) == 0
    assert np.allclose(sf, oracle.size_factors(d["counts"]), rtol=1e-13)


def _ties_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from chicdiff_amd.dist import AllReduceHook, shard_bounds
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = build_harness()
        rng = np.random.default_rng(5)
        n, S = 24000, 4
        counts = rng.poisson(30, size=(n, S)).astype(np.int32)
        counts[: 2 * n // 3] = [10, 20, 30, 41]  # 16 000 identical ratios per column: more than a gathered list holds
        counts = counts[rng.permutation(n)]
        lo, hi = shard_bounds(n, world, rank)
        hook = AllReduceHook(memory="host")
        pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        k = np.asfortranarray(counts[lo:hi])
        sf = np.zeros(S)
        rc = L.harness_size_factors(k.ctypes.data_as(pi), hi - lo, S, world, rank, 0, hook.fn, None, sf.ctypes.data_as(pd))
        assert rc == 0 and hook.error is None
        q.put((rank, sf.tolist(), hook.calls, oracle.size_factors(counts).tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_median_falls_back_when_the_gathered_list_would_overflow():
    """Massive ties: after two rounds more than 4096 candidates share the median's 24 leading key bits, every rank
    sees that from the all-reduced count rows, nobody places candidates, and the remaining histogram rounds run."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_ties_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, sf, calls, ref in res:
        assert calls == 8  # 2 histogram rounds + count rows + (empty) candidate buffer + 4 more histogram rounds
        assert np.allclose(sf, ref, rtol=1e-13)
    assert res[0][1] == res[1][1]


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from chicdiff_amd.dist import AllReduceHook
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hook = AllReduceHook(memory="host")
        count = 1000
        send = np.arange(count, dtype=np.float64) + 10000.0 * rank
        recv = np.full(count * world, -1.0)
        rc = hook.gather_fn(None, send.ctypes.data, recv.ctypes.data, count)
        q.put((rank, rc, repr(hook.error), recv.tolist(), hook.gathers, hook.gather_doubles))
    finally:
        dist.destroy_process_group()


def test_allgather_hook_lays_the_ranks_blocks_out_in_rank_order():
    """chicdiff_allgather_fn as chicdiff_amd.dist implements it (host memory, gloo, world_size 2): rank r's block lands at
    r * count on every rank — the layout api.hip's gathered_trend / trend_compact_kernel read."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = np.concatenate([np.arange(1000.0), np.arange(1000.0) + 10000.0])
    for rank, rc, err, recv, gathers, doubles in res:
        assert rc == 0 and err == "None", (rank, err)
        assert np.array_equal(np.array(recv), want) and gathers == 1 and doubles == 1000


class _StubGrid:
    """Stands in for HipContext.theta_grid: a 'total deviance' that is a known function of theta, NaN at theta = 0.75, and a record
    of what this rank was asked to fit."""

    def __init__(self):
        self.asked = []

    def theta_grid(self, d_counts, d_fullmean, size_factors, thetas, opts=None):
        self.asked.append(list(thetas))
        return np.array([np.nan if t == 0.75 else 1000.0 + 7.0 * t * t - 3.0 * t for t in thetas])


class _FailingGrid(_StubGrid):
    """... and a fit that fails on the rank it is told to"""

    def __init__(self, fail):
        super().__init__()
        self.fail = fail

    def theta_grid(self, d_counts, d_fullmean, size_factors, thetas, opts=None):
        if self.fail:
            raise RuntimeError("device fell over (injected)")
        return super().theta_grid(d_counts, d_fullmean, size_factors, thetas, opts)


def _replica_failure_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from chicdiff_amd.dist import theta_grid_replicas
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dummy = torch.zeros(2, 2)
        try:
            theta_grid_replicas(_FailingGrid(fail=rank == 1), dummy, dummy, [1.0, 1.0], [0.0, 0.25, 0.5, 0.75, 1.0])
            q.put((rank, "returned"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_theta_grid_replicas_a_failure_on_one_rank_raises_on_every_rank():
    """a fit that raises on ONE rank used to leave the others inside dist.all_gather for ever.  Now the local exception is
    caught, a status flag travels with the totals, and every rank raises — the failing rank its own exception, the others a
    message naming it."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_failure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert "injected" in res[1] and "failed on rank(s) [1]" in res[0], res


def _replica_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from chicdiff_amd.dist import theta_grid_replicas
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stub = _StubGrid()
        dummy = torch.zeros(2, 2)
        out = theta_grid_replicas(stub, dummy, dummy, [1.0, 1.0], [0.0, 0.25, 0.5, 0.75, 1.0])
        q.put((rank, out.tolist(), stub.asked))
    finally:
        dist.destroy_process_group()


def test_theta_grid_replicas_every_point_once_and_every_total_everywhere():
    """theta grid in replica mode (chicdiff_amd.dist.theta_grid_replicas, world_size 2 over gloo, the fit replaced by a stub):
    point k is fitted by rank k mod world and by nobody else, one all-gather puts every total — a NaN one included, as NaN —
    on every rank."""
    import socket
    import torch.multiprocessing as mp
    from chicdiff_amd.dist import theta_replica_plan
    assert theta_replica_plan(5, 2) == [[0, 2, 4], [1, 3]] and theta_replica_plan(5, 8) == [[0], [1], [2], [3], [4], [], [], []]
    assert theta_replica_plan(5, 1) == [[0, 1, 2, 3, 4]]
    from chicdiff_amd.dist import theta_grid_replicas
    sharded = _StubGrid()
    sharded._sharded = True  # what HipContext.set_process_group / init_rccl leave behind
    with pytest.raises(ValueError, match="WITHOUT a process group"):
        theta_grid_replicas(sharded, None, None, [1.0], [0.0, 1.0])
    ctx = mp.get_context("spawn")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = [1000.0, 1000.0 + 7.0 / 16 - 0.75, 1000.0 + 7.0 / 4 - 1.5, float("nan"), 1004.0]
    assert res[0][2] == [[0.0, 0.5, 1.0]] and res[1][2] == [[0.25, 0.75]]
    for rank, out, _ in res:
        assert np.array_equal(np.array(out), np.array(want), equal_nan=True), (rank, out)
