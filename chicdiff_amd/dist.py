"""Row sharding and the sum-all-reduce hook.

Interactions are independent except for a handful of global scalars, so a fit shards by
contiguous row blocks, one process per GPU, and the library asks its host for a sum-all-reduce
whenever it has per-rank partial sums (nf column means, trend IRLS sums, radix-select
histograms, deviance sums).  This module turns a ``torch.distributed`` process group into the
C callback of ``include/chicdiff_hip.h`` (``chicdiff_allreduce_fn``).

The buffer the library passes is device memory on the GPU path (backend ``nccl`` = RCCL over
xGMI) and host memory in the CPU test harness (backend ``gloo``); ``memory`` says which
(``device_via_host`` stages a device buffer through the host so that gloo can carry it).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)  # chicdiff_allgather_fn: user, send, recv, count


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row block [start, stop) of rank `rank`: the first n % world ranks get one extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def theta_replica_plan(ntheta: int, world: int) -> list[list[int]]:
    """Which grid points each rank fits in replica mode: point k goes to rank k mod world (round robin, so that with
    world >= ntheta every rank fits at most one)."""
    if world < 1 or ntheta < 0:
        raise ValueError("bad world/ntheta")
    return [[k for k in range(ntheta) if k % world == r] for r in range(world)]


def theta_grid_replicas(ctx, d_counts, d_fullmean, size_factors, thetas, group=None, opts=None) -> np.ndarray:
    """The theta grid of DESeq2Wrap (chicdiff.R:1641-1647: one design-~1 fit per mixing parameter, the total deviance of each)
    as a REPLICA problem: every rank holds ALL rows (`d_counts`, `d_fullmean` are the complete matrices on its own GPU), rank r
    fits the points k = r, r + world, ... on its own, and ONE all-gather of ntheta doubles puts every total on every rank.
    With world >= len(thetas) the grid costs one fit instead of len(thetas).

    `ctx` must be a context WITHOUT a process group attached (its fits are complete, nothing is sharded); the fits are the ones
    `ctx.theta_grid` makes.  A rank that fits ONE point runs the single-launch trend kernel (with the MAD of the residuals in the
    same launch), a single-process grid of several points runs its fits side by side with the trend as one launch per pass: the
    same sums in another order, so the totals agree with the single-process grid's to the trend's summation order (~1e-13
    relative; tests/test_gpu_sharded.py asserts 1e-10), not bit for bit.  The alternative when rows ARE sharded — len(thetas)
    sharded fits one after the other on a context with `set_process_group` — needs no copy of the rows and is what `DESeq2Wrap`
    uses; this function is for callers whose ranks hold the whole interaction set.

    A fit that fails on ONE rank must not leave the others inside the all-gather: the local exception is caught, a status flag
    travels with the totals, and every rank raises."""
    import torch
    import torch.distributed as dist

    if getattr(ctx, "_sharded", False):
        raise ValueError("theta_grid_replicas needs a context WITHOUT a process group (its fits are complete, every rank holds all rows); "
                         "on a sharded context call ctx.theta_grid: the grid's fits then run one after the other, each sharded")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    thetas = [float(t) for t in thetas]
    mine = theta_replica_plan(len(thetas), world)[rank]
    part = np.zeros(len(thetas) + 1, dtype=np.float64)  # the totals, then this rank's status (0 = fine)
    failure = None
    if mine:
        try:
            part[mine] = ctx.theta_grid(d_counts, d_fullmean, size_factors, [thetas[k] for k in mine], opts=opts)
        except Exception as e:  # noqa: BLE001 — whatever it is, the peers must hear of it
            failure = e
            part[-1] = 1.0
    # every point is fitted by exactly one rank: gather the ranks' vectors and pick each point from its owner (a sum of zero-filled
    # vectors would do as well, but a NaN total — a region with zero counts in every sample — must arrive as NaN, not poison the rest)
    backend = str(dist.get_backend(group))
    t = torch.from_numpy(part)
    if "nccl" in backend and getattr(d_counts, "is_cuda", False):  # (a mixed "cpu:gloo,cuda:nccl" group carries device tensors over RCCL)
        t = t.to(d_counts.device)
    got = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(got, t, group=group)
    bad = [r for r in range(world) if float(got[r][-1]) != 0.0]
    if bad:
        if failure is not None:
            raise failure
        raise RuntimeError(f"theta_grid_replicas: the fit failed on rank(s) {bad}")
    out = np.empty(len(thetas), dtype=np.float64)
    for k in range(len(thetas)):
        out[k] = float(got[k % world][k])
    return out


class _RawDevice:
    """Raw device pointer exposed to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class AllReduceHook:
    """Owns the ctypes callback (keep the object alive as long as the library may call it)."""

    def __init__(self, group=None, memory: str = "device", device=None):
        import torch
        import torch.distributed as dist

        if memory not in ("device", "host", "device_via_host"):
            raise ValueError("memory must be 'device', 'host' or 'device_via_host'")
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.calls = 0
        self.doubles = 0
        self.error = None

        views = {}  # (ptr, count) -> tensor view; the library reuses a few workspace buffers for every fit

        def cb(_user, ptr, count):
            try:
                count = int(count)
                if memory == "device":
                    t = views.get((ptr, count))
                    if t is None:
                        if len(views) > 256:
                            views.clear()
                        t = views[(ptr, count)] = torch.as_tensor(_RawDevice(int(ptr), count), device=device)
                elif memory == "device_via_host":
                    # device buffer, CPU-side transport (gloo): several ranks may share one GPU this way,
                    # which RCCL does not allow — used to test the sharded HIP path on a single-GPU box
                    t = torch.as_tensor(_RawDevice(int(ptr), count), device=device)
                    h = t.cpu()  # synchronises with the stream the library enqueued on (torch's current stream)
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(h)
                    self.calls += 1
                    self.doubles += count
                    return 0
                else:
                    buf = (C.c_double * count).from_address(int(ptr))
                    t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64))  # shares the memory
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                self.calls += 1
                self.doubles += count
                return 0
            except Exception as e:  # an exception must not cross the C boundary
                self.error = e
                return 1

        self.fn = ALLREDUCE_FN(cb)
        self.gathers = 0
        self.gather_doubles = 0

        def gather_cb(_user, send, recv, count):
            """chicdiff_allgather_fn: `count` doubles from every rank, rank r's block at recv + r * count."""
            try:
                count = int(count)
                if memory == "host":
                    s_ = torch.from_numpy(np.frombuffer((C.c_double * count).from_address(int(send)), dtype=np.float64))
                    r_ = torch.from_numpy(np.frombuffer((C.c_double * (count * self.world)).from_address(int(recv)), dtype=np.float64))
                    dist.all_gather_into_tensor(r_, s_, group=group)
                else:
                    s_ = torch.as_tensor(_RawDevice(int(send), count), device=device)
                    r_ = torch.as_tensor(_RawDevice(int(recv), count * self.world), device=device)
                    if memory == "device_via_host":  # gloo carries host tensors
                        hs = s_.cpu()
                        hr = torch.empty(count * self.world, dtype=torch.float64)
                        dist.all_gather_into_tensor(hr, hs, group=group)
                        r_.copy_(hr)
                    else:
                        dist.all_gather_into_tensor(r_, s_, group=group)
                self.gathers += 1
                self.gather_doubles += count
                return 0
            except Exception as e:
                self.error = e
                return 1

        self.gather_fn = ALLGATHER_FN(gather_cb)
