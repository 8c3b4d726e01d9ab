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
