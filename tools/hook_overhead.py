import sys, time, os, numpy as np
sys.path.insert(0, '.')
import torch, torch.distributed as dist
from chicdiff_amd import hip, synth
n, S = 2_000_000, 8
d = synth.make(n, S)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1)
ctx = hip.HipContext(0)
dk = ctx.to_device(d["counts"], np.int32); dfm = ctx.to_device(d["nf"] * (d["mu"][:, None] / S), np.float64)
out = {}
def bench(tag):
    for _ in range(2): ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    torch.cuda.synchronize(); print(tag, "ms/step", (time.perf_counter() - t0) / 10 * 1e3)
    ctx.enable_timing(True); ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    print("   ", {k: round(v[0], 3) for k, v in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][0])}); ctx.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(10): ctx.wald_test(dk, dfm, d["group"], theta=0.5, outputs=out)
    print("    host enqueue ms/step", (time.perf_counter() - t0) / 10 * 1e3); torch.cuda.synchronize()
bench("no hook")
ctx.set_process_group()
bench("torch hook (1-rank RCCL)")
print("collectives per step", ctx._hook.calls / 23, "doubles per step", ctx._hook.doubles / 23)
ctx.init_rccl()
bench("direct RCCL (1-rank communicator owned by the library)")
dist.destroy_process_group()
