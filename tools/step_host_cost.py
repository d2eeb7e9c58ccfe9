"""Diagnostic: host-side microseconds per bench step component (one rank, RCCL).
python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 tools/step_host_cost.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcalf_amd
from mcalf_amd import _lib, workloads, dist as mdist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29521")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


kw, batch, seed = workloads.config("B", synth)
P = workloads.draw_P(kw, batch, np.random.default_rng(seed))
fit = mcalf_amd.als_fitter(None, **kw)
_lib.check(fit._lib.mcalf_reserve(fit._ctx, batch), fit._ctx)
dP = torch.from_numpy(P).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
launch = fit._lib.mcalf_loglike_batch_device
out = torch.empty(batch, dtype=torch.float64, device=dev)
for nb in (8, 1024):
    for _ in range(20):
        launch(fit._ctx, dP.data_ptr(), nb, out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        launch(fit._ctx, dP.data_ptr(), nb, out.data_ptr(), st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("launch call, batch %4d: host %.1f us/call (30 calls into an empty queue)" % (nb, (t1 - t0) / 30 * 1e6), flush=True)

plan = mdist.LogLGather(batch, dev, depth=2, always_collective=True)
for _ in range(20):
    plan.local
    plan.gather_async()
plan.finish()
torch.cuda.synchronize()
t_local = t_gather = 0.0
for _ in range(200):
    a = time.perf_counter()
    o = plan.local
    b = time.perf_counter()
    plan.gather_async()
    c = time.perf_counter()
    t_local += b - a
    t_gather += c - b
plan.finish()
print("plan.local (wait on old work) %.1f us, plan.gather_async %.1f us" % (t_local / 200 * 1e6, t_gather / 200 * 1e6), flush=True)
dist.barrier()
dist.destroy_process_group()
