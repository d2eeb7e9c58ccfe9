"""Diagnostic: host-side cost per call of the collectives LogLGather could use (one rank, RCCL).
Launch: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 tools/coll_overhead.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29519")
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
n = 1024
send = torch.zeros(n, dtype=torch.float64, device=dev)
recv = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(world)]
flat = torch.empty(n * world, dtype=torch.float64, device=dev)


def timeit(name, fn, reps=2000):
    for _ in range(50):
        w = fn()
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    works = []
    for _ in range(reps):
        works.append(fn())
        if len(works) > 2:
            w = works.pop(0)
            if w is not None:
                w.wait()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-40s host %.1f us/call, with drain %.1f us/call" % (name, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6), flush=True)


timeit("gather async", lambda: dist.gather(send, recv, dst=0, async_op=True))
timeit("all_gather_into_tensor async", lambda: dist.all_gather_into_tensor(flat, send, async_op=True))
timeit("all_gather (list) async", lambda: dist.all_gather(recv, send, async_op=True))
timeit("reduce async (sum into rank 0)", lambda: dist.reduce(send, dst=0, async_op=True))
dist.barrier()
dist.destroy_process_group()
