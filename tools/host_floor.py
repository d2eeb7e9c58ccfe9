#!/usr/bin/env python3
"""Diagnostic (GPU box): what a synchronous host-pointer step can cost at best.

One config (default C).  Per variant the median of repeated passes of K steps:
  device_async   mcalf_loglike_batch_device back to back, one sync per pass (what bench.py's `value` times)
  device_sync    the same entry with a stream synchronise after EVERY step: launch latency + completion wake-up
                 of a step whose inputs never move -- the floor of any host-pointer entry
  host_pageable  mcalf_loglike_batch from pageable numpy arrays
  host_pinned    mcalf_loglike_batch from page-locked arrays
  cube_host      mcalf_loglike_cube_batch from pageable unit cubes: prior transform while the rows are decoded, theta
                 formed on the host under the launch (MCALF_STREAM=0: the staged form, theta back from the device)
  memcpy         a plain numpy copy of P into a page-locked block (what one host thread needs to stage the batch)
  zero_copy_read the set-up + fused kernels reading P straight from a page-locked, device-mapped block (device entry
                 given the mapped pointer): PCIe reads by the kernels instead of a copy command
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcalf_amd  # noqa: E402
from mcalf_amd import _lib, workloads  # noqa: E402


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--passes", type=int, default=9)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    kw, batch, seed = workloads.config(args.config, synth)
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if args.config == "E" else 0)
    dev = torch.device("cuda", 0)
    fit = mcalf_amd.als_fitter(None, **kw)
    _lib.check(fit._lib.mcalf_reserve(fit._ctx, batch), fit._ctx)
    dP = torch.from_numpy(P).to(dev)
    out_d = torch.empty(batch, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    st = C.c_void_p(stream.cuda_stream)
    launch = fit._lib.mcalf_loglike_batch_device
    ctx = fit._ctx
    P_pin_t = torch.from_numpy(P).pin_memory()
    out_pin_t = torch.empty(batch, dtype=torch.float64).pin_memory()
    P_pin, out_pin = P_pin_t.numpy(), out_pin_t.numpy()
    out_host = np.empty(batch)
    stage = torch.empty(P.shape, dtype=torch.float64).pin_memory().numpy()
    cubes = np.random.default_rng(seed + 1).random(P.shape)

    def dev_async():
        launch(ctx, dP.data_ptr(), batch, out_d.data_ptr(), st)

    def dev_sync():
        launch(ctx, dP.data_ptr(), batch, out_d.data_ptr(), st)
        stream.synchronize()

    def zero_copy():
        # torch's page-locked allocations are device-mapped on ROCm: the host address is valid on the device
        launch(ctx, P_pin_t.data_ptr(), batch, out_pin_t.data_ptr(), st)
        stream.synchronize()

    variants = {
        "device_async": dev_async,
        "device_sync": dev_sync,
        "host_pageable": lambda: fit.loglike_batch(P, out=out_host),
        "host_pinned": lambda: fit.loglike_batch(P_pin, out=out_pin),
        "cube_host": lambda: fit.loglike_cube_batch(cubes),                       # unit cubes in, (theta, logL) out
        "cube_host_no_theta": lambda: fit.loglike_cube_batch(cubes, return_theta=False),
        "memcpy": lambda: np.copyto(stage, P),
        "zero_copy_read": zero_copy,
    }
    res = {}
    for name, fn in variants.items():
        if args.only and name not in args.only.split(","):
            continue
        try:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            times = []
            for _ in range(args.passes):
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / args.steps * 1e3)
            times.sort()
            ll = fit.last_launch()
            res[name] = {"ms_per_step_median": times[len(times) // 2], "min": times[0], "max": times[-1],
                         "path": ll.path, "row_blocks": ll.row_blocks, "stream_wgs": ll.stream_setup_wgs, "polled": ll.stream_polled}
            if name == "host_pageable":
                res[name]["bit_equal_to_device"] = bool(np.array_equal(out_host, out_d.cpu().numpy()))
            if name == "host_pinned":
                res[name]["bit_equal_to_device"] = bool(np.array_equal(out_pin, out_d.cpu().numpy()))
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": str(e)}
        print(name, res[name], flush=True)
    if "zero_copy_read" in res and "error" not in res["zero_copy_read"]:
        ref = out_d.cpu().numpy()
        res["zero_copy_bit_equal"] = bool(np.array_equal(out_pin, ref))
    ll = fit.last_launch()
    res["last_launch"] = {"path": ll.path, "row_blocks": ll.row_blocks, "persistent": ll.persistent}
    print(json.dumps(res))
    fit.close()


if __name__ == "__main__":
    main()
