#!/usr/bin/env python3
"""Diagnostic: per-call durations of the pipelined host-pointer entry on page-locked rows (config E), in passes of 20 calls
bracketed by torch.cuda.synchronize() as bench.py times them -- is the 'slow mode' (+0.35 ms per step) ONE 7 ms call per pass?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import mcalf_amd
from mcalf_amd import workloads

bench.config_leg("B", None, steps=20)                       # (the leg that precedes E's in the runs that showed the slow mode)
kw, batch, seed = workloads.config("E", bench.hip_synth)
P = np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2))
with mcalf_amd.als_fitter(None, **kw) as fit:
    out = np.empty(batch)
    for _ in range(3):
        fit.loglike_batch(P, out=out)
    P_pin = torch.from_numpy(P).pin_memory().numpy()
    out_pin = torch.empty(batch, dtype=torch.float64).pin_memory().numpy()
    for kind, (a, b) in (("pageable", (P, out)), ("pinned", (P_pin, out_pin)), ("pinned rows, pageable results", (P_pin, out)),
                         ("pageable rows, pinned results", (P, out_pin))):
        fit.loglike_batch(a, out=b)
        for p in range(3):
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                fit.loglike_batch(a, out=b)
                ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            print("%-30s pass %d: mean %.4f  median %.4f  max %.4f at call %d   first three %s" % (
                kind, p, np.mean(ts), np.median(ts), max(ts), int(np.argmax(ts)), " ".join("%.3f" % t for t in ts[:3])))
    print(fit.get_config())
