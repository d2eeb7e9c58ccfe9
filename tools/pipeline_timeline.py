#!/usr/bin/env python3
"""Diagnostic: the GPU-side timeline of one pipelined host-pointer call (MCALF_HOST_TRACE=2 prints it when the context is
destroyed).   python tools/pipeline_timeline.py <config> <pageable|pinned> [rows] [calls]"""
import os
import sys
import time

os.environ.setdefault("MCALF_HOST_TRACE", "2")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcalf_amd
from mcalf_amd import workloads


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


cfg, kind = sys.argv[1], sys.argv[2]
kw, batch, seed = workloads.config(cfg, synth)
rows = int(sys.argv[3]) if len(sys.argv) > 3 else batch
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 12
P = np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)[:rows])
out = np.empty(rows)
if kind == "pinned":
    P = torch.from_numpy(P).pin_memory().numpy()
    out = torch.empty(rows, dtype=torch.float64).pin_memory().numpy()
with mcalf_amd.als_fitter(None, **kw) as fit:
    for _ in range(3):
        fit.loglike_batch(P, out=out)
    t0 = time.perf_counter()
    for _ in range(calls):
        fit.loglike_batch(P, out=out)
    dt = (time.perf_counter() - t0) / calls * 1e3
    print("%s %s %d rows: %.4f ms per call, %s" % (cfg, kind, rows, dt, fit.get_config()), file=sys.stderr)
