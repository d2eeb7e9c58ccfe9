#!/usr/bin/env python3
"""Diagnostic (GPU box): what of a one-theta call is the Python wrapper -- lnlhood_dy against the bare C entry through ctypes."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcalf_amd
from mcalf_amd import workloads


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


for cfg in (sys.argv[1:] or ["A", "B"]):
    kw, _, seed = workloads.config(cfg, synth)
    P = workloads.draw_P(kw, 64, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        p = P[0].copy()
        out = np.empty(1)
        fn, ctx, pp, po = fit._lib.mcalf_loglike_batch, fit._ctx, p.ctypes.data, out.ctypes.data
        for _ in range(3000):
            fit.lnlhood_dy(p)
        res = {}
        for name, f in (("lnlhood_dy", lambda: fit.lnlhood_dy(p)), ("lnlhood_pc", lambda: fit.lnlhood_pc(p)),
                        ("bare C entry through ctypes", lambda: fn(ctx, pp, 1, po))):
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(2000):
                    f()
                t.append((time.perf_counter() - t0) / 2000 * 1e6)
            res[name] = sorted(t)[2]
        print("config %s:" % cfg, ", ".join("%s %.2f us" % kv for kv in res.items()))
