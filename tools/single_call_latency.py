#!/usr/bin/env python3
"""Diagnostic: latency of the one-theta-at-a-time callables a PolyChord / dynesty style solver uses."""
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


for cfg in (sys.argv[1:] or ["A", "B", "E"]):
    kw, _, seed = workloads.config(cfg, synth)
    P = workloads.draw_P(kw, 256, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        for p in P[:20]:
            fit.lnlhood_dy(p)
        t0 = time.perf_counter()
        for p in P:
            fit.lnlhood_dy(p)
        dt_first = (time.perf_counter() - t0) / len(P)       # the process's first few hundred calls (round 3 reported this one)
        for _ in range(8):                                   # ~2000 more calls: the clocks of an idle GPU take that long to come up
            for p in P:
                fit.lnlhood_dy(p)
        t0 = time.perf_counter()
        for p in P:
            fit.lnlhood_dy(p)
        dt = (time.perf_counter() - t0) / len(P)
        print("config %s: lnlhood_dy %.1f us per call over the context's FIRST %d calls, %.1f us after 2000 more" % (cfg, dt_first * 1e6, len(P), dt * 1e6))
        for b in (1, 8, 64):
            fit.loglike_batch(P[:b])
            t1 = time.perf_counter()
            for _ in range(50):
                fit.loglike_batch(P[:b])
            print("config %s batch %3d: %.1f us per call" % (cfg, b, (time.perf_counter() - t1) / 50 * 1e6))
        print("config %s: lnlhood_dy %.1f us per call (%.0f logL/s), tiles/sample %d" % (cfg, dt * 1e6, 1 / dt, fit.info.ntiles))
