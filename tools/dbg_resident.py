#!/usr/bin/env python3
"""Diagnostic (GPU box): the resident one-theta evaluator against the launched form -- bits, latency, idle exit."""
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
    P = workloads.draw_P(kw, 512, np.random.default_rng(seed + 7))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = np.array([fit.lnlhood_dy(p) for p in P])
        t0 = time.perf_counter()
        for p in P:
            fit.lnlhood_dy(p)
        t_launch = (time.perf_counter() - t0) / len(P) * 1e6
        for idle in (200, 2000):
            fit.set_resident(idle)
            got = np.array([fit.lnlhood_dy(p) for p in P])
            t0 = time.perf_counter()
            for p in P:
                fit.lnlhood_dy(p)
            t_res = (time.perf_counter() - t0) / len(P) * 1e6
            time.sleep(0.01)                                   # longer than the idle limit: the kernel has left
            again = np.array([fit.lnlhood_dy(p) for p in P[:8]])
            batch = fit.loglike_batch(P)                        # other entries while the evaluator is alive
            ll = fit.last_launch()
            print("config %s idle %4d us: launched %.2f us per call, resident %.2f us; bit-equal %s / after idle %s / batch %s"
                  % (cfg, idle, t_launch, t_res, np.array_equal(got, want), np.array_equal(again, want[:8]), np.array_equal(batch, want)), flush=True)
            fit.set_resident(0)
        back = np.array([fit.lnlhood_dy(p) for p in P[:8]])
        print("   off again: bit-equal", np.array_equal(back, want[:8]))
