#!/usr/bin/env python3
"""Soak (GPU box): the resident evaluator's leave / restart protocol under gaps that straddle its idle limit -- every
answer compared with the launched form's bits.   python tools/soak_resident.py [seconds] [idle_us]"""
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


seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
idle = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kw, _, seed = workloads.config("A", synth)
P = workloads.draw_P(kw, 64, np.random.default_rng(seed + 31))
rng = np.random.default_rng(5)
with mcalf_amd.als_fitter(None, **kw) as fit:
    want = [fit.lnlhood_dy(p) for p in P]
    fit.set_resident(idle)
    gaps = np.array([0, 0, 0, 0.2, 0.5, 0.8, 0.9, 0.95, 1.0, 1.05, 1.1, 1.3, 2.0, 5.0]) * idle * 1e-6
    n = bad = 0
    t_end = time.perf_counter() + seconds
    t_report = time.perf_counter() + 10
    while time.perf_counter() < t_end:
        k = int(rng.integers(64))
        g = float(gaps[int(rng.integers(gaps.size))])
        if g > 0:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < g:
                pass
        got = fit.lnlhood_dy(P[k])
        n += 1
        if got != want[k]:
            bad += 1
            print("MISMATCH call", n, "row", k, got, want[k], flush=True)
        if n % 97 == 0:                                   # other entries in between
            b = fit.loglike_batch(P[:8])
            if list(b) != want[:8]:
                bad += 1
                print("BATCH MISMATCH at call", n, flush=True)
        if time.perf_counter() > t_report:
            print("...", n, "calls,", bad, "bad", flush=True)
            t_report += 10
    import ctypes as C
    print("DONE: %d calls, %d bad, idle limit %d us" % (n, bad, idle))
sys.exit(1 if bad else 0)
