#!/usr/bin/env python3
"""Timing of the wide-LSF path (an LSF whose halo does not fit a workgroup tile: host_abi.cpp launch_wide, kernels.hip
mcalf_wide_taps_kernel / mcalf_wide_conv_kernel) on the three geometries of tests/test_gpu_wide_lsf.py: ms per call of
loglike_batch for 64 and 1024 live points, and what that is per live point and per tap x pixel.
    python tools/wide_lsf_timing.py        (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mcalf_amd
from mcalf_amd import workloads
from test_gpu_wide_lsf import _problem

for npix, velstep, specres, contval, nfill in [(1500, 0.004, (8.0,), (1.0,), 0), (333, 0.0031, (6.0, 9.0), (0.9, 1.1), 2),
                                               (4500, 0.0045, (8.0, 8.5), (1.0,), 0)]:
    kw = _problem(npix, velstep, specres, contval, nfill, seed=npix)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        n_cap = fit.info.n_cap
        for rows in (64, 1024):
            P = workloads.draw_P(kw, rows, np.random.default_rng(rows))
            out = np.empty(rows)
            for _ in range(3):
                fit.loglike_batch(P, out=out)
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                fit.loglike_batch(P, out=out)
            ms = (time.perf_counter() - t0) / reps * 1e3
            R = P[:, 0] if len(specres) > 1 else np.full(rows, specres[0])
            taps = 2 * np.ceil(3.0348 * (R / 2.354820) / velstep) + 1
            work = float(taps.sum()) * npix                      # tap x pixel products per call
            print("npix %4d, provisioned half-width %4d px (taps per live point %.0f .. %.0f), %4d live points: %.3f ms per call, %.2f us per live point, "
                  "%.1f G tap-pixel products / s" % (npix, n_cap, taps.min(), taps.max(), rows, ms, ms * 1e3 / rows, work / ms / 1e6))
