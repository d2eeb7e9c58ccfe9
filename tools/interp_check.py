#!/usr/bin/env python3
"""Diagnostic: model spectra with and without the far-wing interpolation (two builds), written to .npy.
usage: MCALF_HIP_LIB=<lib> python tools/interp_check.py out.npy"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mcalf_amd
from mcalf_amd import workloads

out = []
rng = np.random.default_rng(5)
# strong / narrow / damped single lines on fine and coarse grids, plus the BASELINE configs
for step in (0.2, 0.5, 1.2, 3.0):
    wl = 4862.0 * np.exp(np.arange(12000) * step / 2.9979245e5)
    kw = dict(fitrange=[[wl[0] - 1, wl[-1] + 1]], fitlines=["HI 1215"], linepars=workloads.HI, ncomp=[3, 3],
              specres=[6.0], Nrange=[12.0, 21.5], brange=[2.0, 120.0],
              zrange=[wl[2000] / 1215.67 - 1, wl[-2000] / 1215.67 - 1],
              spectrum=(wl, np.ones_like(wl), np.full_like(wl, 0.02)), velstep=step)
    P = workloads.draw_P(kw, 24, rng)
    P[:8, 1::3][:, :3] = rng.uniform(19.0, 21.5, (8, 3))      # damped
    P[8:16, 3::3][:, :3] = rng.uniform(2.0, 6.0, (8, 3))      # narrow
    with mcalf_amd.als_fitter(None, **kw) as fit:
        m = fit.model_batch(P)
        tau = -np.log(np.maximum(m, 1e-300))
        out.append(m.ravel())
np.save(sys.argv[1], np.concatenate(out))
