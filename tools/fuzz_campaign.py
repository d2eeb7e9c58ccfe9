#!/usr/bin/env python3
"""One-off wide fuzz (GPU box): many seeded random problems through the same generator the test suite
uses, plus tiny / odd pixel counts; prints every disagreement with the oracle instead of stopping.
python tools/fuzz_campaign.py [first_seed] [count]"""
import os
import sys
import time
import warnings

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import mcalf_amd
from mcalf_amd import workloads
from cases import problem_from_kwargs
from oracle import numpy_oracle as o
import test_gpu_fuzz as tg

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
warnings.simplefilter("ignore")
bad = refused = done = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kw = tg.random_problem(rng)
    if seed % 3 == 0:                                   # shrink to a tiny / awkward pixel count
        wl, fl, er = kw["spectrum"]
        m = min(int(rng.choice([1, 2, 3, 7, 31, 63, 64, 65, 127, 129, 511, 513])), wl.size)
        kw["spectrum"] = (wl[:m], fl[:m], er[:m])
        kw["fitrange"] = [[wl[0] - 1e-3, wl[m - 1] + 1e-3]]
    P = workloads.draw_P(kw, 6, rng)
    for mode in ("numpy", "jax"):
        try:
            prob = problem_from_kwargs(kw)
            fit = mcalf_amd.als_fitter(None, conv_mode=mode, **kw)
        except RuntimeError as exc:
            refused += 1
            if not (mode == "jax" and ("MCALF_ERR_RANGE" in str(exc) or "MCALF_ERR_INVALID" in str(exc))):
                bad += 1
                print("seed", seed, mode, "unexpected refusal:", exc, flush=True)
            continue
        with fit:
            got = fit.loglike_batch(P)
            m2 = fit.model_batch(P[:2])
        with np.errstate(all="ignore"):
            if mode == "numpy":
                want = o.loglike_batch(prob, P)
                ref = [o.reconstruct_spec(prob, p) for p in P[:2]]
            else:
                want = np.array([o.jax_loglike_f64(prob, p) for p in P])
                ref = [o.jax_reconstruct_spec_f64(prob, p) for p in P[:2]]
        done += 1
        ok = np.all((np.abs(got - want) < 1e-7 + 2e-9 * np.abs(want)) | (np.isnan(got) & np.isnan(want)) | (got == want))
        okm = all(np.nanmax(np.abs(a - b)) < 2e-10 and np.array_equal(np.isnan(a), np.isnan(b)) for a, b in zip(m2, ref))
        if not (ok and okm):
            bad += 1
            print("seed", seed, mode, "npix", kw["spectrum"][0].size, "ncomp", kw["ncomp"], "nfill", kw["nfill"],
                  "specres", kw["specres"], "velstep", kw["velstep"], "\n   got ", got, "\n   want", want,
                  "\n   model max diff", [float(np.nanmax(np.abs(a - b))) for a, b in zip(m2, ref)], flush=True)
    if (seed - first) % 25 == 24:
        print("... %d seeds, %d comparisons, %d refused, %d bad, %.0f s" % (seed - first + 1, done, refused, bad, time.time() - t0), flush=True)
print("DONE: %d comparisons, %d refused, %d bad" % (done, refused, bad))
sys.exit(1 if bad else 0)
