#!/usr/bin/env python3
"""Which XCDs does a CU-masked stream reach?  For a handful of mask shapes: the XCD set the context's probe kernel saw
(mcalf_last_launch().xcd_mask) and the time of one 2600-row host call (a mask that is honoured makes it slower)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mcalf_amd  # noqa: E402
from mcalf_amd import workloads  # noqa: E402
from cases import oracle_synth  # noqa: E402

kw, _, seed = workloads.config("C", oracle_synth)
P = workloads.draw_P(kw, 2600, np.random.default_rng(seed))
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nw = (ncu + 31) // 32


def words(cus):
    w = np.zeros(nw, dtype=np.uint32)
    for c in cus:
        w[c // 32] |= np.uint32(1 << (c % 32))
    return w


shapes = {
    "none": None,
    "cu 0": [0],
    "cus 0..7": range(8),
    "cus 0..31": range(32),
    "cus 32..63": range(32, 64),
    "every 8th": range(0, ncu, 8),
    "every 8th + 1": range(1, ncu, 8),
    "cus 0..127": range(128),
    "even": range(0, ncu, 2),
}
with mcalf_amd.als_fitter(None, **kw) as fit:
    ref = fit.loglike_batch(P)
    for name, cus in shapes.items():
        try:
            fit.set_cu_mask(None if cus is None else words(cus))
        except RuntimeError as exc:
            print(f"{name:16s} refused: {exc}")
            continue
        fit.loglike_batch(P)
        t0 = time.perf_counter()
        for _ in range(5):
            got = fit.loglike_batch(P)
        dt = (time.perf_counter() - t0) / 5
        ll = fit.last_launch()
        print(f"{name:16s} xcd_mask {ll.xcd_mask:#06x} path {ll.path} fallback {ll.stream_fallback}  {dt * 1e3:8.3f} ms  equal {np.array_equal(got, ref)}", flush=True)
