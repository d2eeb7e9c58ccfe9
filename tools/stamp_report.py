#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timestamps of the fused kernel (needs the instrumented build of tools/make_acc_build.py:
MCALF_HIP_LIB=build/abl/stamps.so python tools/stamp_report.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcalf_amd
from mcalf_amd import _lib, workloads


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


kw, batch, seed = workloads.config(sys.argv[1] if len(sys.argv) > 1 else "B", synth)
if len(sys.argv) > 2:
    batch = int(sys.argv[2])
P = workloads.draw_P(kw, batch, np.random.default_rng(seed))
fit = mcalf_amd.als_fitter(None, **kw)
for _ in range(3):
    fit.loglike_batch(P)
lib = _lib.load()
n = min(batch * fit.info.ntiles, 8192)          # the diagnostic build records the first 8192 workgroups
buf = (C.c_ulonglong * (n * 8))()
lib.mcalf_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mcalf_diag_read_stamps(buf, n * 8) == 0
st = np.array(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
names = ["start", "item set-up (LDS writes)", "barrier", "component loop", "exp+flux store", "conv+terms+prefetch", "reduce"]
print("workgroups", n)
for k in range(1, 7):
    d = st[:, k] - st[:, k - 1]
    print("%-28s mean %9.0f  median %9.0f  max %9.0f cycles" % (names[k], d.mean(), np.median(d), d.max()))
# global timeline from s_memrealtime (100 MHz): stamps 0 (start) and 7 (end)
t0 = st[:, 0].min()
start = (st[:, 0] - t0) * 10e-3   # us
end = (st[:, 7] - t0) * 10e-3
life = end - start
print("kernel span %.1f us; WG lifetime mean %.1f us (min %.1f, max %.1f); sum(life)/512 slots = %.1f us"
      % (end.max(), life.mean(), life.min(), life.max(), life.sum() / 512))
print("start time percentiles [us]:", np.percentile(start, [0, 25, 50, 75, 100]).round(1).tolist())
print("end   time percentiles [us]:", np.percentile(end, [0, 25, 50, 75, 90, 99, 100]).round(1).tolist())
nb = np.resize(P[:, 3::3][:, :8].sum(axis=1), n)
ss = np.sort(start)
print("sorted start times [us] every 64th:", ss[::64].round(1).tolist())
o = np.argsort(start)
print("blockIdx of first 16 starters:", o[:16].tolist())
print("lifetimes of first-round (start<3us) mean %.1f, later mean %.1f; n_first=%d" % (life[start < 3].mean(), life[start >= 3].mean(), (start < 3).sum()))
xcd = np.arange(n) % 8
print("lifetime by blockIdx%8 (XCD group):", [round(float(life[xcd == k].mean()), 1) for k in range(8)])
first = start < 3 if (start < 3).any() and (start >= 3).any() else (np.arange(n) < n // 2)
print("round-1 lifetime mean %.1f std %.1f; round-2 mean %.1f std %.1f" % (life[first].mean(), life[first].std(), life[~first].mean(), life[~first].std()))
loop = (st[:, 3] - st[:, 2])
print("loop cycles by XCD group:", [int(loop[xcd == k].mean()) for k in range(8)])


try:
    dbg = (C.c_ulonglong * 4)()
    lib.mcalf_diag_read_dbg.argtypes = [C.c_void_p]
    lib.mcalf_diag_read_dbg(dbg)
    print("far-wing interpolation: %d of %d (line, segment) pairs interpolated (%.1f %%); segOk bits per 8 = %.2f"
          % (dbg[0], dbg[1], 100.0 * dbg[0] / max(1, dbg[1]), 8.0 * dbg[2] / max(1, dbg[1])))
except AttributeError:
    pass

try:
    acc = (C.c_ulonglong * (n * 32))()
    lib.mcalf_diag_read_acc.argtypes = [C.c_void_p, C.c_int]
    lib.mcalf_diag_read_acc(acc, n * 32)
    a = np.array(acc, dtype=np.uint64).reshape(n, 8, 4).astype(np.int64)
    for w in range(8):
        print("wave %d loop accounting [cycles]: fold %6.0f  barrier wait %6.0f  node pass %6.0f  direct %6.0f   (sum %6.0f)"
              % (w, a[:, w, 0].mean(), a[:, w, 1].mean(), a[:, w, 2].mean(), a[:, w, 3].mean(), a[:, w].sum(axis=1).mean()))
except AttributeError:
    pass
