#!/usr/bin/env python3
"""Diagnostic: timeline of ONE persistent launch of the fused kernel (needs the instrumented build of tools/make_acc_build.py:
MCALF_HIP_LIB=build/abl/stamps.so python tools/timeline_report.py C 4096).  Every work item leaves its start and end
(s_memrealtime, 100 MHz) and the workgroup slot that ran it; from these: ramp, tail, idle share of the slots, the
spread of item durations, how many items each slot took."""
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


cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
kw, batch, seed = workloads.config(cfg, synth)
if len(sys.argv) > 2:
    batch = int(sys.argv[2])
order = sys.argv[3] if len(sys.argv) > 3 else "asdrawn"        # asdrawn | sorted (rows sorted by ncomp, descending)
P = workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)
fit = mcalf_amd.als_fitter(None, **kw)
if order == "sorted":
    P = P[np.argsort(-P[:, fit.startind], kind="stable")]
dP = torch.from_numpy(P).cuda()
out = torch.empty(batch, dtype=torch.float64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = _lib.load()
for _ in range(4):
    _lib.check(lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), batch, out.data_ptr(), st), fit._ctx)
torch.cuda.synchronize()
nitems = batch * fit.info.ntiles
n = min(nitems, 8192)
buf = (C.c_ulonglong * (n * 8))()
lib.mcalf_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mcalf_diag_read_stamps(buf, n * 8) == 0
s = np.array(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
t0 = s[:, 0].min()
start, end, slot = (s[:, 0] - t0) * 0.01, (s[:, 7] - t0) * 0.01, s[:, 6]      # microseconds
dur = end - start
span = end.max()
slots = np.unique(slot)
print(f"config {cfg}, {batch} rows ({order}), {nitems} items ({n} recorded) on {slots.size} slots; kernel span {span:.1f} us")
print("item duration [us]: mean %.2f  std %.2f  min %.2f  max %.2f" % (dur.mean(), dur.std(), dur.min(), dur.max()))
first = np.array([start[slot == b].min() for b in slots])
last = np.array([end[slot == b].max() for b in slots])
cnt = np.array([(slot == b).sum() for b in slots])
busy = np.array([dur[slot == b].sum() for b in slots])
print("slot first start [us]: mean %.2f  max %.2f" % (first.mean(), first.max()))
print("slot last end    [us]: mean %.2f  min %.2f  percentiles 10/50/90: %s" % (last.mean(), last.min(), np.percentile(last, [10, 50, 90]).round(1).tolist()))
print("items per slot: min %d  max %d  histogram %s" % (cnt.min(), cnt.max(), np.bincount(cnt).tolist()))
print("slot time budget %.1f us: busy %.1f %%  ramp %.1f %%  tail %.1f %%  gaps between items %.1f %%"
      % (span, 100 * busy.mean() / span, 100 * first.mean() / span, 100 * (span - last).mean() / span,
         100 * (last - first - busy).mean() / span))
nc = P[:, fit.startind].astype(int)
if fit.info.ntiles == 1:
    for v in np.unique(nc):
        m = nc[:n] == v
        print("  ncomp %2d: %5d items, duration mean %.2f us" % (v, m.sum(), dur[m].mean()))
    k = max(1, n // 8)
    for q in range(8):
        sel = np.argsort(start)[q * k:(q + 1) * k]
        print("  items starting in octile %d of the launch: duration mean %.2f us" % (q, dur[sel].mean()))
fit.close()
