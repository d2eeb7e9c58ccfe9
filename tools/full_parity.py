#!/usr/bin/env python3
"""GPU box: logL of EVERY row of BASELINE configs B, C, D (all 32768 rows), E against the plain-C/OpenMP oracle (and the numpy/scipy
oracle on a sample), written as one JSON object.  The rows go through the host-pointer entry (pageable arrays: the path a
solver's arrays take -- recorded per config) AND through the device entry bench.py times; the two must agree bit for bit.
python tools/full_parity.py > gpurun_out/full_parity.json"""
import json
import os
import sys
import time

import ctypes as C

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import mcalf_amd
from mcalf_amd import _lib, build, workloads
from cases import problem_from_kwargs
from oracle import c_oracle, numpy_oracle as o


def synth(kw, p):
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, float))


out = {}
for cfg in (sys.argv[1:] or ["B", "C", "D", "E"]):
    kw, batch, seed = workloads.config(cfg, synth)
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
        host_path = fit.last_launch().path
        dP = torch.from_numpy(P).cuda()
        dout = torch.empty(batch, dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream()
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), batch, dout.data_ptr(), C.c_void_p(st.cuda_stream)), fit._ctx)
        st.synchronize()
        ll = fit.last_launch()
        got_dev = dout.cpu().numpy()
        models = fit.model_batch(P[:64])
    t0 = time.time()
    want = c_oracle.COracle(prob, threads=min(16, os.cpu_count() or 1)).loglike_batch(P)
    tc = time.time() - t0
    k = min(batch, 256)
    wnp = o.loglike_batch(prob, P[:k])
    mref = np.array([o.reconstruct_spec(prob, p) for p in P[:64]])
    d = np.abs(got - want)
    names = {_lib.MCALF_PATH_HOST_ZEROCOPY: "zero-copy small call", _lib.MCALF_PATH_HOST_PIPELINED: "row-block pipeline",
             _lib.MCALF_PATH_HOST_STREAM: "one streaming launch", _lib.MCALF_PATH_HOST_STAGED: "staged"}
    out[cfg] = {"rows": int(batch), "host_entry_path": names.get(host_path, str(host_path)),
                "device_entry": {"persistent": bool(ll.persistent), "ordered": bool(ll.ordered), "grid": int(ll.grid)},
                "host_entry_bit_equal_to_device_entry": bool(np.array_equal(got, got_dev, equal_nan=True)),
                "max_abs_dlogL_vs_c_oracle": float(d.max()),
                "max_rel_dlogL_vs_c_oracle": float((d / np.maximum(1.0, np.abs(want))).max()),
                "rows_beyond_1e-4": int((d > 1e-4).sum()), "c_oracle_seconds": round(tc, 1),
                "max_abs_dlogL_vs_numpy_oracle_first_%d" % k: float(np.abs(got[:k] - wnp).max()),
                "max_abs_dflux_first_64_models": float(np.abs(models - mref).max()),
                "max_rel_dflux_first_64_models": float((np.abs(models - mref) / np.maximum(np.abs(mref), 1e-300)).max()),
                "logL_range": [float(got.min()), float(got.max())]}
    print(cfg, out[cfg], file=sys.stderr, flush=True)
out["kernel_source_hash"] = build.source_hash()
print(json.dumps(out, indent=1))
