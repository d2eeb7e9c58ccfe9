#!/usr/bin/env python3
"""A/B of two builds of the library: logL of the first N rows of configs C and E through the device entry, each build in
a process of its own (MCALF_HIP_LIB), compared bit for bit.   python tools/explore/ab_bits.py libA.so libB.so"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, ctypes as C
import numpy as np, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth
res = {}
for cfg, n in (("C", 4096), ("E", 1024), ("B", 1024)):
    kw, _, seed = workloads.config(cfg, oracle_synth)
    P = workloads.draw_P(kw, n, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        dP = torch.from_numpy(P).cuda(); out = torch.empty(n, dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        small = fit.loglike_batch(P[:7])               # the one-launch variant
        host = fit.loglike_batch(P)                    # streaming launch / pipeline
        res[cfg] = {"dev": out.cpu().numpy().view(np.uint64).tolist(), "small": small.view(np.uint64).tolist(), "host": host.view(np.uint64).tolist()}
json.dump(res, open(sys.argv[1], "w"))
'''
outs = []
for k, lib in enumerate(sys.argv[1:3]):
    path = "/tmp/ab_bits_%d.json" % k
    env = dict(os.environ, MCALF_HIP_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, path], env=env, check=True)
    outs.append(json.load(open(path)))
for cfg in outs[0]:
    for key in outs[0][cfg]:
        a, b = outs[0][cfg][key], outs[1][cfg][key]
        print(cfg, key, "rows", len(a), "bit-equal" if a == b else "DIFFERENT in %d rows" % sum(x != y for x, y in zip(a, b)))
    print(cfg, "within build A: host == dev", outs[0][cfg]["host"] == outs[0][cfg]["dev"], "; build B:", outs[1][cfg]["host"] == outs[1][cfg]["dev"])
