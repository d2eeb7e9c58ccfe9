# GPU box: wide-LSF timings + multi-device overhead on one GPU (entries [0], [0,0], [0,0,0] -- NOT a multi-GPU measurement)
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out
timeout -k 10 300 python3 tools/wide_lsf_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/wide_lsf.txt
timeout -k 10 300 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/multi_one_gpu.txt
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, mcalf_amd
from mcalf_amd import workloads
for cfg in ("C", "E"):
    kw, batch, seed = workloads.config(cfg, bench.hip_synth)
    P = np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if cfg == "E" else 0))
    ref = None
    for devs in (0, [0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
        with mcalf_amd.als_fitter(None, device=devs, **kw) as fit:
            out = np.empty(batch)
            for _ in range(3):
                fit.loglike_batch(P, out=out)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); fit.loglike_batch(P, out=out); ts.append((time.perf_counter() - t0) * 1e3)
            if ref is None:
                ref = out.copy()
            print("config %s, %5d rows, device=%-12s median %.4f ms per call (min %.4f), devices_used %d, bit-equal to the single context: %s" % (
                cfg, batch, devs, np.median(ts), min(ts), fit.last_launch().devices_used, np.array_equal(out, ref)))
PY
