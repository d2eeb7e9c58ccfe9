"""Worker of tests/test_gpu_stream_shape.py, run with MCALF_HIP_LIB = the TEST variant of the library (only it reads
MCALF_TEST_XCD_MASK): one process, one context, config C's spectrum.

    python tests/stream_shape_worker.py <out.json> <n> [cus8]

Evaluates n rows through the device entry (reference bits) and twice through the host-pointer entry, and reports the path
the host calls took (the environment carries MCALF_TEST_XCD_MASK / MCALF_TEST_STARVE).  `cus8`: the context's own streams
are first restricted to eight compute units (mcalf_set_cu_mask: one per XCD)."""
import ctypes as C
import json
import os
import sys

out_path, n = sys.argv[1], int(sys.argv[2])
cus8 = len(sys.argv) > 3 and sys.argv[3] == "cus8"

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mcalf_amd  # noqa: E402
from mcalf_amd import _lib, workloads  # noqa: E402
from cases import oracle_synth  # noqa: E402

kw, _, seed = workloads.config("C", oracle_synth)
P = workloads.draw_P(kw, n, np.random.default_rng(seed + 1234))
res = {"lib": os.environ.get("MCALF_HIP_LIB"), "forced": os.environ.get("MCALF_TEST_XCD_MASK"), "starve": os.environ.get("MCALF_TEST_STARVE")}
with mcalf_amd.als_fitter(None, **kw) as fit:
    dP = torch.from_numpy(P).cuda()
    dout = torch.empty(n, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, dout.data_ptr(), st), fit._ctx)
    torch.cuda.synchronize()
    ref = dout.cpu().numpy()
    if cus8:
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        words = np.zeros((ncu + 31) // 32, dtype=np.uint32)
        words[0] = 0xFF                                   # consecutive mask bits go round the XCDs: one CU of each
        try:
            fit.set_cu_mask(words)
            res["cu_mask"] = "set"
        except RuntimeError as exc:
            res["cu_mask"] = "refused: %s" % exc
    calls = []
    for rep in range(2):
        got = fit.loglike_batch(P if rep == 0 else P[::-1].copy())
        ll = fit.last_launch()
        calls.append({"path": ll.path, "fallback": ll.stream_fallback, "xcd_mask": ll.xcd_mask, "row_blocks": ll.row_blocks,
                      "wgs_min": ll.stream_wgs_min, "wgs_max": ll.stream_wgs_max,
                      "equal": bool(np.array_equal(got if rep == 0 else got[::-1], ref))})
    res["calls"] = calls
    res["finite"] = bool(np.isfinite(ref).all())
with open(out_path, "w") as fh:
    json.dump(res, fh)
