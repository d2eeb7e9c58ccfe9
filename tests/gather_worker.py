"""Worker of tests/test_gpu_gather_two_ranks.py: ONE rank of a two-process job on the one GPU of a box, driving
mcalf_comm_init + mcalf_loglike_gatherv_device directly through the C ABI.  The transport is the test-only
stand-in tests/stubs/fake_rccl.cpp (MCALF_RCCL_LIB); the 128-byte id travels through a file.

    python tests/gather_worker.py <rank> <world> <dir> <ok|fail|racy>

Every step evaluates a DIFFERENT parameter matrix (variant k = draw k of tests' generator), so that a block that travels
late -- after a later step's kernels have rewritten the buffer it is read from -- or lands in the wrong pair shows.
`racy` breaks the documented contract of overlap mode on purpose (ONE local buffer for all steps instead of two
alternating ones): the negative control that proves the stand-in transport is asynchronous enough to show such a fault.
"""
import ctypes as C
import json
import os
import sys
import time

rank, world, work, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
if mode == "fail" and rank == 1:
    os.environ["MCALF_TEST_FAIL_PREFLIGHT"] = "1"          # this rank's workspace growth "fails" in every call

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mcalf_amd  # noqa: E402
from mcalf_amd import _lib, workloads  # noqa: E402
from mcalf_amd.dist import shard_bounds, shard_counts  # noqa: E402
from cases import oracle_synth  # noqa: E402

kw, _, seed = workloads.config("C", oracle_synth)
batch = 1001                                              # ragged over two ranks: 501 + 500
NVAR = 8
lo, hi = shard_bounds(batch, world, rank)
n = hi - lo
counts = (C.c_int64 * world)(*shard_counts(batch, world))
fit = mcalf_amd.als_fitter(None, **kw)
lib, ctx = fit._lib, fit._ctx
idfile = os.path.join(work, "id.bin")
if rank == 0:
    buf = C.create_string_buffer(_lib.MCALF_COMM_ID_BYTES)
    _lib.check(lib.mcalf_comm_unique_id(buf))
    with open(idfile + ".tmp", "wb") as fh:
        fh.write(buf.raw)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60, "rank 0 never wrote the id"
        time.sleep(0.01)
    buf = C.create_string_buffer(open(idfile, "rb").read(), _lib.MCALF_COMM_ID_BYTES)
_lib.check(lib.mcalf_comm_init(ctx, buf, world, rank), ctx)
dP = [torch.from_numpy(np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed + 99 + v))[lo:hi])).cuda()
      for v in range(NVAR)]
local = [torch.full((n,), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
full = [torch.full((batch,), 7.0, dtype=torch.float64, device="cuda") if rank == 0 else None for _ in range(2)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
codes = []


def step(v, k, kl=None):
    """Variant v into buffer pair k (local buffer kl)."""
    kl = k if kl is None else kl
    return lib.mcalf_loglike_gatherv_device(ctx, dP[v].data_ptr(), n, local[kl].data_ptr(),
                                            full[k].data_ptr() if full[k] is not None else None, counts, 0, st)


def gathered():
    return [full[k].cpu().numpy().tolist() for k in range(2)] if rank == 0 else None


# default: every call ends with the stream behind its exchange -- the SAME pair may be reused at once, and a copy
# enqueued behind a call sees that call's gather
codes.append(step(0, 0))
first = full[0].clone() if rank == 0 else None
codes.append(step(1, 0))
torch.cuda.synchronize()
res = {"plain": [first.cpu().numpy().tolist(), full[0].cpu().numpy().tolist()] if rank == 0 else None}
_lib.check(lib.mcalf_comm_set_overlap(ctx, 1), ctx)      # overlap: two buffer pairs, explicit join
one_local = mode == "racy"
for v in (2, 3):
    codes.append(step(v, v & 1, 0 if one_local else None))
_lib.check(lib.mcalf_comm_join(ctx, st), ctx)
torch.cuda.synchronize()
res["overlap_2_3"] = gathered()
for v in (4, 5, 6, 7):                                    # pairs reused while their previous exchange may still be in flight
    codes.append(step(v, v & 1, 0 if one_local else None))
_lib.check(lib.mcalf_comm_join(ctx, st), ctx)
torch.cuda.synchronize()
res["overlap_6_7"] = gathered()
res["codes"] = codes
res["err"] = lib.mcalf_last_error(ctx).decode()
nr, rk = C.c_int32(), C.c_int32()
_lib.check(lib.mcalf_comm_info(ctx, C.byref(nr), C.byref(rk)), ctx)
res["comm"] = [nr.value, rk.value]
_lib.check(lib.mcalf_comm_destroy(ctx), ctx)
fit.close()
with open(os.path.join(work, f"rank{rank}.json"), "w") as fh:
    json.dump(res, fh)
