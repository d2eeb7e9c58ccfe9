"""GPU: the N > 1 branch of the in-library gather (mcalf_loglike_gatherv_device: grouped receives on the root, send
on the others, ragged counts, the exchange stream with and without overlap, the NaN-block error path) driven by TWO
processes on the one GPU of a box.  RCCL itself refuses two ranks on one device, so the transport is the test-only
stand-in tests/stubs/fake_rccl.cpp behind MCALF_RCCL_LIB -- asynchronous and stream-ordered like the real library, with
every receive landing 2 ms AFTER its message is there (FAKE_RCCL_DELAY_US), so that the gathered vectors are right only
if the library's stream / event chain around the exchange is: every step evaluates a different parameter matrix, and a
negative control (one local buffer for all overlapped steps, against the documented contract) must come out wrong.
This checks the library's control flow, offsets, ordering and error handling, NOT RCCL -- RCCL with more than one rank
runs only on the driver's multi-GPU node."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                    os.path.join(ROOT, "tests", "stubs", "fake_rccl.cpp"), "-lrt"], check=True, capture_output=True)
    return out


def _run(mode, work, fake, lib=None):
    env = dict(os.environ, MCALF_RCCL_LIB=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_DELAY_US="2000")
    if lib:
        env["MCALF_HIP_LIB"] = lib                        # (the test variant of the library: failure injection)
    env.pop("MCALF_TEST_FAIL_PREFLIGHT", None)
    env.pop("FAKE_RCCL_SYNC", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gather_worker.py"), str(r), "2", work, mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append((p.returncode, se[-1500:]))
    assert all(rc == 0 for rc, _ in outs), outs
    return [json.load(open(os.path.join(work, f"rank{r}.json"))) for r in range(2)]


@pytest.fixture(scope="module")
def expected():
    """logL of the eight parameter matrices the worker's steps evaluate, from one process."""
    kw, _, seed = workloads.config("C", oracle_synth)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return [fit.loglike_batch(workloads.draw_P(kw, 1001, np.random.default_rng(seed + 99 + v))) for v in range(8)]


def _vectors(r0):
    """step variant -> gathered vector, for the steps whose gather the worker reads back."""
    return {0: r0["plain"][0], 1: r0["plain"][1], 2: r0["overlap_2_3"][0], 3: r0["overlap_2_3"][1],
            6: r0["overlap_6_7"][0], 7: r0["overlap_6_7"][1]}


def test_two_ranks_ragged_gather_equals_the_single_process_result(tmp_path, fake_rccl, expected):
    r0, r1 = _run("ok", str(tmp_path), fake_rccl)
    assert r0["codes"] == [0] * 8 and r1["codes"] == [0] * 8
    assert r0["comm"] == [2, 0] and r1["comm"] == [2, 1]
    assert not np.array_equal(expected[6], expected[7])
    for v, got in _vectors(r0).items():                              # 501 + 500 rows, bit for bit, every step its own values
        assert np.array_equal(np.array(got), expected[v]), v


def test_negative_control_one_local_buffer_in_overlap_mode_shows(tmp_path, fake_rccl, expected):
    """Against the contract of overlap mode the worker hands every step the SAME local buffer: step 6's block leaves rank 1
    only after step 7's kernels have rewritten it.  The stand-in transport must be asynchronous enough to show that --
    otherwise the test above could not tell a missing event wait from a present one."""
    r0, r1 = _run("racy", str(tmp_path), fake_rccl)
    assert r0["codes"] == [0] * 8 and r1["codes"] == [0] * 8
    got = _vectors(r0)
    assert np.array_equal(np.array(got[7]), expected[7])             # the last step is right in any case ...
    assert not np.array_equal(np.array(got[6])[501:], expected[6][501:])   # ... the one before it is not


def test_a_rank_that_fails_locally_sends_nans_and_nobody_hangs(tmp_path, fake_rccl, expected, testing_lib):
    r0, r1 = _run("fail", str(tmp_path), fake_rccl, testing_lib)
    assert r0["codes"] == [0] * 8                                    # the root is fine ...
    assert r1["codes"] == [_lib.MCALF_ERR_NOMEM] * 8                 # ... rank 1 reports its failure every time
    assert "NaN" in r1["err"] and "MCALF_TEST_FAIL_PREFLIGHT" in r1["err"]
    for v, got in _vectors(r0).items():
        g = np.array(got)
        assert np.array_equal(g[:501], expected[v][:501]) and np.isnan(g[501:]).all(), v


def test_argument_errors_return_a_code_and_the_context_destroys_cleanly():
    """One-rank communicator, real RCCL library: a count that contradicts batch_local, a missing dlogL_all on the
    root and a bad root are refused before anything is enqueued; the communicator stays usable afterwards."""
    import ctypes as C
    import torch
    kw, _, seed = workloads.config("C", oracle_synth)
    P = workloads.draw_P(kw, 64, np.random.default_rng(seed + 5))
    dP = torch.from_numpy(P).cuda()
    out = torch.zeros(64, dtype=torch.float64, device="cuda")
    full = torch.zeros(64, dtype=torch.float64, device="cuda")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = fit.loglike_batch(P)
        lib, ctx = fit._lib, fit._ctx
        buf = C.create_string_buffer(_lib.MCALF_COMM_ID_BYTES)
        _lib.check(lib.mcalf_comm_unique_id(buf))
        _lib.check(lib.mcalf_comm_init(ctx, buf, 1, 0), ctx)
        bad = (C.c_int64 * 1)(63)
        assert lib.mcalf_loglike_gatherv_device(ctx, dP.data_ptr(), 64, out.data_ptr(), full.data_ptr(), bad, 0, None) == _lib.MCALF_ERR_INVALID
        assert lib.mcalf_loglike_gatherv_device(ctx, dP.data_ptr(), 64, out.data_ptr(), None, None, 0, None) == _lib.MCALF_ERR_INVALID
        assert lib.mcalf_loglike_gatherv_device(ctx, dP.data_ptr(), 64, out.data_ptr(), full.data_ptr(), None, 1, None) == _lib.MCALF_ERR_INVALID
        assert lib.mcalf_loglike_gatherv_device(ctx, dP.data_ptr(), -1, out.data_ptr(), full.data_ptr(), None, 0, None) == _lib.MCALF_ERR_INVALID
        good = (C.c_int64 * 1)(64)
        _lib.check(lib.mcalf_loglike_gatherv_device(ctx, dP.data_ptr(), 64, out.data_ptr(), full.data_ptr(), good, 0, None), ctx)
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), want) and np.array_equal(out.cpu().numpy(), want)
        _lib.check(lib.mcalf_comm_destroy(ctx), ctx)


def _bench_two_ranks(fake, *extra):
    env = dict(os.environ, MCALF_RCCL_LIB=fake, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "D",
           "--batch", "4608", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", *extra]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_n2_times_the_three_gathers_in_one_process_group(fake_rccl):
    """`bench.py --gpus 2` (gloo rehearsal on the one GPU, the library's exchange over the stand-in transport): the torch
    gather, the library's exchange on the launch stream and the library's exchange on its side stream are timed one after
    the other in the ONE process group, each with its gather check and the ranks' kernel times; the best is `value`."""
    res, out = _bench_two_ranks(fake_rccl)
    assert res.returncode == 0, res.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and list(out["gathers"]) == ["torch", "inlib", "inlib_overlap"]
    for name, g in out["gathers"].items():
        assert isinstance(g, dict), (name, g)
        assert g["gather_check"] == {"rows": 4608, "own_block_equal": True, "all_finite": True}, name
        assert g["ms_per_step"] > 0 and 0 < g["kernel_ms_min_over_ranks"] <= g["kernel_ms_max_over_ranks"]
        # the library's legs count their ranks through the library's own communicator
        assert (g["rccl_ranks"], g["rccl_ranks_source"]) == (2, "torch.distributed.get_world_size" if name == "torch" else "mcalf_comm_info")
        sp = g["rank0_split"]
        assert sp["kernels_ms"] > 0 and sp["exchange_ms"] >= 0 and sp["join_ms"] >= 0
        assert sp["step_ms_synchronous"] == pytest.approx(sp["kernels_ms"] + sp["exchange_ms"] + sp["join_ms"])
    best = min(out["gathers"], key=lambda k: out["gathers"][k]["ms_per_step"])
    assert out["gather_reported"] == best and out["ms_per_step"] == pytest.approx(out["gathers"][best]["ms_per_step"])
    assert out["value"] == pytest.approx(out["gathers"][best]["value"])


def test_bench_watchdog_prints_what_was_measured_and_exits_nonzero(fake_rccl):
    """A library-gather leg that does not finish in time (here: a limit no leg can meet) is recorded as "timed_out"; the
    line carries the torch leg that was already measured; the process exits non-zero -- nobody waits for a hung rank."""
    res, out = _bench_two_ranks(fake_rccl, "--leg-timeout", "0.0001")
    assert res.returncode != 0
    assert out is not None and out["gathers"]["inlib"] == "timed_out" and out["gather_reported"] == "torch"
    assert out["gathers"]["torch"]["gather_check"]["own_block_equal"] is True and "did not finish" in out["error"]
