"""GPU: the streaming launch of the host-pointer entries is taken only on the device shape it was built for, and a call
is right on every other one.

`run_host_stream` (SURVEY 8(d)'s own metric: P from host memory, logL back) deals the rows of a batch to the eight XCDs of
an unpartitioned MI355X, and only workgroups that RUN on an XCD evaluate its rows.  On a DPX / QPX / CPX partition or on a
CU-masked stream some XCDs receive no workgroup.  The context therefore (1) asks the hardware which XCDs its stream
reaches (a probe kernel at mcalf_create / mcalf_set_cu_mask) and streams only on exactly eight, and (2) checks the
arrival counts after every streaming launch and fails over to the row-block pipeline when an XCD got none.  Both are
exercised here -- (2) through the TEST variant of the library, which can be made to believe a wrong probe answer
(MCALF_TEST_XCD_MASK; the product library has no such switch).  Reference: the solver hands over host arrays and trusts
the float it gets back (hires_fitter.py:250-262,287-294)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth, require_streaming_shape

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAM, PIPELINED = _lib.MCALF_PATH_HOST_STREAM, _lib.MCALF_PATH_HOST_PIPELINED


def _device_logl(fit, P):
    n = P.shape[0]
    dP = torch.from_numpy(P).cuda()
    out = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _xcd_words(xcds):
    """CU mask selecting the compute units of the given XCDs: consecutive mask bits go round the XCDs."""
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = np.zeros((ncu + 31) // 32, dtype=np.uint32)
    for cu in range(ncu):
        if cu % 8 in xcds:
            words[cu // 32] |= np.uint32(1 << (cu % 32))
    return words


def test_a_larger_batch_after_exactly_one_streamed_call_is_not_answered_by_the_old_completion_word():
    """The completion word of a streaming launch is its generation stamp in page-locked memory.  When the batch outgrows
    the streaming workspaces they are re-allocated -- and the generation count must NOT restart: after exactly one
    streamed call the word still holds 1, and a second launch stamped 1 again would be "complete" before it has written
    anything (the host would copy unwritten results out with rc 0).  Also with three sizes in a row and back."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 2100
    P = workloads.draw_P(kw, 3 * n, np.random.default_rng(seed + 4321))
    with mcalf_amd.als_fitter(None, **kw) as ref_fit:
        want = _device_logl(ref_fit, P)
    assert np.isfinite(want).all()
    for sizes in ((n, 2 * n), (n, 2 * n, 3 * n, n), (2 * n, n, 3 * n)):
        with mcalf_amd.als_fitter(None, **kw) as fit:           # a FRESH context per sequence: its first streamed call is call 1
            require_streaming_shape(fit)
            for m in sizes:
                out = np.full(m, np.nan)
                fit.loglike_batch(P[:m], out=out)
                ll = fit.last_launch()
                assert (ll.path, ll.stream_fallback, ll.xcd_mask) == (STREAM, 0, 0xFF), (sizes, m)
                assert np.array_equal(out, want[:m]), (sizes, m)


def test_cu_masked_context_gives_the_same_bits_and_never_streams_onto_xcds_it_cannot_reach():
    """The context's own stream restricted by CU masks (`mcalf_set_cu_mask`, as an application that shares a GPU between
    ranks does): every eighth CU (all of one XCD's in the runtime's numbering), those of four XCDs, the first 32 bits.
    Whatever the probe then sees decides the path -- fewer than eight XCDs: the row-block pipeline (`stream_fallback` =
    SHAPE); eight: the streaming launch (measured on ROCm 7.2 / MI355X: an XCD whose share of a mask is empty runs
    unrestricted, so a mask cannot take an XCD away; tools/explore/cu_mask_probe.py) -- and the bits are the device
    entry's either way; without the mask it streams again."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 2600
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 99))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        require_streaming_shape(fit)
        want = _device_logl(fit, P)
        assert np.array_equal(fit.loglike_batch(P), want)
        ll = fit.last_launch()
        assert (ll.path, ll.xcd_mask, ll.stream_fallback) == (STREAM, 0xFF, 0)
        for xcds in ({0}, {1, 3, 5, 7}, None):
            words = _xcd_words(xcds) if xcds else np.array([0xFFFFFFFF] + [0] * 7, dtype=np.uint32)
            try:
                fit.set_cu_mask(words)
            except RuntimeError as exc:
                pytest.skip(f"this runtime refuses CU-masked streams: {exc}")
            got = fit.loglike_batch(P[::-1].copy())[::-1]
            ll = fit.last_launch()
            assert np.array_equal(got, want), xcds
            assert ll.path in (STREAM, PIPELINED)
            seen = {i for i in range(16) if ll.xcd_mask >> i & 1}
            assert seen, "the probe saw no workgroup at all"
            if ll.xcd_mask != 0xFF:                           # (the mask did confine the stream: then it must not stream)
                assert (ll.path, ll.stream_fallback) == (PIPELINED, _lib.MCALF_STREAM_FALLBACK_SHAPE), xcds
                assert xcds is None or seen == xcds, (seen, xcds)     # consecutive CU-mask bits go round the XCDs
            else:
                assert (ll.path, ll.stream_fallback) == (STREAM, 0) and ll.stream_wgs_min >= 1
            assert np.array_equal(fit.chi2_batch(P), fit.chi2_batch(P[::-1].copy())[::-1])
            # small calls and model output are not affected; the row-block pipeline runs on streams created under the mask
            assert np.array_equal(fit.loglike_batch(P[:5]), want[:5])
            fit.set_chunks(3)
            assert np.array_equal(fit.loglike_batch(P), want) and fit.last_launch().path == PIPELINED
            fit.set_chunks(0)
        fit.set_cu_mask(None)
        assert np.array_equal(fit.loglike_batch(P), want)
        ll = fit.last_launch()
        assert (ll.path, ll.xcd_mask, ll.stream_fallback) == (STREAM, 0xFF, 0)
        with pytest.raises(RuntimeError, match="no compute unit"):
            fit.set_cu_mask(np.zeros(8, dtype=np.uint32))


def _worker(tmp_path, lib, env_extra, *extra):
    tag = "_".join("%s%s" % kv for kv in sorted(env_extra.items())) + "_".join(extra)
    out = os.path.join(str(tmp_path), "res_%s.json" % tag.replace("MCALF_TEST_", ""))
    env = dict(os.environ, MCALF_HIP_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MCALF_TEST_XCD_MASK", "MCALF_TEST_STARVE"):
        env.pop(k, None)
    env.update(env_extra)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_shape_worker.py"), out, "2600", *extra],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.load(open(out))


def test_a_launch_that_reports_an_xcd_without_workgroups_is_discarded(tmp_path, testing_lib):
    """The check BEHIND every streaming launch: the kernel counts the workgroups every XCD received, and a launch in which
    an XCD that was dealt rows got none is not an answer.  On the unpartitioned device of a test box no stream can be
    made to miss an XCD (see above), so the TEST variant of the library (a) is told that its stream reaches four XCDs
    only (MCALF_TEST_XCD_MASK=0x0F: no streaming launch is attempted, `stream_fallback` = SHAPE) and (b) is made to read
    the kernel's report as "an XCD got nothing" (MCALF_TEST_STARVE=1): the launch's results are discarded, the
    row-block pipeline answers with the device entry's bits (STARVED), and the context does not stream again.  The
    control run of the same library streams, and its kernel reports every XCD served."""
    ok = _worker(tmp_path, testing_lib, {})
    assert ok["finite"] and [c["path"] for c in ok["calls"]] == [STREAM, STREAM]
    assert all(c["equal"] and c["fallback"] == 0 and c["xcd_mask"] == 0xFF for c in ok["calls"])
    assert all(1 <= c["wgs_min"] <= c["wgs_max"] <= 512 for c in ok["calls"])        # counted by the kernel: every XCD was there
    four = _worker(tmp_path, testing_lib, {"MCALF_TEST_XCD_MASK": "0x0F"})
    assert [(c["path"], c["fallback"], c["xcd_mask"], c["equal"]) for c in four["calls"]] == \
        [(PIPELINED, _lib.MCALF_STREAM_FALLBACK_SHAPE, 0x0F, True)] * 2
    starved = _worker(tmp_path, testing_lib, {"MCALF_TEST_STARVE": "1"})
    first, second = starved["calls"]
    assert first["equal"] and second["equal"]                  # right bits whatever happened underneath
    assert (first["path"], first["fallback"]) == (PIPELINED, _lib.MCALF_STREAM_FALLBACK_STARVED), starved
    assert first["wgs_min"] == 0 and first["xcd_mask"] == 0     # ... and what the probe said is no longer believed:
    assert (second["path"], second["fallback"]) == (PIPELINED, _lib.MCALF_STREAM_FALLBACK_SHAPE)
    # one compute unit per XCD (the smallest mask that leaves every XCD one): streams, eight XCDs, same bits
    few = _worker(tmp_path, testing_lib, {}, "cus8")
    if few["cu_mask"] == "set":
        assert all(c["equal"] and c["path"] == STREAM and c["xcd_mask"] == 0xFF and c["wgs_min"] >= 1 for c in few["calls"]), few
