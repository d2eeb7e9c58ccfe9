"""GPU: the row-block issue (mcalf_set_chunks) and the pipelined host-pointer entry change nothing in the
results -- every live point is evaluated independently (shards == whole, bit for bit)."""
import ctypes as C

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth, problem_from_kwargs

pytestmark = pytest.mark.gpu


def _device_logl(fit, dP, n):
    out = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cfg,n", [("C", 1031), ("B", 517), ("E", 67)])
def test_row_blocks_do_not_change_results(cfg, n):
    kw, _, seed = workloads.config(cfg, oracle_synth)
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 31), damped=2 if cfg == "E" else 0)
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, n), fit._ctx)
        fit.set_chunks(1)
        assert fit.chunks_for(n) == 1
        whole = _device_logl(fit, dP, n)
        assert np.isfinite(whole).all()
        for k in (2, 3, 8):
            fit.set_chunks(k)
            assert fit.chunks_for(n) == k
            for _ in range(2):                      # twice: the second call reuses streams, events, workspaces
                assert np.array_equal(_device_logl(fit, dP, n), whole)
        fit.set_chunks(0)
        assert fit.chunks_for(n) == 1
        assert np.array_equal(_device_logl(fit, dP, n), whole)
        # host-pointer entry at these sizes: the zero-copy small path (batch * ndim <= 65536 doubles); the pipelined
        # path -- pageable / page-locked, every block plan -- is covered by tests/test_gpu_timed_path.py
        for k in (0, 1, 2, 5):
            fit.set_chunks(k)
            assert np.array_equal(fit.loglike_batch(P), whole)
            assert np.array_equal(fit.chi2_batch(P), fit.chi2_batch(P[::-1].copy())[::-1])
        Ppin = torch.from_numpy(P).pin_memory().numpy()
        opin = torch.empty(n, dtype=torch.float64).pin_memory().numpy()
        fit.set_chunks(0)
        fit.loglike_batch(Ppin, out=opin)
        assert np.array_equal(opin, whole)
        # a batch smaller than the requested block count
        fit.set_chunks(8)
        assert np.array_equal(fit.loglike_batch(P[:3]), whole[:3])
        assert np.array_equal(_device_logl(fit, dP, 5), whole[:5])


def test_chunked_call_replays_inside_a_hip_graph():
    kw, _, seed = workloads.config("C", oracle_synth)
    P = workloads.draw_P(kw, 600, np.random.default_rng(seed + 5))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        fit.set_chunks(1)
        eager = fit.loglike_batch(P)
        fit.set_chunks(3)
        dP = torch.from_numpy(P).cuda()
        out = torch.zeros(600, dtype=torch.float64, device="cuda")
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, 600), fit._ctx)
        _device_logl(fit, dP, 600)                  # creates the auxiliary streams / events outside the capture
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), 600, out.data_ptr(), st), fit._ctx)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), eager)


def test_resolution_beyond_the_provisioned_kernel_is_not_a_valid_likelihood():
    """R > specres_max: the reference would build a longer kernel; here the row is flagged -- model NaN,
    logL -inf, chi2 (+inf, []) -- instead of the -0.0 an all-NaN model gives through nansum."""
    from oracle import numpy_oracle as orc
    kw, _, seed = workloads.config("C", oracle_synth)
    P = workloads.draw_P(kw, 4, np.random.default_rng(seed))
    P[1, 0] = 30.0                                  # far outside [8, 9] km/s
    with mcalf_amd.als_fitter(None, **kw) as fit:
        logL = fit.loglike_batch(P)
        want = orc.loglike_batch(problem_from_kwargs(kw), P[[0, 2, 3]])
        assert logL[1] == -np.inf
        assert np.abs(logL[[0, 2, 3]] - want).max() < 1e-4
        assert fit.chi2_batch(P)[1] == np.inf
        assert np.isnan(fit.model_batch(P[1:2])).all()
        with pytest.raises(ValueError, match="specres"):
            fit.reconstruct_onecomp(30.0, 1.0, 13.5, 3.0, 20.0)
