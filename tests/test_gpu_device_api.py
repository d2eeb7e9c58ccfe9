"""GPU: the *_device entry points fed with torch tensors on torch's stream (the path bench.py
and a vectorised sampler use) agree bit for bit with the host-pointer entries."""
import ctypes as C

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth

pytestmark = pytest.mark.gpu


def test_device_pointer_entries_match_host_entries():
    kw, _, seed = workloads.config("C", oracle_synth)
    P = workloads.draw_P(kw, 300, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        host = fit.loglike_batch(P)
        hmodel = fit.model_batch(P[:5], targonly=True)
        dP = torch.from_numpy(P).cuda()
        out = torch.empty(P.shape[0], dtype=torch.float64, device="cuda")
        dm = torch.empty((5, fit.obj_wl.size), dtype=torch.float64, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(fit._lib.mcalf_reserve(fit._ctx, P.shape[0]), fit._ctx)
            for _ in range(3):
                _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), P.shape[0], out.data_ptr(), st),
                           fit._ctx)
            _lib.check(fit._lib.mcalf_model_batch_device(fit._ctx, dP.data_ptr(), 5, 1, dm.data_ptr(), st), fit._ctx)
        side.synchronize()
        assert np.array_equal(out.cpu().numpy(), host)
        assert np.array_equal(dm.cpu().numpy(), hmodel)
