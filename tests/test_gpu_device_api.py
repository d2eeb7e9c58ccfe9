"""GPU: the *_device entry points fed with torch tensors on torch's stream (the path bench.py
and a vectorised sampler use) agree bit for bit with the host-pointer entries."""
import ctypes as C

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth, require_streaming_shape

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


def test_loglike_replays_inside_a_hip_graph():
    """After mcalf_reserve the *_device entry allocates nothing and synchronises nothing, so it can be
    captured into a hipGraph (torch.cuda.CUDAGraph) and replayed with new parameters in place."""
    kw, _, seed = workloads.config("C", oracle_synth)
    rng = np.random.default_rng(seed + 9)
    P1, P2 = workloads.draw_P(kw, 257, rng), workloads.draw_P(kw, 257, rng)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        eager1, eager2 = fit.loglike_batch(P1), fit.loglike_batch(P2)
        dP = torch.from_numpy(P1).cuda()
        out = torch.zeros(257, dtype=torch.float64, device="cuda")
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, 257), fit._ctx)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), 257, out.data_ptr(), st), fit._ctx)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), eager1)
        dP.copy_(torch.from_numpy(P2))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), eager2)


def test_context_lifecycle_and_capacity_error():
    kw, _, seed = workloads.config("A")
    P = workloads.draw_P(kw, 3, np.random.default_rng(seed))
    first = None
    for _ in range(40):                                   # create / destroy must not leak or cross-talk
        with mcalf_amd.als_fitter(None, **kw) as fit:
            v = fit.loglike_batch(P)
        first = v if first is None else first
        assert np.array_equal(v, first)
    a = mcalf_amd.als_fitter(None, **kw)
    b = mcalf_amd.als_fitter(None, **dict(kw, specres=[20.0]))
    assert not np.array_equal(a.loglike_batch(P), b.loglike_batch(P))    # independent contexts
    a.close(); a.close(); b.close()
    # an LSF that cannot fit a workgroup tile is taken by the wide-LSF path under BOTH boundary modes (round 5: numpy; round 6:
    # JAX semantics too -- tests/test_gpu_wide_lsf.py); the context reports the half-width it provisions
    wl = 6200.0 * np.exp(np.arange(12000) * 0.01 / 2.9979245e5)
    args = ([[wl[0] - 1, wl[-1] + 1]], ["CIV 1548"], [1, 1])
    wide = dict(specres=[20.0], spectrum=(wl, np.ones_like(wl), np.full_like(wl, 0.02)), velstep=0.01)
    for mode in ("jax", "numpy"):
        with mcalf_amd.als_fitter(None, *args, conv_mode=mode, **wide) as fit:
            assert 2 * fit.info.n_cap > 4096
            assert fit.info.ntiles == 3                      # (the fused stage tiles the 12000 pixels without a halo)
            if mode == "numpy":
                assert fit.info.n_cap == int(np.ceil(3.0348 * (20.0 / 2.354820) / 0.01))


def test_cube_in_logl_out_matches_two_step_path_and_oracle():
    """mcalf_loglike_cube_batch == lnlhood_pc(_scale_cube_pc(cube)) row by row (hires_fitter.py:202-209,
    250-262): theta bit-equal to the numpy transform, logL bit-equal to the two-step device path and within
    the parity tolerance of the oracle evaluated on the numpy-transformed rows."""
    from cases import problem_from_kwargs
    from oracle import numpy_oracle as orc
    kw, _, seed = workloads.config("C", oracle_synth)
    rng = np.random.default_rng(seed + 100)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        cubes = rng.random((257, fit.ndim))
        cubes[0, :] = 0.0                               # box corners
        cubes[1, :] = np.nextafter(1.0, 0.0)
        theta, logL = fit.loglike_cube_batch(cubes)
        want_theta = np.array([fit._scale_cube_pc(c.copy()) for c in cubes])
        assert np.array_equal(theta, want_theta)
        assert np.array_equal(logL, fit.loglike_batch(want_theta))
        only = fit.loglike_cube_batch(cubes, return_theta=False)
        assert np.array_equal(only, logL)
        # MultiNest flavour: the ncomp slot stays fractional in theta, int() happens in the decode
        theta_mn, logL_mn = fit.loglike_cube_batch(cubes, int_ncomp=False)
        assert np.array_equal(theta_mn, np.array([fit._scale_cube_mn(c.copy(), fit.ndim, fit.ndim) for c in cubes]))
        assert np.array_equal(logL_mn, logL)
        # device-pointer flavour on torch's stream, without theta
        dc = torch.from_numpy(cubes).cuda()
        out = torch.empty(cubes.shape[0], dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fit._lib.mcalf_loglike_cube_batch_device(fit._ctx, dc.data_ptr(), cubes.shape[0], None,
                                                            out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), logL_mn)
    prob = problem_from_kwargs(kw)
    want = orc.loglike_batch(prob, np.array([orc.scale_cube_pc(prob, c) for c in cubes[:40]]))
    assert np.max(np.abs(logL[:40] - want)) < 1e-4


@pytest.mark.parametrize("cfg,n,path", [("C", 4096, "stream"), ("C", 700, "small"), ("E", 1400, "stream"), ("E", 13200, "staged")])
def test_cube_host_entry_takes_the_fast_paths_of_the_theta_entry(cfg, n, path):
    """The host-pointer cube entry goes the ways of mcalf_loglike_batch -- ONE streaming launch for a large batch of a
    single-tile spectrum and for a tiled one up to 65536 work items, the zero-copy small call below 512 KB of parameters
    (completion read off the results), staged copies for a larger tiled batch -- with the prior transform applied while the rows are decoded and theta formed on the
    host under the launch: theta bit-equal to the numpy transform, logL bit-equal to the two-step path, both int() flavours."""
    kw, _, seed = workloads.config(cfg, oracle_synth)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        if path == "stream":
            require_streaming_shape(fit)
        cubes = np.random.default_rng(seed + 300).random((n, fit.ndim))
        for int_ncomp in (True, False):
            theta, logL = fit.loglike_cube_batch(cubes, int_ncomp=int_ncomp)
            ll = fit.last_launch()
            assert ll.path == {"stream": _lib.MCALF_PATH_HOST_STREAM, "small": _lib.MCALF_PATH_HOST_ZEROCOPY,
                               "staged": _lib.MCALF_PATH_HOST_STAGED}[path]
            if path != "staged":
                assert ll.stream_polled == 1
            want_theta = (np.array([fit._scale_cube_pc(c.copy()) for c in cubes]) if int_ncomp else
                          np.array([fit._scale_cube_mn(c.copy(), fit.ndim, fit.ndim) for c in cubes]))
            assert np.array_equal(theta, want_theta)
            assert np.array_equal(logL, fit.loglike_batch(np.array([fit._scale_cube_pc(c.copy()) for c in cubes])))
            assert np.array_equal(fit.loglike_cube_batch(cubes, int_ncomp=int_ncomp, return_theta=False), logL)


def test_cube_entry_requires_a_prior():
    kw, _, _ = workloads.config("A")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        cube = np.full((1, fit.ndim), 0.5)
        out = np.empty(1)
        pd = C.POINTER(C.c_double)
        rc = fit._lib.mcalf_loglike_cube_batch(fit._ctx, cube.ctypes.data_as(pd), 1, None, out.ctypes.data_as(pd))
        assert rc == _lib.MCALF_ERR_INVALID
        assert b"mcalf_set_prior" in fit._lib.mcalf_last_error(fit._ctx)


def test_batch_pool_turns_a_sampler_map_into_one_device_call():
    from mcalf_amd import adapters
    kw, _, seed = workloads.config("A")
    P = workloads.draw_P(kw, 64, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        pool = adapters.BatchPool(fit, size=64)
        got = pool.map(fit.lnlhood_dy, list(P))
        assert got == [float(v) for v in fit.loglike_batch(P)] and pool.batched_calls == 1
        assert got[:3] == [fit.lnlhood_dy(p) for p in P[:3]]
        loglike, transform = adapters.batch_functions(fit)
        cubes = np.random.default_rng(3).random((16, fit.ndim))
        assert np.array_equal(loglike(transform(cubes)), fit.loglike_cube_batch(cubes)[1])


def test_loglike_batch_fills_a_caller_buffer_including_page_locked_memory():
    kw, _, seed = workloads.config("A")
    P = workloads.draw_P(kw, 33, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = fit.loglike_batch(P)
        out = np.full(33, np.nan)
        assert fit.loglike_batch(P, out=out) is out and np.array_equal(out, want)
        Pp = torch.from_numpy(P).pin_memory().numpy()
        outp = torch.empty(33, dtype=torch.float64).pin_memory().numpy()
        fit.loglike_batch(Pp, out=outp)
        assert np.array_equal(outp, want)
        for bad in (np.empty(32), np.empty(33, dtype=np.float32), np.empty(66)[::2]):
            with pytest.raises(ValueError):
                fit.loglike_batch(P, out=bad)
