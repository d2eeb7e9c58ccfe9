"""GPU: an LSF wider than a workgroup tile.  The fused kernel convolves inside a 4096-pixel LDS tile, halo included; the
reference simply builds a longer kernel (hires_fitter.py:458-464, astropy `convolve(..., boundary='wrap')`).  Round 5:
such a context is no longer refused (MCALF_ERR_RANGE) -- the fused kernel runs without convolution and continuum, and two
more kernels form every live point's taps and convolve periodically (host_abi.cpp: launch_wide; round 6: flux window
and taps staged through LDS, taps accumulated -- and `bot` formed -- in tap order as in the fused kernel, so the suite's
usual bars hold here too) -- same entry points, same conventions.  Checked against the numpy oracle: kernels several times longer than the spectrum (the window
wraps round it more than once), free resolution and continuum, fillers, logL / chi2 / model / single components / unit-cube
input, host and device entries, a resolution beyond the provisioned maximum."""
import ctypes as C

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import problem_from_kwargs
from oracle import numpy_oracle as o

pytestmark = pytest.mark.gpu

CIV = [(1548.204, 0.1899, 2.643e8), (1550.781, 0.09475, 2.628e8)]


def _problem(npix, velstep, specres, contval=(1.0,), nfill=0, seed=0):
    rng = np.random.default_rng(seed)
    wl = np.linspace(6180.0, 6220.0, npix + 2)[1:-1]
    flux = 1 + rng.normal(0, 0.03, npix)
    err = rng.uniform(0.01, 0.05, npix)
    return dict(fitrange=[[6180.0, 6220.0]], fitlines=["CIV 1548", "CIV 1550"], linepars=CIV, ncomp=[1, 3], nfill=nfill,
                specres=list(specres), contval=list(contval), Nrange=[12.0, 14.5], brange=[5.0, 40.0], zrange=[2.995, 3.012],
                spectrum=(wl, flux, err), velstep=float(velstep))


@pytest.mark.parametrize("npix,velstep,specres,contval,nfill", [
    (1500, 0.004, (8.0,), (1.0,), 0),            # n = 2578: wider than a tile, 1.7 x the spectrum on each side
    (333, 0.0031, (6.0, 9.0), (0.9, 1.1), 2),    # free resolution and continuum, fillers; the window wraps ~ 22 times
    (4500, 0.0045, (8.0, 8.5), (1.0,), 0),       # two pixel tiles in the fused stage, n up to 2438
])
def test_wide_lsf_context_matches_the_oracle(npix, velstep, specres, contval, nfill):
    kw = _problem(npix, velstep, specres, contval, nfill, seed=npix)
    prob = problem_from_kwargs(kw)
    rng = np.random.default_rng(7 + npix)
    P = workloads.draw_P(kw, 6, rng)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        n_cap = fit.info.n_cap
        assert 2 * n_cap + 64 > 4096                            # the case the fused kernel's tile cannot hold
        assert n_cap == int(np.ceil(3.0348 * (max(specres) / 2.354820) / velstep))
        got = fit.loglike_batch(P)
        want = o.loglike_batch(prob, P)
        assert np.all(np.abs(got - want) < 1e-10 * np.abs(want) + 1e-9), (got, want)
        for targ in (False, True):
            m = fit.model_batch(P[:2], targonly=targ)
            for a, p in zip(m, P[:2]):
                assert np.abs(a - o.reconstruct_spec(prob, p, targonly=targ)).max() < 1e-11
        # an absorber-free model convolves to EXACTLY the continuum (taps and `bot` are added up in the same order):
        # columns of 1e-30 cm^-2 leave tau == 0
        flat = P[:1].copy()
        flat[0, fit.startind + 1::3] = -30.0
        cont = flat[0, 1 if fit.freespecres else 0] if fit.freecont else float(contval[0])
        assert np.array_equal(fit.model_batch(flat)[0], np.full(npix, 1.0 * cont))
        chi2 = fit.chi2_batch(P)
        want_chi2 = np.array([o.chi2(prob, p) for p in P])
        assert np.allclose(chi2, want_chi2, rtol=1e-11, atol=1e-9)
        # the reference's one-theta callables and the single-component spectra
        assert fit.lnlhood_pc(P[3])[0] == got[3] and fit.lnlhood_dy(P[4]) == got[4]
        s = fit.startind
        one = fit.reconstruct_onecomp(max(specres), 0.95, P[0][s + 1], P[0][s + 2], P[0][s + 3])
        assert np.abs(one - o.reconstruct_onecomp(prob, max(specres), 0.95, P[0][s + 1], P[0][s + 2], P[0][s + 3])).max() < 1e-11
        # device entry == host entry, bit for bit; unit-cube input == the two-step path
        dP = torch.from_numpy(P).cuda()
        out = torch.full((len(P),), float("nan"), dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), len(P), out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), got)
        cubes = rng.random((5, P.shape[1]))
        theta, ll = fit.loglike_cube_batch(cubes)
        assert np.array_equal(theta, fit.scale_cube_batch(cubes)) and np.array_equal(ll, fit.loglike_batch(theta))
        # a larger batch than the one-launch variant takes (set-up kernel + fused kernel in the first stage): same bits
        big = workloads.draw_P(kw, 700, rng)
        big[:6] = P
        assert np.array_equal(fit.loglike_batch(big)[:6], got)


def test_large_unit_cube_batch_on_a_wide_context_does_not_take_the_streaming_launch():
    """A unit-cube batch large enough for the streaming launch (one tile, >= 4 items per workgroup slot, > 65536 doubles)
    on a wide-LSF context: the streaming workspaces and the tile's LDS carry no halo for such a context, so the call must
    go the staged way through launch_wide (round 5's cube entry called the streaming path unguarded).  Compared with the
    two-step path on the same rows."""
    kw = _problem(333, 0.0031, (6.0, 9.0), (1.0,), 0, seed=9)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        assert fit.info.ntiles == 1 and 2 * fit.info.n_cap + 64 > 4096
        cubes = np.random.default_rng(3).random((6144, fit.ndim))
        assert cubes.size > 65536
        theta, ll = fit.loglike_cube_batch(cubes)
        assert fit.last_launch().path == _lib.MCALF_PATH_HOST_STAGED
        assert np.array_equal(theta, fit.scale_cube_batch(cubes))
        two_step = fit.loglike_batch(theta)
        assert fit.last_launch().path != _lib.MCALF_PATH_HOST_STREAM
        assert np.array_equal(ll, two_step) and np.isfinite(ll).all()
        prob = problem_from_kwargs(kw)
        idx = [0, 1000, 6143]
        want = o.loglike_batch(prob, theta[idx])
        assert np.all(np.abs(ll[idx] - want) < 1e-10 * np.abs(want) + 1e-9)


def test_wide_lsf_resolution_beyond_the_provisioned_maximum_and_no_convolution():
    """Free resolution in [6, 9] km/s on a 0.0031 km/s grid: a row with R = 12 is beyond the provisioned half-width --
    NaN model, logL = -inf, chi2 = +inf, as in every other context -- and a row with R below the velocity step is not
    convolved at all (hires_fitter.py:445)."""
    kw = _problem(333, 0.0031, (6.0, 9.0), (1.0,), 0, seed=5)
    prob = problem_from_kwargs(kw)
    P = workloads.draw_P(kw, 4, np.random.default_rng(11))
    P[1, 0] = 12.0
    P[2, 0] = 0.003
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
        chi2 = fit.chi2_batch(P)
        m = fit.model_batch(P)
        assert got[1] == -np.inf and chi2[1] == np.inf and np.isnan(m[1]).all()
        keep = [0, 2, 3]
        want = o.loglike_batch(prob, P[keep])
        assert np.all(np.abs(got[keep] - want) < 1e-10 * np.abs(want) + 1e-9)
        assert np.abs(m[2] - o.reconstruct_spec(prob, P[2])).max() < 1e-11


@pytest.mark.parametrize("npix,velstep,specres,contval,nfill", [
    (6000, 0.0045, (8.0,), (1.0,), 0),             # fixed grid of 2 x 2292 + 1 taps on two pixel tiles
    (5200, 0.0045, (7.0, 8.0), (0.9, 1.1), 1),     # free resolution (grid from specres[1], hires_fitter.py:549-550), free continuum, a filler
])
def test_wide_lsf_under_jax_semantics(npix, velstep, specres, contval, nfill):
    """Round 6: the JAX path's fixed kernel grid (hires_fitter.py:549-560, 667-681) may be wider than a workgroup tile too:
    the fused kernel runs with a grid of half-width 0 (the single tap 1) and the wide kernels do the zero-padded 'same'
    convolution, the reset of the first / last n pixels to the unconvolved model and the continuum.  Against the oracle's
    float64 restatement of that path; device entry == host entry."""
    kw = dict(_problem(npix, velstep, specres, contval, nfill, seed=npix), conv_mode="jax")
    prob = problem_from_kwargs(kw)
    P = workloads.draw_P(kw, 5, np.random.default_rng(3 + npix))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        n = o.jax_half_size(prob)
        assert fit.info.n_cap == n and 2 * n + 64 > 4096 and 2 * n + 1 <= npix
        got = fit.loglike_batch(P)
        want = np.array([o.jax_loglike_f64(prob, p) for p in P])
        assert np.all(np.abs(got - want) < 1e-10 * np.abs(want) + 1e-9), (got, want)
        m = fit.model_batch(P[:2])
        for a, p in zip(m, P[:2]):
            ref = o.jax_reconstruct_spec_f64(prob, p)
            assert np.abs(a - ref).max() < 1e-11
            assert np.array_equal(a[:3], ref[:3]) or np.abs(a[:3] - ref[:3]).max() < 1e-13      # (edge pixels: the unconvolved model)
        dP = torch.from_numpy(P).cuda()
        out = torch.full((len(P),), float("nan"), dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), len(P), out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), got)
        big = workloads.draw_P(kw, 600, np.random.default_rng(5))
        big[:5] = P
        assert np.array_equal(fit.loglike_batch(big)[:5], got)
        assert fit.get_jax_likelihood(use_jax=False)(P[0].astype(np.float32)).dtype == np.float32


def test_jax_grid_longer_than_the_spectrum_is_refused():
    """A JAX-semantics kernel grid LONGER than the spectrum stays an error (MCALF_ERR_INVALID): the reference's
    jnp.convolve(..., 'same') / jnp.where (hires_fitter.py:674-681) cannot broadcast it either."""
    kw = _problem(1500, 0.004, (8.0,), (1.0,), 0, seed=1)
    with pytest.raises(RuntimeError, match="MCALF_ERR_INVALID"):
        mcalf_amd.als_fitter(None, conv_mode="jax", **kw)


def test_wide_context_is_not_served_by_resident_evaluators():
    """The resident one-theta evaluator runs the one-launch variant's item, which convolves inside the tile: a wide-LSF
    context keeps launching (same values as without the switch), and the resident broker refuses it with a message."""
    kw = _problem(333, 0.0031, (8.0,), (1.0,), 0, seed=3)
    P = workloads.draw_P(kw, 3, np.random.default_rng(1))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = fit.loglike_batch(P)
        fit.set_resident(300)
        got = np.array([fit.lnlhood_dy(p) for p in P])
        assert np.array_equal(got, want) and fit.last_launch().inline_setup != 3
        fit.set_resident(0)
        boxes = (C.c_char * (2 * 576 + 64))()
        base = (C.addressof(boxes) + 63) & ~63
        stop = C.c_uint64(0)
        rc = fit._lib.mcalf_broker_serve_resident(fit._ctx, base, 1, C.addressof(stop), 100, None, 0.01)
        assert rc == _lib.MCALF_ERR_RANGE and b"wider than a tile" in fit._lib.mcalf_last_error(fit._ctx)
