"""GPU: parity on spectra/layouts beyond the BASELINE configs -- masked (two-range) grids, many
lines per component, long LSF kernels, tiny spectra, odd batch sizes, many tiles."""
import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import workloads
from cases import problem_from_kwargs
from oracle import numpy_oracle as o

pytestmark = pytest.mark.gpu

SIII = [(1190.4158, 0.292, 4.08e9), (1193.2897, 0.582, 4.07e9), (1260.4221, 1.18, 2.95e9),
        (1304.3702, 0.0863, 1.01e9), (1526.7070, 0.133, 1.13e9), (1808.0129, 0.00208, 2.38e6)]


def _check(kw, P, atol=1e-4):
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
        m = fit.model_batch(P[:2])
        info = fit.info
    want = o.loglike_batch(prob, P)
    assert np.abs(got - want).max() < atol
    assert (np.abs(got - want) / np.maximum(1.0, np.abs(want))).max() < 1e-10
    for i in range(2):
        assert np.abs(m[i] - o.reconstruct_spec(prob, P[i])).max() < 1e-11
    return info


def _spectrum(wl, rng):
    flux = 1.0 + rng.normal(0, 0.03, wl.size)
    err = rng.uniform(0.02, 0.05, wl.size)
    return wl, flux, err


def test_two_fit_ranges_masked_grid_and_six_lines():
    """Masked grid (gap between ranges -> the periodic LSF wraps across unrelated pixels, as in the
    reference) with a six-line multiplet: ncl = 6 lines x 3 components."""
    rng = np.random.default_rng(11)
    wl_all = 4750.0 * np.exp(np.arange(9000) * 1.2 / 2.9979245e5)      # constant 1.2 km/s pixels
    spec = _spectrum(wl_all, rng)
    z0 = 2.99
    kw = dict(fitrange=[[1190.0 * (1 + z0) - 6, 1193.5 * (1 + z0) + 6], [1260.4 * (1 + z0) - 8, 1260.4 * (1 + z0) + 8]],
              fitlines=["SiII %d" % int(l[0]) for l in SIII], linepars=SIII, ncomp=[1, 3], nfill=1,
              specres=[6.0, 9.0], contval=[0.95, 1.05], Nrange=[12.0, 15.5], brange=[3.0, 25.0],
              zrange=[z0 - 0.0015, z0 + 0.0015], spectrum=spec, velstep=1.2)
    P = workloads.draw_P(kw, 13, rng)
    info = _check(kw, P)
    assert info.ntiles == 1 and info.ndim == 2 + 1 + 9 + 3


def test_long_lsf_kernel_and_many_tiles():
    """Oversampled spectrum: 0.2 km/s pixels with R up to 40 km/s -> n = 258 taps half-width,
    9000 pixels -> 3 tiles with 516-pixel halos."""
    rng = np.random.default_rng(12)
    wl = 6190.0 * np.exp(np.arange(9000) * 0.2 / 2.9979245e5)
    kw = dict(fitrange=[[wl[0] - 1, wl[-1] + 1]], fitlines=["CIV 1548", "CIV 1550"], linepars=workloads.CIV,
              ncomp=[2, 4], specres=[20.0, 40.0], Nrange=[12.5, 14.5], brange=[5.0, 30.0],
              zrange=[2.9985, 3.0005], spectrum=_spectrum(wl, rng))
    P = workloads.draw_P(kw, 9, rng)
    info = _check(kw, P)
    assert info.n_cap >= 250 and info.ntiles >= 3


def test_tiny_spectrum_and_odd_batches():
    rng = np.random.default_rng(13)
    wl = np.linspace(6190, 6194, 97)
    kw = dict(fitrange=[[6189, 6195]], fitlines=["CIV 1548"], linepars=workloads.CIV[:1], ncomp=[1, 2],
              specres=[12.0], Nrange=[12.5, 14.5], brange=[5.0, 30.0], zrange=[2.9985, 3.0005],
              spectrum=_spectrum(wl, rng))
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        for batch in (1, 2, 63, 65, 513):
            P = workloads.draw_P(kw, batch, rng)
            got = fit.loglike_batch(P)
            idx = rng.choice(batch, size=min(batch, 6), replace=False)
            want = o.loglike_batch(prob, P[idx])
            assert np.abs(got[idx] - want).max() < 1e-6


def test_sixteen_component_lines_cap_and_saturation():
    """ncompmax = 20 doublets (40 records + 6 fillers) with saturated columns."""
    rng = np.random.default_rng(14)
    wl = np.linspace(6170, 6230, 3001)[1:-1]
    kw = dict(fitrange=[[6170, 6230]], fitlines=["CIV 1548", "CIV 1550"], linepars=workloads.CIV, ncomp=[10, 20],
              nfill=6, specres=[7.0], Nrange=[12.0, 17.5], brange=[4.0, 60.0], zrange=[2.985, 3.015],
              spectrum=_spectrum(wl, rng))
    P = workloads.draw_P(kw, 7, rng)
    _check(kw, P, atol=1e-3)      # |logL| ~ 1e6 here; the 1e-10 relative bar still applies


def test_many_records_take_the_second_record_trip():
    """30 components x 6 lines + 3 fillers = 183 records (1464 doubles): more than the two registers per
    thread the fused kernel's set-up holds in flight, so the remainder goes through its follow-up loop;
    the LDS left for the flux tile shrinks and the 2600-pixel spectrum needs several tiles."""
    rng = np.random.default_rng(14)
    z0 = 2.2
    wl = 1180.0 * (1 + z0) * np.exp(np.arange(2600) * 2.0 / 2.9979245e5)
    kw = dict(fitrange=[[wl[0] - 1, wl[-1] + 1]], fitlines=["SiII %d" % int(l[0]) for l in SIII], linepars=SIII,
              ncomp=[24, 30], nfill=3, specres=[7.0], Nrange=[11.5, 13.5], brange=[4.0, 20.0],
              zrange=[z0 - 0.001, z0 + 0.012], spectrum=_spectrum(wl, rng), velstep=2.0)
    P = workloads.draw_P(kw, 7, rng)
    info = _check(kw, P)
    assert info.ndim == 1 + 90 + 9
