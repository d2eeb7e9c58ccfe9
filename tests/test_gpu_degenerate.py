"""GPU: degenerate parameter values.  Per-sample numerical failures are VALUES (nan dropped by nansum,
inf propagated), exactly as the reference's numpy path produces them (hires_fitter.py:294, 357-365);
the single-point callables raise where the reference's `int()` raises (:428)."""
import warnings

import numpy as np
import pytest
from scipy.special import wofz

import mcalf_amd
from mcalf_amd import workloads
from cases import oracle_synth, problem_from_kwargs
from oracle import numpy_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    kw, _, _ = workloads.config("C", oracle_synth)
    prob = problem_from_kwargs(kw)
    P0 = workloads.draw_P(kw, 1, np.random.default_rng(7))[0]
    with mcalf_amd.als_fitter(None, **kw) as fit:
        yield kw, prob, P0, fit


VALUES = [np.nan, np.inf, -np.inf, 0.0, -1.0, 1e300, -1e300]


def test_degenerate_line_parameters_give_the_reference_values(setup):
    """logN, z, b of a target component and of a filler set to nan / +-inf / 0 / negative / huge: the model
    has NaN exactly where the oracle's has, and logL agrees (a NaN model is logL = -0.0 through nansum)."""
    kw, prob, P0, fit = setup
    s, e = fit.startind, fit.endind
    rows, names = [], []
    for v in VALUES:
        for nm, idx in [("logN", s + 1), ("z", s + 2), ("b", s + 3), ("fillN", e), ("fillz", e + 1), ("fillb", e + 2)]:
            p = P0.copy()
            p[idx] = v
            rows.append(p)
            names.append(f"{nm}={v}")
    P = np.array(rows)
    got = fit.loglike_batch(P)
    gm = fit.model_batch(P)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for name, p, g, m in zip(names, P, got, gm):
            want = orc.lnlhood_worker(prob, p)
            wm = orc.reconstruct_spec(prob, p)
            assert np.array_equal(np.isnan(m), np.isnan(wm)), name
            ok = ~np.isnan(wm)
            big = np.abs(wm[ok]) > 1e30                      # b < 0: exp(+tau) up to 1e17 and beyond
            assert np.allclose(m[ok][~big], wm[ok][~big], rtol=1e-6, atol=1e-12), name
            assert np.allclose(m[ok][big], wm[ok][big], rtol=1e-6) or not big.any(), name
            if np.isnan(want):
                assert np.isnan(g), name
            else:
                assert g == want or abs(g - want) <= 1e-4 + 1e-9 * abs(want), (name, g, want)


def test_resolution_and_ncomp_slot_edge_values(setup):
    kw, prob, P0, fit = setup
    s = fit.startind
    # R = nan / <= velstep / negative: `R > velstep` is False -> no convolution (:445)
    for v in (np.nan, 0.0, -1.0, -np.inf, -1e300):
        p = P0.copy()
        p[0] = v
        with np.errstate(all="ignore"):
            want = orc.lnlhood_worker(prob, p)
        assert abs(fit.lnlhood_worker(p) - want) < 1e-4 + 1e-12 * abs(want)
    # ncomp slot <= 0: no target component
    for v in (0.0, -1.0, -1e300):
        p = P0.copy()
        p[s] = v
        want = orc.lnlhood_worker(prob, p)
        assert abs(fit.lnlhood_worker(p) - want) < 1e-4 + 1e-12 * abs(want)
    # nan / inf ncomp: the reference's int() raises; the batched entry clamps to [0, ncompmax]
    for v, exc in ((np.nan, ValueError), (np.inf, OverflowError), (-np.inf, OverflowError)):
        p = P0.copy()
        p[s] = v
        with pytest.raises(exc):
            int(p[s])                                        # what hires_fitter.py:428 does
        for call in (fit.lnlhood_worker, fit.reconstruct_spec, fit.chi2, fit.lnlhood_dy):
            with pytest.raises(exc):
                call(p)
        lo, hi = P0.copy(), P0.copy()
        lo[s], hi[s] = 0.0, float(kw["ncomp"][1])
        want = fit.loglike_batch(hi if v == np.inf else lo)[0]
        assert fit.loglike_batch(p)[0] == want


def test_device_voigt_in_the_lower_half_plane():
    """a < 0 (b < 0): H follows scipy's reflection w(z) = 2 exp(-z^2) - w(-z)."""
    from mcalf_amd import _lib
    import ctypes as C
    lib = _lib.load()
    x = np.concatenate([np.linspace(0, 12, 2401), np.linspace(12, 400, 500)])
    pd = C.POINTER(C.c_double)
    for yv in (-1e-5, -3.25e-3, -2e-2, -0.7, -3.0):
        y = np.full_like(x, yv)
        out = np.empty_like(x)
        _lib.check(lib.mcalf_voigt_hjerting(x.ctypes.data_as(pd), y.ctypes.data_as(pd), x.size,
                                            out.ctypes.data_as(pd), -1))
        ref = wofz(x + 1j * y).real
        scale = np.maximum(np.abs(ref), np.abs(2 * np.exp(yv * yv - x * x)))     # cancellation in the reflection
        assert np.max(np.abs(out - ref) / scale) < 5e-13, yv
