"""CPU: pin the oracle on the reference's own known-answer data."""
import json
import os

import mpmath as mp
import numpy as np
from scipy.special import wofz

from cases import seeded_noise
from oracle import numpy_oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TZ = [2.999, 2.9995, 3.0, 3.001, 3.0005, 3.0015, 3.002, 3.0025, 3.0035, 3.0039]
TN = [13.6, 13.0, 13.8, 13.6, 13.2, 13.4, 13.5, 14.0, 14.2, 13.7]
TB = [17.5, 8.0, 20.0, 25.0, 15.0, 30.0, 10.0, 25.0, 15.0, 20.0]


def _load(name):
    return np.loadtxt(os.path.join(GOLD, name))


def test_fixture_grid_facts():
    d = _load("civ_mock_spec_multicomp.txt")
    assert d.shape == (1998, 3)
    assert np.array_equal(d[:, 0], np.linspace(6180, 6220, 2000)[1:-1])
    assert np.all(d[:, 2] == 0.02)
    assert o.velstep_of(d[:, 0]) == 0.9675546360962316
    ker = o.lsf_kernel(8.0, 0.9675546360962316)
    assert ker.size == 23


def test_G1_single_component_fixture():
    """civ_mock_spec.txt - seeded noise == reconstruct_spec([1, 13.8, 3.0, 15.0])."""
    d = _load("civ_mock_spec.txt")
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (1, 1), specres=[8.0])
    model = o.reconstruct_spec(prob, np.array([1.0, 13.8, 3.0, 15.0]))
    assert np.abs(d[:, 1] - seeded_noise() - model).max() < 2e-15


def test_G2_multicomponent_fixture():
    """civ_mock_spec_multicomp.txt - noise == product of the 10 single-component models
    (testdata/generate_from_model.py:12-14,27-45)."""
    d = _load("civ_mock_spec_multicomp.txt")
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (1, 1), specres=[8.0])
    parts = [o.reconstruct_spec(prob, np.array([1.0, TN[i], TZ[i], TB[i]])) for i in range(10)]
    model = np.prod(np.array(parts), axis=0)
    assert np.abs(d[:, 1] - seeded_noise() - model).max() < 5e-15


def test_G3_derived_goldens_are_stable():
    d = _load("civ_mock_spec_multicomp.txt")
    g = json.load(open(os.path.join(GOLD, "derived_goldens.json")))
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (10, 10), specres=[8.0], fitrange=[[6180, 6220]])
    p = np.array([10.0] + [v for i in range(10) for v in (TN[i], TZ[i], TB[i])])
    assert abs(o.lnlhood_worker(prob, p) - g["G3_logL_truth"]) < 1e-9
    assert abs(o.chi2(prob, p) - g["G3_chi2_truth"]) < 1e-9
    # survey-measured values
    assert abs(g["G3_logL_truth"] - 4991.860095571162) < 1e-8
    assert abs(g["G3_chi2_truth"] - 1976.6453598626758) < 1e-8
    # JAX-path semantics in float64 differ from the numpy path by ~1e-6 in logL
    dj = o.jax_loglike_f64(prob, p) - g["G3_logL_truth"]
    assert 1e-8 < abs(dj) < 1e-4
    pA = np.array(g["A_P16"])
    probA = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (2, 2), specres=[8.0], Nrange=[12.0, 14.5],
                      brange=[10.0, 40.0], zrange=[2.99, 3.01], fitrange=[[6180, 6220]])
    assert np.abs(o.loglike_batch(probA, pA) - np.array(g["A_logL16"])).max() < 1e-7


def test_wofz_spot_values_against_mpmath():
    """The third-party Faddeeva function the oracle leans on, checked at spot points."""
    mp.mp.dps = 40
    for (x, y) in [(0.0, 1e-4), (0.7, 3e-4), (2.5, 1.2e-3), (5.5, 1e-5), (12.0, 1e-3), (150.0, 2e-4), (1.0, 0.5)]:
        z = mp.mpf(x) + 1j * mp.mpf(y)
        ex = mp.re(mp.exp(-z * z) * mp.erfc(-1j * z))
        assert abs(wofz(x + 1j * y).real / float(ex) - 1) < 1e-13


def test_layout_and_cube_map():
    d = _load("civ_mock_spec_multicomp.txt")
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (8, 11), nfill=4, specres=[8, 9],
                     Nrange=[12, 14.5], brange=[10, 40], zrange=[2.99, 3.01], fitrange=[[6180, 6220]])
    assert (prob.ndim, prob.startind, prob.endind) == (47, 1, 35)
    th = o.scale_cube_pc(prob, np.full(47, 0.5))
    assert th[0] == 8.5 and th[1] == 9.0 and th[2] == 13.25      # int(9.5) = 9
    zf = prob.bounds[35 + 1]
    assert abs(zf[0] - ((d[0, 0] + 0.25) / 250 - 1)) < 1e-15


def test_edge_cases_numpy_path():
    d = _load("civ_mock_spec.txt")
    # R <= velstep: no convolution (hires_fitter.py:445)
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (1, 1), specres=[0.5])
    p = np.array([1.0, 13.8, 3.0, 15.0])
    m = o.reconstruct_spec(prob, p)
    t = o.voigt_model(d[:, 0], 13.8, 15.0, 3.0, *o.CIV_LINES[0]) * o.voigt_model(d[:, 0], 13.8, 15.0, 3.0, *o.CIV_LINES[1])
    assert np.array_equal(m, t)
    # zero active components -> flat continuum
    assert np.all(o.reconstruct_spec(prob, np.array([0.0, 13.8, 3.0, 15.0])) == 1.0)
    # a NaN pixel in the data is dropped by nansum
    flux = d[:, 1].copy()
    flux[100] = np.nan
    prob2 = o.Problem(d[:, 0], flux, d[:, 2], o.CIV_LINES, (1, 1), specres=[8.0])
    assert np.isfinite(o.lnlhood_worker(prob2, p))


def test_distance_between_the_references_own_two_paths_is_reproduced():
    """Informational pin: the JAX path as shipped -- float32, voigt_jax.hjert -- restated in
    numpy lands where the survey's probe found it on the multicomponent fixture at the truth parameters
    (max |dflux| 2.1e-3, max rel 5.6e-3, dlogL -0.61), and the same algorithm in float64 within 1e-8 / -5.7e-6.
    This is why parity is asserted against the float64 numpy path only."""
    from mcalf_amd import workloads
    d = np.loadtxt(os.path.join(GOLD, "civ_mock_spec_multicomp.txt"))
    prob = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (10, 10), specres=[8.0])
    p = workloads.truth_vector(10)
    m_np, l_np = o.reconstruct_spec(prob, p), o.lnlhood_worker(prob, p)
    m32, l32 = o.jax_reconstruct_spec_f32(prob, p), o.jax_loglike_f32(prob, p)
    assert 1.5e-3 < np.abs(m32 - m_np).max() < 3e-3
    assert 4e-3 < (np.abs(m32 - m_np) / m_np).max() < 8e-3
    assert -0.8 < l32 - l_np < -0.45
    m64, l64 = o.jax_reconstruct_spec_f32(prob, p, np.float64), o.jax_loglike_f32(prob, p, np.float64)
    assert np.abs(m64 - m_np).max() < 5e-8 and -1e-5 < l64 - l_np < -1e-6
    # the Voigt function of that path: ~1e-6 relative from its three-term asymptotic branch
    x = np.linspace(0, 30, 1501)
    from scipy.special import wofz
    rel = np.abs(o.jax_hjert(x, 1.2e-3, np.float64) - wofz(x + 1.2e-3j).real) / wofz(x + 1.2e-3j).real
    assert 1e-7 < rel.max() < 3e-6
