"""GPU: the HIP path through the C ABI (via the als_fitter mirror) against the CPU oracle on
the same seeded inputs, against the golden fixtures, and size-independent properties at
BASELINE.json's full batch sizes.

Tolerances (BASELINE.json north_star): flux 1e-6 relative, logL 1e-4 absolute.  The asserts
below are tighter where the arithmetic allows, so a regression shows before the bar is hit.
"""
import json
import os

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import workloads
from cases import oracle_synth, problem_from_kwargs, seeded_noise
from oracle import c_oracle
from oracle import numpy_oracle as o

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLUX_RTOL = 1e-6
LOGL_ATOL = 1e-4


def relflux(got, ref):
    return np.abs(got - ref).max() / 1.0, np.abs(got / np.where(ref == 0, 1, ref) - 1)[ref > 1e-300].max()


def check_batch(kw, P, n_model=4):
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
        models = fit.model_batch(P[:n_model])
    want = o.loglike_batch(prob, P)
    assert np.abs(got - want).max() < LOGL_ATOL, np.abs(got - want).max()
    # also a relative bar: 1e-10 of |logL| (bad fits have |logL| ~ 1e6)
    assert (np.abs(got - want) / np.maximum(1.0, np.abs(want))).max() < 1e-10
    # second, independent oracle: plain C with its own Faddeeva routine
    want_c = c_oracle.COracle(prob, threads=8).loglike_batch(P)
    assert np.abs(got - want_c).max() < LOGL_ATOL
    assert (np.abs(got - want_c) / np.maximum(1.0, np.abs(want_c))).max() < 1e-10
    for i in range(n_model):
        ref = o.reconstruct_spec(prob, P[i])
        assert np.abs(models[i] - ref).max() < 1e-11
        ok = ref > 1e-290
        assert np.abs(models[i][ok] / ref[ok] - 1).max() < FLUX_RTOL
    return got, want


def test_golden_G1_G2_models_from_the_device():
    d1 = np.loadtxt(os.path.join(GOLD, "civ_mock_spec.txt"))
    d2 = np.loadtxt(os.path.join(GOLD, "civ_mock_spec_multicomp.txt"))
    noise = seeded_noise()
    kw = dict(fitrange=[[6180, 6220]], fitlines=["CIV 1548", "CIV 1550"], ncomp=[1, 1], specres=[8.0],
              spectrum=(d1[:, 0], d1[:, 1], d1[:, 2]))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        m1 = fit.reconstruct_spec(np.array([1.0, 13.8, 3.0, 15.0]))
        P = np.array([[1.0, workloads.TRUTH_N[i], workloads.TRUTH_Z[i], workloads.TRUTH_B[i]] for i in range(10)])
        parts = fit.model_batch(P)
        one = fit.reconstruct_onecomp(8.0, [1.0], 13.8, 3.0, 15.0)
    assert np.abs(d1[:, 1] - noise - m1).max() < 2e-11   # u = (nu zp1 - nu0)/dnu cancellation noise, both sides
    assert np.abs(d2[:, 1] - noise - np.prod(parts, axis=0)).max() < 2e-11
    assert np.array_equal(one, m1)


def test_golden_G3_logL_chi2_at_truth_and_callable_conventions():
    g = json.load(open(os.path.join(GOLD, "derived_goldens.json")))
    d2 = np.loadtxt(os.path.join(GOLD, "civ_mock_spec_multicomp.txt"))
    kw = dict(fitrange=[[6180, 6220]], fitlines=["CIV 1548", "CIV 1550"], ncomp=[10, 10], specres=[8.0],
              spectrum=(d2[:, 0], d2[:, 1], d2[:, 2]))
    p = workloads.truth_vector(10)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        pc = fit.lnlhood_pc(p)
        assert isinstance(pc, tuple) and pc[1] == [] and isinstance(pc[0], float)
        assert abs(pc[0] - g["G3_logL_truth"]) < 1e-7
        assert fit.lnlhood_dy(p) == pc[0]
        assert fit.lnlhood_mn(list(p), 31, 31) == pc[0]
        assert abs(fit.chi2(p) - g["G3_chi2_truth"]) < 1e-7
        # an opaque model is chi2 = (+inf, [])   (hires_fitter.py:241-242)
        pz = p.copy()
        pz[1::3][:10] = 30.0
        pz[3::3][:10] = 4000.0
        assert fit.chi2(pz) == (np.inf, [])
    with mcalf_amd.als_fitter(None, conv_mode="jax", **kw) as fj:
        assert abs(fj.lnlhood_dy(p) - g["G3_jaxsem_f64_logL_truth"]) < 1e-7


def test_config_A_fixture_batch():
    g = json.load(open(os.path.join(GOLD, "derived_goldens.json")))
    kw, _, seed = workloads.config("A")
    P = workloads.draw_P(kw, 64, np.random.default_rng(seed))
    assert np.array_equal(P[:16], np.array(g["A_P16"]))
    got, _ = check_batch(kw, P)
    assert np.abs(got[:16] - np.array(g["A_logL16"])).max() < LOGL_ATOL


@pytest.fixture(scope="module")
def cfgB():
    return workloads.config("B", oracle_synth)


@pytest.fixture(scope="module")
def cfgC():
    return workloads.config("C", oracle_synth)


def test_config_B_sample(cfgB):
    kw, _, seed = cfgB
    P = workloads.draw_P(kw, 48, np.random.default_rng(seed))
    check_batch(kw, P)


def test_config_C_free_specres_fillers_variable_ncomp(cfgC):
    kw, _, seed = cfgC
    P = workloads.draw_P(kw, 48, np.random.default_rng(seed))
    assert set(np.unique(P[:, 1])) <= {8.0, 9.0, 10.0, 11.0}
    check_batch(kw, P)
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        t = fit.reconstruct_spec(P[0], targonly=True)
        f = fit.reconstruct_onecomp_fill(P[0][0], [1.0], *P[0][35:38])
    assert np.abs(t - o.reconstruct_spec(prob, P[0], targonly=True)).max() < 1e-11
    assert np.abs(f - o.reconstruct_onecomp(prob, P[0][0], 1.0, *P[0][35:38], fill=True)).max() < 1e-11


def test_config_E_damped_lya_multi_tile():
    kw, _, seed = workloads.config("E", oracle_synth)
    P = workloads.draw_P(kw, 6, np.random.default_rng(seed), damped=2)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        assert fit.info.ntiles > 1
    check_batch(kw, P, n_model=2)


def test_edge_cases(cfgB):
    kw, _, _ = cfgB
    kw = dict(kw)
    wl, flux, err = kw["spectrum"]
    prob = problem_from_kwargs(kw)
    rng = np.random.default_rng(7)
    P = workloads.draw_P(kw, 4, rng)
    # (1) R <= velstep: convolution skipped
    kw1 = dict(kw, specres=[0.4])
    p1 = problem_from_kwargs(kw1)
    with mcalf_amd.als_fitter(None, **kw1) as fit:
        assert fit.info.n_cap == 0
        assert np.abs(fit.loglike_batch(P) - o.loglike_batch(p1, P)).max() < LOGL_ATOL
    # (2) zero active components, ncomp slot fractional / negative
    with mcalf_amd.als_fitter(None, **kw) as fit:
        q = P[0].copy()
        for v in (0.0, 0.9, -3.0):
            q[0] = v
            # no absorber: the model is the LSF applied to a constant, sum(w) / bot.  astropy forms both sums in
            # one loop, tap after tap, so the reference gives exactly 1 (hires_fitter.py:463-464); the set-up code
            # adds the taps up in the order of the fused kernel's numerator chain for the same reason
            assert np.all(fit.reconstruct_spec(q) == 1.0)
        q[0] = 3.7                      # int() -> 3
        assert np.abs(fit.reconstruct_spec(q) - o.reconstruct_spec(prob, q)).max() < 1e-11
        # (3) empty batch
        assert fit.loglike_batch(np.empty((0, fit.ndim))).shape == (0,)
        with pytest.raises(ValueError):
            fit.loglike_batch(np.zeros((2, fit.ndim + 1)))
    # (4) NaN pixel in the data is dropped (np.nansum), zero error -> inf propagates
    f2 = flux.copy()
    f2[10] = np.nan
    kw2 = dict(kw, spectrum=(wl, f2, err))
    with mcalf_amd.als_fitter(None, **kw2) as fit:
        got = fit.loglike_batch(P)
    assert np.abs(got - o.loglike_batch(problem_from_kwargs(kw2), P)).max() < LOGL_ATOL
    # (5) free continuum + free specres
    kw3 = dict(kw, specres=[6.0, 12.0], contval=[0.9, 1.1])
    P3 = workloads.draw_P(kw3, 4, rng)
    with mcalf_amd.als_fitter(None, **kw3) as fit:
        assert fit.startind == 2
        got = fit.loglike_batch(P3)
    assert np.abs(got - o.loglike_batch(problem_from_kwargs(kw3), P3)).max() < LOGL_ATOL
    # (6) general-damping path (tiny b -> a > 2^-8) and strong saturated line
    q = P[0].copy()
    q[3] = 0.05          # b = 0.05 km/s
    q[4] = 17.5          # logN of component 2
    with mcalf_amd.als_fitter(None, **kw) as fit:
        m = fit.reconstruct_spec(q)
    assert np.abs(m - o.reconstruct_spec(prob, q)).max() < 1e-10


def test_jax_semantics_mode(cfgC):
    kw, _, seed = cfgC
    P = workloads.draw_P(kw, 8, np.random.default_rng(seed + 1))
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, conv_mode="jax", **kw) as fit:
        got = fit.loglike_batch(P)
        m = fit.model_batch(P[:2])
    want = np.array([o.jax_loglike_f64(prob, p) for p in P])
    assert np.abs(got - want).max() < LOGL_ATOL
    for i in range(2):
        assert np.abs(m[i] - o.jax_reconstruct_spec_f64(prob, P[i])).max() < 1e-11


def test_scale_cube_batch_matches_host(cfgC):
    kw, _, _ = cfgC
    cubes = np.random.default_rng(3).random((33, 47))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        dev = fit.scale_cube_batch(cubes)
        host = np.array([fit._scale_cube_pc(c) for c in cubes])
        mn = fit.scale_cube_batch(cubes, int_ncomp=False)
    assert np.array_equal(dev, host)
    assert np.array_equal(mn, np.array([fit._scale_cube_mn(c.copy(), 47, 47) for c in cubes]))


def test_full_size_properties_config_B(cfgB):
    """BASELINE batch (1024): sharded == unsharded bit for bit, row order independence,
    and logL recomputed on the host from the device's own model spectra."""
    kw, batch, seed = cfgB
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed))
    wl, flux, err = kw["spectrum"]
    with mcalf_amd.als_fitter(None, **kw) as fit:
        full = fit.loglike_batch(P)
        shards = np.concatenate([fit.loglike_batch(P[i * 128:(i + 1) * 128]) for i in range(8)])
        perm = np.random.default_rng(0).permutation(batch)
        permd = fit.loglike_batch(P[perm])
        models = fit.model_batch(P[:256])
    assert np.array_equal(full, shards)
    assert np.array_equal(full[perm], permd)
    ispec2 = 1.0 / err ** 2
    host = -0.5 * np.nansum(ispec2 * (flux - models) ** 2 - np.log(ispec2) + np.log(2 * np.pi), axis=1)
    assert (np.abs(host - full[:256]) / np.abs(host)).max() < 1e-13
    assert np.all(np.isfinite(full)) and np.all(models >= 0) and np.all(models <= 1 + 1e-12)


def test_asymmetric_likelihood_veto():
    """hires_fitter.py:296-303 with pinned thresholds (the reference draws them unseeded)."""
    kw, _, seed = workloads.config("A")
    prob = problem_from_kwargs(kw)
    rng = np.random.default_rng(21)
    P = workloads.draw_P(kw, 24, rng)
    P[0] = [2.0, 13.6, 2.999, 17.5, 13.8, 3.0, 20.0]            # two of the true components: a fair fit
    P[1] = [2.0, 12.0, 2.9905, 10.0, 12.0, 2.9906, 10.0]        # nearly no absorption: data far below model
    for cdf in ([5, 0, 0], [50, 30, 10], [3000, 3000, 3000]):
        with mcalf_amd.als_fitter(None, Asymmlike=True, gauss_cdf=cdf, **kw) as fit:
            got = fit.loglike_batch(P)
            assert fit.gauss_cdf == cdf and fit.gracenum == 0.01 * fit.obj.size
        want = np.array([o.lnlhood_worker(prob, p, asymm_thresholds=(cdf[1], cdf[2])) for p in P])
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        ok = ~np.isneginf(want)
        assert np.abs(got[ok] - want[ok]).max() < LOGL_ATOL
    # the veto only looks at positive residuals (model BELOW the data): force some
    with mcalf_amd.als_fitter(None, Asymmlike=True, gauss_cdf=[0, 0, 0], **kw) as fit:
        q = np.array([2.0, 14.5, 3.005, 40.0, 14.5, 3.006, 40.0])   # strong absorption where the data has none
        assert fit.lnlhood_dy(q) == -np.inf
        assert o.lnlhood_worker(prob, q, asymm_thresholds=(0, 0)) == -np.inf


def test_single_line_models_and_equivalent_width(cfgC):
    """mcalf_onecomp_batch(which = 2 + k) and the derived quantities built on it."""
    kw, _, seed = cfgC
    prob = problem_from_kwargs(kw)
    P = workloads.draw_P(kw, 3, np.random.default_rng(seed + 5))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        for k in range(2):
            m = fit.onecomp_batch([0.0, 1.0, 13.7, 3.0005, 18.0], line=k)[0]
            ref = o.voigt_model(prob.wl, 13.7, 18.0, 3.0005, *prob.lines[k])
            assert np.abs(m - ref).max() < 2e-11      # u-cancellation noise, both sides
        with pytest.raises(RuntimeError):
            fit.onecomp_batch([0.0, 1.0, 13.7, 3.0005, 18.0], line=2)
        for p in P:
            for k in range(2):
                assert abs(fit.calc_w(p, lineid=k, reference_indexing=False) - o.calc_w_intended(prob, p, lineid=k)) < 1e-11
                # the reference as written (hires_fitter.py:481-489: all ncompmax slots, sliced from the ncomp slot)
                want = o.calc_w_reference(prob, p, lineid=k)
                got = fit.calc_w(p, lineid=k)
                assert np.isfinite(want) and abs(got - want) <= 1e-11 + 1e-9 * abs(want)
            assert abs(fit.calc_N(p, reference_indexing=False) - o.calc_N_intended(prob, p)) < 1e-13
            with pytest.raises(IndexError):                       # :499-503, in the reference as here
                fit.calc_N(p)


def test_full_size_properties_configs_C_and_E(cfgC):
    """Size-independent properties at BASELINE.json's full batches (C: 4096 x 47, E: 16384 x 49 on
    20000 pixels): shards == whole bit for bit, exact continuum scaling, component-order symmetry,
    a zero-column component is a no-op."""
    kw, batch, seed = cfgC
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        full = fit.loglike_batch(P)
        assert np.all(np.isfinite(full))
        parts = np.concatenate([fit.loglike_batch(P[i:i + 512]) for i in range(0, batch, 512)])
        assert np.array_equal(full, parts)
        # swap the first two components of every row: same physics, different summation order
        Q = P.copy()
        Q[:, 2:5], Q[:, 5:8] = P[:, 5:8], P[:, 2:5]
        swapped = fit.loglike_batch(Q)
        assert (np.abs(swapped - full) / np.abs(full)).max() < 1e-12
    kwc = dict(kw, contval=[0.5, 2.0])                      # free continuum: slot 1
    Pc = workloads.draw_P(kwc, 64, np.random.default_rng(seed + 1))
    with mcalf_amd.als_fitter(None, **kwc) as fit:
        m1 = fit.model_batch(Pc)
        Pc2 = Pc.copy()
        Pc2[:, 1] *= 2.0
        assert np.array_equal(fit.model_batch(Pc2), 2.0 * m1)          # exact power-of-two scaling
        Pz = Pc.copy()
        Pz[:, 3] = -np.inf                                              # first component: zero column
        Pd = Pc.copy()
        Pd[:, 3] = -400.0                                               # 10**-400 underflows to the same zero
        assert np.array_equal(fit.model_batch(Pz), fit.model_batch(Pd))
    kwE, batchE, seedE = workloads.config("E", oracle_synth)
    PE = workloads.draw_P(kwE, batchE, np.random.default_rng(seedE), damped=2)
    with mcalf_amd.als_fitter(None, **kwE) as fit:
        fullE = fit.loglike_batch(PE)
        assert np.all(np.isfinite(fullE)) and fit.info.ntiles == 5
        idx = np.random.default_rng(0).choice(batchE, 96, replace=False)
        assert np.array_equal(fit.loglike_batch(PE[idx]), fullE[idx])
        m = fit.model_batch(PE[:32])
        assert np.all(m >= 0.0) and np.all(m <= 1.0 + 1e-12)
        assert np.any(m == 0.0)                                         # damped cores underflow to exactly 0
