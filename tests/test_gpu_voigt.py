"""GPU: the device Voigt-Hjerting function (through the C ABI) against scipy.special.wofz on a
dense (u, a) grid and against 40-digit mpmath at spot points."""
import ctypes as C

import mpmath as mp
import numpy as np
import pytest
from scipy.special import wofz

from mcalf_amd import _lib

pytestmark = pytest.mark.gpu


def hjert_gpu(x, y, nodes=False):
    """nodes=False: the arithmetic of a directly evaluated pixel (core table for every |x| < 8);
    nodes=True: that of an interpolation node (zone-1 wing polynomial between x_c and 8)."""
    x = np.ascontiguousarray(np.broadcast_to(x, np.broadcast(x, y).shape), dtype=float).ravel()
    y = np.ascontiguousarray(np.broadcast_to(y, x.shape), dtype=float).ravel()
    out = np.empty_like(x)
    pd = C.POINTER(C.c_double)
    fn = _lib.load().mcalf_voigt_hjerting_nodes if nodes else _lib.load().mcalf_voigt_hjerting
    _lib.check(fn(x.ctypes.data_as(pd), y.ctypes.data_as(pd), x.size, out.ctypes.data_as(pd), -1))
    return out


def _mp_H(x, y):
    z = mp.mpf(float(x)) + 1j * mp.mpf(float(y))
    return mp.re(mp.exp(-z * z) * mp.erfc(-1j * z))


def test_dense_grid_vs_scipy_fast_path():
    rng = np.random.default_rng(0)
    u = np.concatenate([np.linspace(0, 8, 4001), np.linspace(8, 40, 2001), 10 ** rng.uniform(1, 3.5, 4000),
                        -rng.uniform(0, 30, 500)])
    for a in [1e-7, 1.8e-5, 1e-4, 3.3e-4, 1.2e-3, 2 ** -8]:
        ref = wofz(u + 1j * a).real
        for nodes in (False, True):
            got = hjert_gpu(u, a, nodes=nodes)
            # scipy itself is ~2e-14; the node form drops exp(-x^2) once it is < 2e-17 (K = 1)
            assert (np.abs(got - ref) / (3e-17 + 6e-14 * ref)).max() < 1, (a, nodes)


def test_dense_grid_vs_scipy_general_path():
    u = np.concatenate([np.linspace(0, 8, 801), np.linspace(8, 60, 400), [200.0, 3000.0]])
    for a in [0.004, 0.01, 0.05, 0.1, 0.7, 3.0, 12.0]:
        got = hjert_gpu(u, a)
        ref = wofz(u + 1j * a).real
        assert np.abs(got / ref - 1).max() < 6e-14, a


def test_spot_points_vs_mpmath():
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 8, 120), rng.uniform(8, 16, 40), 10 ** rng.uniform(1.2, 3.5, 40),
                         [0.0, 7.9999999, 8.0, 8.0000001]])
    for a in [1e-9, 1.8e-5, 1.2e-3, 2 ** -8, 0.00391, 0.02, 1.5]:
        ex = np.array([float(_mp_H(x, a)) for x in xs])
        rtol = 5e-15 if a <= 2 ** -8 else 2e-14
        for nodes in (False, True):
            got = hjert_gpu(xs, a, nodes=nodes)
            err = np.abs(got - ex) / (3e-17 + rtol * ex)
            assert err.max() < 1, (a, nodes, xs[err.argmax()], err.max())


def test_limits():
    assert hjert_gpu(0.0, 0.0)[0] == 1.0
    g = hjert_gpu(np.array([0.5, 3.0]), 0.0)
    assert np.allclose(g, np.exp(-np.array([0.25, 9.0])), rtol=1e-14, atol=1e-17)
    assert np.isnan(hjert_gpu(np.nan, 1e-4)[0])
    # symmetric in u
    assert np.array_equal(hjert_gpu(np.array([-2.5, -50.0]), 1e-3), hjert_gpu(np.array([2.5, 50.0]), 1e-3))
