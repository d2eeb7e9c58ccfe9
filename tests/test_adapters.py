"""CPU: the vectorised-sampler adapters, driven through the calling conventions of the samplers they
target (a `pool.map` over a wrapped callable; batch functions).  The fitter is a stand-in that records
what reaches the batched entries -- the device work itself is covered by tests/test_gpu_*.py."""
import functools

import numpy as np

import mcalf_amd  # noqa: F401
from mcalf_amd import adapters


class FakeFit:
    ndim = 4

    def __init__(self):
        self.calls = []

    def loglike_batch(self, P):
        P = np.asarray(P, dtype=float).reshape(-1, self.ndim)
        self.calls.append(("logl", P.shape[0]))
        return -0.5 * (P ** 2).sum(axis=1)

    def scale_cube_batch(self, cubes, int_ncomp=True):
        cubes = np.asarray(cubes, dtype=float).reshape(-1, self.ndim)
        self.calls.append(("cube", cubes.shape[0]))
        return cubes * 2.0 - 1.0

    def lnlhood_dy(self, p):
        return float(self.loglike_batch(p)[0])

    def lnlhood_pc(self, p):
        return self.lnlhood_dy(p), []

    def _scale_cube_pc(self, cube):
        return self.scale_cube_batch(cube)[0]


class SamplerSideWrapper:
    """What dynesty puts around user callables: `.func`, `.args`, `.kwargs`, `__call__`."""

    def __init__(self, func, args=(), kwargs=None):
        self.func, self.args, self.kwargs = func, args, kwargs or {}

    def __call__(self, x):
        return self.func(x, *self.args, **self.kwargs)


def test_pool_map_batches_the_mirrors_own_callables():
    fit = FakeFit()
    pool = adapters.BatchPool(fit, size=64)
    pts = [np.random.default_rng(i).random(4) for i in range(37)]
    want = [-0.5 * float((p ** 2).sum()) for p in pts]
    assert pool.map(SamplerSideWrapper(fit.lnlhood_dy), pts) == want
    assert fit.calls == [("logl", 37)]
    assert pool.map(fit.lnlhood_pc, pts) == [(w, []) for w in want]
    cubes = pool.map(SamplerSideWrapper(SamplerSideWrapper(fit._scale_cube_pc)), pts)
    assert np.allclose(np.array(cubes), np.array(pts) * 2 - 1)
    assert fit.calls[-1] == ("cube", 37) and pool.batched_calls == 3 and pool.serial_calls == 0


def test_pool_map_falls_back_to_serial_for_anything_else():
    fit = FakeFit()
    other = FakeFit()
    pool = adapters.BatchPool(fit)
    pts = [np.full(4, float(i)) for i in range(5)]
    # a callable of a different fitter, a wrapper that carries extra arguments, a partial, a lambda
    assert pool.map(other.lnlhood_dy, pts) == [other.lnlhood_dy(p) for p in pts]
    assert pool.map(SamplerSideWrapper(lambda p, s: s * p.sum(), args=(2.0,)), pts) == [2.0 * p.sum() for p in pts]
    assert pool.map(functools.partial(lambda p, s=1.0: s * p.sum(), s=3.0), pts) == [3.0 * p.sum() for p in pts]
    assert pool.map(fit.lnlhood_dy, []) == []
    assert pool.batched_calls == 0 and not any(c[1] > 1 for c in fit.calls)
    with pool as p2:
        assert p2 is pool
    pool.close(); pool.join()


def test_batch_functions_shapes():
    fit = FakeFit()
    loglike, transform = adapters.batch_functions(fit)
    cubes = np.random.default_rng(0).random((11, 4))
    theta = transform(cubes)
    assert theta.shape == (11, 4) and loglike(theta).shape == (11,)
    assert fit.calls == [("cube", 11), ("logl", 11)]
