"""Vectorised-sampler adapters.

The reference hands its solvers one-point callables (`cli.py:110,153,179,196`); the device only pays
when whole batches of live points arrive in one call.  Two calling conventions cover the samplers
that can do that without any change to the sampler itself:

* **batch functions** -- UltraNest-style `loglike(points[n, ndim]) -> [n]` and
  `transform(cubes[n, ndim]) -> [n, ndim]` (`vectorized=True`): `batch_functions(fit)`.
* **a pool with `map`** -- dynesty-style `pool=..., queue_size=n`: the sampler maps its wrapped
  likelihood over a list of proposals; `BatchPool.map` recognises the mirror's own callables and turns
  the list into one `loglike_batch` / `scale_cube_batch` call.  Anything else is mapped serially, so
  the pool is safe to pass wherever a `multiprocessing.Pool` is accepted.

No sampler is imported here; the tests drive the adapters through the calling conventions alone.
"""
from __future__ import annotations

import numpy as np


def batch_functions(fit, int_ncomp=True):
    """(loglike, transform) taking and returning arrays with one row per live point."""

    def loglike(points):
        return fit.loglike_batch(np.asarray(points, dtype=float))

    def transform(cubes):
        return fit.scale_cube_batch(np.asarray(cubes, dtype=float), int_ncomp=int_ncomp)

    return loglike, transform


def _unwrap(fn):
    """Follow the `.func` chain of sampler-side wrappers (dynesty's `_function_wrapper` keeps the user
    callable in `.func` and extra arguments in `.args` / `.kwargs`) and functools.partial objects."""
    seen = 0
    while seen < 8:
        inner = getattr(fn, "func", None)
        if inner is None or getattr(fn, "args", ()) or getattr(fn, "kwargs", None) or getattr(fn, "keywords", None):
            break
        fn = inner
        seen += 1
    return fn


class BatchPool:
    """Drop-in for the `pool` argument of samplers that evaluate proposals with `pool.map`.

    `size` is what the sampler reads to decide how many proposals to queue (dynesty: `queue_size`
    defaults to `pool.size`); set it to the batch the GPU should see per call."""

    def __init__(self, fit, size=1024):
        self.fit = fit
        self.size = int(size)
        self.batched_calls = 0
        self.serial_calls = 0

    def _kind(self, fn):
        target = _unwrap(fn)
        owner = getattr(target, "__self__", None)
        name = getattr(target, "__name__", "")
        if owner is self.fit:
            if name in ("lnlhood_dy", "lnlhood_worker"):
                return "logl"
            if name == "lnlhood_pc":
                return "logl_pc"
            if name == "_scale_cube_pc":
                return "cube_pc"
        return None

    def map(self, fn, iterable):
        items = list(iterable)
        kind = self._kind(fn)
        if kind is None or not items:
            self.serial_calls += 1
            return [fn(x) for x in items]
        self.batched_calls += 1
        rows = np.asarray(items, dtype=float)
        if kind == "cube_pc":
            return list(self.fit.scale_cube_batch(rows, int_ncomp=True))
        logl = self.fit.loglike_batch(rows)
        if kind == "logl_pc":
            return [(float(v), []) for v in logl]
        return [float(v) for v in logl]

    # the rest of the multiprocessing.Pool surface samplers touch
    def close(self):
        pass

    def join(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
