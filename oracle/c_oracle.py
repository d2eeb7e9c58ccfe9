"""ctypes wrapper of oracle/c/libmcalf_oracle.so (plain-C restatement of the reference's numpy path).
TEST INFRASTRUCTURE ONLY -- imported by tests/ and by the cpu_baseline leg of bench.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIB = os.path.join(_DIR, "libmcalf_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "mcalf_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", _LIB, src, "-lm"], check=True)
    return _LIB


class _Problem(C.Structure):
    _fields_ = [("npix", C.c_long), ("wl", C.POINTER(C.c_double)), ("flux", C.POINTER(C.c_double)),
                ("err", C.POINTER(C.c_double)), ("velstep", C.c_double), ("nlines", C.c_int),
                ("lines", C.POINTER(C.c_double)), ("fill", C.c_double * 3), ("ncompmax", C.c_int),
                ("nfill", C.c_int), ("freespecres", C.c_int), ("freecont", C.c_int),
                ("specres_fixed", C.c_double), ("contval_fixed", C.c_double)]


class COracle:
    """Same inputs as oracle.numpy_oracle.Problem (pass one)."""

    def __init__(self, prob, threads: int = 1):
        self.lib = C.CDLL(build())
        os.environ["OMP_NUM_THREADS"] = str(threads)
        try:
            omp = C.CDLL("libgomp.so.1")
            omp.omp_set_num_threads(int(threads))
        except OSError:
            pass
        self.threads = threads
        self.ndim = prob.ndim
        self.npix = prob.wl.size
        self._keep = [np.ascontiguousarray(a, dtype=float) for a in (prob.wl, prob.flux, prob.err)]
        lines = np.ascontiguousarray(np.array(prob.lines, dtype=float).reshape(-1))
        self._keep.append(lines)
        pd = C.POINTER(C.c_double)
        self.pb = _Problem(self.npix, self._keep[0].ctypes.data_as(pd), self._keep[1].ctypes.data_as(pd),
                           self._keep[2].ctypes.data_as(pd), float(prob.velstep), len(prob.lines),
                           lines.ctypes.data_as(pd), (C.c_double * 3)(*prob.linefill), int(prob.ncompmax),
                           int(prob.nfill), int(prob.freespecres), int(prob.freecont),
                           float(max(prob.specres)), float(np.asarray(prob.contval).reshape(-1)[0]))
        self.lib.oracle_re_w.restype = C.c_double
        self.lib.oracle_re_w.argtypes = [C.c_double, C.c_double]

    def loglike_batch(self, P):
        P = np.ascontiguousarray(P, dtype=float).reshape(-1, self.ndim)
        out = np.empty(P.shape[0])
        pd = C.POINTER(C.c_double)
        self.lib.oracle_loglike_batch(C.byref(self.pb), P.ctypes.data_as(pd), C.c_long(P.shape[0]), self.ndim,
                                      out.ctypes.data_as(pd))
        return out

    def model_batch(self, P, targonly=False):
        P = np.ascontiguousarray(P, dtype=float).reshape(-1, self.ndim)
        out = np.empty((P.shape[0], self.npix))
        pd = C.POINTER(C.c_double)
        self.lib.oracle_model_batch(C.byref(self.pb), P.ctypes.data_as(pd), C.c_long(P.shape[0]), self.ndim,
                                    int(bool(targonly)), out.ctypes.data_as(pd))
        return out

    def re_w(self, x, y):
        return self.lib.oracle_re_w(float(x), float(y))
