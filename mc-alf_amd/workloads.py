"""Synthetic workload definitions: BASELINE.json configs A-E.

Pure numpy; no oracle and no device code.  A config is a dict of `als_fitter` keyword
arguments plus the batch size / seed of the parameter draw.  Spectra that need a model
(`B`..`E`: truth profile + noise) take a `synth(kwargs, p_truth) -> flux[npix]` callback,
so the bench synthesises the truth with the HIP path and the CPU tests with the oracle.
"""
from __future__ import annotations

import os

import numpy as np

CIV = [(1548.204, 0.1899, 2.643e8), (1550.781, 0.09475, 2.628e8)]
HI = [(1215.67, 0.4164, 6.265e8)]

# truth table of the reference's mock generator (testdata/generate_from_model.py:12-14)
TRUTH_Z = [2.999, 2.9995, 3.0, 3.001, 3.0005, 3.0015, 3.002, 3.0025, 3.0035, 3.0039]
TRUTH_N = [13.6, 13.0, 13.8, 13.6, 13.2, 13.4, 13.5, 14.0, 14.2, 13.7]
TRUTH_B = [17.5, 8.0, 20.0, 25.0, 15.0, 30.0, 10.0, 25.0, 15.0, 20.0]

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def truth_vector(ncomp=10):
    p = [float(ncomp)]
    for i in range(ncomp):
        p += [TRUTH_N[i], TRUTH_Z[i], TRUTH_B[i]]
    return np.array(p)


def _civ_spectrum(npix_total, synth, noise_seed):
    wl = np.linspace(6180, 6220, npix_total)[1:-1]
    err = np.full_like(wl, 0.02)
    kw = dict(fitrange=[[6180, 6220]], fitlines=["CIV 1548", "CIV 1550"], linepars=CIV,
              ncomp=[10, 10], specres=[8.0], spectrum=(wl, np.ones_like(wl), err))
    flux = synth(kw, truth_vector(10))
    flux = flux + np.random.default_rng(noise_seed).normal(0, 0.02, wl.size)
    return wl, flux, err


def config(name, synth=None):
    """Return (als_fitter kwargs, batch, seed) for config `name` in 'A'..'E'."""
    name = name.upper()
    if name == "A":
        d = np.loadtxt(os.path.join(_GOLDEN, "civ_mock_spec_multicomp.txt"))
        kw = dict(fitrange=[[6180, 6220]], fitlines=["CIV 1548", "CIV 1550"], linepars=CIV, ncomp=[2, 2],
                  specres=[8.0], Nrange=[12.0, 14.5], brange=[10.0, 40.0], zrange=[2.99, 3.01],
                  spectrum=(d[:, 0], d[:, 1], d[:, 2]))
        return kw, 1024, 0
    if name in ("B", "C", "D"):
        wl, flux, err = _civ_spectrum(4002, synth, 42)
        kw = dict(fitrange=[[6180, 6220]], fitlines=["CIV 1548", "CIV 1550"], linepars=CIV,
                  Nrange=[12.0, 14.5], brange=[10.0, 40.0], zrange=[2.99, 3.01], spectrum=(wl, flux, err))
        if name == "B":
            kw.update(ncomp=[8, 8], specres=[8.0])
            return kw, 1024, 1
        kw.update(ncomp=[8, 11], nfill=4, specres=[8.0, 9.0], Nrangefill=[11.5, 16], brangefill=[1, 30])
        return (kw, 4096, 2) if name == "C" else (kw, 32768, 3)
    if name == "E":
        wl = np.linspace(4662.68, 5062.68, 20002)[1:-1]
        err = np.full_like(wl, 0.02)
        kw = dict(fitrange=[[4662.68, 5062.68]], fitlines=["HI 1215"], linepars=HI, ncomp=[16, 16],
                  specres=[8.0], Nrange=[12.0, 21.0], brange=[5.0, 100.0], zrange=[2.85, 3.15],
                  spectrum=(wl, np.ones_like(wl), err))
        rng = np.random.default_rng(44)
        truth = draw_P(kw, 1, rng, damped=2)[0]
        flux = synth(kw, truth) + rng.normal(0, 0.02, wl.size)
        kw["spectrum"] = (wl, flux, err)
        return kw, 16384, 4
    raise ValueError(name)


def bounds_of(kw):
    """Prior box implied by the kwargs, in the reference's parameter order
    (hires_fitter.py:184-198); mirrors als_fitter without needing a device."""
    wl = np.asarray(kw["spectrum"][0], dtype=float)
    ok = np.zeros(wl.size, dtype=bool)                  # range selection, hires_fitter.py:75-82
    for lo_w, hi_w in kw["fitrange"]:
        ok |= (wl > lo_w) & (wl < hi_w)
    wl = wl[ok]
    specres = list(np.atleast_1d(kw.get("specres", [7.0])))
    contval = list(np.atleast_1d(kw.get("contval", [1.0])))
    nmax = kw["ncomp"][1]
    nfill = kw.get("nfill", 0)
    b = []
    if len(specres) > 1:
        b.append(specres)
    if len(contval) > 1:
        b.append(contval)
    b.append(list(kw["ncomp"]))
    zr = kw.get("zrange")
    w0 = kw["linepars"][0][0]
    for k in range(nmax):
        if zr is None:
            z = [(kw["fitrange"][0][0] + 0.25) / w0 - 1.0, (kw["fitrange"][0][1] - 0.25) / w0 - 1.0]
        elif len(zr) == 2:
            z = list(zr)
        else:
            z = list(zr[2 * k:2 * k + 2])
        b += [list(kw.get("Nrange", [11.5, 16])), z, list(kw.get("brange", [1, 30]))]
    zf = [(wl.min() + 0.25) / 250.0 - 1.0, (wl.max() - 0.25) / 250.0 - 1.0]
    for _ in range(nfill):
        b += [list(kw.get("Nrangefill", [11.5, 16])), zf, list(kw.get("brangefill", [1, 30]))]
    return np.array(b, dtype=float)


def draw_P(kw, batch, rng, damped=0):
    """Uniform draws from the prior box with int() on the ncomp slot, exactly what
    `_scale_cube_pc` (hires_fitter.py:202-209) does to a unit-cube sample.  `damped` forces
    that many components per row to N in [20, 21] (config E, damping-wing regime)."""
    b = bounds_of(kw)
    lo, hi = b.min(axis=1), b.max(axis=1)
    P = rng.random((batch, b.shape[0])) * (hi - lo) + lo
    start = int(len(np.atleast_1d(kw.get("specres", [7.0]))) > 1) + int(len(np.atleast_1d(kw.get("contval", [1.0]))) > 1)
    P[:, start] = np.trunc(P[:, start])
    if damped:
        nmax = kw["ncomp"][1]
        for r in range(batch):
            which = rng.choice(nmax, size=damped, replace=False)
            for c in which:
                P[r, start + 1 + 3 * c] = rng.uniform(20.0, 21.0)
    return P
