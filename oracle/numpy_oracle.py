"""CPU oracle for the MC-ALF likelihood hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float64 numpy/scipy restatement of the reference's *numpy path*
(Voigt optical depth -> transmitted flux -> Gaussian-LSF convolution -> Gaussian
log-likelihood) plus a float64 restatement of the *JAX-path semantics*.  It exists
to check the HIP implementation; it is never imported by the product package
(`mc-alf_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.

Parity status: PINNED.  The numpy-path restatement reproduces the two data files the
reference ships (`testdata/civ_mock_spec.txt`, `testdata/civ_mock_spec_multicomp.txt`,
committed under `tests/golden/`) once the seeded noise of
`testdata/generate_from_model.py:52-54` is subtracted -- see
`tests/test_oracle_golden.py`.  The reference itself cannot be imported here
(astropy / linetools / jax are not installed: ordinary ModuleNotFoundError), so the
third-party pieces are restated:

* `scipy.special.wofz` (call site hires_fitter.py:365) -- available, called directly.
* `astropy.convolution.convolve(..., boundary='wrap', normalize_kernel=True)` with a
  `Gaussian1DKernel(sigma, x_size)` (call site hires_fitter.py:463-464) -- restated in
  `convolve_model` below, pinned by the fixtures.
* `linetools` atomic data (hires_fitter.py:90-96) -- line triples are explicit inputs;
  the CIV doublet constants are pinned by the fixtures (`CIV_LINES`).

Every function cites the reference lines it follows (paths relative to
/root/reference/mcalf/routines/).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy.special import wofz

# hires_fitter.py:65-66
CLIGHT_KMS = 2.9979245e5
CCGS = 2.9979245e10

# (wrest [Angstrom], f, gamma [1/s]); pinned by the fixtures
CIV_LINES = ((1548.204, 0.1899, 2.643e8), (1550.781, 0.09475, 2.628e8))
# NOT pinned by any fixture (linetools absent): Morton (2003) values
HI_LYA_LINE = ((1215.67, 0.4164, 6.265e8),)


@dataclass
class Problem:
    """Arrays + layout flags the likelihood reads (hires_fitter.py:30-200).

    `specres` / `contval` follow the reference convention: a length-1 sequence means
    "fixed", a length-2 sequence means "free parameter with these bounds"
    (hires_fitter.py:54-62).
    """
    wl: np.ndarray            # obj_wl  [Angstrom]
    flux: np.ndarray          # obj
    err: np.ndarray           # obj_noise
    lines: Sequence[Tuple[float, float, float]]
    ncomp: Tuple[int, int]    # (ncompmin, ncompmax)
    nfill: int = 0
    specres: Sequence[float] = (7.0,)
    contval: Sequence[float] = (1.0,)
    velstep: Optional[float] = None
    linefill: Optional[Tuple[float, float, float]] = None
    Nrange: Sequence[float] = (11.5, 16.0)
    brange: Sequence[float] = (1.0, 30.0)
    zrange: Optional[Sequence[float]] = None
    Nrangefill: Sequence[float] = (11.5, 16.0)
    brangefill: Sequence[float] = (1.0, 30.0)
    fitrange: Optional[Sequence[Sequence[float]]] = None
    bounds: List[np.ndarray] = field(default_factory=list)

    def __post_init__(self):
        self.wl = np.asarray(self.wl, dtype=float)
        self.flux = np.asarray(self.flux, dtype=float)
        self.err = np.asarray(self.err, dtype=float)
        self.ncompmin, self.ncompmax = int(self.ncomp[0]), int(self.ncomp[1])
        self.freecont = len(self.contval) > 1        # :54-57
        self.freespecres = len(self.specres) > 1     # :59-62
        if self.velstep is None:
            self.velstep = velstep_of(self.wl)
        self.numlines = len(self.lines)
        if self.linefill is None:                    # :120-121  filler = line 0 at 250 A
            self.linefill = (250.0, self.lines[0][1], self.lines[0][2])
        # :169-176
        self.startind = int(self.freecont) + int(self.freespecres)
        self.endind = self.startind + 3 * self.ncompmax + 1
        self.bounds = self._make_bounds()
        self.ndim = len(self.bounds)

    def _make_bounds(self):
        # hires_fitter.py:133-166 (z boxes) and :184-198 (ordering)
        if self.fitrange is None:
            lo_w, hi_w = float(self.wl.min()), float(self.wl.max())
        else:
            lo_w, hi_w = self.fitrange[0][0], self.fitrange[0][1]
        zl = []
        for k in range(self.ncompmax):
            if self.zrange is None:
                zl.append(np.array(((lo_w + 0.25) / self.lines[0][0] - 1.0,
                                    (hi_w - 0.25) / self.lines[0][0] - 1.0)))
            elif len(self.zrange) == 2:
                zl.append(np.array(self.zrange, dtype=float))
            else:
                zl.append(np.array(self.zrange[2 * k:2 * k + 2], dtype=float))
        zf = np.array(((self.wl.min() + 0.25) / self.linefill[0] - 1.0,
                       (self.wl.max() - 0.25) / self.linefill[0] - 1.0))
        b = []
        if self.freespecres:
            b.append(np.array(self.specres, dtype=float))
        if self.freecont:
            b.append(np.array(self.contval, dtype=float))
        b.append(np.array(self.ncomp, dtype=float))
        for k in range(self.ncompmax):
            b += [np.array(self.Nrange, dtype=float), zl[k], np.array(self.brange, dtype=float)]
        for _ in range(self.nfill):
            b += [np.array(self.Nrangefill, dtype=float), zf, np.array(self.brangefill, dtype=float)]
        return b


def velstep_of(wl: np.ndarray) -> float:
    """hires_fitter.py:84-87.  The reference takes the median of the 3-sigma-clipped
    per-pixel velocity steps (astropy `sigma_clipped_stats`, absent here).  On the
    fixture grids no pixel is clipped, so this is the plain median; for grids where
    clipping would remove pixels the result is NOT pinned -- callers pass `velstep`."""
    wl = np.asarray(wl, dtype=float)
    steps = (wl[1:] - wl[:-1]) / wl[1:] * CLIGHT_KMS
    return float(np.median(steps))


# ----------------------------------------------------------------------------------
# numpy path
# ----------------------------------------------------------------------------------

def voigt_tau(wave_cm: np.ndarray, logN, z, b_cms, wrest_cm, f, gamma) -> np.ndarray:
    """Optical depth of one (component, line); cgs inputs.  hires_fitter.py:357-365."""
    cold = 10.0 ** logN
    zp1 = z + 1.0
    nujk = CCGS / wrest_cm
    dnu = b_cms / wrest_cm
    avoigt = gamma / (4 * np.pi * dnu)
    uvoigt = ((CCGS / (wave_cm / zp1)) - nujk) / dnu
    cne = 0.014971475 * cold * f
    return cne * wofz(uvoigt + 1j * avoigt).real / dnu


def voigt_model(wave_A: np.ndarray, logN, b_kms, z, wrest_A, f, gamma) -> np.ndarray:
    """Transmitted flux exp(-tau) of one line; hires_fitter.py:369-377 (unit conversions
    Angstrom->cm and km/s->cm/s are done exactly as there)."""
    return np.exp(-1 * voigt_tau(wave_A / 1e8, logN, z, b_kms * 1e5, wrest_A / 1e8, f, gamma))


def lsf_kernel(fwhm: float, velstep: float) -> np.ndarray:
    """Gaussian LSF taps: hires_fitter.py:454-459 + astropy `Gaussian1DKernel(sigma,
    x_size)` (Gaussian1D of unit area sampled at integer offsets -n..n).  The kernel is
    returned UN-normalised, as astropy holds it before `normalize_kernel`."""
    sigma = (fwhm / 2.354820) / velstep
    n = np.ceil(3.0348 * sigma)
    x_size = int(2 * n) + 1
    half = (x_size - 1) // 2
    x = np.arange(-half, half + 1, dtype=float)
    return np.exp(-0.5 * x * x / (sigma * sigma)) / (np.sqrt(2 * np.pi) * sigma)


def convolve_model(spec: np.ndarray, fwhm: float, velstep: float) -> np.ndarray:
    """hires_fitter.py:452-464.  astropy `convolve(spec, kernel, boundary='wrap',
    normalize_kernel=True)` restated: the kernel is divided by its sum, the array is
    padded periodically by the kernel half-width, each output is the tap-ordered sum
    `top = sum_k spec[(i+k) mod npix] * w_k` divided by `bot = sum_k w_k` (astropy's
    default nan_treatment='interpolate' code path; with no NaN present bot is the
    normalised-kernel sum, i.e. 1 to rounding)."""
    ker = lsf_kernel(fwhm, velstep)
    ker = ker / ker.sum()
    half = (ker.size - 1) // 2
    npix = spec.size
    idx = (np.arange(-half, npix + half)) % npix
    padded = spec[idx]
    top = np.zeros(npix)
    bot = 0.0
    # astropy's C loop walks the window left to right with the kernel index flipped
    for j in range(ker.size):
        w = ker[ker.size - 1 - j]
        top = top + padded[j:j + npix] * w
        bot = bot + w
    return top / bot


def decode(prob: Problem, p) -> Tuple[float, object, int]:
    """(specresolution, continuum, thisncomp) from a parameter vector;
    hires_fitter.py:412-428."""
    if prob.freespecres:
        R = p[0]
    else:
        R = float(max(prob.specres))
    if prob.freecont:
        cont = p[1] if prob.freespecres else p[0]
    else:
        cont = np.asarray(prob.contval, dtype=float)  # length-1 -> broadcasts (:425)
    return R, cont, int(p[prob.startind])


def reconstruct_spec(prob: Problem, p, targonly: bool = False) -> np.ndarray:
    """Model spectrum for one parameter vector; hires_fitter.py:409-449."""
    R, cont, nc = decode(prob, p)
    s = prob.startind
    model = np.ones_like(prob.flux)
    for c in range(nc):
        N, z, b = p[1 + 3 * c + s: 1 + 3 * c + 3 + s]          # :431  (N, z, b)
        for (wrest, f, gam) in prob.lines:                      # :433-435
            model = model * voigt_model(prob.wl, N, b, z, wrest, f, gam)
    if not targonly:
        wrest, f, gam = prob.linefill
        for k in range(prob.nfill):                             # :438-442
            N, z, b = p[3 * k + prob.endind: 3 * k + 3 + prob.endind]
            model = model * voigt_model(prob.wl, N, b, z, wrest, f, gam)
    if R > prob.velstep:                                        # :445
        return convolve_model(model, R, prob.velstep) * cont
    return model * cont


def reconstruct_onecomp(prob: Problem, R, cont, N, z, b, fill: bool = False) -> np.ndarray:
    """hires_fitter.py:379-392 (`fill=False`) and :394-406 (`fill=True`)."""
    model = np.ones_like(prob.flux)
    for (wrest, f, gam) in ([prob.linefill] if fill else prob.lines):
        model = model * voigt_model(prob.wl, N, b, z, wrest, f, gam)
    if R > prob.velstep:
        return convolve_model(model, R, prob.velstep) * cont
    return model * cont


def lnlhood_worker(prob: Problem, p, asymm_thresholds=None) -> float:
    """hires_fitter.py:287-328.  `asymm_thresholds=(n_gt4, n_gt5)` enables the
    asymmetric veto of :296-303 with explicit thresholds (the reference derives them
    from an unseeded random draw, :179-181, so they are inputs here)."""
    model = reconstruct_spec(prob, p)
    ispec2 = 1.0 / (prob.err ** 2)
    logl = -0.5 * np.nansum(ispec2 * (prob.flux - model) ** 2 - np.log(ispec2) + np.log(2.0 * np.pi))
    if asymm_thresholds is not None:
        resid = (prob.flux - model) / prob.err
        gracenum = 0.01 * prob.flux.size
        if (resid > 5).sum() > asymm_thresholds[1] + gracenum:
            return -np.inf
        if (resid > 4).sum() > asymm_thresholds[0] + gracenum:
            return -np.inf
    return float(logl)


def chi2(prob: Problem, p):
    """hires_fitter.py:236-248."""
    model = reconstruct_spec(prob, p)
    if np.all(model == 0.0):
        return +np.inf, []
    ispec2 = 1.0 / (prob.err ** 2)
    return float(np.nansum(ispec2 * (prob.flux - model) ** 2))


def scale_cube_pc(prob: Problem, cube) -> np.ndarray:
    """Unit cube -> parameters; hires_fitter.py:202-209 (int() on the ncomp slot)."""
    out = np.array(cube, dtype=float, copy=True)
    for i in range(out.size):
        out[i] = out[i] * np.ptp(prob.bounds[i]) + np.min(prob.bounds[i])
        if i == prob.startind:
            out[i] = int(out[i])
    return out


def scale_cube_mn(prob: Problem, cube, ndim, nparam):
    """hires_fitter.py:211-216 (in place, no int())."""
    for i in range(ndim):
        cube[i] = cube[i] * np.ptp(prob.bounds[i]) + np.min(prob.bounds[i])
    return cube


def loglike_batch(prob: Problem, P: np.ndarray) -> np.ndarray:
    return np.array([lnlhood_worker(prob, row) for row in np.asarray(P, dtype=float)])


def model_batch(prob: Problem, P: np.ndarray, targonly: bool = False) -> np.ndarray:
    return np.stack([reconstruct_spec(prob, row, targonly) for row in np.asarray(P, dtype=float)])


# ----------------------------------------------------------------------------------
# JAX-path semantics, evaluated in float64 with scipy's Faddeeva
# (hires_fitter.py:521-695).  The shipped JAX path is float32 with the Algorithm-916
# Voigt-Hjerting function of voigt_jax.py; SURVEY.md section 8(a) row J explains why
# float32 is not a usable parity target (u loses ~1e-2 to cancellation).  This mode
# reproduces the *semantic* differences of that path:
#   * thisncomp = floor(p[startind])                                     (:616)
#   * tau summed over all components/lines/fillers, ONE exp              (:625-663)
#   * kernel on the FIXED grid -half_size..half_size from max specres    (:549-560,667-670)
#   * zero-padded 'same' convolution, ALWAYS applied                     (:674)
#   * first/last half_size pixels reset to the unconvolved model         (:677-681)
#   * continuum is a scalar                                              (:571)
# ----------------------------------------------------------------------------------

def jax_half_size(prob: Problem) -> int:
    """hires_fitter.py:549-559: `res_lims[1]` -- the SECOND entry of specres, not its maximum -- when the
    resolution is free (:550), `np.max(specres)` otherwise (:552-555)."""
    max_res = float(np.asarray(prob.specres, dtype=float)[1]) if prob.freespecres else float(np.max(prob.specres))
    sigma_max = (max_res / 2.354820) / prob.velstep
    return int(np.ceil(np.float32(3.0348 * sigma_max)))


def jax_reconstruct_spec_f64(prob: Problem, p) -> np.ndarray:
    if prob.freespecres:
        R = p[0]
    else:
        R = float(prob.specres[0])                              # :572
    if prob.freecont:
        cont = p[1] if prob.freespecres else p[0]
    else:
        cont = float(prob.contval[0])                           # :571
    nc = int(np.floor(p[prob.startind]))                        # :616
    s = prob.startind
    tau = np.zeros_like(prob.wl)
    for c in range(prob.ncompmax):                              # :628-649 (masked loop)
        if c >= nc:
            continue
        N, z, b = p[1 + 3 * c + s: 1 + 3 * c + 3 + s]
        for (wrest, f, gam) in prob.lines:
            tau = tau + voigt_tau(prob.wl / 1e8, N, z, b * 1e5, wrest / 1e8, f, gam)
    wrest, f, gam = prob.linefill
    for k in range(prob.nfill):                                 # :652-661
        N, z, b = p[3 * k + prob.endind: 3 * k + 3 + prob.endind]
        tau = tau + voigt_tau(prob.wl / 1e8, N, z, b * 1e5, wrest / 1e8, f, gam)
    model = np.exp(-tau)                                        # :663
    half = jax_half_size(prob)
    sigma = (R / 2.354820) / prob.velstep                       # :667
    kx = np.arange(-half, half + 1, dtype=float)
    ker = np.exp(-kx ** 2 / (2 * sigma ** 2))                   # :669
    ker = ker / ker.sum()                                       # :670
    conv = np.convolve(model, ker, mode="same")                 # :674
    idx = np.arange(model.size)
    edge = (idx < half) | (idx >= model.size - half)            # :680
    return np.where(edge, model, conv) * cont                   # :681-683


def jax_loglike_f64(prob: Problem, p) -> float:
    """hires_fitter.py:685-693."""
    model = jax_reconstruct_spec_f64(prob, p)
    ispec2 = 1.0 / (prob.err ** 2)
    return float(-0.5 * np.nansum(ispec2 * (prob.flux - model) ** 2 - np.log(ispec2) + np.log(2.0 * np.pi)))


# ----------------------------------------------------------------------------------
# The JAX path AS SHIPPED: float32 arithmetic and voigt_jax.hjert (Algorithm 916 / asymptotic form).
# Informational only: float32 loses ~1e-2 in u to cancellation, so this is not a
# parity target; tests/test_oracle_golden.py records how far the reference's own two paths are apart.
# numpy float32 arrays keep every intermediate in float32 (XLA may fuse / contract differently, at the
# 6e-8 level -- three orders of magnitude below the effect being shown).
# ----------------------------------------------------------------------------------

_AN = np.arange(1, 28) * 0.5                                   # voigt_jax.py:60-66
_SIG1_W = np.array([7.78800786e-01, 3.67879450e-01, 1.05399221e-01, 1.83156393e-02, 1.93045416e-03,
                    1.23409802e-04, 4.78511765e-06, 1.12535176e-07])    # :77-86
_ERFCX_P = [5.92470169e-5, 1.61224554e-4, -3.46481771e-4, -1.39681227e-3, 1.20588380e-3, 8.69014394e-3,
            -8.01387429e-3, -5.42122945e-2, 1.64048523e-1, -1.66031078e-1, -9.27637145e-2, 2.76978403e-1]


def jax_erfcx(x, dt=np.float32):
    """voigt_jax.py:5-57 (Shepherd & Laframboise 1981)."""
    x = np.asarray(x, dtype=dt)
    one, two, half = dt(1.0), dt(2.0), dt(0.5)
    a = np.abs(x)
    b = (a - two) / (a + two)
    q = (-a * b - two * (b + one) + a) / (a + two) + b
    pp = dt(_ERFCX_P[0])
    for c in _ERFCX_P[1:]:
        pp = pp * q + dt(c)
    q = (pp + one) / (one + two * a)
    d = (pp + one) - q * (one + two * a)
    f = half * d / (a + half) + q
    with np.errstate(over="ignore"):
        return np.where(x >= 0, f, two * np.exp(x * x) - f).astype(dt)


def jax_hjert(x, a, dt=np.float32):
    """voigt_jax.hjert (voigt_jax.py:89-127): Algorithm 916 with a = 0.5 inside r^2 < 111, the
    three-term asymptotic series outside; x array, a scalar."""
    x = np.asarray(x, dtype=dt)
    a = dt(a)
    pi = dt(np.pi)
    with np.errstate(all="ignore"):
        xy = x * a
        exx = np.exp(-x * x)
        sinc = np.where(xy == 0, dt(1.0), np.sin(xy) / np.where(xy == 0, dt(1.0), xy)).astype(dt)   # jnp.sinc(xy/pi)
        f = exx * (jax_erfcx(a, dt) * np.cos(dt(2.0) * xy) + x * np.sin(xy) / pi * sinc)
        y2 = a * a
        an = _AN.astype(dt)
        a2n2 = (an * an).astype(dt)
        s23 = ((np.exp(-((an[None, :] + x[:, None]) ** 2)) + np.exp(-((an[None, :] - x[:, None]) ** 2)))
               / (a2n2[None, :] + y2)).sum(axis=1, dtype=dt)
        s1 = exx * (_SIG1_W.astype(dt) / (a2n2[:8] + y2)).sum(dtype=dt)
        small = f + a / pi * (-np.cos(dt(2.0) * xy) * s1 + dt(0.5) * s23)
        cdt = np.complex64 if dt == np.float32 else np.complex128
        z = (x + 1j * a).astype(cdt)
        aa = (1.0 / (2.0 * z * z)).astype(cdt)
        q = ((1j) / (z * cdt(np.sqrt(np.pi))) * (1.0 + aa * (1.0 + aa * (3.0 + aa * 15.0)))).astype(cdt)
        r2 = x * x + a * a
        return np.where(r2 < dt(111.0), small, q.real.astype(dt)).astype(dt)


def jax_reconstruct_spec_f32(prob: Problem, p, dt=np.float32) -> np.ndarray:
    """hires_fitter.py:521-683 with every captured array and the parameter vector in `dt`."""
    p = np.asarray(p, dtype=dt)
    wl = prob.wl.astype(dt)
    R = p[0] if prob.freespecres else dt(float(prob.specres[0]))
    cont = (p[1] if prob.freespecres else p[0]) if prob.freecont else dt(float(prob.contval[0]))
    nc = int(np.floor(p[prob.startind]))
    s = prob.startind

    def tau_of(N, z, b, wrest, f, gam):                         # :576-599
        with np.errstate(all="ignore"):
            cold = dt(10.0) ** N
            zp1 = z + dt(1.0)
            w_cm = wl / dt(1e8)
            wrest_cm = dt(wrest) / dt(1e8)
            nujk = dt(CCGS) / wrest_cm
            dnu = (b * dt(1e5)) / wrest_cm
            av = dt(gam) / (dt(4.0) * dt(np.pi) * dnu)
            uv = ((dt(CCGS) / (w_cm / zp1)) - nujk) / dnu
            cne = dt(0.014971475) * cold * dt(f)
            return (cne * jax_hjert(uv, av, dt) / dnu).astype(dt)

    tau = np.zeros_like(wl)
    for c in range(min(nc, prob.ncompmax)):
        N, z, b = p[1 + 3 * c + s: 1 + 3 * c + 3 + s]
        comp = np.zeros_like(wl)
        for (wrest, f, gam) in prob.lines:
            comp = (comp + tau_of(N, z, b, wrest, f, gam)).astype(dt)
        tau = tau + comp
    wrest, f, gam = prob.linefill
    for k in range(prob.nfill):
        N, z, b = p[3 * k + prob.endind: 3 * k + 3 + prob.endind]
        tau = (tau + tau_of(N, z, b, wrest, f, gam)).astype(dt)
    model = np.exp(-tau)
    half = jax_half_size(prob)
    sigma = (R / dt(2.354820)) / dt(prob.velstep)
    kx = np.arange(-half, half + 1).astype(dt)
    ker = np.exp(-kx ** 2 / (dt(2.0) * sigma ** 2))
    ker = (ker / ker.sum(dtype=dt)).astype(dt)
    conv = np.convolve(model, ker, mode="same").astype(dt)
    idx = np.arange(model.size)
    edge = (idx < half) | (idx >= model.size - half)
    return (np.where(edge, model, conv) * cont).astype(dt)


def jax_loglike_f32(prob: Problem, p, dt=np.float32) -> float:
    """hires_fitter.py:685-693 in `dt`."""
    model = jax_reconstruct_spec_f32(prob, p, dt)
    err = prob.err.astype(dt)
    obj = prob.flux.astype(dt)
    ispec2 = dt(1.0) / (err ** 2)
    return float(dt(-0.5) * np.nansum(ispec2 * (obj - model) ** 2 - np.log(ispec2) + np.log(dt(2.0) * dt(np.pi)), dtype=dt))


# ----------------------------------------------------------------------------------
# Analysis helpers ("next" rows)
# ----------------------------------------------------------------------------------

def calc_w_reference(prob: Problem, p, lineid: int = 0) -> float:
    """hires_fitter.py:467-491 AS WRITTEN: every one of the ncompmax slots, sliced from `3*comp + startind`
    (:482) -- one short of the layout reconstruct_spec uses (:431), so the triples are (ncomp, N1, z1),
    (b1, N2, z2), ... read as (N, z, b)."""
    p = np.asarray(p, dtype=float)
    if prob.freecont:                                            # :473-479
        cont = p[1] if prob.freespecres else p[0]
    else:
        cont = np.asarray(prob.contval, dtype=float)             # `self.contval`, a length-1 sequence
    wrest, f, gam = prob.lines[lineid]
    Wtot = 0
    for comp in range(prob.ncompmax):                            # :481
        _N, _z, _b = p[3 * comp + prob.startind: 3 * comp + 3 + prob.startind]                 # :482
        absorption = (np.zeros_like(prob.flux) + cont) * voigt_model(prob.wl, _N, _b, _z, wrest, f, gam)   # :483
        dlambda = np.diff(prob.wl)                               # :485-486
        dlambda = np.insert(dlambda, 0, dlambda[0])
        Wtemp = np.sum((1 - (absorption / cont)) * dlambda)      # :488
        Wtot += Wtemp / (1 + _z)                                 # :489
    return float(Wtot)


def calc_N_reference(prob: Problem, p):
    """hires_fitter.py:493-505 AS WRITTEN.  The two strides start at the ncomp slot (:499-500) and differ in
    length by one for every valid parameter vector, so the boolean index raises IndexError -- in the reference
    exactly as here."""
    p = np.asarray(p, dtype=float)
    allN = p[prob.startind::3]                                   # :499
    allz = p[prob.startind + 1::3]                               # :500
    okN = (allz < 10)                                            # :502
    allN = 10 ** allN[okN]                                       # :503
    return np.log10(np.sum(allN))                                # :505


def calc_N_intended(prob: Problem, p) -> float:
    """hires_fitter.py:493-505 with the stride started at the first N slot (the reference starts
    at the ncomp slot, :499-500, and therefore always returns -inf for real columns)."""
    p = np.asarray(p, dtype=float)
    allN = p[prob.startind + 1::3]
    allz = p[prob.startind + 2::3]
    n = min(allN.size, allz.size)
    ok = allz[:n] < 10
    return float(np.log10(np.sum(10 ** allN[:n][ok])))


def calc_w_intended(prob: Problem, p, lineid: int = 0) -> float:
    """Equivalent width as hires_fitter.py:467-491 evidently intends it: the reference slices
    p[3*comp+startind : +3] (:482, missing the +1 of the ncomp slot) over ncompmax components;
    this uses the layout of reconstruct_spec (:431) over the active components."""
    p = np.asarray(p, dtype=float)
    cont = (p[1] if prob.freespecres else p[0]) if prob.freecont else float(np.asarray(prob.contval).reshape(-1)[0])
    wrest, f, gam = prob.lines[lineid]
    dl = np.diff(prob.wl)
    dl = np.insert(dl, 0, dl[0])
    tot = 0.0
    for c in range(int(p[prob.startind])):
        N, z, b = p[1 + 3 * c + prob.startind: 1 + 3 * c + 3 + prob.startind]
        absorption = (np.zeros_like(prob.flux) + cont) * voigt_model(prob.wl, N, b, z, wrest, f, gam)
        tot += np.sum((1 - (absorption / cont)) * dl) / (1 + z)
    return float(tot)


def pc_sort_components(postsamples: np.ndarray) -> np.ndarray:
    """What hires_fitter.py:726-743 leaves in `postsorted`: per posterior sample the (N, z, b) triples of
    its ACTIVE components ordered by increasing redshift, every slot beyond them NaN (and any entry that
    happens to equal the reference's 99 placeholder NaN as well, :743).  The slots before the first triple
    (resolution / continuum / ncomp) are untouched; their count is `(ncols - 1) % 3` (:728)."""
    rows = np.asarray(postsamples, dtype=float)
    first = (rows.shape[1] - 1) % 3 + 1
    result = []
    for row in rows.tolist():
        active = int(row[first - 1])
        triples = [row[first + 3 * c: first + 3 * c + 3] for c in range(active)]
        triples.sort(key=lambda t: t[1])                    # stable, like argsort on distinct redshifts
        flat = [v for t in triples for v in t]
        tail = [float("nan")] * (len(row) - first - len(flat))
        result.append(row[:first] + flat + tail)
    out = np.array(result, dtype=float).reshape(rows.shape)
    out[out == 99] = np.nan
    return out
