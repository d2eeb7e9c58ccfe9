"""Host-side mirror of `mcalf.routines.hires_fitter.als_fitter` for the likelihood hot path.

Same class name, constructor keywords, method names, argument order and return
conventions as the reference (hires_fitter.py:30-518) for everything the solver branches of
cli.py call on the likelihood path, so the solve step of an existing driver can switch
`from mcalf.routines import hires_fitter` to this module.  Every likelihood / model method is
a batch-of-one call into the HIP library (`libmcalf_hip.so`); `loglike_batch` /
`model_batch` are the vectorised entries.

NOT a complete replacement of the reference module (out of scope; the
list a maintainer needs is in INTEGRATION.md section 2): no solver dispatch, no plotting, no
chain readers -- the module-level `pc_analyzer` and `get_parnames` (hires_fitter.py:704-760),
which cli.py:345-348 calls after a PolyChord run, stay in `mcalf` -- and no prior log-density (`lnprior` and the `__call__`
built on it, hires_fitter.py:218-234,509-518, which no solver branch calls).  linetools is
replaced by an explicit `linepars=` argument plus a tiny built-in table (`LINE_TABLE`).
"""
from __future__ import annotations

import ctypes as C
import gc
from typing import Optional, Sequence

import numpy as np

from .. import _lib

# (wrest [A], f, gamma [1/s]).  CIV: pinned by the reference's own test spectra
# (SURVEY.md section 4).  HI 1215: Morton (2003), NOT pinned (linetools absent).
LINE_TABLE = {
    "CIV 1548": (1548.204, 0.1899, 2.643e8),
    "CIV 1550": (1550.781, 0.09475, 2.628e8),
    "HI 1215": (1215.67, 0.4164, 6.265e8),
}

# The atomic constants the reference itself holds: after the database look-up it REPLACES f and gamma of three CrII
# lines (hires_fitter.py:100-110, "from atomic database in RCooke ALIS code"); wrest stays the database's.
LINE_OVERRIDES = {
    "CrII 2066": dict(f=0.0512, gamma=4.17e8),
    "CrII 2062": dict(f=0.0759, gamma=4.06e8),
    "CrII 2056": dict(f=0.103, gamma=4.07e8),
}


def apply_line_overrides(fitlines, linepars):
    """`linepars` = [(wrest, f, gamma), ...] as a database returned them for `fitlines`, with the reference's in-file
    overrides applied by line name (hires_fitter.py:100-110).  `wrest` must still come from the caller: the linetools
    'ISM' list the reference reads it from is not available here."""
    out = []
    for name, (w, f, g) in zip(fitlines, linepars):
        o = LINE_OVERRIDES.get(name)
        out.append((float(w), float(o["f"]), float(o["gamma"])) if o else (float(w), float(f), float(g)))
    return out


def sigma_clipped_median(x, sigma=3.0, maxiters=5):
    """Median after iterative sigma clipping about the median (the `med` that
    `astropy.stats.sigma_clipped_stats` returns with its defaults; hires_fitter.py:85).
    Pinned only for grids where nothing is clipped (both reference fixtures)."""
    x = np.asarray(x, dtype=float)
    x = x[np.isfinite(x)]
    for _ in range(maxiters):
        med, std = np.median(x), np.std(x)
        keep = np.abs(x - med) <= sigma * std
        if keep.all():
            break
        x = x[keep]
    return float(np.median(x))


def _read_ascii_table(path, coldef):
    """Minimal stand-in for `astropy.io.ascii.read` (hires_fitter.py:69-72) on plain-text tables: whitespace-,
    comma- or tab-separated columns whose names are either a `# Wave Flux Err` comment line (np.savetxt header,
    as in the reference's testdata) or a bare first line (what `ascii.write` / a CSV export produce); without any
    header the columns are taken in `coldef` order."""
    names, skip, delim = None, 0, None

    def split(text):
        return [t.strip() for t in text.split(",")] if "," in text else text.split()

    with open(path) as fh:
        for line in fh:
            s = line.strip()
            if not s:
                skip += 1
                continue
            if s.startswith("#"):
                names = split(s.lstrip("#").strip())
                skip += 1
                continue
            if "," in s:
                delim = ","
            try:
                [float(t) for t in split(s)]
            except ValueError:
                names = split(s)
                skip += 1
            break
    data = np.loadtxt(path, comments="#", skiprows=skip, ndmin=2, delimiter=delim)
    if names is None or len(names) != data.shape[1]:
        names = list(coldef)
    idx = [names.index(c) for c in coldef]
    return data[:, idx[0]], data[:, idx[1]], data[:, idx[2]]


class als_fitter:
    """Drop-in for `mcalf.routines.hires_fitter.als_fitter` (hires_fitter.py:30).

    Extra keywords (all optional): `spectrum=(wl, flux, err)` arrays instead of a file;
    `linepars=[(wrest, f, gamma), ...]` instead of a linetools lookup; `velstep=`;
    `conv_mode='numpy'|'jax'`; `device=` HIP ordinal, or a LIST of ordinals for one context that drives
    several GPUs of this process (`mcalf_create_multi`: the batched entries cut their rows into contiguous
    blocks, one per device, results bit-identical to one device; entries may repeat);
    `gauss_cdf=[n3, n4, n5]` to pin the asymmetric-veto thresholds the reference draws at random.
    """

    def __init__(self, specfile, fitrange, fitlines, ncomp, nfill=0, specres=[7.0], contval=[1.0],
                 Nrange=[11.5, 16], brange=[1, 30], zrange=None, Nrangefill=[11.5, 16], brangefill=[1, 30],
                 wrangefill=None, coldef=['Wave', 'Flux', 'Err'], Gpriors=None, Asymmlike=False, debug=False,
                 *, spectrum=None, linepars=None, velstep=None, conv_mode="numpy", device=-1, gauss_cdf=None,
                 database_linepars=False):
        # public attributes under the names the reference's callers read (hires_fitter.py:46-60)
        self.specfile, self.fitrange, self.fitlines = specfile, fitrange, fitlines
        self.Gpriors, self.Asymmlike, self.debug = Gpriors, bool(Asymmlike), debug
        self.specres = list(np.atleast_1d(specres))
        self.contval = list(np.atleast_1d(contval))
        self.ncompmin, self.ncompmax = ncomp
        self.nfill = nfill
        if self.Asymmlike:
            print("Running asymmetric likelihood")
        self.freecont = len(self.contval) > 1          # hires_fitter.py:54-57
        self.freespecres = len(self.specres) > 1       # :59-62
        self.clight = 2.9979245e5                      # :65
        self.ccgs = 2.9979245e10                       # :66

        if spectrum is not None:
            obj_wl, obj, obj_noise = (np.asarray(a, dtype=float) for a in spectrum)
        else:
            obj_wl, obj, obj_noise = _read_ascii_table(specfile, coldef)
        okrange = np.zeros_like(obj_wl, dtype=bool)    # :75-78
        self.numfitranges = len(self.fitrange)
        for lo, hi in self.fitrange:
            okrange[(obj_wl > lo) & (obj_wl < hi)] = True
        self.obj = np.ascontiguousarray(obj[okrange])
        self.obj_noise = np.ascontiguousarray(obj_noise[okrange])
        self.obj_wl = np.ascontiguousarray(obj_wl[okrange])

        if velstep is None:                            # :84-87
            velsteps = (self.obj_wl[1:] - self.obj_wl[:-1]) / self.obj_wl[1:] * self.clight
            velstep = sigma_clipped_median(velsteps)
        self.velstep = float(velstep)

        self.numlines = len(fitlines)                  # :93-116
        if linepars is None:
            try:
                linepars = [LINE_TABLE[name] for name in fitlines]
            except KeyError as exc:
                raise KeyError(f"line {exc} is not in the built-in table; pass linepars=[(wrest,f,gamma),...]")
        if len(linepars) != self.numlines:
            raise ValueError(f"{self.numlines} fit lines but {len(linepars)} (wrest, f, gamma) triples")
        if database_linepars:                          # :100-110: the caller's triples are raw database values
            linepars = apply_line_overrides(fitlines, linepars)
        self.linepars = [dict(wrest=float(w), f=float(f), gamma=float(g)) for (w, f, g) in linepars]
        self.linefill = dict(self.linepars[0])         # :120-121
        self.linefill["wrest"] = 250.0

        # prior box, in the reference's order [R?][cont?][ncomp][N,z,b]*ncompmax [N,z,b]*nfill
        # (hires_fitter.py:124-200)
        self.cont_lims, self.res_lims = np.array(self.contval), np.array(self.specres)
        self.N_lims, self.b_lims = np.array(Nrange), np.array(brange)
        self.N_lims_fill, self.b_lims_fill = np.array(Nrangefill), np.array(brangefill)
        w0 = self.linepars[0]["wrest"]
        wf = self.linefill["wrest"]
        self.z_lims = [self._zbox(zrange, k, self.ncompmax, w0, as_wave=False,
                                  default=(self.fitrange[0][0] + 0.25, self.fitrange[0][1] - 0.25),
                                  what="Zrange") for k in range(self.ncompmax)]
        self.z_lims_fill = [self._zbox(wrangefill, k, self.nfill, wf, as_wave=True,
                                       default=(np.min(self.obj_wl) + 0.25, np.max(self.obj_wl) - 0.25),
                                       what="Wrangefill") for k in range(self.nfill)]
        self.startind = int(self.freecont) + int(self.freespecres)     # :169-174
        self.endind = self.startind + 3 * self.ncompmax + 1            # :176
        head = ([self.res_lims] if self.freespecres else []) + ([self.cont_lims] if self.freecont else [])
        comps = [lim for k in range(self.ncompmax) for lim in (self.N_lims, self.z_lims[k], self.b_lims)]
        fills = [lim for k in range(self.nfill) for lim in (self.N_lims_fill, self.z_lims_fill[k], self.b_lims_fill)]
        self.bounds = head + [ncomp] + comps + fills
        self.ndim = len(self.bounds)
        self._lo = np.array([np.min(b) for b in self.bounds], dtype=float)
        self._hi = np.array([np.max(b) for b in self.bounds], dtype=float)

        # Asymmetric-veto thresholds (hires_fitter.py:179-181): counts of a standard-normal draw
        # above 3, 4, 5 sigma.  The reference's draw is unseeded; pass gauss_cdf=[n3, n4, n5] to pin it.
        if gauss_cdf is None:
            gauss = np.random.normal(size=len(self.obj))
            gauss_cdf = [(gauss > 3).sum(), (gauss > 4).sum(), (gauss > 5).sum()]
        self.gauss_cdf = [int(v) for v in gauss_cdf]
        self.gracenum = 0.01 * len(self.obj)

        self.conv_mode = conv_mode
        self._ctx = None
        self._open_context(device)

    @staticmethod
    def _zbox(spec, k, count, wrest, as_wave, default, what):
        """Redshift box of slot k.  `spec` None -> from the wavelength window `default`
        (hires_fitter.py:138-139,155-156); 2 numbers -> shared box; 2*count numbers -> per slot.
        `as_wave`: numbers are observed wavelengths (fillers, :158-162) rather than redshifts."""
        if spec is None:
            lo, hi, as_wave = default[0], default[1], True
        elif len(spec) == 2:
            lo, hi = spec[0], spec[1]
        elif (as_wave and len(spec) == 2 * count) or (not as_wave and len(spec) >= 2 * count):
            lo, hi = spec[2 * k], spec[2 * k + 1]
        else:
            raise ValueError(f"{what} keyword not understood.")
        if as_wave:
            lo, hi = lo / wrest - 1., hi / wrest - 1.
        return np.array((lo, hi))

    # ------------------------------------------------------------------ device context
    def _open_context(self, device):
        lib = _lib.load()
        sp = _lib.mcalf_spec()
        sp.npix = self.obj_wl.size
        self._keep = (np.ascontiguousarray(self.obj_wl), np.ascontiguousarray(self.obj),
                      np.ascontiguousarray(self.obj_noise))
        pd = C.POINTER(C.c_double)
        sp.wl, sp.flux, sp.err = (a.ctypes.data_as(pd) for a in self._keep)
        sp.velstep = self.velstep
        sp.nlines = self.numlines
        lines = (_lib.mcalf_line * self.numlines)()
        for i, lp in enumerate(self.linepars):
            lines[i] = _lib.mcalf_line(lp["wrest"], lp["f"], lp["gamma"])
        sp.lines = lines
        sp.fill = _lib.mcalf_line(self.linefill["wrest"], self.linefill["f"], self.linefill["gamma"])
        sp.ncompmax = int(self.ncompmax)
        sp.nfill = int(self.nfill)
        sp.freespecres = int(self.freespecres)
        sp.freecont = int(self.freecont)
        jax = (self.conv_mode == "jax")
        # numpy path: float(max(specres)) (:415-417); JAX path: specres[0] (:572)
        sp.specres_fixed = float(self.specres[0]) if jax else float(max(self.specres))
        # LSF reach: numpy path max(specres); the JAX path sizes its fixed kernel grid from res_lims[1] when the
        # resolution is free (hires_fitter.py:549-550: the SECOND entry, not the maximum) and from max otherwise
        sp.specres_max = float(self.specres[1]) if (jax and self.freespecres) else float(max(self.specres))
        sp.contval_fixed = float(self.contval[0])
        sp.conv_mode = _lib.MCALF_CONV_SAME_EDGE_JAX if jax else _lib.MCALF_CONV_WRAP_NUMPY
        devices = None if np.ndim(device) == 0 else [int(d) for d in device]
        sp.device = int(device) if devices is None else devices[0]
        sp.asymmlike = int(bool(self.Asymmlike))
        sp.asymm_n4 = float(self.gauss_cdf[1])
        sp.asymm_n5 = float(self.gauss_cdf[2])
        ctx = C.c_void_p()
        if devices is None:
            _lib.check(lib.mcalf_create(C.byref(sp), C.byref(ctx)))
        else:
            _lib.check(lib.mcalf_create_multi(C.byref(sp), (C.c_int32 * len(devices))(*devices), len(devices), C.byref(ctx)))
        self._ctx = ctx
        self._lib = lib
        info = _lib.mcalf_info_t()
        _lib.check(lib.mcalf_info(ctx, C.byref(info)), ctx)
        assert info.ndim == self.ndim and info.startind == self.startind and info.endind == self.endind
        self.info = info
        # host-side call overhead matters for the one-theta-at-a-time solvers (a call is ~40 us on the GPU side):
        # one persistent parameter row / result slot with their addresses taken once, and a small cache of the
        # addresses of the caller's arrays (a sampler hands over the same buffers call after call)
        self._p1 = np.empty((1, self.ndim))
        self._o1 = np.empty(1)
        self._p1_ptr, self._o1_ptr = self._p1.ctypes.data, self._o1.ctypes.data

    def _ptr(self, arr):
        """Address of a C-contiguous array's data.  Only the context's OWN persistent one-theta buffers have their address
        cached (`ndarray.ctypes` costs a microsecond of a 15 us call): a caller's array may be reallocated in place
        (`ndarray.resize(refcheck=False)`) without changing its identity, so its address is read on every call."""
        if arr is self._p1:
            return self._p1_ptr
        if arr is self._o1:
            return self._o1_ptr
        return arr.__array_interface__["data"][0]

    def close(self):
        twin = getattr(self, "_twin", None)
        if twin is not None:
            twin.close()
            self._twin = None
        if self._ctx is not None:
            self._lib.mcalf_destroy(self._ctx)
            self._ctx = None

    def set_chunks(self, nchunks):
        """Row blocks a batch is issued in inside the library (0 = automatic, 1 = one launch per batch); results
        do not depend on it."""
        _lib.check(self._lib.mcalf_set_chunks(self._ctx, int(nchunks)), self._ctx)

    def chunks_for(self, batch):
        return int(self._lib.mcalf_get_chunks(self._ctx, int(batch)))

    def set_resident(self, idle_us):
        """Resident one-theta evaluator (mcalf_set_resident): with `idle_us` > 0 the one-theta callables are answered by a
        workgroup that stays on the GPU between calls -- no kernel launch per call -- and leaves by itself after `idle_us`
        microseconds without one; 0 turns it off.  Same bits as the launched form."""
        _lib.check(self._lib.mcalf_set_resident(self._ctx, int(idle_us)), self._ctx)

    def set_cu_mask(self, mask_words):
        """Restrict the context's own streams to the compute units of `mask_words` (uint32 words, bit i of word i // 32 =
        CU i; empty / None removes the mask) -- `mcalf_set_cu_mask`.  Results never depend on it; on anything but the
        eight XCDs of an unpartitioned MI355X large host batches take the row-block pipeline (`last_launch()`)."""
        words = np.ascontiguousarray([] if mask_words is None else mask_words, dtype=np.uint32)
        ptr = words.ctypes.data_as(C.POINTER(C.c_uint32)) if words.size else None
        _lib.check(self._lib.mcalf_set_cu_mask(self._ctx, ptr, int(words.size)), self._ctx)

    def last_launch(self, sub=None):
        """What the last call of this context did (`mcalf_last_launch`): entry plan, row blocks, whether the fused
        kernel ran as the persistent grid, its grid and work-item count, the device entries it was cut over.
        `sub=k`: device entry k of a multi-device context (`mcalf_last_launch_sub`)."""
        info = _lib.mcalf_launch_info_t()
        if sub is None:
            _lib.check(self._lib.mcalf_last_launch(self._ctx, C.byref(info)), self._ctx)
        else:
            _lib.check(self._lib.mcalf_last_launch_sub(self._ctx, int(sub), C.byref(info)), self._ctx)
        return info

    def get_config(self):
        """The knobs the context runs under and the MCALF_* variables its environment had set (`mcalf_get_config`)."""
        buf = C.create_string_buffer(2048)
        _lib.check(self._lib.mcalf_get_config(self._ctx, buf, len(buf)), self._ctx)
        return buf.value.decode()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, type, value, trace):
        self.close()
        gc.collect()

    # ------------------------------------------------------------------ prior transforms
    def _scale_cube_pc(self, cube):
        """Prior transform for PolyChord / dyPolyChord (hires_fitter.py:202-209): a new vector
        `cube * ptp(bounds) + min(bounds)` (separately rounded multiply and add, as numpy evaluates it),
        with the ncomp slot truncated like Python's int()."""
        theta = np.array(cube, dtype=float, copy=True)
        theta *= self._hi - self._lo
        theta += self._lo
        theta[self.startind] = int(theta[self.startind])
        return theta

    def _scale_cube_mn(self, cube, ndim, nparam):
        """Prior transform for MultiNest (hires_fitter.py:211-216): rescales the first `ndim` entries of
        `cube` -- possibly a C double pointer -- in place; the ncomp slot keeps its fraction."""
        for k in range(ndim):
            cube[k] = cube[k] * (self._hi[k] - self._lo[k]) + self._lo[k]
        return cube

    def scale_cube_batch(self, cubes, int_ncomp=True):
        """Vectorised prior transform on the device (rows of `cubes` in [0,1]^ndim)."""
        cubes = np.ascontiguousarray(cubes, dtype=float).reshape(-1, self.ndim)
        out = np.empty_like(cubes)
        pd = C.POINTER(C.c_double)
        _lib.check(self._lib.mcalf_scale_cube_batch(
            self._ctx, self._lo.ctypes.data_as(pd), self._hi.ctypes.data_as(pd), cubes.ctypes.data_as(pd),
            cubes.shape[0], int(bool(int_ncomp)), out.ctypes.data_as(pd)), self._ctx)
        return out

    def loglike_cube_batch(self, cubes, int_ncomp=True, return_theta=True):
        """Unit cube in -> (theta, logL) out: lnlhood_pc(_scale_cube_pc(cube)) for every row
        (hires_fitter.py:202-209, 250-262) in one device pass; the prior transform runs inside the
        per-sample set-up kernel.  `int_ncomp=False` gives the MultiNest flavour (:211-216)."""
        cubes = self._rows(cubes, self.ndim)
        pd = C.POINTER(C.c_double)
        key = bool(int_ncomp)
        if getattr(self, "_prior_key", None) != key:
            _lib.check(self._lib.mcalf_set_prior(self._ctx, self._lo.ctypes.data_as(pd),
                                                 self._hi.ctypes.data_as(pd), int(key)), self._ctx)
            self._prior_key = key
        logL = np.empty(cubes.shape[0])
        theta = np.empty_like(cubes) if return_theta else None
        _lib.check(self._lib.mcalf_loglike_cube_batch(
            self._ctx, cubes.ctypes.data_as(pd), cubes.shape[0],
            theta.ctypes.data_as(pd) if return_theta else None, logL.ctypes.data_as(pd)), self._ctx)
        return (theta, logL) if return_theta else logL

    # ------------------------------------------------------------------ batched entries
    def _rows(self, P, width):
        P = np.ascontiguousarray(P, dtype=float)
        if P.ndim == 1:
            P = P.reshape(1, -1)
        if P.shape[1] != width:
            raise ValueError(f"parameter rows must have {width} entries, got {P.shape[1]}")
        return P

    def loglike_batch(self, P, out=None):
        """logL[i] = lnlhood_worker(P[i]) for every row.  `out` (float64, C-contiguous, one entry per
        row) is filled in place when given; with `P` and `out` in page-locked host memory (e.g.
        `torch.empty(..., pin_memory=True).numpy()`) the two PCIe copies of the call are DMA transfers
        instead of staged ones."""
        P = self._rows(P, self.ndim)
        if out is None:
            out = np.empty(P.shape[0])
        elif out.dtype != np.float64 or not out.flags.c_contiguous or out.size != P.shape[0]:
            raise ValueError("out must be a C-contiguous float64 array with one entry per row")
        rc = self._lib.mcalf_loglike_batch(self._ctx, self._ptr(P), P.shape[0], self._ptr(out))
        if rc:
            _lib.check(rc, self._ctx)
        return out

    def chi2_batch(self, P):
        P = self._rows(P, self.ndim)
        out = np.empty(P.shape[0])
        pd = C.POINTER(C.c_double)
        _lib.check(self._lib.mcalf_chi2_batch(self._ctx, P.ctypes.data_as(pd), P.shape[0],
                                              out.ctypes.data_as(pd)), self._ctx)
        return out

    def model_batch(self, P, targonly=False):
        """model[i, :] = reconstruct_spec(P[i], targonly)."""
        P = self._rows(P, self.ndim)
        out = np.empty((P.shape[0], self.obj_wl.size))
        pd = C.POINTER(C.c_double)
        _lib.check(self._lib.mcalf_model_batch(self._ctx, P.ctypes.data_as(pd), P.shape[0], int(bool(targonly)),
                                               out.ctypes.data_as(pd)), self._ctx)
        return out

    def onecomp_batch(self, Q, fill=False, line=None):
        """Rows (R, cont, N, z, b) -> single-component spectra.  `fill`: the filler line;
        `line=k`: line k of the multiplet alone; default: every line of the component."""
        Q = self._rows(Q, 5)
        # The LDS halo of the context is provisioned from max(specres); the reference would simply build a longer
        # kernel (hires_fitter.py:458-459), so a wider request is an error here rather than a NaN spectrum.
        R = Q[:, 0]
        with np.errstate(invalid="ignore"):
            half = np.where(R > self.velstep, np.ceil(3.0348 * (R / 2.354820) / self.velstep), 0.0)
        if self.conv_mode != "jax" and np.any(half > self.info.n_cap):
            raise ValueError(f"specresolution {float(np.nanmax(R)):g} km/s needs an LSF kernel wider than the context was "
                             f"provisioned for (max(specres) = {max(self.specres):g} km/s); construct als_fitter with a "
                             "larger specres")
        out = np.empty((Q.shape[0], self.obj_wl.size))
        which = 1 if fill else (0 if line is None else 2 + int(line))
        pd = C.POINTER(C.c_double)
        _lib.check(self._lib.mcalf_onecomp_batch(self._ctx, Q.ctypes.data_as(pd), Q.shape[0], which,
                                                 out.ctypes.data_as(pd)), self._ctx)
        return out

    # ------------------------------------------------------------------ reference callables
    def _check_scalar(self, p):
        """The single-point callables of the reference raise where Python's `int()` does: a NaN or
        infinite ncomp slot (`int(p[startind])`, :428) is a ValueError / OverflowError, not a value.
        (The batched entries clamp the slot to [0, ncompmax] instead and never raise.)"""
        int(p[self.startind])

    def chi2(self, p):
        """hires_fitter.py:236-248 (returns `(+inf, [])` for an all-zero model, else a float)."""
        self._check_scalar(p)
        v = float(self.chi2_batch(p)[0])
        if v == np.inf:
            return +np.inf, []
        return v

    def lnlhood_pc(self, p):
        """hires_fitter.py:250-262 -> (logL, [])."""
        return self.lnlhood_worker(p), []

    def lnlhood_dy(self, p):
        """hires_fitter.py:264-272 -> float."""
        return self.lnlhood_worker(p)

    def lnlhood_mn(self, p, ndim, nparam):
        """hires_fitter.py:274-285: `p` may be a C double pointer, hence the copy."""
        parr = np.array([p[x] for x in range(self.ndim)])
        return self.lnlhood_worker(parr)

    def lnlhood_worker(self, p):
        """hires_fitter.py:287-328."""
        self._check_scalar(p)
        p = np.asarray(p, dtype=float)
        if p.shape != (self.ndim,):
            return float(self.loglike_batch(p)[0])          # (raises for a wrong length, as the batched entry does)
        self._p1[0, :] = p
        rc = self._lib.mcalf_loglike_batch(self._ctx, self._p1_ptr, 1, self._o1_ptr)
        if rc:
            _lib.check(rc, self._ctx)
        return float(self._o1[0])

    def reconstruct_spec(self, p, targonly=False):
        """hires_fitter.py:409-449."""
        self._check_scalar(p)
        return self.model_batch(p, targonly)[0]

    def reconstruct_onecomp(self, specresolution, continuum, N, z, b):
        """hires_fitter.py:379-392."""
        return self.onecomp_batch([specresolution, _scalar(continuum), N, z, b], fill=False)[0]

    def reconstruct_onecomp_fill(self, specresolution, continuum, N, z, b):
        """hires_fitter.py:394-406."""
        return self.onecomp_batch([specresolution, _scalar(continuum), N, z, b], fill=True)[0]

    def calc_w(self, p, lineid=0, reference_indexing=True):
        """Rest-frame equivalent width of line `lineid` (hires_fitter.py:467-491).

        `reference_indexing=True` (default) reproduces the reference AS WRITTEN: it loops over all `ncompmax`
        slots and slices `p[3*comp+startind : 3*comp+3+startind]` (:482), i.e. it starts at the ncomp slot
        instead of one past it, so the triples it integrates are (ncomp, N1, z1), (b1, N2, z2), ... read as
        (N, z, b).  `reference_indexing=False` follows the parameter layout of reconstruct_spec (:431) over the
        ACTIVE components -- what the routine evidently intends.  No solver calls either."""
        p = np.asarray(p, dtype=float)
        cont = (p[1] if self.freespecres else p[0]) if self.freecont else _scalar(self.contval)
        if reference_indexing:
            nc, first = int(self.ncompmax), self.startind
        else:
            nc, first = min(max(int(p[self.startind]), 0), self.ncompmax), self.startind + 1
        if nc == 0:
            return 0 if reference_indexing else 0.0
        comps = p[first: first + 3 * nc].reshape(nc, 3)                                    # read as (N, z, b)
        Q = np.column_stack([np.zeros(nc), np.full(nc, cont), comps])                      # R = 0: unconvolved (:483)
        absorption = self.onecomp_batch(Q, line=lineid)
        dlambda = np.diff(self.obj_wl)
        dlambda = np.insert(dlambda, 0, dlambda[0])
        w = np.sum((1 - (absorption / cont)) * dlambda, axis=1)                            # :488
        return float(np.sum(w / (1 + comps[:, 1])))                                        # :489

    def calc_N(self, p, reference_indexing=True):
        """Total column density log10(sum 10**N) of the slots with z < 10 (hires_fitter.py:493-505).

        `reference_indexing=True` (default) evaluates the reference's own expressions: `p[startind::3]` as N and
        `p[startind+1::3]` as z (:499-500) -- strides that start at the ncomp slot, so "N" is (ncomp, b1, b2, ...)
        and "z" is (N1, N2, ...); the two differ in length by one for every valid layout and numpy raises the same
        IndexError the reference raises.  `reference_indexing=False` strides from the first N slot, as the layout
        (:431) implies; the z cut then drops the fillers, which sit at z ~ 24."""
        p = np.asarray(p, dtype=float)
        if reference_indexing:
            allN = p[self.startind::3]
            allz = p[self.startind + 1::3]
            okN = (allz < 10)
            try:
                allN = 10 ** allN[okN]
            except IndexError as exc:
                raise IndexError(f"{exc} -- calc_N(reference_indexing=True) evaluates the reference's expressions as "
                                 "written (hires_fitter.py:499-503), which fail like this for every valid parameter "
                                 "vector; pass reference_indexing=False for the total column density") from exc
            return np.log10(np.sum(allN))
        allN = p[self.startind + 1::3]
        allz = p[self.startind + 2::3]
        n = min(allN.size, allz.size)
        allN, allz = allN[:n], allz[:n]
        return np.log10(np.sum(10 ** allN[allz < 10]))

    @classmethod
    def from_config(cls, run_params, **extra):
        """Construct from the dict `readconfig` returns, exactly as cli.py:73-76 does."""
        return cls(run_params['specfile'], run_params['wavefit'], run_params['linelist'], run_params['ncomp'],
                   nfill=run_params['nfill'], specres=run_params['specres'], contval=run_params['contval'],
                   Nrange=run_params['Nrange'], brange=run_params['brange'], zrange=run_params['zrange'],
                   Nrangefill=run_params['Nrangefill'], brangefill=run_params['brangefill'],
                   wrangefill=run_params['wrangefill'], coldef=run_params['coldef'],
                   Asymmlike=run_params['asymmlike'], **extra)

    def get_jax_likelihood(self, use_jax=None):
        """Counterpart of the closure hires_fitter.py:521-695 returns (`log_likelihood` of cli.py:237,256):
        `log_likelihood(p) -> float32 scalar` with the JAX path's semantics -- theta in float32, floor() on the
        ncomp slot (:616), one exp of the summed optical depth (:663), LSF kernel on the fixed grid of the
        largest resolution (:549-560), zero-padded 'same' convolution that is always applied (:674), first / last
        half_size pixels reset to the unconvolved model (:677-681) -- evaluated by the HIP kernel
        (`conv_mode='jax'` context) in float64 and rounded to float32 on return.  float32 arithmetic itself is
        not reproduced: the reference's own float32 path loses 0.6 in logL to cancellation.

        The closure is batch-capable: `p` of shape [..., ndim] gives float32 [...].  With JAX installed (and
        `use_jax` not False) it is wrapped in `jax.pure_callback`, so it can be traced, jitted and vmapped by
        jaxns exactly like the reference's closure; without JAX it is the plain host function.  `use_jax=True`
        raises ImportError when JAX is absent, as the reference does (:523-524)."""
        twin = self if self.conv_mode == "jax" else self._jax_twin()
        ndim = self.ndim

        def host_loglike(p):
            p32 = np.asarray(p, dtype=np.float32)                        # jaxns hands float32 live points (:529-536)
            if p32.shape[-1:] != (ndim,):
                raise ValueError(f"parameter vectors must have {ndim} entries")
            rows = np.ascontiguousarray(p32.reshape(-1, ndim), dtype=np.float64)
            ll = twin.loglike_batch(rows).astype(np.float32)
            return ll.reshape(p32.shape[:-1]) if p32.ndim > 1 else ll.reshape(())[()]

        try:
            if use_jax is False:
                raise ImportError
            import jax
            import jax.numpy as jnp
        except ImportError:
            if use_jax:
                raise ImportError("JAX is not available.")
            host_loglike.fitter = twin
            return host_loglike

        def log_likelihood(p):                                           # pragma: no cover - needs JAX
            p = jnp.asarray(p, dtype=jnp.float32)
            shape = jax.ShapeDtypeStruct(p.shape[:-1], jnp.float32)
            return jax.pure_callback(lambda q: np.asarray(host_loglike(q), dtype=np.float32).reshape(q.shape[:-1]),
                                     shape, p, vmap_method="expand_dims")
        log_likelihood.fitter = twin
        log_likelihood.host = host_loglike
        return log_likelihood

    def _jax_twin(self):
        """A second context over the same arrays with the JAX path's boundary semantics (kept for the
        lifetime of this object)."""
        if getattr(self, "_twin", None) is None:
            self._twin = als_fitter(
                None, self.fitrange, self.fitlines, [self.ncompmin, self.ncompmax], nfill=self.nfill,
                specres=self.specres, contval=self.contval, Nrange=list(self.N_lims), brange=list(self.b_lims),
                zrange=[z for lim in self.z_lims for z in lim], Nrangefill=list(self.N_lims_fill),
                brangefill=list(self.b_lims_fill),
                wrangefill=[(1 + z) * self.linefill["wrest"] for lim in self.z_lims_fill for z in lim] or None,
                Asymmlike=False, spectrum=(self.obj_wl, self.obj, self.obj_noise),
                linepars=[(lp["wrest"], lp["f"], lp["gamma"]) for lp in self.linepars], velstep=self.velstep,
                conv_mode="jax", device=self.info.device, gauss_cdf=self.gauss_cdf)
        return self._twin


def _scalar(v):
    return float(np.asarray(v, dtype=float).reshape(-1)[0])


# ---------------------------------------------------------------------------------------------
# Module-level helpers of the reference module (usable without an als_fitter instance)
# ---------------------------------------------------------------------------------------------

def write_equal_weights(path, logl, samples):
    """Chain file in the layout the CLI writes (cli.py:314-325):
    columns [weight = 1, -2 logL, parameters...]."""
    logl = np.asarray(logl, dtype=float).reshape(-1)
    samples = np.asarray(samples, dtype=float).reshape(logl.size, -1)
    np.savetxt(path, np.column_stack([np.ones_like(logl), -2.0 * logl, samples]))


_BOOL = {'True': True, 'False': False}


def _floats(txt):
    return np.array(txt.split(','), dtype=float)


# (section, option, key, default, converter); converter=None keeps the string
_CONFIG_TABLE = [
    ('input', 'coldef', 'coldef', ['Wave', 'Flux', 'Err'], lambda t: [x.strip() for x in t.split(',')]),
    ('input', 'specres', 'specres', np.array([7.0]), _floats),
    ('input', 'asymmlike', 'asymmlike', False, lambda t: _BOOL[t]),
    ('input', 'solver', 'solver', 'polychord', None),
    ('components', 'ncomp', 'ncomp', np.array((1, 1), dtype=int), lambda t: np.array(t.split(','), dtype=int)),
    ('components', 'nfill', 'nfill', 0, int),
    ('components', 'contval', 'contval', np.array([1]), _floats),
    ('components', 'Nrange', 'Nrange', np.array((11.5, 16)), _floats),
    ('components', 'brange', 'brange', np.array((1, 30)), _floats),
    ('components', 'zrange', 'zrange', None, _floats),
    ('components', 'Nrangefill', 'Nrangefill', np.array((11.5, 16)), _floats),
    ('components', 'brangefill', 'brangefill', np.array((1, 30)), _floats),
    ('components', 'wrangefill', 'wrangefill', None, _floats),
    ('plots', 'nmaxcols', 'nmaxcols', 5, lambda t: int(t[0])),        # first character only (:886)
    ('plots', 'yrange', 'yrange', np.array((-0.1, 1.2)), _floats),
    ('run', 'dofit', 'dofit', True, lambda t: _BOOL[t]),
    ('run', 'doplot', 'doplot', True, lambda t: _BOOL[t]),
    ('run', 'showprogress', 'showprogress', False, lambda t: _BOOL[t]),
    ('run', 'device', 'device', 'cpu', None),
]


def readconfig(configfile=None, logger=None):
    """INI file -> run-parameter dict with the keys, defaults and quirks of hires_fitter.py:762-969
    (mandatory input.specfile / wavefit / linelist; literal 'True'/'False' booleans; chaindir and
    plotdir prefixed by outdir; solver sections copied verbatim)."""
    try:
        import configparser
    except ImportError:  # pragma: no cover
        import ConfigParser as configparser
    cfg = configparser.ConfigParser()
    cfg.read(configfile)
    for opt in ('specfile', 'wavefit', 'linelist'):
        if not cfg.has_option('input', opt):
            raise configparser.NoOptionError('input', opt)
    edges = cfg.get('input', 'wavefit').split(',')
    if len(edges) % 2 == 1:
        raise ValueError("Number of wavefit values must be even")
    wavefit = [(float(edges[2 * i]), float(edges[2 * i + 1])) for i in range(len(edges) // 2)]

    def opt(section, name, default, conv=None):
        if not cfg.has_option(section, name):
            return default
        txt = cfg.get(section, name)
        return txt if conv is None else conv(txt)

    datadir = opt('pathing', 'datadir', './')
    outdir = opt('pathing', 'outdir', './')
    run = {'specfile': datadir + cfg.get('input', 'specfile'),
           'wavefit': wavefit,
           'linelist': [x.strip() for x in cfg.get('input', 'linelist').split(',')],
           'chaindir': outdir + opt('pathing', 'chaindir', 'fits/'),
           'plotdir': outdir + opt('pathing', 'plotdir', 'plots/'),
           'chainfmt': opt('pathing', 'chainfmt', 'pc_fits_{}_{1}')}
    for section, name, key, default, conv in _CONFIG_TABLE:
        run[key] = opt(section, name, default, conv)
    for section in ('mn_settings', 'pc_settings', 'jaxns_settings'):
        if cfg.has_section(section):
            run[section] = {o: _BOOL.get(cfg.get(section, o), cfg.get(section, o)) for o in cfg.options(section)}
    return run
