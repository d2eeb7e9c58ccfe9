"""CPU: host-side logic of the als_fitter mirror (layout, prior box, cube maps, table
reader) against the oracle restatement -- no device calls."""
import os

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import workloads
from mcalf_amd.routines import hires_fitter
from cases import problem_from_kwargs
from oracle import numpy_oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def nodev(monkeypatch):
    monkeypatch.setattr(mcalf_amd.als_fitter, "_open_context", lambda self, dev: None)


def test_reads_reference_table_and_matches_oracle_layout(nodev):
    path = os.path.join(GOLD, "civ_mock_spec_multicomp.txt")
    f = mcalf_amd.als_fitter(path, [[6180, 6220]], ["CIV 1548", "CIV 1550"], [8, 11], nfill=4, specres=[8, 9],
                             Nrange=[12, 14.5], brange=[10, 40], zrange=[2.99, 3.01])
    d = np.loadtxt(path)
    pr = o.Problem(d[:, 0], d[:, 1], d[:, 2], o.CIV_LINES, (8, 11), nfill=4, specres=[8, 9], Nrange=[12, 14.5],
                   brange=[10, 40], zrange=[2.99, 3.01], fitrange=[[6180, 6220]])
    assert (f.ndim, f.startind, f.endind) == (pr.ndim, pr.startind, pr.endind) == (47, 1, 35)
    assert f.velstep == pr.velstep
    for a, b in zip(f.bounds, pr.bounds):
        assert np.array_equal(np.asarray(a, float), np.asarray(b, float))
    cube = np.random.default_rng(0).random(47)
    assert np.array_equal(f._scale_cube_pc(cube), o.scale_cube_pc(pr, cube))
    c2 = cube.copy()
    assert np.array_equal(f._scale_cube_mn(c2, 47, 47), o.scale_cube_mn(pr, cube.copy(), 47, 47))


def test_fitrange_mask_and_default_zbox(nodev):
    path = os.path.join(GOLD, "civ_mock_spec.txt")
    f = mcalf_amd.als_fitter(path, [[6185, 6190], [6200, 6210]], ["CIV 1548", "CIV 1550"], [1, 2], contval=[0.9, 1.1])
    assert f.freecont and not f.freespecres and f.startind == 1 and f.ndim == 2 + 6
    assert np.all(((f.obj_wl > 6185) & (f.obj_wl < 6190)) | ((f.obj_wl > 6200) & (f.obj_wl < 6210)))
    assert abs(f.z_lims[0][0] - ((6185 + 0.25) / 1548.204 - 1)) < 1e-15
    assert f.linefill["wrest"] == 250.0 and f.linefill["f"] == 0.1899


def test_sigma_clipped_median_clips_outliers():
    x = np.concatenate([np.full(100, 1.0) + np.linspace(-1e-3, 1e-3, 100), [50.0, 80.0]])
    assert abs(hires_fitter.sigma_clipped_median(x) - 1.0) < 1e-3
    assert hires_fitter.sigma_clipped_median([1.0, 2.0, 3.0]) == 2.0


def test_unknown_line_needs_explicit_linepars(nodev):
    wl = np.linspace(6180, 6220, 50)
    with pytest.raises(KeyError):
        mcalf_amd.als_fitter(None, [[6180, 6220]], ["XX 9999"], [1, 1], spectrum=(wl, wl * 0 + 1, wl * 0 + 0.02))
    f = mcalf_amd.als_fitter(None, [[6180, 6220]], ["XX 9999"], [1, 1], spectrum=(wl, wl * 0 + 1, wl * 0 + 0.02),
                             linepars=[(1500.0, 0.1, 1e8)])
    assert f.linepars[0]["wrest"] == 1500.0


def test_workload_boxes_match_the_oracle_problem():
    kw, batch, seed = workloads.config("A")
    pr = problem_from_kwargs(kw)
    assert np.array_equal(workloads.bounds_of(kw), np.array([np.asarray(b, float) for b in pr.bounds]))
    P = workloads.draw_P(kw, 32, np.random.default_rng(seed))
    assert P.shape == (32, 7) and np.all(P[:, 0] == 2.0)
    assert np.all((P >= workloads.bounds_of(kw)[:, 0]) & (P <= workloads.bounds_of(kw)[:, 1]))


@pytest.mark.parametrize("style", ["savetxt_header", "bare_header", "csv", "tabs", "no_header", "reordered_csv"])
def test_table_reader_accepts_the_usual_plain_text_layouts(tmp_path, style):
    """The stand-in for astropy's ascii.read (hires_fitter.py:69-72): same three columns whatever the layout."""
    wl = np.linspace(6180.0, 6181.0, 7)
    fl = 1.0 + 0.01 * np.arange(7)
    er = np.full(7, 0.02)
    rows = [(float(w), float(f), float(e)) for w, f, e in zip(wl, fl, er)]      # plain floats: repr round-trips
    path = tmp_path / "spec.txt"
    if style == "savetxt_header":
        np.savetxt(path, np.c_[wl, fl, er], header="Wave Flux Err")
    elif style == "bare_header":
        path.write_text("Wave Flux Err\n" + "\n".join("%r %r %r" % t for t in rows) + "\n")
    elif style == "csv":
        path.write_text("Wave,Flux,Err\n" + "\n".join("%r,%r,%r" % t for t in rows) + "\n")
    elif style == "tabs":
        path.write_text("Wave\tFlux\tErr\n" + "\n".join("%r\t%r\t%r" % t for t in rows) + "\n")
    elif style == "no_header":
        np.savetxt(path, np.c_[wl, fl, er])
    else:
        path.write_text("Err, Wave, Flux\n" + "\n".join("%r, %r, %r" % (e, w, f) for w, f, e in rows) + "\n")
    got = hires_fitter._read_ascii_table(str(path), ["Wave", "Flux", "Err"])
    assert np.array_equal(got[0], wl) and np.array_equal(got[1], fl) and np.array_equal(got[2], er)


@pytest.mark.parametrize("nxcd", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("nrows", [0, 1, 7, 8, 9, 63, 64, 65, 1000, 4096, 32768])
def test_streaming_launch_deals_every_row_to_exactly_one_xcd(nrows, nxcd):
    """The streaming launch of the host-pointer entries deals the rows of a batch to the XCDs in blocks of eight (block k ->
    XCD k % nxcd) and only workgroups on an XCD evaluate its rows: `stream_rows_of` / `stream_row` (kernel_args.h -- the
    constexpr functions the kernels use, reached here through mcalf_stream_partition, pure host arithmetic) must be a
    bijection between the rows and the (XCD, local index) pairs, with the local indices of an XCD contiguous from 0."""
    import ctypes as C
    from mcalf_amd import _lib
    lib = _lib.load()
    owner = np.full(max(nrows, 1), -7, dtype=np.int32)
    local = np.full(max(nrows, 1), -7, dtype=np.int32)
    pi = C.POINTER(C.c_int32)
    rc = lib.mcalf_stream_partition(nrows, nxcd, owner.ctypes.data_as(pi), local.ctypes.data_as(pi))
    assert rc == 0, lib.mcalf_last_error(None)
    owner, local = owner[:nrows], local[:nrows]
    assert ((owner >= 0) & (owner < nxcd)).all()                         # every row has an owner (none left at -1)
    r = np.arange(nrows)
    assert np.array_equal(owner, (r // 8) % nxcd)                        # block k of eight rows -> XCD k % nxcd
    for x in range(nxcd):
        mine = np.sort(local[owner == x])
        assert np.array_equal(mine, np.arange(mine.size))                # local indices 0 .. n_x - 1, each once
        rows = r[owner == x]
        assert np.array_equal(rows[np.argsort(local[owner == x])], rows)  # and in row order: the host stages rows in order


def test_crii_overrides_are_the_reference_s_own_constants(nodev):
    """hires_fitter.py:100-110: after the database look-up the reference REPLACES f and gamma of CrII 2056 / 2062 / 2066
    (the only atomic constants it holds itself); wrest stays the database's.  With `database_linepars=True` the caller's
    triples are treated as raw database values and get the same treatment, by line name."""
    assert hires_fitter.LINE_OVERRIDES == {"CrII 2066": dict(f=0.0512, gamma=4.17e8),
                                           "CrII 2062": dict(f=0.0759, gamma=4.06e8),
                                           "CrII 2056": dict(f=0.103, gamma=4.07e8)}
    names = ["CrII 2056", "CIV 1548", "CrII 2066"]
    db = [(2056.2569, 0.1, 1.0e8), (1548.204, 0.1899, 2.643e8), (2066.164, 0.05, 1.0e8)]     # (f, gamma: placeholders to be replaced)
    got = hires_fitter.apply_line_overrides(names, db)
    assert got == [(2056.2569, 0.103, 4.07e8), (1548.204, 0.1899, 2.643e8), (2066.164, 0.0512, 4.17e8)]
    wl = np.linspace(8220.0, 8270.0, 400)
    kw = dict(spectrum=(wl, np.ones_like(wl), np.full_like(wl, 0.02)), linepars=db, zrange=[2.99, 3.01])
    fit = mcalf_amd.als_fitter(None, [[8220.0, 8270.0]], names, [1, 1], database_linepars=True, **kw)
    assert [(d["wrest"], d["f"], d["gamma"]) for d in fit.linepars] == got
    assert fit.linefill == dict(wrest=250.0, f=0.103, gamma=4.07e8)       # :120-121: a copy of line 0, AFTER the override
    raw = mcalf_amd.als_fitter(None, [[8220.0, 8270.0]], names, [1, 1], **kw)       # explicit triples are taken as given
    assert [(d["wrest"], d["f"], d["gamma"]) for d in raw.linepars] == db
    with pytest.raises(ValueError, match="triples"):
        mcalf_amd.als_fitter(None, [[8220.0, 8270.0]], names, [1, 1], spectrum=kw["spectrum"], linepars=db[:2], zrange=[2.99, 3.01])


def test_multi_device_sharding_is_dist_shard_bounds():
    """mcalf_shard_bounds (pure host arithmetic; what a mcalf_create_multi context cuts its batches with): the entries in use
    cover the batch exactly once with contiguous blocks whose sizes differ by at most one -- the split of
    mc-alf_amd/dist.py::shard_bounds, i.e. of the one-process-per-GPU form -- every entry in use gets at least 256 rows, and
    entries beyond those sit the call out."""
    import ctypes as C
    from mcalf_amd import _lib
    from mcalf_amd import dist as mdist
    lib = _lib.load()
    used, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
    for nent in (1, 2, 3, 8, 16):
        for batch in (0, 1, 255, 256, 511, 512, 1000, 4096, 4097, 16384, 32768, 100003):
            rows = []
            for k in range(nent):
                assert lib.mcalf_shard_bounds(batch, nent, k, C.byref(used), C.byref(lo), C.byref(hi)) == 0
                n = used.value
                assert n == max(1, min(nent, batch // 256))
                if k < n:
                    assert (lo.value, hi.value) == mdist.shard_bounds(batch, n, k)
                    assert n == 1 or hi.value - lo.value >= 256
                    rows += list(range(lo.value, hi.value)) if batch <= 5000 else [lo.value, hi.value]
                else:
                    assert lo.value == hi.value == batch
            if batch <= 5000:
                assert rows == list(range(batch))
    assert lib.mcalf_shard_bounds(10, 0, 0, C.byref(used), C.byref(lo), C.byref(hi)) == -1
    assert lib.mcalf_shard_bounds(10, 2, 2, C.byref(used), C.byref(lo), C.byref(hi)) == -1
    assert lib.mcalf_shard_bounds(10, 2, 0, None, None, None) == -1
