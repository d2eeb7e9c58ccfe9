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
