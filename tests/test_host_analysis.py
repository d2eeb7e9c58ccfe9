"""CPU: the host-side helpers the mirror adds beyond the likelihood -- INI reader, chain writer /
reader with the redshift sort, parameter names, total column -- against hand-checked values and
the oracle's loop-for-loop restatements."""
import os

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd.routines import hires_fitter as h
from oracle import numpy_oracle as o

CFG = """
[input]
specfile = data/spec.txt
wavefit = 6180,6220, 6300, 6310
linelist = CIV 1548, CIV 1550
coldef = Wave, Flux, Err
solver = jaxns
specres = 8.0, 9.0
asymmlike = False

[pathing]
datadir = ./in/
outdir = out/
chainfmt = pc_fits_{0}

[components]
ncomp = 8,11
nfill = 4
contval  = 1
Nrange = 12.0,14.5
brange = 10.0, 40.0
zrange = 2.99, 3.01

[run]
dofit = True
doplot = False
device = cpu

[jaxns_settings]
max_samples = 2000
difficult_model = True

[pc_settings]
nlive = 150
do_clustering = False

[plots]
nmaxcols = 37
"""


def test_readconfig_keys_defaults_and_quirks(tmp_path):
    f = tmp_path / "fit.cfg"
    f.write_text(CFG)
    r = h.readconfig(str(f))
    assert r["specfile"] == "./in/data/spec.txt"
    assert r["wavefit"] == [(6180.0, 6220.0), (6300.0, 6310.0)]
    assert r["linelist"] == ["CIV 1548", "CIV 1550"] and r["coldef"] == ["Wave", "Flux", "Err"]
    assert r["solver"] == "jaxns" and r["asymmlike"] is False and r["device"] == "cpu"
    assert np.array_equal(r["specres"], [8.0, 9.0]) and np.array_equal(r["ncomp"], [8, 11]) and r["nfill"] == 4
    assert np.array_equal(r["contval"], [1.0]) and np.array_equal(r["zrange"], [2.99, 3.01])
    assert np.array_equal(r["Nrangefill"], [11.5, 16]) and r["wrangefill"] is None
    assert r["chaindir"] == "out/fits/" and r["plotdir"] == "out/plots/" and r["chainfmt"] == "pc_fits_{0}"
    assert r["nmaxcols"] == 3                      # only the first character is read (hires_fitter.py:886)
    assert r["dofit"] is True and r["doplot"] is False and r["showprogress"] is False
    assert r["jaxns_settings"] == {"max_samples": "2000", "difficult_model": True}
    assert r["pc_settings"] == {"nlive": "150", "do_clustering": False} and "mn_settings" not in r
    bad = tmp_path / "bad.cfg"
    bad.write_text("[input]\nspecfile = a\nlinelist = x\n")
    with pytest.raises(Exception):
        h.readconfig(str(bad))
    odd = tmp_path / "odd.cfg"
    odd.write_text("[input]\nspecfile = a\nlinelist = x\nwavefit = 1,2,3\n")
    with pytest.raises(ValueError):
        h.readconfig(str(odd))


def test_chain_file_layout(tmp_path):
    """write_equal_weights: the `_equal_weights.txt` layout of cli.py:314-325, columns [1, -2 logL, theta...]."""
    rng = np.random.default_rng(3)
    samples = rng.uniform(1, 5, (25, 14))
    logl = rng.normal(-100, 5, 25)
    path = str(tmp_path / "chain_equal_weights.txt")
    h.write_equal_weights(path, logl, samples)
    back = np.loadtxt(path, ndmin=2)
    assert back.shape == (25, 16) and np.all(back[:, 0] == 1.0)
    assert np.allclose(-0.5 * back[:, 1], logl, rtol=0, atol=1e-12) and np.allclose(back[:, 2:], samples, rtol=0, atol=1e-12)


def test_total_column_reference_and_intended(monkeypatch):
    monkeypatch.setattr(mcalf_amd.als_fitter, "_open_context", lambda self, dev: None)
    wl = np.linspace(6180, 6220, 200)
    f = mcalf_amd.als_fitter(None, [[6180, 6220]], ["CIV 1548", "CIV 1550"], [3, 3], nfill=2,
                             spectrum=(wl, wl * 0 + 1, wl * 0 + 0.02))
    prob = o.Problem(wl, wl * 0 + 1, wl * 0 + 0.02, o.CIV_LINES, (3, 3), nfill=2, fitrange=[[6180, 6220]])
    p = np.array([3.0, 13.0, 3.0, 10.0, 13.5, 3.001, 12.0, 14.0, 3.002, 15.0, 12.0, 23.8, 5.0, 12.5, 23.81, 6.0])
    assert abs(f.calc_N(p, reference_indexing=False) - o.calc_N_intended(prob, p)) < 1e-14
    assert abs(f.calc_N(p, reference_indexing=False) - np.log10(10 ** 13.0 + 10 ** 13.5 + 10 ** 14.0)) < 1e-13   # fillers (z ~ 24) excluded
    # the reference as written (hires_fitter.py:499-503): strides of unequal length -> IndexError, here as there
    with pytest.raises(IndexError):
        o.calc_N_reference(prob, p)
    with pytest.raises(IndexError):
        f.calc_N(p)
    # a layout whose strides happen to have equal length (not reachable from als_fitter's own ndim, but the
    # expressions are the reference's): the same number out of both
    q = np.array([3.0, 13.0, 3.0, 10.0, 13.5, 3.001, 12.0, 14.0, 3.002, 15.0, 12.0, 23.8, 5.0, 12.5, 23.81, 6.0, 1.0])
    assert f.calc_N(q) == o.calc_N_reference(prob, q)


def test_readconfig_on_the_reference_s_own_example_configuration():
    """tests/golden/fit.cfg is the reference's testdata/fit.cfg, byte for byte: the INI file `mcalf` is started with
    (cli.py:60-76).  Expected values by reading the reference's parser (hires_fitter.py:762-969): strings split on commas
    and stripped, numeric lists as float / int arrays, `nmaxcols` = the first character only, settings sections with
    'True' / 'False' turned into booleans and everything else left a string, `device` stripped by configparser."""
    r = h.readconfig(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fit.cfg"))
    assert r["specfile"] == "./testdata/civ_mock_spec_multicomp.txt"          # datadir + specfile (:909)
    assert r["wavefit"] == [(6180.0, 6220.0)] and r["linelist"] == ["CIV 1548", "CIV 1550"]
    assert r["coldef"] == ["Wave", "Flux", "Err"] and r["solver"] == "jaxns" and r["asymmlike"] is False
    assert np.array_equal(r["specres"], [8.0]) and np.array_equal(r["ncomp"], [8, 11]) and r["ncomp"].dtype.kind == "i"
    assert r["nfill"] == 0 and np.array_equal(r["contval"], [1.0])
    assert np.array_equal(r["Nrange"], [12.0, 14.5]) and np.array_equal(r["brange"], [10.0, 40.0])
    assert np.array_equal(r["zrange"], [2.99, 3.01])
    assert np.array_equal(r["Nrangefill"], [11.5, 16.0]) and np.array_equal(r["brangefill"], [1.0, 30.0]) and r["wrangefill"] is None
    assert r["chaindir"] == "testdata/output/fits/" and r["plotdir"] == "testdata/output/plots/" and r["chainfmt"] == "pc_fits_{0}"
    assert r["nmaxcols"] == 3 and np.array_equal(r["yrange"], [-0.1, 1.2])
    assert r["dofit"] is True and r["doplot"] is True and r["showprogress"] is False and r["device"] == "cpu"
    assert r["jaxns_settings"] == {"max_samples": "2000", "num_live_points": "200", "difficult_model": True}
    assert r["pc_settings"]["nlive"] == "150" and r["pc_settings"]["do_clustering"] is False and r["pc_settings"]["equals"] is True
    assert len(r["pc_settings"]) == 13 and "mn_settings" not in r
    assert set(r) == {"specfile", "wavefit", "linelist", "coldef", "asymmlike", "solver", "specres", "chaindir", "plotdir", "chainfmt",
                      "ncomp", "nfill", "Nrange", "brange", "zrange", "Nrangefill", "brangefill", "wrangefill", "contval", "nmaxcols",
                      "yrange", "dofit", "doplot", "showprogress", "pc_settings", "jaxns_settings", "device"}
