"""CPU: the C-ABI library loads and exports every symbol include/mcalf_hip.h declares, and
refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mcalf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mcalf_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    names = _declared_symbols()
    assert names, "no declarations found"
    assert sorted(_lib.SYMBOLS) == names


def test_library_exports_every_symbol():
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.mcalf_version()


def test_struct_sizes_match_header_layout():
    assert C.sizeof(_lib.mcalf_line) == 24
    assert C.sizeof(_lib.mcalf_spec) == 8 + 3 * 8 + 8 + 8 + 8 + 24 + 4 * 4 + 3 * 8 + 2 * 4 + 8 + 2 * 8
    assert C.sizeof(_lib.mcalf_info_t) == 8 * 4 + 8 + 32


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wl = np.linspace(6180, 6220, 200)
    with pytest.raises(RuntimeError, match="NODEVICE|no HIP device|HIP"):
        mcalf_amd.als_fitter(None, [[6180, 6220]], ["CIV 1548"], [1, 1],
                             spectrum=(wl, np.ones_like(wl), np.full_like(wl, 0.02)))
    x = np.zeros(4)
    out = np.zeros(4)
    pd = C.POINTER(C.c_double)
    rc = _lib.load().mcalf_voigt_hjerting(x.ctypes.data_as(pd), x.ctypes.data_as(pd), 4, out.ctypes.data_as(pd), -1)
    assert rc != 0


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mc-alf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
