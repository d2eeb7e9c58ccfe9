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
    # (header-only helpers -- `static inline` -- are code for the caller's side, not exports of the library)
    inline = set(re.findall(r"static\s+inline\s+[a-z0-9_ \*]+?\b(mcalf_[a-z0-9_]+)\s*\(", txt))
    return sorted(set(re.findall(r"\b(mcalf_[a-z0-9_]+)\s*\(", txt)) - inline)


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
    assert C.sizeof(_lib.mcalf_info_t) == 8 * 4 + 8 + 32 + 4 + 16 * 4 + 4          # (ABI 7: ndevices, devices[16]; padded to 8)
    assert C.sizeof(_lib.mcalf_broker_t) == 2 * 4 + 10 * 8
    assert C.sizeof(_lib.mcalf_launch_info_t) == 4 * 4 + 8 + 13 * 4 + 4 == 80            # (ABI 7: devices_used; padded to 8)


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
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def _spec(npix=100, **over):
    """A valid mcalf_spec (kept alive by the returned tuple)."""
    wl = np.linspace(6180, 6220, max(npix, 1))
    one = np.ones_like(wl)
    err = np.full_like(wl, 0.02)
    pd = C.POINTER(C.c_double)
    lines = (_lib.mcalf_line * 1)(_lib.mcalf_line(1548.204, 0.1899, 2.643e8))
    sp = _lib.mcalf_spec(npix=npix, wl=wl.ctypes.data_as(pd), flux=one.ctypes.data_as(pd), err=err.ctypes.data_as(pd),
                         velstep=1.0, nlines=1, lines=lines, fill=_lib.mcalf_line(250.0, 0.1899, 2.643e8),
                         ncompmax=2, nfill=0, freespecres=0, freecont=0, specres_fixed=8.0, specres_max=8.0,
                         contval_fixed=1.0, conv_mode=0, device=-1, asymmlike=0, asymm_n4=0.0, asymm_n5=0.0)
    for k, v in over.items():
        setattr(sp, k, v)
    return sp, (wl, one, err, lines)


@pytest.mark.parametrize("over,needle", [
    (dict(npix=0), "npix"), (dict(nlines=0), "line"), (dict(ncompmax=-1), "negative"),
    (dict(velstep=0.0), "velstep"), (dict(conv_mode=7), "conv_mode"), (dict(specres_max=float("nan"), freespecres=1), "specres_max"),
])
def test_invalid_specs_are_refused_with_a_message(over, needle):
    """Argument validation happens before any device is touched, so it is testable without a GPU;
    errors come back as codes + mcalf_last_error, never as exceptions across the ABI."""
    lib = _lib.load()
    sp, keep = _spec(**over)
    ctx = C.c_void_p()
    rc = lib.mcalf_create(C.byref(sp), C.byref(ctx))
    assert rc == -1 and not ctx.value                      # MCALF_ERR_INVALID, no half-built context
    assert needle in lib.mcalf_last_error(None).decode()


def test_null_arguments_do_not_crash():
    lib = _lib.load()
    assert lib.mcalf_create(None, None) == -1
    ctx = C.c_void_p()
    assert lib.mcalf_create(None, C.byref(ctx)) == -1
    lib.mcalf_destroy(None)
    assert lib.mcalf_info(None, None) == -1
    assert lib.mcalf_loglike_batch(None, None, 4, None) == -1
    assert lib.mcalf_reserve(None, 4) == -1
    assert lib.mcalf_broker_serve(None, 0, None, 0.0) == -1
    assert lib.mcalf_broker_serve_resident(None, None, 0, None, 0, None, 0.0) == -1
    assert lib.mcalf_set_resident(None, 100) == -1
    one = (C.c_void_p * 1)(None)
    assert lib.mcalf_broker_serve(one, 1, None, 0.0) == -1
    assert lib.mcalf_voigt_hjerting(None, None, -1, None, -1) == -1
    assert lib.mcalf_voigt_hjerting(None, None, 0, None, -1) == 0
    assert lib.mcalf_set_cu_mask(None, None, 0) == -1
    assert lib.mcalf_stream_partition(8, 0, None, None) == -1 and lib.mcalf_stream_partition(8, 4, None, None) == -1
    assert lib.mcalf_get_config(None, None, 0) == -1 and lib.mcalf_last_launch_sub(None, 0, None) == -1
    assert lib.mcalf_create_multi(None, None, 0, C.byref(ctx)) == -1 and not ctx.value
    sp, keep = _spec()
    assert lib.mcalf_create_multi(C.byref(sp), (C.c_int32 * 1)(0), 17, C.byref(ctx)) == -1 and not ctx.value    # 1 .. 16 entries


def test_the_environment_is_read_in_one_place():
    """One configuration object instead of getenv calls scattered over the library: host_config.cpp takes ONE snapshot
    per context (read_environment, called from mcalf_create) and mcalf_get_config prints what the context runs under."""
    csrc = os.path.join(ROOT, "mc-alf_amd", "csrc")
    users = sorted(f for f in os.listdir(csrc) if f.endswith((".cpp", ".hip", ".h")) and "getenv" in open(os.path.join(csrc, f)).read())
    assert users == ["host_config.cpp"], users


def test_product_library_carries_no_failure_injection_hooks():
    """MCALF_TEST_* switches exist only in the -DMCALF_TESTING variant the GPU tests build into a temporary directory
    (mc-alf_amd/build.py: build(testing=True, target=...)); the in-tree product library does not read them."""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"MCALF_TEST_" not in blob
    assert b"MCALF_STREAM_TIMEOUT" in blob                 # (the check does see environment names that ARE there)
