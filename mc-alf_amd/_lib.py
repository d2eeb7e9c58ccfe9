"""ctypes binding of libmcalf_hip.so (C ABI in include/mcalf_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
present when a context is created, this fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCALF_HIP_LIB") or os.path.join(_HERE, "csrc", "libmcalf_hip.so")

MCALF_OK = 0
MCALF_ERR_INVALID, MCALF_ERR_HIP, MCALF_ERR_NODEVICE, MCALF_ERR_RANGE, MCALF_ERR_NOMEM, MCALF_ERR_COMM = -1, -2, -3, -4, -5, -6
MCALF_COMM_ID_BYTES = 128
MCALF_CONV_WRAP_NUMPY = 0
MCALF_CONV_SAME_EDGE_JAX = 1

_ERR_NAMES = {-1: "MCALF_ERR_INVALID", -2: "MCALF_ERR_HIP", -3: "MCALF_ERR_NODEVICE",
              -4: "MCALF_ERR_RANGE", -5: "MCALF_ERR_NOMEM", -6: "MCALF_ERR_COMM"}


class mcalf_line(C.Structure):
    _fields_ = [("wrest_A", C.c_double), ("f", C.c_double), ("gamma", C.c_double)]


class mcalf_spec(C.Structure):
    _fields_ = [
        ("npix", C.c_int64),
        ("wl", C.POINTER(C.c_double)),
        ("flux", C.POINTER(C.c_double)),
        ("err", C.POINTER(C.c_double)),
        ("velstep", C.c_double),
        ("nlines", C.c_int32),
        ("lines", C.POINTER(mcalf_line)),
        ("fill", mcalf_line),
        ("ncompmax", C.c_int32),
        ("nfill", C.c_int32),
        ("freespecres", C.c_int32),
        ("freecont", C.c_int32),
        ("specres_fixed", C.c_double),
        ("specres_max", C.c_double),
        ("contval_fixed", C.c_double),
        ("conv_mode", C.c_int32),
        ("device", C.c_int32),
        ("asymmlike", C.c_int32),
        ("asymm_n4", C.c_double),
        ("asymm_n5", C.c_double),
    ]


class mcalf_info_t(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("ndim", C.c_int32),
        ("startind", C.c_int32),
        ("endind", C.c_int32),
        ("n_cap", C.c_int32),
        ("tile", C.c_int32),
        ("ntiles", C.c_int32),
        ("device", C.c_int32),
        ("npix", C.c_int64),
        ("arch", C.c_char * 32),
        ("ndevices", C.c_int32),
        ("devices", C.c_int32 * 16),
    ]


(MCALF_PATH_NONE, MCALF_PATH_DEVICE, MCALF_PATH_HOST_ZEROCOPY, MCALF_PATH_HOST_PIPELINED, MCALF_PATH_HOST_STAGED,
 MCALF_PATH_HOST_STREAM) = range(6)


class mcalf_launch_info_t(C.Structure):
    _fields_ = [
        ("path", C.c_int32),
        ("row_blocks", C.c_int32),
        ("persistent", C.c_int32),
        ("grid", C.c_int32),
        ("items", C.c_int64),
        ("lines_per_sync", C.c_int32),
        ("selfhalo", C.c_int32),
        ("pinned_in", C.c_int32),
        ("pinned_out", C.c_int32),
        ("inline_setup", C.c_int32),
        ("ordered", C.c_int32),
        ("stream_setup_wgs", C.c_int32),
        ("stream_polled", C.c_int32),
        ("xcd_mask", C.c_int32),
        ("stream_wgs_min", C.c_int32),
        ("stream_wgs_max", C.c_int32),
        ("stream_fallback", C.c_int32),
        ("devices_used", C.c_int32),
    ]


MCALF_STREAM_FALLBACK_SHAPE, MCALF_STREAM_FALLBACK_TIMEOUT, MCALF_STREAM_FALLBACK_STARVED = 1, 2, 3


class mcalf_broker_t(C.Structure):
    _fields_ = [
        ("slots", C.c_int32),
        ("ndim", C.c_int32),
        ("req", C.c_void_p),
        ("ack", C.c_void_p),
        ("counter_stride", C.c_int64),
        ("theta", C.c_void_p),
        ("theta_stride", C.c_int64),
        ("logl", C.c_void_p),
        ("logl_stride", C.c_int64),
        ("stop", C.c_void_p),
        ("stats", C.c_void_p),
        ("idle_sleep_after_s", C.c_double),
    ]


# every symbol include/mcalf_hip.h declares: name -> (restype, argtypes)
_PD = C.POINTER(C.c_double)
_CTX = C.c_void_p
SYMBOLS = {
    "mcalf_create": (C.c_int, [C.POINTER(mcalf_spec), C.POINTER(_CTX)]),
    "mcalf_create_multi": (C.c_int, [C.POINTER(mcalf_spec), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_CTX)]),
    "mcalf_shard_bounds": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mcalf_destroy": (None, [_CTX]),
    "mcalf_get_config": (C.c_int, [_CTX, C.c_char_p, C.c_int64]),
    "mcalf_last_launch_sub": (C.c_int, [_CTX, C.c_int32, C.POINTER(mcalf_launch_info_t)]),
    "mcalf_info": (C.c_int, [_CTX, C.POINTER(mcalf_info_t)]),
    "mcalf_last_error": (C.c_char_p, [_CTX]),
    "mcalf_version": (C.c_char_p, []),
    "mcalf_reserve": (C.c_int, [_CTX, C.c_int64]),
    "mcalf_set_chunks": (C.c_int, [_CTX, C.c_int32]),
    "mcalf_get_chunks": (C.c_int32, [_CTX, C.c_int64]),
    # (host pointers as void*: plain addresses are accepted, which spares the hot wrappers two ctypes pointer objects per call)
    "mcalf_loglike_batch": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_void_p]),
    "mcalf_model_batch": (C.c_int, [_CTX, _PD, C.c_int64, C.c_int32, _PD]),
    "mcalf_chi2_batch": (C.c_int, [_CTX, _PD, C.c_int64, _PD]),
    "mcalf_onecomp_batch": (C.c_int, [_CTX, _PD, C.c_int64, C.c_int32, _PD]),
    "mcalf_loglike_batch_device": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mcalf_model_batch_device": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mcalf_last_launch": (C.c_int, [_CTX, C.POINTER(mcalf_launch_info_t)]),
    "mcalf_set_cu_mask": (C.c_int, [_CTX, C.POINTER(C.c_uint32), C.c_int32]),
    "mcalf_stream_partition": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mcalf_profile_begin": (C.c_int, [_CTX, C.c_int32]),
    "mcalf_profile_end": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "mcalf_scale_cube_batch": (C.c_int, [_CTX, _PD, _PD, _PD, C.c_int64, C.c_int32, _PD]),
    "mcalf_set_prior": (C.c_int, [_CTX, _PD, _PD, C.c_int32]),
    "mcalf_loglike_cube_batch": (C.c_int, [_CTX, _PD, C.c_int64, _PD, _PD]),
    "mcalf_loglike_cube_batch_device": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcalf_set_resident": (C.c_int, [_CTX, C.c_int32]),
    "mcalf_broker_serve_resident": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_double]),
    "mcalf_broker_serve": (C.c_int, [C.POINTER(_CTX), C.c_int32, C.POINTER(mcalf_broker_t), C.c_double]),
    "mcalf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mcalf_comm_init": (C.c_int, [_CTX, C.c_void_p, C.c_int32, C.c_int32]),
    "mcalf_comm_info": (C.c_int, [_CTX, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mcalf_comm_destroy": (C.c_int, [_CTX]),
    "mcalf_comm_set_overlap": (C.c_int, [_CTX, C.c_int32]),
    "mcalf_comm_join": (C.c_int, [_CTX, C.c_void_p]),
    "mcalf_loglike_gather_device": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mcalf_loglike_gatherv_device": (C.c_int, [_CTX, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                               C.c_int32, C.c_void_p]),
    "mcalf_voigt_hjerting": (C.c_int, [_PD, _PD, C.c_int64, _PD, C.c_int32]),
    "mcalf_voigt_hjerting_nodes": (C.c_int, [_PD, _PD, C.c_int64, _PD, C.c_int32]),
}

_lib = None


def load():
    """Load libmcalf_hip.so once and set the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mc-alf_amd/csrc`). There is no CPU fallback for this package.")
    # If torch is (or will be) in the process, its bundled HIP runtime must be the one both
    # sides use, otherwise device pointers cannot be shared: import it first when available.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the host-pointer API
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None):
    if rc == MCALF_OK:
        return
    lib = load()
    msg = lib.mcalf_last_error(ctx)
    raise RuntimeError(f"{_ERR_NAMES.get(rc, rc)}: {msg.decode() if msg else ''}")
