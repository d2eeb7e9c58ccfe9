"""Build libmcalf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["mcalf_hip.hip"]
HEADERS = ["voigt_device.h", "voigt_tables.h", os.path.join("..", "..", "include", "mcalf_hip.h")]
HASHED = ["mcalf_hip.hip", "voigt_device.h", "voigt_tables.h"]
HOST_MARKER = b"// Host side: context + C ABI"
TARGET = os.path.join(CSRC, "libmcalf_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"]


def _stale():
    if not os.path.exists(TARGET):
        return True
    t = os.path.getmtime(TARGET)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_hash() -> str:
    """sha256 over the kernel sources (the device part of mcalf_hip.hip, voigt_device.h, voigt_tables.h), 16 hex digits.  The library
    carries it (mcalf_version()), the PMC-derived files under profiles/ are stamped with it, and bench.py drops
    their figures when the two differ -- a kernel edit cannot ship stale utilisation numbers."""
    h = hashlib.sha256()
    for f in HASHED:
        with open(os.path.join(CSRC, f), "rb") as fh:
            data = fh.read()
        if f == "mcalf_hip.hip":
            # the DEVICE part of the file only (everything above the host-side section): an edit of the C ABI's
            # host code does not change what the counters measured
            cut = data.find(HOST_MARKER)
            if cut < 0:
                raise RuntimeError("mcalf_hip.hip no longer contains its host-section marker")
            data = data[:cut]
        h.update(data)
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if it is missing or older than its sources."""
    if not force and not _stale():
        return TARGET
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + [f'-DMCALF_SRC_HASH="{source_hash()}"', "-o", TARGET] + SOURCES
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return TARGET


if __name__ == "__main__":
    import sys
    if "--hash" in sys.argv:                  # (the Makefile stamps the library with it)
        print(source_hash())
    else:
        print(build(force=True, verbose=True))
