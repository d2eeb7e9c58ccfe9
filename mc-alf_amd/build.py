"""Build libmcalf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is seven translation units: kernels.hip (every kernel: the only file compiled for the device) and the host
side of the C ABI -- host_abi.cpp, host_stream.cpp, host_config.cpp, host_multi.cpp, broker.cpp, comm.cpp -- compiled as plain C++ against the HIP
runtime API.  Objects are kept under csrc/obj/ so that an edit of a host file does not recompile the kernels (22 s)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCE = "kernels.hip"
HOST_SOURCES = ["host_abi.cpp", "host_stream.cpp", "host_config.cpp", "host_multi.cpp", "broker.cpp", "comm.cpp"]
SOURCES = [KERNEL_SOURCE] + HOST_SOURCES
# what the DEVICE code is made of: the kernel-source hash covers exactly these
HASHED = ["kernels.hip", "kernel_args.h", "voigt_device.h", "voigt_tables.h"]
HOST_HEADERS = ["host_ctx.h", "kernel_args.h", "voigt_tables.h", os.path.join("..", "..", "include", "mcalf_hip.h")]
TARGET = os.path.join(CSRC, "libmcalf_hip.so")
OBJDIR = os.path.join(CSRC, "obj")
ROCM_INCLUDE = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
KERNEL_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
HOST_FLAGS = ["-x", "c++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__", "-I" + ROCM_INCLUDE]


def _hipcc() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _mtime(name: str) -> float:
    return os.path.getmtime(os.path.join(CSRC, name))


def _deps(src: str):
    return [src] + (HASHED if src == KERNEL_SOURCE else HOST_HEADERS)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(_mtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 over the kernel sources (kernels.hip, kernel_args.h, voigt_device.h, voigt_tables.h), 16 hex digits.  The
    library carries it (mcalf_version()), the PMC-derived files under profiles/ are stamped with it, and bench.py drops
    their figures when the two differ -- a kernel edit cannot ship stale utilisation numbers, and an edit of the host
    side of the C ABI (the .cpp files) does not change what the counters measured."""
    h = hashlib.sha256()
    for f in HASHED:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _run(cmd, verbose=False):
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed (%s):\n%s%s" % (" ".join(cmd), res.stdout, res.stderr))
    if verbose:
        print(res.stderr)


def build(force: bool = False, verbose: bool = False, testing: bool = False, target: str | None = None) -> str:
    """Compile the HIP library if it is missing or older than its sources.

    testing=True builds the TEST variant (-DMCALF_TESTING: the failure-injection hooks MCALF_TEST_FAIL_PREFLIGHT and
    MCALF_TEST_XCD_MASK exist only there) into `target` -- never the in-tree product library; it shares the product's
    kernel object, so only the host files are recompiled."""
    if testing and not target:
        raise ValueError("a testing build needs an explicit target outside the product path")
    target = target or TARGET
    objdir = OBJDIR + ("_testing" if testing else "")
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    stamp = source_hash()
    objs = []
    relinked = False
    # the kernel object (shared by both variants); the hash it was built from is kept next to it
    kobj = os.path.join(OBJDIR, "kernels.o")
    kstamp = kobj + ".hash"
    have = open(kstamp).read().strip() if os.path.exists(kstamp) else ""
    if (force and not testing) or have != stamp or _stale(kobj, _deps(KERNEL_SOURCE)):
        cmd = [hipcc] + KERNEL_FLAGS + ["-c", KERNEL_SOURCE, "-o", kobj]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        _run(cmd, verbose)
        with open(kstamp, "w") as fh:
            fh.write(stamp)
        relinked = True
    objs.append(kobj)
    for src in HOST_SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        # (host_abi.cpp carries the hash string: it is recompiled when the kernels changed)
        if force or relinked or _stale(obj, _deps(src)):
            cmd = [hipcc] + HOST_FLAGS + [f'-DMCALF_SRC_HASH="{stamp}"'] + (["-DMCALF_TESTING"] if testing else []) + ["-c", src, "-o", obj]
            _run(cmd)
            relinked = True
        objs.append(obj)
    if relinked or not os.path.exists(target) or any(os.path.getmtime(o) > os.path.getmtime(target) for o in objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-ldl"])
    return target


def build_tree(workdir: str, target: str, stamp: str = "patched-copy", defines=(), kernel_flags=(), report: bool = False) -> str:
    """Compile a COPY of the sources (tools/make_acc_build.py and friends patch one: the product source carries no
    instrumentation) into `target`; with `report` the kernels' resource usage is returned instead of the path."""
    hipcc = _hipcc()
    objs = []
    log = ""
    for src in SOURCES:
        obj = os.path.splitext(src)[0] + ".o"
        if src == KERNEL_SOURCE:
            cmd = [hipcc] + KERNEL_FLAGS + list(kernel_flags) + (["-Rpass-analysis=kernel-resource-usage"] if report else [])
        else:
            cmd = [hipcc] + HOST_FLAGS + [f'-DMCALF_SRC_HASH="{stamp}"']
        res = subprocess.run(cmd + [f"-D{d}" for d in defines] + ["-c", src, "-o", obj], cwd=workdir, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, res.stderr[-4000:]))
        if src == KERNEL_SOURCE:
            log = res.stderr
        objs.append(obj)
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-ldl"], cwd=workdir,
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    return log if report else target


def copy_sources(workdir: str) -> None:
    """A copy of everything the library is built from, laid out so that the copies' includes resolve inside `workdir`."""
    os.makedirs(os.path.join(workdir, "include"), exist_ok=True)
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip", ".cpp")):
            shutil.copy(os.path.join(CSRC, f), workdir)
    shutil.copy(os.path.join(CSRC, "..", "..", "include", "mcalf_hip.h"), os.path.join(workdir, "include"))
    hp = os.path.join(workdir, "host_ctx.h")
    txt = open(hp).read().replace('#include "../../include/mcalf_hip.h"', '#include "include/mcalf_hip.h"')
    open(hp, "w").write(txt)


def resource_table(log: str, only: str = "fused"):
    """(kernel, VGPRs, scratch bytes per lane, occupancy, SGPR spills, VGPR spills) from a -Rpass-analysis log."""
    import re
    rows = []
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)"
                         r".*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", log, re.S):
        if only in m.group(1):
            rows.append((m.group(1),) + tuple(int(m.group(k)) for k in range(2, 7)))
    return rows


if __name__ == "__main__":
    import sys
    if "--hash" in sys.argv:                  # (the Makefile stamps the library with it)
        print(source_hash())
    else:
        print(build(force=True, verbose=True))
