"""CPU: the workload selection and bookkeeping helpers of bench.py (no GPU, nothing is launched)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_default_workloads_follow_baseline_json():
    cfg = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "batch=4096" in cfg[2] and "batch=32768" in cfg[3]                  # C: largest 1-GPU config, D: the 8-GPU one
    assert bench.pick_workload(1) == ("C", "strong")
    for n in (2, 4, 8):
        assert bench.pick_workload(n) == ("D", "strong")
    assert bench.pick_workload(8, "E", "weak") == ("E", "weak")
    assert bench.WORKLOAD_LABEL["C"].startswith("BASELINE config C") and bench.WORKLOAD_LABEL["D"].startswith("BASELINE config D")


def test_rows_per_rank():
    assert [bench.rows_per_rank(32768, n, "strong") for n in (1, 2, 4, 8)] == [32768, 16384, 8192, 4096]
    assert bench.rows_per_rank(4096, 8, "weak") == 4096
    with pytest.raises(SystemExit):
        bench.rows_per_rank(1000, 3, "strong")


def test_usable_cpus_is_bounded_by_the_affinity_mask():
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_profile_json_lookup_tolerates_missing_files():
    assert bench.load_profile_json("does_not_exist.json", "C") is None
    got = bench.load_profile_json("pmc.json", "C")
    assert got is None or 0.0 < got["valu_busy"] < 1.0


def test_pmc_figures_are_only_quoted_for_the_kernel_they_were_measured_on():
    """bench.py drops `traffic` / `valu_busy` / the issue model when the loaded library's kernel-source hash differs
    from the stamp of the profiles/ record (a kernel edit must not ship stale utilisation numbers)."""
    entry = {"source_hash": "abc", "valu_busy": 0.6}
    assert bench.stamped(entry, "abc") == (entry, None)
    got, why = bench.stamped(entry, "def")
    assert got is None and "abc" in why and "def" in why
    assert bench.stamped({"valu_busy": 0.6}, "abc")[0] is None            # an unstamped (round-2) record
    assert bench.stamped(entry, "unstamped")[0] is None                   # a library built outside mc-alf_amd/build.py
    assert bench.stamped(None, "abc")[0] is None


def test_library_carries_the_hash_of_the_device_sources():
    import importlib
    import mcalf_amd
    bld = importlib.import_module("mc-alf_amd.build")
    lib = mcalf_amd._lib.load()
    assert bench.library_source_hash(lib) == bld.source_hash()            # built by __graft_entry__.build()
    # the hash covers what the DEVICE code is made of -- the kernels, the argument block they share with the host, the
    # Voigt function and its tables -- and none of the host side of the C ABI
    assert bld.HASHED == ["kernels.hip", "kernel_args.h", "voigt_device.h", "voigt_tables.h"]
    assert not set(bld.HASHED) & set(bld.HOST_SOURCES)
    kernels = open(os.path.join(bld.CSRC, "kernels.hip")).read()
    assert 'extern "C"' not in kernels and "__global__" in kernels
    for f in bld.HOST_SOURCES:                                            # (no device code outside kernels.hip)
        assert "__global__" not in open(os.path.join(bld.CSRC, f)).read(), f


def test_issue_model_arithmetic():
    iss = {"insts": {"valu_f64": 4.0e6, "salu": 2.0e6}, "simds": 1000, "waves_per_simd": 4, "clock_mhz": 2000.0,
           "cycles": {"wave": {"valu_f64": 5.0, "salu": 4.0}, "pipe": {"valu_f64": 4.0, "salu": 4.0}},
           "lds_pipe_cycles_per_cu": 10000.0}
    out = bench.issue_model(iss, kern_ms=0.02)
    # pipes: valu 4000 wave-instructions per SIMD x 4 cycles = 16000 cycles = 0.008 ms; salu 2000 x 4 = 0.004 ms
    assert abs(out["class_ms"]["valu_f64"] - 0.008) < 1e-12 and abs(out["class_ms"]["salu"] - 0.004) < 1e-12
    assert abs(out["pipe_ms"]["valu"] - 0.008) < 1e-12 and abs(out["pipe_ms"]["scalar"] - 0.004) < 1e-12
    assert abs(out["pipe_ms"]["lds"] - 0.005) < 1e-12 and out["busiest_pipe"] == "valu"
    # serial issue: (4000 x 5 + 2000 x 4) / 4 waves = 7000 cycles = 0.0035 ms
    assert abs(out["serial_issue_ms"] - 0.0035) < 1e-12
    assert abs(out["peak"] - 0.008) < 1e-12 and abs(out["frac"] - 0.4) < 1e-12
