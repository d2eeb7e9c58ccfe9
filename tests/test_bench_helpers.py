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
