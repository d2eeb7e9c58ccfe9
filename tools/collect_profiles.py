#!/usr/bin/env python3
"""Copy the summaries of tools/profiles.sh runs into profiles/ under the round's names and merge their PMC figures --
stamped with the kernel-source hash -- into profiles/pmc.json and profiles/traffic.json (keyed by config).  bench.py
quotes them only while the loaded library carries the same hash.

    python tools/collect_profiles.py [--round r05] [--src <dir of ONE config's run>] B C E"""
import argparse
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
pmc_path, tr_path = os.path.join(prof, "pmc.json"), os.path.join(prof, "traffic.json")
pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
traffic = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
ap = argparse.ArgumentParser()
ap.add_argument("--round", default="r05")
ap.add_argument("--src", default=None, help="directory of the run (with ONE config); default gpurun_out/prof_<cfg>")
ap.add_argument("configs", nargs="*", default=["B", "C", "E"])
args = ap.parse_args()
tag, done = args.round, []
for cfg in args.configs:
    src = args.src if args.src else os.path.join(root, "gpurun_out", "prof_" + cfg)
    if not os.path.isdir(src) or not os.path.exists(os.path.join(src, "pmc.json")):
        continue
    done.append(cfg)
    for name, dst in (("bench.json", f"{tag}_bench_{cfg}.json"), ("kernel_stats.csv", f"{tag}_bench_{cfg}_kernel_stats.csv"),
                      ("pmc_summary.txt", f"{tag}_bench_{cfg}_pmc_summary.txt"), ("ktrace_bench.json", f"{tag}_bench_{cfg}_under_tracer.json"),
                      ("issue_rate.json", f"{tag}_issue_rate_{cfg}.json"), ("lane_unit.json", f"{tag}_lane_unit_{cfg}.json")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(prof, dst))
    p = json.load(open(os.path.join(src, "pmc.json")))
    pmc[cfg] = {k: p[k] for k in ("source_hash", "valu_busy", "wave_wait_any_frac", "insts_valu_per_launch", "insts_salu_per_launch",
                                  "executed_flops_per_launch", "executed_flops_note", "valu_active_lane_fraction",
                                  "valu_active_lane_fraction_note", "executed_flops_lane_weighted_estimate", "issue") if k in p}
    pmc[cfg]["source"] = f"profiles/{tag}_bench_{cfg}_pmc_summary.txt ({p['source']})"
    if "traffic_bytes_per_launch" in p:
        traffic[cfg] = {"source_hash": p["source_hash"], "fetch_size_kib_raw": p["fetch_size_kib_raw"], "write_size_kib": p["write_size_kib"],
                        "traffic_bytes_per_launch": p["traffic_bytes_per_launch"],
                        "source": f"profiles/{tag}_bench_{cfg}_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
                                  "passes, full-batch mcalf_fused_kernel dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md's "
                                  "gfx950 correction; KiB units",
                        "note": "above the algorithmic bytes because the per-sample records / taps / headers the set-up kernel "
                                "leaves in HBM (64 B per component-line, read once per tile) are intermediate traffic the "
                                "SURVEY 8(d) formula does not count"}
json.dump(pmc, open(pmc_path, "w"), indent=1)
json.dump(traffic, open(tr_path, "w"), indent=1)
print("profiles updated for", done)
