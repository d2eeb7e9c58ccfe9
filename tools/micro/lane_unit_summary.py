#!/usr/bin/env python3
"""Reads the counter CSV of `rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU
-- build/lane_unit` and prints, per kernel with a known share of active lanes, thread-cycles / (instructions x 64):
the unit tools/profiles_summary.py needs.   python tools/micro/lane_unit_summary.py <dir> [out.json]"""
import collections
import csv
import glob
import json
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(rows.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_INSTS_VALU"):
        known = 0.25 if "lanes16" in k else 1.0
        ratio = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_INSTS_VALU"] * 64.0)
        out[k] = {"known_active_lane_share_of_the_fma_loop": known, "thread_cycles_over_insts_x64": ratio,
                  "thread_cycles_over_active_inst_x4x64": m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 4 * 64.0) if m.get("SQ_ACTIVE_INST_VALU") else None,
                  "counters": m}
        print("%-16s known %.2f  THREAD_CYCLES/(INSTS x 64) = %.4f   per-instruction unit = %.3f" % (k, known, ratio, ratio / known))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
