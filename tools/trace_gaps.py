#!/usr/bin/env python3
"""Diagnostic: gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = lambda r: r["Kernel_Name"][:60]
# take the steady-state: last 600 dispatches
rows = rows[-600:]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[(names(a)[:28], names(b)[:28])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    dur[names(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print("dur  %-62s n=%4d mean %8.0f ns" % (k, len(v), sum(v) / len(v)))
for k, v in gap.items():
    print("gap  %-28s -> %-28s n=%4d mean %8.0f ns  (negative = overlap)" % (k[0], k[1], len(v), sum(v) / len(v)))
for r in rows[-12:]:
    print(names(r)[:40], r.get("Queue_Id"), r.get("Stream_Id", ""), int(r["Start_Timestamp"]) - int(rows[-12]["Start_Timestamp"]), int(r["End_Timestamp"]) - int(rows[-12]["Start_Timestamp"]))
