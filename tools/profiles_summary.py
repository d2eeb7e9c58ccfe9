#!/usr/bin/env python3
"""Summarise one tools/profiles.sh output directory: per-counter means over the FULL-batch dispatches of the
fused kernel (smaller launches are dropped by grid size), VALU busy, executed FP64 flops, HBM traffic per launch
(FETCH_SIZE doubled: gfx950 reports half the bytes of coalesced reads, MI355X_MICROARCH.md), the wave-instruction
counts by class for the issue model -- all stamped with the hash of the kernel sources the library was built from."""
import collections
import csv
import glob
import json
import os
import sys

cfg, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mcalf_fused" not in row["Kernel_Name"]:
            continue
        rows[row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
mean = {}
for k, lst in rows.items():
    gmax = max(g for g, _ in lst)
    full = [v for g, v in lst if g == gmax]
    mean[k] = (sum(full) / len(full), len(full))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k in sorted(mean):
        line = "%-28s per-dispatch mean %.6g  (full-batch dispatches %d)" % (k, mean[k][0], mean[k][1])
        print(line)
        fh.write(line + "\n")
bench = json.load(open(out + "/bench.json"))
res = {"config": cfg, "source_hash": bench["roofline"]["kernel_source_hash"]}
g = lambda k: mean[k][0] if k in mean else None
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res["valu_busy"] = g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / (g("GRBM_GUI_ACTIVE") / 8)
    res["wave_wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else None
    res["insts_valu_per_launch"] = g("SQ_INSTS_VALU")
    res["insts_salu_per_launch"] = g("SQ_INSTS_SALU")
f64 = None
if g("SQ_INSTS_VALU_FMA_F64") is not None:
    f64 = g("SQ_INSTS_VALU_FMA_F64") + (g("SQ_INSTS_VALU_MUL_F64") or 0) + (g("SQ_INSTS_VALU_ADD_F64") or 0) + (g("SQ_INSTS_VALU_TRANS_F64") or 0)
    res["executed_flops_per_launch"] = 64.0 * (f64 + g("SQ_INSTS_VALU_FMA_F64"))
    res["executed_flops_note"] = ("64 lanes x (2 FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64) wave instructions: an UPPER BOUND -- exec masks "
                                  "are ignored (a wave instruction counts 64 lanes whether 1 or 64 of them are enabled)")
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
        # Share of lanes enabled over ALL vector instructions (not only the FP64 ones).  SQ_THREAD_CYCLES_VALU adds, per
        # vector instruction, (its enabled lanes) x (a per-instruction unit): the unit is MEASURED on the same lease with
        # kernels whose lane share is known (tools/micro/lane_unit.hip: 64 of 64 lanes and 16 of 64 in an FP64 FMA loop) --
        # round 4 divided by SQ_ACTIVE_INST_VALU x 4 x 64 instead, a factor of four too much (0.22 where 0.89 was meant).
        unit, how = 1.0, "unit 1 per instruction assumed (no lane_unit.json in this run)"
        try:
            cal = json.load(open(out + "/lane_unit.json"))
            full = cal["lanes64_f64"]["thread_cycles_over_insts_x64"]
            part = cal["lanes16_f64"]["thread_cycles_over_insts_x64"]
            unit = full
            how = ("unit %.3f per instruction: tools/micro/lane_unit.hip on the same lease gives THREAD_CYCLES / (INSTS x 64) = %.3f with "
                   "64 of 64 lanes enabled and %.3f with 16 of 64 (FP64 FMA loops)" % (full, full, part))
        except (OSError, ValueError, KeyError):
            pass
        lane = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_INSTS_VALU") * 64.0 * unit)
        res["valu_active_lane_fraction"] = lane
        res["valu_active_lane_fraction_note"] = "SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64 lanes x unit); " + how
        res["executed_flops_lane_weighted_estimate"] = res["executed_flops_per_launch"] * min(lane, 1.0)
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    res["fetch_size_kib_raw"], res["write_size_kib"] = g("FETCH_SIZE"), g("WRITE_SIZE")
    res["traffic_bytes_per_launch"] = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
# wave-instructions by issue class, for bench.py's roofline_issue
try:
    rate = json.load(open(out + "/issue_rate.json"))
except (OSError, ValueError):
    rate = None
if rate and f64 is not None and g("SQ_INSTS_VALU") and g("SQ_INSTS_LDS") is not None:
    scalar = (g("SQ_INSTS_SALU") or 0) + (g("SQ_INSTS_SMEM") or 0)
    insts = {"valu_f64": f64, "valu_other": g("SQ_INSTS_VALU") - f64, "salu": scalar, "lds": g("SQ_INSTS_LDS"),
             "branch": g("SQ_INSTS_BRANCH") or 0}
    clock = rate["pipe"].pop("clock_ghz")
    rate["wave"].pop("clock_ghz", None)
    res["issue"] = {"insts": insts, "simds": 1024, "waves_per_simd": 4, "clock_mhz": clock * 1e3, "cycles": rate,
                    "lds_pipe_cycles_per_cu": (g("SQ_LDS_IDX_ACTIVE") / 256.0) if g("SQ_LDS_IDX_ACTIVE") else None,
                    "note": "SQ_INSTS_* are wave-instructions summed over the chip; SALU includes scalar memory; the SALU count "
                            "already contains the branches (they are listed, not added twice: branch cycles price the taken-"
                            "branch overhead per branch instruction)",
                    "source": "tools/micro/issue_rate.hip on the same lease"}
    # (the branch instructions are part of SQ_INSTS_SALU: remove them from `salu` so that every instruction is priced once)
    insts["salu"] = max(0.0, scalar - insts["branch"])
res["source"] = "tools/profiles.sh %s (rocprofv3 --pmc, one counter set per run, full-batch mcalf_fused_kernel dispatches)" % cfg
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res))
try:
    for row in csv.DictReader(open(out + "/kernel_stats.csv")):
        if "mcalf" in row["Name"]:
            print(row["Name"][:60], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"])
    b = json.load(open(out + "/ktrace_bench.json"))
    print("bench under the tracer: kernel_ms %.4f ms_per_step %.4f" % (b["kernel_ms"], b["ms_per_step"]))
    print("bench: kernel_ms %.4f ms_per_step %.4f host_api %.4f" % (bench["kernel_ms"], bench["ms_per_step"], bench.get("ms_per_step_host_api", float("nan"))))
except Exception as exc:  # noqa: BLE001
    print("summary:", exc)
