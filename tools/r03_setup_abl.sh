#!/bin/bash
# Round 3, GPU box: where the per-sample set-up kernel's time goes (ablation builds build/abl/setup_abl{0..3}.so:
# 0 = product, 1 = no taps, 2 = no records, 3 = neither) and the timeline of one persistent launch.
out=${1:-gpurun_out/r03b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
common="--steps 100 --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref"
for v in 0 1 2 3 0; do
  d="$out/kt_$v"; rm -rf "$d"
  MCALF_HIP_LIB=$PWD/build/abl/setup_abl$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 bench.py $common > "$out/bench_$v.json" 2> "$out/bench_$v.err" || echo "variant $v failed"
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "== variant $v" >> "$out/summary.txt"
  python3 - "$f" >> "$out/summary.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mcalf' in r['Name']:
        print("%-70s calls %5s avg %9.1f ns  min %9s max %9s" % (r['Name'][:70], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
done
for order in asdrawn sorted; do
  MCALF_HIP_LIB=$PWD/build/abl/stamps.so timeout -k 10 120 python3 tools/timeline_report.py C 4096 $order > "$out/timeline_C_$order.txt" 2>&1
done
MCALF_HIP_LIB=$PWD/build/abl/stamps.so timeout -k 10 120 python3 tools/timeline_report.py C 8192 asdrawn > "$out/timeline_C_8192.txt" 2>&1
cat "$out/summary.txt"
