#!/bin/bash
# Round 3, GPU box: set-up kernel A/B (old vs lean arithmetic, workgroup geometry) and the ordered hand-out experiment.
out=${1:-gpurun_out/r03c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
common="--steps 100 --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref"
stats() {   # label, lib, setup block
  d="$out/kt_$1"; rm -rf "$d"
  MCALF_SETUP_BLOCK=$3 MCALF_HIP_LIB=$PWD/build/abl/$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 bench.py $common > "$out/kt_$1.json" 2> "$out/kt_$1.err" || echo "$1 failed"
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "== $1" >> "$out/summary.txt"
  python3 - "$f" >> "$out/summary.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mcalf' in r['Name']:
        print("%-70s calls %5s avg %9.1f ns" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])))
PY
}
stats old lib_old.so 64
for b in 64 128 256 512; do stats lean$b lib_lean.so $b; done
stats old_again lib_old.so 64
# plain bench lines (no profiler), interleaved, 3 rounds
for r in 1 2 3; do
  for v in "old lib_old.so 64 none" "lean256 lib_lean.so 256 none" "lean512 lib_lean.so 512 none" "lean256_desc lib_lean.so 256 desc" "lean256_asc lib_lean.so 256 asc"; do
    set -- $v
    MCALF_BENCH_SORT=$4 MCALF_SETUP_BLOCK=$3 MCALF_HIP_LIB=$PWD/build/abl/$2 timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['kernel_ms'], d['ms_per_step'])" >> "$out/bench_lines.txt"
  done
done
cat "$out/summary.txt"; sort "$out/bench_lines.txt"
