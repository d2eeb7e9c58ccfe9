#!/bin/bash
# Round 3, GPU box: A/B of library variants build/abl/lib_<name>.so given as arguments: rocprofv3 kernel averages and
# interleaved plain bench lines (config C).   tools/r03_lib_ab.sh <outdir> <name> [<name> ...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
common="--steps 100 --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg"
for v in "$@"; do
  d="$out/kt_$v"; rm -rf "$d"
  MCALF_HIP_LIB=$PWD/build/abl/lib_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 bench.py $common ${BENCH_ARGS} > "$out/kt_$v.json" 2> "$out/kt_$v.err" || echo "$v failed"
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "== $v" >> "$out/summary.txt"; grep mcalf "$f" | cut -d, -f1-4 >> "$out/summary.txt"
done
for r in 1 2 3; do
  for v in "$@"; do
    MCALF_HIP_LIB=$PWD/build/abl/lib_$v.so timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg ${BENCH_ARGS} 2>>"$out/err.txt" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['kernel_ms'], d['ms_per_step'], d['parity'] if 'parity' in d else '')" >> "$out/bench_lines.txt"
  done
done
cat "$out/summary.txt"; sort "$out/bench_lines.txt"
