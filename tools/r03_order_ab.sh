#!/bin/bash
# Round 3, GPU box: ordered hand-out on/off (MCALF_ORDER), interleaved, + the GPU test-suite.
out=${1:-gpurun_out/r03d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for r in 1 2 3; do
  for v in "order1 1 C" "order0 0 C" "order1 1 E" "order0 0 E"; do
    set -- $v
    MCALF_ORDER=$2 timeout -k 10 200 python3 bench.py --config $3 --steps 100 --warmup 10 --cpu-seconds 0 --no-host-api --no-strong-ref 2>>"$out/err.txt" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$3 $1', d['kernel_ms'], d['ms_per_step'])" >> "$out/bench_lines.txt"
  done
done
sort "$out/bench_lines.txt"
d="$out/kt"; rm -rf "$d"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref > "$out/kt.json" 2> "$out/kt.err"
f=$(find "$d" -name '*kernel_stats.csv' | head -1); grep mcalf "$f" | cut -c1-160
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; tail -5 "$out/pytest.log"
