#!/bin/bash
# Round 3, GPU box: lines folded per barrier (MCALF_LINES_PER_SYNC = 4 / 5 / 6), interleaved, configs C, E, B.
out=${1:-gpurun_out/r03s}; lib=${2:-build/abl/lib_lps6.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for r in 1 2 3; do
  for cfg in C E B; do
    for l in 4 5 6; do
      st=100; [ $cfg = E ] && st=20
      MCALF_LINES_PER_SYNC=$l MCALF_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config $cfg --steps $st --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg 2>>"$out/err.txt" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg lps$l', d['kernel_ms'], d['ms_per_step'], d['launch']['lines_per_sync'])" >> "$out/bench_lines.txt"
    done
  done
done
sort "$out/bench_lines.txt"
