#!/bin/bash
# Round 3, GPU box: ONE GPU evaluating the per-GPU shard of configs D and E at N = 1, 2, 4, 8 ranks (job batch / N rows):
# what each rank of a strong-scaling job would take per step before any collective.  NOT a multi-GPU measurement.
out=${1:-gpurun_out/r03v}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for r in 1 2; do
  for v in "D 32768 10" "D 16384 20" "D 8192 40" "D 4096 80" "E 16384 10" "E 8192 20" "E 4096 40" "E 2048 80"; do
    set -- $v
    timeout -k 10 300 python3 bench.py --config $1 --batch $2 --steps $3 --warmup 3 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg 2>>"$out/err.txt" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, d['ms_per_step'], d['kernel_ms'], d['launch']['persistent'], d['launch']['ordered_handout'])" >> "$out/lines.txt"
  done
done
sort -k1,1 -k2,2n "$out/lines.txt"
