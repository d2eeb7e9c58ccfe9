#!/bin/bash
# Round 3, GPU box: tests + single-call latency (inline set-up on/off) + drop-in ranks + order on/off at 32768 rows + bench
out=${1:-gpurun_out/r03h}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "rc $?" >> "$out/pytest.log"; tail -4 "$out/pytest.log"
for v in 0 512; do
  echo "== MCALF_INLINE_MAX=$v" >> "$out/latency.txt"
  MCALF_INLINE_MAX=$v timeout -k 10 200 python tools/single_call_latency.py A B E >> "$out/latency.txt" 2>> "$out/latency.err"
done
cat "$out/latency.txt"
timeout -k 10 400 python tools/dropin_ranks.py --config B --ranks 1,2,4,6 --calls 2000 --out "$out/dropin_B.json" 2> "$out/dropin.err"
for r in 1 2; do for o in 1 0; do
  MCALF_ORDER=$o timeout -k 10 200 python3 bench.py --config D --steps 20 --warmup 3 --cpu-seconds 0 --no-host-api --no-model-leg 2>>"$out/err.txt" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('D32768 order$o', d['kernel_ms'], d['ms_per_step'])" >> "$out/bench_lines.txt"
done; done
cat "$out/bench_lines.txt"
timeout -k 10 300 python bench.py > "$out/bench.json" 2> "$out/bench.err"; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['kernel_ms'], d['strong_scaling_reference']['ms_per_step'], d['model_output']['kernel_ms'], d['model_output']['ms_per_step'])"
