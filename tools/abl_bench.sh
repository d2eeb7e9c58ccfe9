#!/bin/bash
# Run bench.py against each ablation/variant library in build/abl (GPU box).  Diagnostic only.
# ABL_ROUNDS interleaved rounds (default 3); the minimum over rounds is what gets compared
# (run-to-run spread of a single bench is ~1.5 %).
rounds=${ABL_ROUNDS:-3}
tmp=$(mktemp)
for r in $(seq $rounds); do
  for lib in build/abl/lib_*.so; do
    MCALF_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref --no-other-configs --no-multi-device "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['kernel_ms'], d['ms_per_step'])" >> $tmp
  done
done
python - $tmp <<'PY'
import sys, collections
k = collections.defaultdict(list); m = collections.defaultdict(list)
for line in open(sys.argv[1]):
    lib, a, b = line.split(); k[lib].append(float(a)); m[lib].append(float(b))
for lib in sorted(k):
    print(lib, 'kernel_ms min %.4f med %.4f' % (min(k[lib]), sorted(k[lib])[len(k[lib]) // 2]), 'ms_per_step min %.4f med %.4f' % (min(m[lib]), sorted(m[lib])[len(m[lib]) // 2]))
PY
rm -f $tmp
