#!/bin/bash
# Run bench.py against each ablation/variant library in build/abl (GPU box).  Diagnostic only.
for lib in build/abl/lib_*.so; do
  MCALF_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms=%.4f'%d['kernel_ms'], 'ms_per_step=%.4f'%d['ms_per_step'])"
done
