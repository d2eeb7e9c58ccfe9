#!/bin/bash
# GPU box, round 4: one-theta-per-call latency and drop-in mode at up to 36 contexts on one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_dropin; mkdir -p $O
timeout -k 10 200 python tools/single_call_latency.py A B E > $O/single_call_latency.txt 2>&1 || { tail $O/single_call_latency.txt; exit 1; }
grep -v amdgpu.ids $O/single_call_latency.txt
timeout -k 10 500 python tools/dropin_ranks.py --config B --ranks ${1:-1,2,4,6,6x2,6x4,6x6} --calls 1000 --out $O/r04_dropin.json > $O/dropin.txt 2>&1 || { tail -20 $O/dropin.txt; exit 1; }
grep -v amdgpu.ids $O/dropin.txt
