#!/bin/bash
# GPU box, round 4: the broker's serving loop inside the library (one and several contexts) against the Python loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_broker2; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -30 $O/test.txt; exit 1; }
tail -2 $O/test.txt
timeout -k 10 700 python tools/dropin_ranks.py --config B --ranks ${1:-b8,b8l1,b8l2,b15,b15l1,b15l2,b15l4,b32,b32l2,b32l4} --calls 1500 --out $O/r04_dropin_broker_native.json > $O/dropin.txt 2>&1 || { tail -20 $O/dropin.txt; exit 1; }
grep -v amdgpu.ids $O/dropin.txt
