#!/bin/bash
# GPU box, round 4: solver ranks behind ONE likelihood broker against one context per rank
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_broker; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -20 $O/test.txt; exit 1; }
tail -2 $O/test.txt
timeout -k 10 600 python tools/dropin_ranks.py --config B --ranks 4,6,b1,b4,b8,b15,b32 --calls 1500 --out $O/r04_dropin_broker.json > $O/dropin.txt 2>&1 || { tail -20 $O/dropin.txt; exit 1; }
grep -v amdgpu.ids $O/dropin.txt
