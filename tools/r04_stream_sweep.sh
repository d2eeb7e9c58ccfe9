#!/bin/bash
# GPU box, round 4: streaming launch under different numbers of dedicated set-up workgroups / completion modes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_stream; mkdir -p $O
cfg=${1:-C}; shift
for spec in "$@"; do
  env $spec timeout -k 10 120 python tools/host_floor.py --config $cfg --steps 50 --passes 5 --only ${ONLY:-host_pageable,host_pinned} > $O/sweep.txt 2>&1 || { tail -20 $O/sweep.txt; exit 1; }
  echo "$spec :: $(tail -1 $O/sweep.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%s %.4f (eq %s, path %d)'%(k,v['ms_per_step_median'],v.get('bit_equal_to_device'),v['path']) for k,v in d.items() if 'ms_per_step_median' in v))")"
done
