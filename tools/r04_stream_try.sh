#!/bin/bash
# GPU box, round 4: first runs of the streaming single launch (bounded: every step under its own timeout)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_stream; mkdir -p $O
for cfg in ${1:-C}; do
  timeout -k 10 120 python tools/host_floor.py --config $cfg --steps 50 --passes 5 --only device_async,device_sync,host_pageable,host_pinned > $O/floor_$cfg.txt 2>&1 || { tail -20 $O/floor_$cfg.txt; exit 1; }
  tail -1 $O/floor_$cfg.txt
done
