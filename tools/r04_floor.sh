#!/bin/bash
# GPU box, round 4: floors of the synchronous host-pointer step + a kernel / memory-copy timeline of the current pipeline.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_floor; mkdir -p $O
timeout -k 10 300 python tools/host_floor.py --config C > $O/floor_C.txt 2>&1 || exit 1
cat $O/floor_C.txt | tail -12
for v in host_pageable host_pinned device_sync; do
  timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_$v -- python3 tools/host_floor.py --config C --steps 20 --passes 3 --only $v > $O/trace_$v.log 2>&1 || exit 1
done
ls -R $O | head -40
