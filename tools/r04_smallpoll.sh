#!/bin/bash
# GPU box, round 4: small host calls with completion read off the results (MCALF_STREAM_POLL=1, default) against the stream wait (=0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_smallpoll; mkdir -p $O
for rep in 1 2; do
for poll in 0 1; do
  echo "== MCALF_STREAM_POLL=$poll (pass $rep)" >> $O/latency.txt
  MCALF_STREAM_POLL=$poll timeout -k 10 200 python3 tools/single_call_latency.py A B E 2>&1 | grep -v amdgpu.ids >> $O/latency.txt || exit 1
done
done
cat $O/latency.txt
timeout -k 10 300 python3 tools/dropin_ranks.py --config B --ranks 1,4,b15l2 --calls 1500 --out $O/dropin.json 2>&1 | grep "^R ="
