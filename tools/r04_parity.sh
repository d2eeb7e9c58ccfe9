#!/bin/bash
# Round 4, GPU box: every row of configs B / C / D / E against the C oracle through both entries (the host entry's
# streaming launch included), and a wide fuzz of the final build.     tools/r04_parity.sh [fuzz seeds, default 1200]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_parity; mkdir -p $O
timeout -k 10 700 python3 tools/full_parity.py > $O/r04_full_parity.json 2> $O/full_parity.log || { tail -20 $O/full_parity.log; exit 1; }
grep -v amdgpu.ids $O/full_parity.log | cut -c1-400
timeout -k 10 420 python3 tools/fuzz_campaign.py 9000 ${1:-1200} > $O/r04_fuzz.txt 2>&1
rc=$?
tail -3 $O/r04_fuzz.txt
exit $rc
