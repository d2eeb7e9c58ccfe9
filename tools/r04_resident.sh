#!/bin/bash
# GPU box, round 4: the resident evaluators -- a context's own (MCALF_RESIDENT_US) across solver ranks, and the broker's
# (one workgroup per rank's mailbox) against the broker's launched forms
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_resident; mkdir -p $O
timeout -k 10 100 python3 tools/dbg_resident.py A B C 2>&1 | grep config > $O/one_context.txt || exit 1
cat $O/one_context.txt
MCALF_RESIDENT_US=500 timeout -k 10 300 python3 tools/dropin_ranks.py --config B --ranks 1,2,4,6 --calls 1500 --out $O/r04_dropin_resident.json 2>&1 | grep "^R =" > $O/ranks.txt || exit 1
cat $O/ranks.txt
timeout -k 10 600 python3 tools/dropin_ranks.py --config B --ranks b4r500,b8r500,b15r500,b15l2,b32r500,b32l2,b32 --calls 1500 --out $O/r04_dropin_broker_resident.json 2>&1 | grep "^R =" > $O/broker.txt || exit 1
cat $O/broker.txt
