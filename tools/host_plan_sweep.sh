#!/bin/bash
# GPU box: host-pointer entry (H2D + D2H inside the step) under different row-block plans; diagnostic.
#   tools/host_plan_sweep.sh [config] ["plan plan ..."]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
plans=${2:-"1,1,2,4 1,1,2,4,8 1,2,4,9 1,1,1,1,4,8 1,3,12 1,7 1,15 1,1,6"}
for r in 1 2; do for plan in $plans; do
  MCALF_HOST_PLAN=$plan timeout -k 10 200 python bench.py --config ${1:-C} --cpu-seconds 0 --no-strong-ref --no-model-leg --no-other-configs --no-multi-device --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plan $plan', 'host %.4f pinned %.4f device %.4f'%(d['ms_per_step_host_api'], d['host_api']['ms_per_step_pinned'], d['ms_per_step']))"
done; done
