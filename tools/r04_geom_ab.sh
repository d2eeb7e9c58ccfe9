#!/bin/bash
# GPU box, round 4: A/B of the workgroup-geometry experiment builds (tools/make_geom_build.py) against the product library,
# device entry, config C; interleaved rounds; parity of every build against the numpy oracle on a sample.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_geom; mkdir -p $O
common="--config C --steps 100 --warmup 10 --cpu-seconds 2 --cpu-threads -1 --no-host-api --no-strong-ref --no-model-leg"
for r in 1 2; do for v in product geom_640x7 geom_640x7_cflds geom_768x6_cflds; do
  lib=$PWD/build/abl/$v.so; [ $v = product ] && lib=$PWD/mc-alf_amd/csrc/libmcalf_hip.so
  MCALF_STREAM=0 MCALF_HIP_LIB=$lib timeout -k 10 200 python3 bench.py $common > $O/${v}_$r.json 2> $O/${v}_$r.err || { echo "$v failed"; tail -5 $O/${v}_$r.err; continue; }
  python3 -c "
import json,sys
d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1])
print('$v round $r: ms_per_step %.4f kernel_ms %.4f parity %.2e over %d rows' % (d['ms_per_step'], d['kernel_ms'], d['parity']['max_abs_dlogL_vs_oracle'], d['parity']['rows']))"
done; done
