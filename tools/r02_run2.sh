#!/bin/bash
# round 2: persistent kernel on/off A/B (same box, interleaved), then GPU tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02c; mkdir -p $o
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || { echo SMOKE FAILED; tail -20 $o/smoke.log; exit 1; }
tail -1 $o/smoke.log
for r in 1 2 3; do for p in 0 1; do for c in C B; do
  MCALF_PERSIST=$p timeout -k 10 300 python bench.py --config $c --cpu-seconds 0 --no-strong-ref --no-host-api --steps 100 --warmup 10 > $o/b_${c}_p${p}_r$r.json 2> $o/b_${c}_p${p}_r$r.err || echo "bench failed $c $p"
  python - $o/b_${c}_p${p}_r$r.json $c $p <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], 'persist', sys.argv[3], 'ms/step %.4f kernel_ms %.4f'%(d['ms_per_step'], d['kernel_ms']))
PY
done; done; done
MCALF_PERSIST=1 timeout -k 10 300 python bench.py --config E --cpu-seconds 0 --no-host-api --steps 20 --warmup 3 > $o/b_E_p1.json 2> $o/b_E_p1.err; MCALF_PERSIST=0 timeout -k 10 300 python bench.py --config E --cpu-seconds 0 --no-host-api --steps 20 --warmup 3 > $o/b_E_p0.json 2> $o/b_E_p0.err
python - $o <<'PY'
import json,sys
for p in (0,1):
    d=json.load(open(sys.argv[1]+'/b_E_p%d.json'%p)); print('E persist',p,'ms/step %.4f kernel_ms %.4f'%(d['ms_per_step'], d['kernel_ms']))
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.log
