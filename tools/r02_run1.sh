#!/bin/bash
# round 2, first GPU pass: GPU tests, then bench C with 1 / 2 / 3 / 4 row blocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02a; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log
tail -5 $o/pytest.log
for c in 1 2 3 4; do
  timeout -k 10 300 python bench.py --chunks $c --cpu-seconds 0 --no-strong-ref > $o/bench_C_chunks$c.json 2> $o/bench_C_chunks$c.err || echo "bench chunks $c failed"
  python - $o/bench_C_chunks$c.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], 'ms/step %.4f kernel_ms %.4f host %.4f pinned %.4f value %.4g'%(d['ms_per_step'], d['kernel_ms'], d['ms_per_step_host_api'], d['host_api']['ms_per_step_pinned'], d['value']))
PY
done
timeout -k 10 300 python bench.py --config B --chunks 1 --cpu-seconds 0 > $o/bench_B_chunks1.json 2> $o/bench_B_chunks1.err
timeout -k 10 300 python bench.py --config B --chunks 2 --cpu-seconds 0 > $o/bench_B_chunks2.json 2> $o/bench_B_chunks2.err
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err
tail -c 1500 $o/bench_default.json
