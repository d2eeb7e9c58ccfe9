#!/bin/bash
# Round 3, GPU box: the final measurement set on ONE lease -- profile sets of configs C / E / B (bench line, kernel
# trace, PMC passes, issue rates), the one-rank collective plumbing, drop-in ranks, launch timeline, single-call latency.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_final; mkdir -p $out
bash tools/r03_profiles.sh C gpurun_out/r03_prof_C 50 > $out/prof_C.log 2>&1
bash tools/r03_profiles.sh E gpurun_out/r03_prof_E 20 > $out/prof_E.log 2>&1
bash tools/r03_profiles.sh B gpurun_out/r03_prof_B 100 > $out/prof_B.log 2>&1
common="--config C --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg"
for r in 1 2 3; do
  timeout -k 10 200 python3 bench.py $common > $out/nodist_$r.json 2>> $out/err.txt
  MCALF_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2957$r bench.py --gpus 1 $common > $out/dist_torch_$r.json 2>> $out/err.txt
  MCALF_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2958$r bench.py --gpus 1 --gather inlib $common > $out/dist_inlib_$r.json 2>> $out/err.txt
done
timeout -k 10 400 python tools/dropin_ranks.py --config B --ranks 1,2,4,6 --calls 2000 --out $out/dropin_B.json > $out/dropin_B.txt 2>> $out/err.txt
timeout -k 10 200 python tools/single_call_latency.py A B E > $out/latency.txt 2>> $out/err.txt
python3 - <<'PY'
import json, glob
for k in ("nodist", "dist_torch", "dist_inlib"):
    v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("gpurun_out/r03_final/%s_*.json" % k))]
    print(k, ["%.4f" % x for x in v], "min %.4f" % min(v))
PY
cat $out/dropin_B.txt $out/latency.txt
