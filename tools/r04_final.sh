#!/bin/bash
# Round 4, GPU box: the final measurement set -- profile sets of configs C / E / B on ONE lease each call (bench line,
# kernel trace, PMC passes, issue rates, bench line WITH the counters), one-rank collective plumbing of the three gathers.
#   tools/r04_final.sh C|E|B|coll
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_final; mkdir -p $out build
case "${1:-C}" in
  C) bash tools/profiles.sh C gpurun_out/prof_C 50 r04 > $out/prof_C.log 2>&1; tail -8 $out/prof_C.log ;;
  E) bash tools/profiles.sh E gpurun_out/prof_E 20 r04 > $out/prof_E.log 2>&1; tail -8 $out/prof_E.log ;;
  B) bash tools/profiles.sh B gpurun_out/prof_B 100 r04 > $out/prof_B.log 2>&1; tail -8 $out/prof_B.log ;;
  coll)
    common="--config C --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg"
    timeout -k 10 200 python3 bench.py $common > $out/nodist.json 2>> $out/err.txt
    MCALF_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 $common > $out/one_rank_rccl_three_gathers.json 2>> $out/err.txt
    python3 - <<'PY'
import json
a = json.loads(open("gpurun_out/r04_final/nodist.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/r04_final/one_rank_rccl_three_gathers.json").read().strip().splitlines()[-1])
print("no process group: %.4f ms per step" % a["ms_per_step"])
for k, v in b["gathers"].items():
    print(k, v if isinstance(v, str) else "%.4f ms per step (x %.3f), check %s" % (v["ms_per_step"], v["ms_per_step"] / a["ms_per_step"], v["gather_check"]))
PY
    ;;
esac
