#!/bin/bash
# GPU box: the measurement / verification recipes of a round, ONE parameterised script (run through gpurun from the repo
# root; everything lands under gpurun_out/<tag>/).   tools/gpu_session.sh <recipe> [tag, default r06] [args]
#   tests              the whole `-m gpu` suite in one process
#   bench [cfg]        the driver's bench line (default config C) + the same under rocprofv3 --kernel-trace --stats
#   profiles <cfg>     tools/profiles.sh: bench line, kernel trace, PMC passes, issue rates, bench line WITH counters
#   lane_unit          unit of SQ_THREAD_CYCLES_VALU (tools/micro/lane_unit.hip under rocprofv3 --pmc)
#   coll               the three gathers on a one-rank RCCL process group against no process group
#   parity             tools/full_parity.py (every row of B / C / D / E, both entries) + tools/fuzz_campaign.py
#   latency            one theta per call: launched / polled / resident (tools/single_call_latency.py, call_overhead.py)
#   dropin             tools/dropin_ranks.py: ranks with a context each, ranks behind the broker
#   hostfloor          tools/host_floor.py: what a synchronous host-pointer step costs at best
#   ab <cfg> [steps]   tools/abl_bench.sh over build/abl/lib_*.so
#   hostgap [legs]     the host-pointer step against the device-resident one (bench.py --only-other-configs) under the
#                      default plan and its alternatives, interleaved twice: row-block pipeline only, streaming launch for
#                      every size, round 5's 1:1:2:4 row blocks
recipe=${1:-tests}; tag=${2:-r06}; shift 2 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p "$out" build
case "$recipe" in
  tests)
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q -rs > "$out/gputests.txt" 2>&1; echo "pytest rc=$?" >> "$out/gputests.txt"; tail -6 "$out/gputests.txt" ;;
  bench)
    cfg=${1:-C}
    timeout -k 10 500 python bench.py --config $cfg > "$out/bench_$cfg.json" 2> "$out/bench_$cfg.err"; tail -c 600 "$out/bench_$cfg.json"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/ktrace_$cfg" --output-format csv -- python3 bench.py --config $cfg --cpu-seconds 0 --no-strong-ref --no-model-leg --no-other-configs --no-multi-device > "$out/bench_${cfg}_under_tracer.json" 2> "$out/ktrace_$cfg.err"
    find "$out/ktrace_$cfg" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$out/bench_${cfg}_kernel_stats.csv"; head -5 "$out/bench_${cfg}_kernel_stats.csv" ;;
  profiles)
    cfg=${1:-C}; steps=50; [ "$cfg" = E ] && steps=20; [ "$cfg" = B ] && steps=100
    bash tools/profiles.sh $cfg gpurun_out/prof_$cfg $steps $tag > "$out/prof_$cfg.log" 2>&1; tail -8 "$out/prof_$cfg.log" ;;
  lane_unit)
    [ -x build/lane_unit ] || hipcc -O3 --offload-arch=gfx950 -o build/lane_unit tools/micro/lane_unit.hip
    timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -d "$out/lane_unit" --output-format csv -- build/lane_unit > "$out/lane_unit.log" 2>&1
    python3 tools/micro/lane_unit_summary.py "$out/lane_unit" "$out/lane_unit.json" | tee "$out/lane_unit.txt" ;;
  coll)
    common="--config C --steps 200 --warmup 20 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg --no-other-configs --no-multi-device"
    timeout -k 10 200 python3 bench.py $common > "$out/bench_C_no_process_group.json" 2>> "$out/coll.err"
    MCALF_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 $common > "$out/bench_C_one_rank_rccl_three_gathers.json" 2>> "$out/coll.err"
    python3 - "$out" <<'PY'
import json, sys
o = sys.argv[1]
a = json.loads(open(o + "/bench_C_no_process_group.json").read().strip().splitlines()[-1])
b = json.loads(open(o + "/bench_C_one_rank_rccl_three_gathers.json").read().strip().splitlines()[-1])
print("no process group: %.4f ms per step" % a["ms_per_step"])
for k, v in b["gathers"].items():
    print(k, v if isinstance(v, str) else "%.4f ms per step (x %.3f), ranks %s (%s), check %s, split %s" % (
        v["ms_per_step"], v["ms_per_step"] / a["ms_per_step"], v["rccl_ranks"], v["rccl_ranks_source"], v["gather_check"],
        {k2: round(v2, 4) for k2, v2 in v["rank0_split"].items() if k2.endswith("_ms")}))
PY
    ;;
  parity)
    timeout -k 10 1000 python3 tools/full_parity.py > "$out/full_parity.json" 2> "$out/full_parity.err"; tail -c 1200 "$out/full_parity.json"
    timeout -k 10 900 python3 tools/fuzz_campaign.py ${1:-1200} ${2:-60000} > "$out/fuzz.txt" 2>&1; tail -4 "$out/fuzz.txt" ;;
  latency)
    timeout -k 10 300 python3 tools/single_call_latency.py > "$out/single_call_latency.txt" 2>&1; tail -12 "$out/single_call_latency.txt"
    timeout -k 10 200 python3 tools/call_overhead.py > "$out/call_overhead.txt" 2>&1; tail -6 "$out/call_overhead.txt" ;;
  dropin)
    timeout -k 10 900 python3 tools/dropin_ranks.py ${@:-1 2 4 6} > "$out/dropin.txt" 2>&1; tail -12 "$out/dropin.txt" ;;
  hostfloor)
    timeout -k 10 400 python3 tools/host_floor.py > "$out/host_floor.txt" 2>&1; tail -20 "$out/host_floor.txt" ;;
  ab)
    cfg=${1:-C}; steps=${2:-200}
    ABL_ROUNDS=${ABL_ROUNDS:-3} bash tools/abl_bench.sh --config $cfg --no-model-leg --no-other-configs --no-multi-device --steps $steps > "$out/ab_$cfg.txt" 2>&1; cat "$out/ab_$cfg.txt" ;;
  hostgap)
    legs=${1:-B,E,E2048}
    run() {  # name, then VAR=value ...
      name=$1; shift
      env "$@" MCALF_HOST_TRACE=1 MCALF_STREAM_TRACE=1 timeout -k 10 300 python3 bench.py --only-other-configs $legs --steps 20 > "$out/hostgap_$name.json" 2> "$out/hostgap_$name.err"
      python3 - "$out/hostgap_$name.json" "$name" <<'PY' | tee -a "$out/hostgap.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["other_configs"]
except Exception as exc:
    print(sys.argv[2], "FAILED", exc); raise SystemExit(0)
for k, v in d.items():
    print("%-22s %-6s device %.4f  host %.4f (x %.3f)  pinned %.4f (x %.3f)  %s, %d blocks, bits %s, parity %.1e" % (
        sys.argv[2], k, v["ms_per_step_device_resident"], v["ms_per_step_host_api"], v["host_over_device"], v["ms_per_step_host_api_pinned"],
        v["host_over_device_pinned"], v["path_host_api"].split("(")[1].rstrip(")"), v["row_blocks_host_api"], v["bit_equal_host_vs_device_entry"],
        v["parity"]["max_abs_dlogL_vs_oracle"]))
PY
      grep -h "mcalf .* trace" "$out/hostgap_$name.err" | sed "s/^/    $name: /" >> "$out/hostgap.txt"
    }
    : > "$out/hostgap.txt"
    for round in 1 2; do
      run default_$round A=0
      run pipeline_$round MCALF_STREAM=0
      run stream_all_$round MCALF_STREAM=2
      run oldplan_$round MCALF_STREAM=0 MCALF_HOST_PLAN=1,1,2,4
    done
    cat "$out/hostgap.txt" ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
