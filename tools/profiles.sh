#!/bin/bash
# Measurement set for one BASELINE config on ONE lease (GPU box): the bench line, the rocprofv3 kernel summary of the same
# command (one fused launch per step), the PMC passes (each --pmc set its own run, no trace domains), the
# instruction-issue micro-benchmark; then the figures are merged into profiles/pmc.json / traffic.json (stamped with the
# kernel-source hash) and the bench line is taken AGAIN, now carrying them (bench_with_counters.json).
# Usage: tools/profiles.sh <config> <outdir> [steps] [round tag, default r06]
cfg=${1:-C}; out=${2:-gpurun_out/prof_$cfg}; steps=${3:-50}; tag=${4:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 10 600 python bench.py --config $cfg --steps $steps > "$out/bench.json" 2> "$out/bench.err" || echo "bench failed"
common="--config $cfg --steps $steps --warmup 5 --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg --no-other-configs --no-multi-device"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/ktrace" --output-format csv -- python3 bench.py $common > "$out/ktrace_bench.json" 2> "$out/ktrace.err" || echo "kernel trace failed"
find "$out/ktrace" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SENDMSG SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d "$out/pmc$i" --output-format csv -- python3 bench.py $common > "$out/pmc$i.json" 2> "$out/pmc$i.err" || echo "pmc pass $i failed"
done
[ -x build/issue_rate ] || hipcc -O3 --offload-arch=gfx950 -o build/issue_rate tools/micro/issue_rate.hip 2>/dev/null
[ -x build/issue_rate ] && timeout -k 10 120 build/issue_rate > "$out/issue_rate.json" 2> "$out/issue_rate.err"
# (the unit of SQ_THREAD_CYCLES_VALU, measured on the same lease: kernels with a known share of active lanes)
[ -x build/lane_unit ] || hipcc -O3 --offload-arch=gfx950 -o build/lane_unit tools/micro/lane_unit.hip 2>/dev/null
[ -x build/lane_unit ] && timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -d "$out/lane_unit" --output-format csv -- build/lane_unit > "$out/lane_unit.log" 2>&1 \
  && python3 tools/micro/lane_unit_summary.py "$out/lane_unit" "$out/lane_unit.json" > "$out/lane_unit.txt"
python3 tools/profiles_summary.py "$cfg" "$out"
python3 tools/collect_profiles.py --round $tag --src "$out" $cfg
timeout -k 10 600 python bench.py --config $cfg --steps $steps > "$out/bench_with_counters.json" 2> "$out/bench_with_counters.err" || echo "second bench failed"
cp "$out/bench_with_counters.json" profiles/${tag}_bench_${cfg}_with_counters.json
mkdir -p gpurun_out/profiles_${tag}; cp profiles/${tag}_* profiles/pmc.json profiles/traffic.json gpurun_out/profiles_${tag}/ 2>/dev/null
