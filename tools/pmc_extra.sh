#!/bin/bash
# Extra PMC passes (instruction mix, I-cache, scalar cache, branches) for the fused kernel. GPU box.
out=${1:-gpurun_out/pmc_extra}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS SQ_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d "$out/pass$i" --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-other-configs --no-multi-device "$@" > "$out/pass$i.json" 2> "$out/pass$i.err" || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: [0.0,0])
for f in glob.glob(out+'/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'mcalf_fused' not in row['Kernel_Name']: continue
        k=row['Counter_Name']; agg[k][0]+=float(row['Counter_Value']); agg[k][1]+=1
with open(out+'/summary.txt','w') as fh:
    for k,(v,n) in sorted(agg.items()):
        line='%-28s per-dispatch mean %.6g  (dispatches %d)'%(k, v/n, n)
        print(line); fh.write(line+'\n')
PY
