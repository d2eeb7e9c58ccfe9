#!/bin/bash
# rocprofv3 counter passes for bench.py (GPU box). Each --pmc set is its own run (no trace domains).
# Usage: tools/pmc_profile.sh <outdir> [bench args]
out=${1:-gpurun_out/pmc}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
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
        line='%-24s per-dispatch mean %.6g  (dispatches %d)'%(k, v/n, n)
        print(line); fh.write(line+'\n')
# HBM traffic per bench launch for profiles/traffic.json: the fused kernel runs once more than the bench
# launches (truth synthesis, one workgroup), FETCH_SIZE / WRITE_SIZE are KiB, and gfx950 reports half the
# bytes of coalesced reads (MI355X_MICROARCH.md) -> read side doubled.
if 'FETCH_SIZE' in agg and 'WRITE_SIZE' in agg:
    import json
    (fv,fn),(wv,wn)=agg['FETCH_SIZE'],agg['WRITE_SIZE']
    fetch=fv/max(fn-1,1); write=wv/max(wn-1,1)
    json.dump({"fetch_size_kib_raw":fetch,"write_size_kib":write,"traffic_bytes_per_launch":(2*fetch+write)*1024,
               "dispatches":fn}, open(out+'/traffic.json','w'), indent=1)
PY
