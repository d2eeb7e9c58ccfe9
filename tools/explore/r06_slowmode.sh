# GPU box: is the slow mode of the pipelined entry (config E: ~0.35 ms above the fast mode, per process) two of the context's
# three streams sharing ONE hardware queue?  HIP deals streams to GPU_MAX_HW_QUEUES (default 4) queues per process; with 1 every
# stream shares, with 8 none should.  Legs B,E as in the runs that showed the slow mode (the B leg's context comes and goes first).
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out; : > $out/slowmode.txt
for rep in 1 2; do
  for q in default; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    for plan in 1,1,2,4 auto; do
      if [ $plan = auto ]; then unset MCALF_HOST_PLAN; else export MCALF_HOST_PLAN=$plan; fi
      MCALF_HOST_TRACE=1 timeout -k 10 300 python3 bench.py --only-other-configs B,E --steps 20 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
tr = ''
for line in sys.stdin:
    if line.startswith('{'):
        v = json.loads(line)['other_configs']['E']
        print('GPU_MAX_HW_QUEUES=$q plan $plan: device %.4f host %.4f pinned %.4f   %s' % (v['ms_per_step_device_resident'], v['ms_per_step_host_api'], v['ms_per_step_host_api_pinned'], tr))
    elif '43 calls' in line:
        tr = line[line.find('first block'):].rstrip()[:300]
" >> $out/slowmode.txt
    done
  done
done
unset GPU_MAX_HW_QUEUES MCALF_HOST_PLAN
cat $out/slowmode.txt
