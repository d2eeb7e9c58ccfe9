# NOTE: the launch-before-copy variant this script measured (MCALF_SMALL_GATE) was REMOVED from the source after the run (slower:
# profiles/r06_small_call_gate_experiment.txt; commit "Small host calls launch before they copy their rows" has it); kept as the record of how it was measured.
# GPU box: config B's step through host pointers with and without the gated launch (launch first, copy afterwards), interleaved.
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out; : > $out/small_gate.txt
for rep in 1 2 3; do
  for g in 0 1; do
    MCALF_SMALL_GATE=$g MCALF_HOST_TRACE=1 timeout -k 10 200 python3 bench.py --only-other-configs B --steps 50 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
tr = ''
for line in sys.stdin:
    if line.startswith('{'):
        v = json.loads(line)['other_configs']['B']
        print('gate=$g: device %.4f host %.4f (x %.3f) pinned %.4f   %s' % (v['ms_per_step_device_resident'], v['ms_per_step_host_api'], v['host_over_device'], v['ms_per_step_host_api_pinned'], tr))
    elif 'small calls' in line:
        tr = line[line.find('us per call'):].rstrip()[:300]
" >> $out/small_gate.txt
  done
done
cat $out/small_gate.txt
