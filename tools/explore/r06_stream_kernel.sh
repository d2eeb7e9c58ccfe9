# GPU box: (1) what does the streaming launch's machinery cost the KERNEL, with the rows resident in HBM (MCALF_STREAM_DEVICE=1: the
# *_device entry through the streaming kernel) -- single-tile instantiation (config C / D: no register spills) against the tiled one
# (config E: 11 VGPR spills)?  (2) tiled spectra through host pointers: where does the streaming launch stop beating the row-block
# pipeline (config E at 2048 / 4096 / 8192 / 16384 rows)?
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out; : > $out/stream_kernel.txt
for rep in 1 2; do
  for cfg in C D E; do
    for v in 0 1; do
      steps=100; [ $cfg = E ] && steps=20; [ $cfg = D ] && steps=20
      MCALF_STREAM_DEVICE=$v timeout -k 10 300 python3 bench.py --config $cfg --gpus 1 --steps $steps --cpu-seconds 0 --no-host-api --no-strong-ref --no-model-leg --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg stream_device=$v ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['kernel_ms']))" >> $out/stream_kernel.txt
    done
  done
done
cat $out/stream_kernel.txt
: > $out/tiled_crossover.txt
for rep in 1 2; do
  for s in 1 2; do
    MCALF_STREAM=$s timeout -k 10 300 python3 bench.py --only-other-configs E2048,E4096,E8192,E --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['other_configs']
for k,v in d.items(): print('MCALF_STREAM=$s %-6s device %.4f host %.4f (x %.3f) pinned %.4f (x %.3f) %s' % (k, v['ms_per_step_device_resident'], v['ms_per_step_host_api'], v['host_over_device'], v['ms_per_step_host_api_pinned'], v['host_over_device_pinned'], v['path_host_api'][:24]))" >> $out/tiled_crossover.txt
  done
done
cat $out/tiled_crossover.txt
