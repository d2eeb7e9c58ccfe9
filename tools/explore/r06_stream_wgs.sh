# GPU box: the streaming launch's dedicated set-up workgroups (MCALF_STREAM_WGS, default 16) and rows per claim (MCALF_STREAM_CHUNK, 32) against the
# batch size: host step of C (4096 rows), D (32768), E2048 (2048 x 5 tiles), E4096, two interleaved rounds.
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out; : > $out/stream_wgs.txt
for rep in 1 2; do
  for w in 8 16 24 32 48 64; do
    MCALF_STREAM_WGS=$w timeout -k 10 300 python3 bench.py --only-other-configs C,D,E2048,E4096 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['other_configs']
print('wgs=%-3s ' % '$w' + '  '.join('%s %.4f/%.4f (x %.3f)' % (k, v['ms_per_step_host_api'], v['ms_per_step_device_resident'], v['host_over_device']) for k,v in d.items()))" >> $out/stream_wgs.txt
  done
done
for c in 8 16 64 128; do
  MCALF_STREAM_CHUNK=$c timeout -k 10 300 python3 bench.py --only-other-configs C,D,E2048,E4096 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['other_configs']
print('chunk=%-3s ' % '$c' + '  '.join('%s %.4f/%.4f (x %.3f)' % (k, v['ms_per_step_host_api'], v['ms_per_step_device_resident'], v['host_over_device']) for k,v in d.items()))" >> $out/stream_wgs.txt
done
cat $out/stream_wgs.txt
