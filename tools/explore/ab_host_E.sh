cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05e
for r in 1 2; do
  for t in new old; do
    if [ $t = new ]; then d=.; else d=build/r04tree; fi
    ( cd $d && timeout -k 10 300 python bench.py --config E --steps 20 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_api']; print('$t round $r: step %.4f host pageable %.4f pinned %.4f path %s' % (d['ms_per_step'], h['ms_per_step'], h['ms_per_step_pinned'], h['path']))" ) | tee -a gpurun_out/r05e/ab_host_E.txt
  done
done
