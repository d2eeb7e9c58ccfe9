# GPU box: the GPU-side timeline of pipelined host-pointer calls of config E (MCALF_HOST_TRACE=2), several processes per kind --
# the entry's time is bimodal BETWEEN processes (a slow mode ~0.35 ms above the fast one): catch both.
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out; : > $out/timeline_E.txt
for rep in 1 2 3 4 5; do
  for kind in pageable pinned; do
    echo "== E $kind (process $rep)" >> $out/timeline_E.txt
    timeout -k 10 200 python3 tools/pipeline_timeline.py E $kind 2>&1 >/dev/null | grep -v amdgpu.ids | cut -c1-400 >> $out/timeline_E.txt
  done
done
grep "ms per call" $out/timeline_E.txt | cut -c1-60
