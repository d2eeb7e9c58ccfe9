cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06; mkdir -p $out
for kind in pageable pinned; do
  for plan in new old; do
    if [ $plan = old ]; then if [ $kind = pinned ]; then export MCALF_HOST_PLAN=1,7; else export MCALF_HOST_PLAN=1,1,2,4; fi; else unset MCALF_HOST_PLAN; fi
    echo "== E $kind $plan" >> $out/timeline_E.txt
    timeout -k 10 200 python3 tools/pipeline_timeline.py E $kind 2>> $out/timeline_E.txt >/dev/null
  done
done
unset MCALF_HOST_PLAN
cat $out/timeline_E.txt
