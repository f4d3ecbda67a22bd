#!/bin/bash
# Round 4: deferred accumulation -- timings of the frame with the any-hit launches grouped in different ways (INCR_WF_SHADOW_GROUPS) and with / without the side stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab_deferred.txt; : > $O
run() { # label, env, bench args
  r=$(env $2 timeout -k 10 150 python bench.py --no-raster --no-cpu-baseline $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mray/s', d['ms_per_step'], 'ms', json.dumps(d['roofline']['kernel_ms_per_step']))")
  echo "$1 | $3 | $r" >> $O
}
for g in "0,1,2,3" "0,1,3" "1,3" "0,3" "3"; do
  run "groups $g overlap on pipelines 2" "INCR_WF_SHADOW_GROUPS=$g" "--steps 12 --warmup 3"
  run "groups $g overlap on pipelines 1" "INCR_WF_SHADOW_GROUPS=$g" "--steps 12 --warmup 3 --pipelines 1 --overlap on"
done
run "groups 0,1,2,3 overlap off pipelines 2" "" "--steps 12 --warmup 3 --overlap off"
run "groups 0,1,2,3 overlap off pipelines 1" "" "--steps 12 --warmup 3 --pipelines 1"
for g in "0,1,2,3" "1,3" "3"; do
  run "helmet groups $g overlap on pipelines 2" "INCR_WF_SHADOW_GROUPS=$g" "--scene helmet --res 1920x1080 --steps 64 --warmup 8"
  run "helmet groups $g overlap on pipelines 1" "INCR_WF_SHADOW_GROUPS=$g" "--scene helmet --res 1920x1080 --steps 64 --warmup 8 --pipelines 1 --overlap on"
done
cat $O
