#!/bin/bash
# same-box A/B, three rounds: accumulate behind each any-hit launch (side stream) or at the end; one or two pipelines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab_deferred2.txt; : > $O
for rep in 1 2 3; do
for acc in 1 0; do for p in 1 2; do
  r=$(INCR_WF_ACC_SIDE=$acc timeout -k 10 150 python bench.py --no-raster --no-cpu-baseline --steps 16 --warmup 3 --pipelines $p --overlap on 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mray/s', d['ms_per_step'], 'ms')")
  echo "round $rep acc_side $acc pipelines $p | $r" >> $O
done; done; done
sort -k4,4 -k6,6 $O
