#!/bin/bash
# A/B of source variants on one GPU box: tools/ab_src.sh <dir1> <dir2> ...   Each directory holds replacement files for
# incarnation-renderer_amd/csrc/; a variant is built once and benched REPS times (default 2), variants interleaved per repetition.
set -u
C=$GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
mkdir -p /tmp/ab
for v in "$@"; do
  cp $GRAFT_REPO_ROOT/$v/* $C/ && (cd $C && make -s -j16 > /dev/null 2>&1) || { echo "build failed for $v"; exit 1; }
  cp $C/libincr_hip.so /tmp/ab/$(basename $v).so
done
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    cp /tmp/ab/$(basename $v).so $C/libincr_hip.so
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    h=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --scene helmet --res 1920x1080 --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "[$v] 4K: $r   helmet 1080p: $h ms"
  done
done
