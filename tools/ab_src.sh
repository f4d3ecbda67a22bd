#!/bin/bash
# A/B of source variants on one GPU box: tools/ab_src.sh <dir1> <dir2> ...   Each directory holds replacement files for
# incarnation-renderer_amd/csrc/; a variant is built once and benched REPS times (default 2), variants interleaved per repetition.
# Variants are built in a SCRATCH COPY of csrc (/tmp/ab/csrc): the tracked sources are never overwritten, and the library the tree started with is put back on
# exit (also on an error or a signal) -- after a run the tree holds what it held before.
set -u
C=$GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
mkdir -p /tmp/ab
cp $C/libincr_hip.so /tmp/ab/_original.so 2>/dev/null
trap 'if [ -f /tmp/ab/_original.so ]; then cp /tmp/ab/_original.so $C/libincr_hip.so; fi' EXIT INT TERM
for v in "$@"; do
  rm -rf /tmp/ab/csrc && mkdir -p /tmp/ab/tree/incarnation-renderer_amd && cp -r $GRAFT_REPO_ROOT/include /tmp/ab/tree/ 2>/dev/null
  rm -rf /tmp/ab/tree/incarnation-renderer_amd/csrc && cp -r $C /tmp/ab/tree/incarnation-renderer_amd/csrc
  S=/tmp/ab/tree/incarnation-renderer_amd/csrc
  rm -f $S/*.o $S/host/*.o $S/libincr_hip.so
  cp $GRAFT_REPO_ROOT/$v/* $S/ && (cd $S && make -s -j16 > /dev/null 2>&1) || { echo "build failed for $v"; exit 1; }
  cp $S/libincr_hip.so /tmp/ab/$(basename $v).so
done
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    cp /tmp/ab/$(basename $v).so $C/libincr_hip.so
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    h=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --scene helmet --res 1920x1080 --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    a=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 --pipelines 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('alone: shadow %.3f extend %.3f shade %.3f' % (k['wf_shadow_kernel'], k['wf_extend_kernel'], k['wf_shade_kernel']))")
    echo "[$v] 4K: $r   helmet 1080p: $h ms   $a"
  done
done
