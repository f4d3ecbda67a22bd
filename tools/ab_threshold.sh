#!/bin/bash
# A/B a compile-time constant of the path tracer on the GPU box: tools/ab_threshold.sh MACRO v1 v2 ...
set -u
MACRO=$1; shift
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for v in "$@"; do
  touch pathtrace_wf.hip
  make -s EXTRA="-D$MACRO=$v" > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])")
  echo "$MACRO=$v -> $r"
done
touch pathtrace_wf.hip; make -s > /dev/null 2>&1
