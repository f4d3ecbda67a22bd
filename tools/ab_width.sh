#!/bin/bash
# A/B of narrower BVH nodes (INCR_BVH_WIDTH = children per node the builder fills and the traversal tests) on the GPU box:
#   tools/ab_width.sh 8 6 4      -- full rebuild of the library per width (builder + kernels), bench with every kernel timed alone
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for w in "$@"; do
  make -s -j16 -B EXTRA="-DINCR_BVH_WIDTH=$w" > /dev/null 2>&1 || { echo "build failed for width $w"; continue; }
  for pipes in 1 0; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 --pipelines $pipes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['kernel_ms_per_step']))")
    echo "[width $w] pipelines=$pipes -> $r"
  done
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python -m pytest tests/test_gpu_pathtrace.py -m gpu -x -q 2>&1 | tail -1)
done
make -s -j16 -B > /dev/null 2>&1
