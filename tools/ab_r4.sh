#!/bin/bash
# A/B builds of the path-tracing kernels on the GPU box with the DEFAULT issue (one pipeline, side stream): tools/ab_r4.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# each flag set is built once and benched REPS times (default 2): Mray/s, ms per 4K frame, and the helmet 1080p frame
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
make -s -j16 > /dev/null 2>&1
for flags in "$@"; do
  rm -f pathtrace_wf.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in $(seq ${REPS:-2}); do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    h=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --scene helmet --res 1920x1080 --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "[$flags] 4K: $r   helmet 1080p: $h ms"
  done
done
rm -f pathtrace_wf.o; make -s -j16 > /dev/null 2>&1
