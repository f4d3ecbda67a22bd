#!/bin/bash
# A/B builds of the path-tracing kernels only, on the GPU box: tools/ab_wf.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# (rebuilds pathtrace_wf.o with each flag set, relinks, runs the bench twice)
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
make -s -j16 > /dev/null 2>&1
for flags in "$@"; do
  rm -f pathtrace_wf.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])")
    echo "[$flags] -> $r"
  done
done
rm -f pathtrace_wf.o; make -s -j16 > /dev/null 2>&1
