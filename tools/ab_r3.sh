#!/bin/bash
# A/B builds of the path-tracing kernels on the GPU box: tools/ab_r3.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# Rebuilds pathtrace_wf.o with each flag set, relinks, runs the bench once with one pipeline (every kernel timed alone) and once
# with the library's choice; prints Mray/s, ms/frame and the per-kernel milliseconds.  ENVS="A=1 B=2" is exported to the bench.
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
make -s -j16 > /dev/null 2>&1
for flags in "$@"; do
  rm -f pathtrace_wf.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for pipes in ${PIPES:-1 0}; do
    r=$(cd $GRAFT_REPO_ROOT && env ${ENVS:-} timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 --pipelines $pipes ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['kernel_ms_per_step']))")
    echo "[$flags] pipelines=$pipes -> $r"
  done
done
rm -f pathtrace_wf.o; make -s -j16 > /dev/null 2>&1
