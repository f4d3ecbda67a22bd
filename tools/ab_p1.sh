#!/bin/bash
# A/B builds timed with every kernel ALONE (bench.py --pipelines 1: one stream, in order) and with the default issue:  tools/ab_p1.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# prints ms per 4K frame alone, the per-kernel sums alone, and ms per frame of the default issue (twice)
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  rm -f pathtrace_wf.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  a=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --pipelines 1 --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['ms_per_step'], ' '.join('%s %.3f' % (n.replace('wf_','').replace('_kernel',''), v) for n, v in k.items() if v))")
  echo "[$flags] alone: $a"
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "[$flags] default issue: $r ms"
  done
done
rm -f pathtrace_wf.o; make -s -j16 > /dev/null 2>&1
