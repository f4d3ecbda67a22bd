#!/bin/bash
# A/B full builds on the GPU box: tools/ab_flags.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...  (each runs the bench twice)
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  make -s clean > /dev/null 2>&1
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])")
    echo "[$flags] -> $r"
  done
done
make -s clean > /dev/null 2>&1; make -s -j16 > /dev/null 2>&1
