#!/bin/bash
# A/B full builds on the GPU box for the raster frame: tools/ab_raster.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  make -s clean > /dev/null 2>&1
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py --frames 30 2>/dev/null | grep "raster frame")
    echo "[$flags] -> $r"
  done
done
make -s clean > /dev/null 2>&1; make -s -j16 > /dev/null 2>&1
