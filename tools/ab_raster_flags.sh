#!/bin/bash
# A/B builds of raster.hip on the GPU box: tools/ab_raster_flags.sh "<raster_time.py args>" "<EXTRA flags 1>" "<EXTRA flags 2>" ...   (each timed twice)
set -u
ARGS=$1; shift
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  rm -f raster.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py $ARGS 2>&1 | grep -E "raster frame|hbao" | tr '\n' ' ')
    echo "[$flags] $r"
  done
done
rm -f raster.o; make -s -j16 > /dev/null 2>&1
