#!/bin/bash
# Round 5 A/B builds of the raster kernels on the GPU box: tools/ab_raster_r5.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# per flag set (each timed twice): the fused frame of the C++ renderer at 4K with the bench's 1024^2 textures, and its per-pass times
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
make -s -j16 > /dev/null 2>&1
for flags in "$@"; do
  rm -f raster.o raster_api.o
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py --host --texture-size 1024 --frames 30 2>&1 | grep -E "raster frame|passes" | sed -e 's/(C++ renderer, fused passes)//' -e "s/'//g" | tr '\n' ' ')
    echo "[$flags] $r"
  done
done
rm -f raster.o raster_api.o; make -s -j16 > /dev/null 2>&1
