#!/bin/bash
# The sun's shadow rays of the 4K Sponza-stat raster frame, on the GPU box: tools/sun_count.sh
#   builds with -DINCR_DIAG_SUN_COUNT=1 / 2 and prints light_shadow_rays = rays + 65536 x (1: occluded rays, 2: rays the sun cover map answers), plus the map's header;
#   -DINCR_DIAG_SUN_COUNT also prints the map look-up of a grid of sample pixels (stderr of raster_time.py).  Rebuilds the product at the end.
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for f in "" "-DINCR_DIAG_SUN_COUNT=1" "-DINCR_DIAG_SUN_COUNT=2"; do
  rm -f raster.o raster_api.o; make -s -j16 EXTRA="$f" > /dev/null 2>&1 || { echo "build failed for $f"; continue; }
  echo "[$f]"; (cd $GRAFT_REPO_ROOT && python tools/raster_time.py --texture-size 256 --frames 1 2>&1 | grep -E "shadow rays|sun cover header" | sort | uniq -c)
done
rm -f raster.o raster_api.o; make -s -j16 > /dev/null 2>&1
