#!/bin/bash
# A/B of the sun cover map's resolution on the path tracer (the map's size against the L2): tools/ab_sun_cover_g.sh 2048 1024 ...
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for g in "$@"; do
  rm -f pathtrace_wf.o raster.o raster_api.o api.o
  make -s -j16 EXTRA="-DINCR_SUN_COVER_G=$g" > /dev/null 2>&1 || { echo "build failed for $g"; continue; }
  for rep in 1 2; do
    (cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); a=d['roofline']['alone']['kernel_ms_per_step']
print('[G = $g] frame %.3f ms  answered %d  alone: shade %.3f shadow %.3f' % (d['ms_per_step'], d['config']['shadow_queue']['rays_answered_by_sun_cover_per_step'], a['wf_shade_kernel'], a['wf_shadow_kernel']))")
  done
done
rm -f pathtrace_wf.o raster.o raster_api.o api.o; make -s -j16 > /dev/null 2>&1
