#!/bin/bash
# Visibility scan diagnostics (tile counts, per-wave band-phase cycles): tools/vis_diag.sh   -- builds with -DINCR_DIAG_VIS_COUNT, prints, rebuilds the product
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
rm -f raster.o raster_api.o; make -s -j16 EXTRA="-DINCR_DIAG_VIS_COUNT=${DIAG_LEVEL:-1} ${DIAG_EXTRA:-}" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py --texture-size 1024 --frames 5 ${RASTER_ARGS:-} 2>&1 | grep -E "vis diag|visibility queues|raster frame")
rm -f raster.o raster_api.o; make -s -j16 > /dev/null 2>&1
