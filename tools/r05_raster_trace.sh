#!/bin/bash
# round 5: rocprofv3 kernel trace of the fused raster frame (the C++ renderer, 4K, 1024^2 textures, x16 sampler): average launch durations per kernel
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_raster -- python3 $R/tools/raster_time.py --host --texture-size 1024 --frames 30 > $O/rocprof_raster.log 2>&1
cp $(find $O/rocprof_raster -name "*kernel_stats.csv" | head -1) $O/raster_frame_kernel_stats.csv; rm -rf $O/rocprof_raster
head -30 $O/raster_frame_kernel_stats.csv | cut -c1-200
