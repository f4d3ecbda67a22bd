#!/bin/bash
# Occupancy A/B of the fused lighting and the HBAO kernel (launch bounds: minimum workgroups per CU): tools/ab_occupancy.sh "<EXTRA flags>" ...
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  rm -f raster.o; make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  a=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py --host --texture-size 1024 --frames 30 2>&1 | grep -E "raster frame|passes" | sed -e 's/(C++ renderer, fused passes)//' | grep -oE "raster frame +[0-9.]+ ms|deferred_lighting., [0-9.]+" | tr '\n' ' ')
  b=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/raster_time.py --host --hbao --texture-size 1024 --frames 10 2>&1 | grep -E "raster frame|passes" | grep -oE "raster frame[^0-9]+[0-9.]+ ms|hbao., [0-9.]+" | tr '\n' ' ')
  echo "[$flags] $a | with HBAO: $b"
done
rm -f raster.o; make -s -j16 > /dev/null 2>&1
