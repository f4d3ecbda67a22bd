#!/bin/bash
# Average vis_scan_kernel launch (rocprofv3 kernel trace) of diagnostic builds: tools/vis_variants.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
set -u
R=$GRAFT_REPO_ROOT; C=$R/incarnation-renderer_amd/csrc
cd /tmp && export TMPDIR=/tmp
for flags in "$@"; do
  (cd $C && rm -f raster.o raster_api.o && make -s -j16 EXTRA="$flags" > /dev/null 2>&1) || { echo "build failed for $flags"; continue; }
  rm -rf /tmp/visvar; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/visvar -- python3 $R/tools/raster_time.py --host --texture-size 1024 --frames 30 > /tmp/visvar.log 2>&1
  f=$(find /tmp/visvar -name "*kernel_stats.csv" | head -1)
  echo "[$flags] $(python3 - $f <<'PY'
import csv, sys
want = ("vis_setup_kernel", "vis_scan_kernel", "vis_resolve_kernel", "gbuffer_kernel", "deferred_lighting_kernel")
print("  ".join("%s %.4f ms" % (w, float(r["AverageNs"]) / 1e6) for w in want for r in csv.DictReader(open(sys.argv[1])) if "incr::" + w in r["Name"]))
PY
)"
done
(cd $C && rm -f raster.o raster_api.o && make -s -j16 > /dev/null 2>&1)
