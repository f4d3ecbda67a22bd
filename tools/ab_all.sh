#!/bin/bash
# A/B full builds (every translation unit) on the GPU box: tools/ab_all.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...: path tracer alone + default issue, raster frame (tools/raster_time.py --host --texture-size 1024)
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
for flags in "$@"; do
  make -s clean > /dev/null 2>&1
  make -s -j16 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  a=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --pipelines 1 --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['ms_per_step'], ' '.join('%s %.3f' % (n.replace('wf_','').replace('_kernel',''), v) for n, v in k.items() if v))")
  echo "[$flags] alone: $a"
  for rep in 1 2; do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    t=$(cd $GRAFT_REPO_ROOT && timeout -k 10 100 python tools/raster_time.py --host --texture-size 1024 2>&1 | grep -E "raster frame|passes" | tr "\n" " " | sed -E "s/.*fused passes\) ([0-9.]+) ms.*'visibility', ([0-9.]+)\), \('gbuffer', ([0-9.]+)\).*'deferred_lighting', ([0-9.]+)\).*/raster \1 ms (vis \2 gbuffer \3 lighting \4)/")
    echo "[$flags] default issue: $r ms   $t"
  done
done
make -s clean > /dev/null 2>&1; make -s -j16 > /dev/null 2>&1
