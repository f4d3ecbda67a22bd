#!/bin/bash
# A/B of RUNTIME settings (environment variables read by incr_initialize) on the GPU box, one build:  tools/ab_env.sh "VAR=1" "VAR=2 OTHER=x" ...
# each setting is benched REPS times (default 2): Mray/s and ms per 4K frame, and the helmet 1080p frame
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc && make -s -j16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  for rep in $(seq ${REPS:-2}); do
    r=$(env $setting timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    h=$(env $setting timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --scene helmet --res 1920x1080 --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "[$setting] 4K: $r   helmet 1080p: $h ms"
  done
done
