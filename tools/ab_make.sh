#!/bin/bash
# A/B of make variables for the path-tracing kernels on the GPU box: tools/ab_make.sh "" "SCHED_pathtrace_wf='-mllvm -amdgpu-sched-strategy=iterative-ilp'" "EXTRA='-mllvm -amdgpu-set-wave-priority'" ...
# per setting (REPS runs, default 2): the 4K frame under the default issue, and every kernel alone on one stream
set -u
cd $GRAFT_REPO_ROOT/incarnation-renderer_amd/csrc
make -s -j16 > /dev/null 2>&1
for setting in "$@"; do
  rm -f pathtrace_wf.o
  eval make -s -j16 $setting > /dev/null 2>&1 || { echo "build failed for $setting"; continue; }
  for rep in $(seq ${REPS:-2}); do
    r=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 16 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    a=$(cd $GRAFT_REPO_ROOT && timeout -k 10 200 python bench.py --no-cpu-baseline --no-raster --steps 8 --warmup 2 --pipelines 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
g=lambda n: sum(v for kk,v in k.items() if n in kk)
print('alone: frame %.2f shadow %.3f extend %.3f shade %.3f acc %.3f' % (d['ms_per_step'], g('shadow'), g('extend'), g('shade'), g('accumulate')))")
    echo "[$setting] frame $r ms   $a"
  done
done
rm -f pathtrace_wf.o; make -s -j16 > /dev/null 2>&1
