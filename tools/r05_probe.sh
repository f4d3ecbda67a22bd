#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R/tools/microbench && hipcc -O3 --offload-arch=gfx950 -o line_fill line_fill.hip 2>&1 | grep -i error
timeout -k 10 240 ./line_fill > $O/line_fill.txt 2>&1; rc=$?; cat $O/line_fill.txt
if grep -q "Memory access fault" $O/line_fill.txt; then exit 1; fi
exit $rc
