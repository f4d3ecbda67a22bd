#!/bin/bash
# The per-round profile set, on the GPU box:  tools/profile_round.sh <part> <tag>   (parts: a | b | c; results under gpurun_out/<tag>/)
#   a: GPU test log, bench (default + --pipelines 1), rocprofv3 --kernel-trace --stats of both, lane statistics, shard emulation (1 and 8 samples in flight)
#   b: PMC passes of the default bench (tools/pmc.sh)              c: PMC passes of --pipelines 1, the five BASELINE configs (tools/run_configs.py)
set -u
PART=$1; TAG=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R/incarnation-renderer_amd/csrc && make -s -j16 2>&1 | grep -i error; make -s -C $R/oracle 2>&1 | grep -i error
cd $R
if [ "$PART" = a ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
  timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
  timeout -k 10 200 python bench.py --pipelines 1 --no-raster --no-cpu-baseline > $O/bench_pipelines1.json 2>> $O/bench.err; cut -c1-200 $O/bench_pipelines1.json
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --no-cpu-baseline --no-raster > $O/rocprof.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_p1 -- python3 $R/bench.py --no-cpu-baseline --no-raster --pipelines 1 > $O/rocprof_p1.log 2>&1
  cp $(find $O/rocprof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; cp $(find $O/rocprof_p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pipelines1.csv
  rm -rf $O/rocprof $O/rocprof_p1
  cd $R
  timeout -k 10 300 python tools/lane_stats.py > $O/lane_stats.txt 2>&1; tail -4 $O/lane_stats.txt
  timeout -k 10 300 python tools/shard_emulation.py --steps 6 --out $O/shard_emulation.json > /dev/null 2>> $O/bench.err
  timeout -k 10 300 python tools/shard_emulation.py --steps 6 --samples-in-flight 8 --out $O/shard_emulation_8_in_flight.json > /dev/null 2>> $O/bench.err
  python -c "import json; [print(f, json.load(open('$O/' + f))['projected_speedup_compute_only']) for f in ('shard_emulation.json', 'shard_emulation_8_in_flight.json')]"
elif [ "$PART" = b ]; then
  bash tools/pmc.sh $TAG/pmc --no-raster --steps 5 --warmup 1 | tail -3
  cp $O/pmc/summary.txt $O/pmc_summary.txt; cp $O/pmc/hbm_traffic.json $O/hbm_traffic.json; rm -rf $O/pmc/pass*
elif [ "$PART" = c ]; then
  bash tools/pmc.sh $TAG/pmc1 --no-raster --steps 5 --warmup 1 --pipelines 1 | tail -3
  cp $O/pmc1/summary.txt $O/pmc_summary_pipelines1.txt; cp $O/pmc1/hbm_traffic.json $O/hbm_traffic_pipelines1.json; rm -rf $O/pmc1/pass*
  timeout -k 10 600 python tools/run_configs.py > $O/configs.json 2> $O/configs.err; tail -30 $O/configs.json
fi
