cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab_overlap.txt; : > $O
for p in 1 2; do for ov in off on; do
  for sc in "--scene helmet --res 1920x1080 --steps 64 --warmup 8" "--steps 12 --warmup 3"; do
    timeout -k 10 120 python bench.py $sc --no-raster --no-cpu-baseline --pipelines $p --overlap $ov 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench pipelines $p overlap $ov | $sc |', d['value'], 'Mray/s', d['ms_per_step'], 'ms')" >> $O
  done
done; done
for S in 1 8; do for cfg in "1 off" "1 on" "2 off" "2 on" "4 off"; do set -- $cfg
  timeout -k 10 200 python tools/shard_emulation.py --steps 6 --samples-in-flight $S --pipelines $1 --overlap $2 --shard-list 0,3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1/8 shard of 4K, S=$S pipelines $1 overlap $2: full', d['full_frame_ms'], 'shards', d['shard_ms'])" >> $O
done; done
cat $O
