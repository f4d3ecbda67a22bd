#!/bin/bash
# PMC counters of the raster frame's kernels (tools/raster_time.py, Sponza-stat 4K), separate rocprofv3 passes:  tools/pmc_raster.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/raster_time.py --frames 5 --texture-size 1024 ${RASTER_ARGS:---host} > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "incr::" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
with open("$OUT/summary.txt", "w") as o:
    for (kn, cn), (v, n) in sorted(tot.items()):
        o.write("%-36s %-32s per-launch %.6g  (launches %d)\n" % (kn, cn, v / n, n))
PY
rm -rf $OUT/pass*/
grep -E "gbuffer_kernel|vis_scan_kernel|deferred_lighting_kernel" $OUT/summary.txt
