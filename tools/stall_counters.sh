#!/bin/bash
# Round 5: the stall picture of the path tracer's kernels from PMC counters (PC sampling is refused on this pool and the thread-trace decoder is not installed:
# DESIGN 5.1).  Separate rocprofv3 --pmc passes over bench.py with every kernel alone on one stream (--pipelines 1); per-kernel per-launch averages.
# usage: tools/stall_counters.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_COALESCEABLE_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-raster --steps 3 --warmup 1 --pipelines 1 "$@" > $OUT/pass$i.log 2>&1
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
        line = "%-36s %-40s per-launch %.6g  (launches %d)" % (kn, cn, v / n, n)
        o.write(line + "\n")
PY
rm -rf $OUT/pass*/
grep -c . $OUT/summary.txt
