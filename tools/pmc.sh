#!/bin/bash
# Collect PMC counters for the bench in separate rocprofv3 passes (no tracing domains alongside --pmc).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_WAVE32_LDS SQ_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1
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
        line = "%-36s %-32s per-launch %.6g  (launches %d)" % (kn, cn, v / n, n)
        print(line); o.write(line + "\n")
# HBM traffic per launch of every path-tracing kernel, stamped with the hash of the kernel sources it was measured on:
# bench.py attaches roofline.traffic only when the stamp matches the sources it runs (profiles/hbm_traffic_latest.json)
import json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
kernels = {}
for (kn, cn), (v, n) in tot.items():
    if cn in ("FETCH_SIZE", "WRITE_SIZE"):
        kernels.setdefault(kn.split("<")[0], {})[cn + "_KB_per_launch"] = v / n
    if cn in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):     # VALU issue pressure and lane utilisation
        kernels.setdefault(kn.split("<")[0], {})[cn + "_per_launch"] = v / n
for kn, d in kernels.items():
    if "FETCH_SIZE_KB_per_launch" in d and "WRITE_SIZE_KB_per_launch" in d:
        d["hbm_bytes_per_launch"] = int(2 * d["FETCH_SIZE_KB_per_launch"] * 1024 + d["WRITE_SIZE_KB_per_launch"] * 1024)
_args = "$*".split()
_pipes = int(_args[_args.index("--pipelines") + 1]) if "--pipelines" in _args else 1      # the library's choice with the any-hit launches on the side stream
_overlap = (_args[_args.index("--overlap") + 1] == "on") if "--overlap" in _args else ("--pipelines" not in _args or _pipes != 1)      # bench.py: an explicit --pipelines 1 times kernels alone
json.dump({"kernel_source_hash": bench.kernel_source_hash(), "bench_args": "$*", "pipelines": _pipes, "bounce_overlap": _overlap, "kernels": kernels,
           "source": "tools/pmc.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py $*; bytes = 2 x FETCH_SIZE x 1024 + "
                     "WRITE_SIZE x 1024 (MI355X_MICROARCH.md 'HBM': FETCH_SIZE is in KB and counts 128-B requests as 64 B on gfx950; the factor 2 is "
                     "calibrated for wide streaming reads, so for 80-B node gathers this is an upper estimate); counter collection serialises the launches, so "
                     "these are per-launch figures of kernels that have the GPU to themselves, a launch = one pipeline's share of the frame (the whole frame with one pipeline)"},
          open("$OUT/hbm_traffic.json", "w"), indent=1)
PY
