"""Developer tool: the kernel timeline of ONE path-traced frame (argv[2]: its index among the incr_pathtrace calls of the run, default the last; bench.py's last
frames are the roofline.alone leg, the timed ones are [warmup, warmup + steps)) out of a rocprofv3 --kernel-trace csv (start / end relative to the frame's first kernel, stream = queue id).
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-raster --steps 4 --warmup 2
                  python tools/frame_timeline.py $(find $OUT -name "*kernel_trace.csv")"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "incr::wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "wf_init_counters" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
last = rows[starts[k]:(starts[k + 1] if k + 1 < len(starts) else len(rows))]
print("frame %d of %d" % (k, len(starts)))
t0 = int(last[0]["Start_Timestamp"])
queues = {}
for r in last:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    name = r["Kernel_Name"].replace("void ", "").split("(")[0].replace("incr::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-28s queue %d  start %8.3f ms  end %8.3f ms  dur %7.3f ms" % (name, q, s, e, e - s))
print("frame span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
