"""Chronological kernel / copy list of the last `window_ms` of a rocprofv3 --kernel-trace --memory-copy-trace output dir."""
import csv, glob, sys
d, window_ms, limit = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("scrg::", "").replace("void ", "")[:26], "q" + r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), "copy"))
rows.sort()
end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= end - window_ms * 1e6 and "rocprim" not in r[2] and "rocclr" not in r[2] and "totals" not in r[2]]
t0 = sel[0][0]
for s, e, n, q in sel[:limit]:
    print("%8.3f %7.3f  %-28s %s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
