"""Per-kernel and per-copy totals of the LAST `window_ms` of a rocprofv3 --kernel-trace --memory-copy-trace output dir."""
import csv, glob, sys, collections
d, window_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " ))
end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= end - window_ms * 1e6]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, name in sel:
    acc[name][0] += 1
    acc[name][1] += e - s
print("window %.1f ms, %d events" % (window_ms, len(sel)))
for name, (cnt, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%8.3f ms  %5d x  %s" % (ns / 1e6, cnt, name))
