"""Copy the judged part of a rocprofv3 --kernel-trace --stats run into profiles/.
usage: save_profile.py <gpurun_out/prof_dir> <profiles/name.csv> [note]"""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
import glob
f = glob.glob(src + "/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open(dst, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats summary (kernel_stats.csv), %s\n" % note)
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        if "scrg::" in r["Name"] or float(r["Percentage"]) >= 1.0:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
print(open(dst).read())
