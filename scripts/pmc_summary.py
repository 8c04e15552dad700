"""Average the PMC counters of every aligner-kernel (genasm_align_kernel / genasm_lane_kernel) dispatch found under a pmc_run.sh output dir."""
import csv, glob, sys, collections, json
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "genasm_align_kernel" in r.get("Kernel_Name", "") or "genasm_lane_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
out["_dispatches_per_counter"] = {k: len(v) for k, v in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open(d + "/summary.json", "w"), indent=1)
