#!/bin/bash
# usage (GPU box, repo root): scripts/wide_pmc.sh  -> gpurun_out/pmc_wide/{w64_o2,w128_o65,w64_o33}.json
# SQ counters of the align kernel at two knob-sweep points (genasm_lane_wide_kernel) and at the default (genasm_lane_kernel).
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
out=$root/gpurun_out/pmc_wide; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wo in "64 2" "128 65" "64 33"; do
  set -- $wo; tag=w$1_o$2
  timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
      -d $out/$tag -o pmc --output-format csv -- python3 $root/scripts/wide_pmc_probe.py $1 $2 > $out/$tag.log 2> $out/$tag.err
  python3 - $out/$tag $out/$tag.log <<'PY'
import csv, glob, json, sys, collections
d, log = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
name = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "genasm_lane" in r.get("Kernel_Name", ""):
            name = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
info = json.loads([l for l in open(log) if l.startswith("{")][-1])
out = dict(info, kernel=name, counters={k: sum(v) / len(v) for k, v in sorted(acc.items())})
r = info["window_rounds_per_launch"]
out["valu_instructions_per_window_round"] = out["counters"]["SQ_INSTS_VALU"] / r
out["valu_instructions_per_pair"] = out["counters"]["SQ_INSTS_VALU"] * 64 / info["pairs"]
json.dump(out, open(d + ".json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("W", "O", "kernel", "kernel_ms", "windows_per_pair", "valu_instructions_per_window_round", "valu_instructions_per_pair")}))
PY
done
