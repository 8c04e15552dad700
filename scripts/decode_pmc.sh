#!/bin/bash
# usage (GPU box, repo root): scripts/decode_pmc.sh [slots=8]
# HBM traffic of scrg_decode_edit_stream (decode_edits_kernel<true>): one rocprofv3 --pmc pass per counter set (kernel trace only)
# over `scripts/decode_timing.py --reps 1 --slots N --busy 0`; the launch of N x 100 000 pairs is the one with the most traffic.
slots=${1:-8}
kind=${2:-hbm}         # hbm: traffic counters; sq: instruction / wait counters of the shader sequencer
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
out=$root/gpurun_out/pmc_decode
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
if [ "$kind" = sq ]; then
  sets=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS")
else
  sets=("WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "FETCH_SIZE")
fi
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc --output-format csv -- python3 $root/scripts/decode_timing.py --reps 1 --slots $slots --busy 0 > $out/p$i.log 2>&1
done
python3 - $out $slots <<'P'
import csv, glob, sys, json, collections
d, slots = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(list)
for f in glob.glob(d + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if ("decode_edits" in kn) and ("<true>" in kn or "ILb1" in kn):          # the storing form of whichever decoder the launches took
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# the timed launches of the N-slot decode are the ones with the largest values; the one-slot launches the smallest
out = {k: {"largest_launches_mean": sum(sorted(v)[-2:]) / 2, "smallest_launches_mean": sum(sorted(v)[:2]) / 2, "n": len(v)} for k, v in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open(d + "/summary.json", "w"), indent=1)
P
