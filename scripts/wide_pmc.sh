#!/bin/bash
# usage (GPU box, repo root): scripts/wide_pmc.sh  -> gpurun_out/pmc_wide/w<W>_o<O>.json
# SQ counters and HBM requests of the align kernel at knob-sweep points of the other one-pair-per-lane kernels
# (genasm_lane_wide_kernel: 64/2, 128/65; genasm_lane_parts_kernel: 160/81, 256/129, 128/20) next to the default (genasm_lane_kernel):
# rocprofv3 --pmc on the SHIPPED library, one pass per counter set; the window rounds come from one more, unprofiled run with the
# -DSCRG_STATS build (ab_libs/lib_stats.so, built in the container with scripts/ab.sh build stats -DSCRG_STATS).
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
out=$root/gpurun_out/pmc_wide; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wo in "64 2" "128 65" "160 81" "256 129" "128 20" "64 33"; do
  set -- $wo; tag=w$1_o$2
  SCRG_LIB=$root/ab_libs/lib_stats.so python3 $root/scripts/wide_pmc_probe.py $1 $2 > $out/$tag.rounds.log 2> $out/$tag.rounds.err
  export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
    i=$((i+1))
    timeout 180 rocprofv3 --kernel-trace --pmc $set -d $out/$tag/p$i -o pmc --output-format csv -- python3 $root/scripts/wide_pmc_probe.py $1 $2 > $out/$tag.log 2> $out/$tag.err
  done
  unset SCRG_LIB
  python3 - $out/$tag $out/$tag.log $out/$tag.rounds.log <<'PY'
import csv, glob, json, sys, collections
d, log, rlog = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
name = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "genasm_lane" in r.get("Kernel_Name", ""):
            name = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
info = json.loads([l for l in open(log) if l.startswith("{")][-1])
try:
    rinfo = json.loads([l for l in open(rlog) if l.startswith("{")][-1])
    info["window_rounds_per_launch"], info["windows_per_pair"] = rinfo["window_rounds_per_launch"], rinfo["windows_per_pair"]
except Exception:
    pass
c = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
out = dict(info, kernel=name, counters=c)
r = info.get("window_rounds_per_launch")
if r:
    out["valu_instructions_per_window_round"] = c["SQ_INSTS_VALU"] / r
out["valu_instructions_per_pair"] = c["SQ_INSTS_VALU"] * 64 / info["pairs"]
if "TCC_EA0_RDREQ" in c and "TCC_EA0_WRREQ" in c:
    rd = c["TCC_EA0_RDREQ"] * 128.0                     # (MI355X_MICROARCH.md: read requests are 128-byte ones)
    wr = c["TCC_EA0_WRREQ_64B"] * 64.0 + (c["TCC_EA0_WRREQ"] - c["TCC_EA0_WRREQ_64B"]) * 32.0
    out["hbm_read_bytes_per_launch"], out["hbm_write_bytes_per_launch"], out["hbm_bytes_per_launch"] = rd, wr, rd + wr
    out["hbm_GBs_at_kernel_ms"] = (rd + wr) / (min(info["kernel_ms"]) * 1e-3) / 1e9
json.dump(out, open(d + ".json", "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("W", "O", "kernel", "kernel_ms", "windows_per_pair", "valu_instructions_per_window_round", "valu_instructions_per_pair", "hbm_bytes_per_launch", "hbm_GBs_at_kernel_ms")}))
PY
done
