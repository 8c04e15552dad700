#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/collect_profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/
# 2. PMC passes (one counter set per pass, kernel trace only) of the same workload, one stream:
#      HBM traffic  -> gpurun_out/pmc_<tag>_hbm/summary.json
#      SQ counters  -> gpurun_out/pmc_<tag>_sq/summary.json
# The library AND the oracle (the CPU leg of the default bench command) are built first, the library is pinned with
# SCRG_LIB, and bench.py --no-build loads both without ever forking a compiler (oracle/pyoracle.py: allow_compile=False);
# every profiler run has its own time limit.
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library(); from oracle.pyoracle import build; build()" || exit 1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
mkdir -p $root/gpurun_out/prof_$tag
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o prof --output-format csv -- \
    python3 $root/bench.py --no-build > $root/gpurun_out/prof_$tag/bench.json 2> $root/gpurun_out/prof_$tag/bench.err)
tail -c 400 $root/gpurun_out/prof_$tag/bench.json; echo
mkdir -p $root/gpurun_out/prof_${tag}_serial
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_serial -o prof --output-format csv -- \
    python3 $root/bench.py --no-build --serial --cpu-seconds 0 > $root/gpurun_out/prof_${tag}_serial/bench.json 2> $root/gpurun_out/prof_${tag}_serial/bench.err)
# the N > 1 step on this one GPU: the gather path on a one-rank RCCL group, CIGARs as edit streams written by the kernel
mkdir -p $root/gpurun_out/prof_${tag}_gather
(cd /tmp && export TMPDIR=/tmp && export SCRG_BENCH_FORCE_GATHER=1 && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_gather -o prof --output-format csv -- \
    python3 $root/bench.py --no-build --cpu-seconds 0 > $root/gpurun_out/prof_${tag}_gather/bench.json 2> $root/gpurun_out/prof_${tag}_gather/bench.err)
# the receiving side: scrg_decode_edit_stream, one slot alone and eight slots in one launch
mkdir -p $root/gpurun_out/prof_${tag}_decode
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_decode -o prof --output-format csv -- \
    python3 $root/scripts/decode_timing.py > $root/gpurun_out/prof_${tag}_decode/decode_timing.json 2> $root/gpurun_out/prof_${tag}_decode/err.txt)
cd $root
# window rounds per launch: the kernel's own counters exist in a -DSCRG_STATS build only (scripts/ab.sh build stats -DSCRG_STATS,
# built in the container: it travels with the snapshot); not under the profiler
if [ -f ab_libs/lib_stats.so ]; then
  SCRG_LIB=$root/ab_libs/lib_stats.so python3 bench.py --no-build --stats --cpu-seconds 0 --steps 2 2> gpurun_out/prof_$tag/stats.txt > /dev/null
else
  echo "no ab_libs/lib_stats.so: run scripts/ab.sh build stats -DSCRG_STATS before gpurun" >&2
fi
scripts/pmc_run.sh ${tag}_hbm "--serial" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ TCC_READ" | grep -v dispatches
scripts/pmc_run.sh ${tag}_sq "--serial" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" | grep -v dispatches
