#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_run.sh <tag> "<bench args>" "<counters pass 1>" ["<counters pass 2>" ...]
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains than --kernel-trace).
# The library is built BEFORE profiling and pinned with SCRG_LIB, so that nothing under rocprofv3 ever forks a
# compiler: the profiler's preloaded library initialises the GPU, and an exec from such a process takes the box down.
tag=$1; shift
bargs=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
export SCRG_BENCH_NO_STATS_LAUNCH=1      # (the counters launch of bench.py runs extra timing code: keep it out of the averages)
cd /tmp && export TMPDIR=/tmp
mkdir -p $root/gpurun_out/pmc_${tag}
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $set -d $root/gpurun_out/pmc_${tag}/p$i -o pmc --output-format csv -- python3 $root/bench.py --no-build --cpu-seconds 0 --steps 2 --warmup 0 $bargs > $root/gpurun_out/pmc_${tag}/p$i.log 2>&1
done
python3 $root/scripts/pmc_summary.py $root/gpurun_out/pmc_${tag}
