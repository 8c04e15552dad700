#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
export SCRG_BENCH_FORCE_GATHER=1 SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=8 SCRG_BENCH_NO_STATS_LAUNCH=1
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_rot; mkdir -p $root/gpurun_out/prof_rot
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_rot -o rot --output-format csv -- python3 $root/bench.py --no-build --cpu-seconds 0 --pairs 125000 --steps 40 --warmup 8 --other-configs off > $root/gpurun_out/prof_rot/log.txt 2>&1
head -30 $root/gpurun_out/prof_rot/*kernel_stats.csv | cut -c1-150
