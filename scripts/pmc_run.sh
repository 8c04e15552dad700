#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_run.sh <tag> "<bench args>" "<counters pass 1>" ["<counters pass 2>" ...]
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains than --kernel-trace).
tag=$1; shift
bargs=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $root/gpurun_out/pmc_${tag}/p$i -o pmc --output-format csv -- python3 $root/bench.py --cpu-seconds 0 --steps 2 --warmup 0 $bargs > $root/gpurun_out/pmc_${tag}/p$i.log 2>&1
done
python3 $root/scripts/pmc_summary.py $root/gpurun_out/pmc_${tag}
