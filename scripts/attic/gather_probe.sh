#!/bin/bash
# What the gather path costs the receiving rank on ONE GPU (a one-rank RCCL group, SCRG_BENCH_FORCE_GATHER=1), per
# transfer format.  usage: scripts/gather_probe.sh
show() { python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$1', round(j['value']/1e6,2), 'M pairs/s', round(j['ms_per_step'],3), 'ms/step, gather check', j['gather_check'], ', host enqueue', round(j['host_enqueue_ms_per_step'],3), 'ms/step')"; }
python bench.py --cpu-seconds 0 2>/dev/null | show "local (no gather)     "
export SCRG_BENCH_FORCE_GATHER=1
for f in edits edits-from-runs packed runs; do
  python bench.py --cpu-seconds 0 --gather-format $f 2>gpurun_out/gp_err.txt | show "$f      " || tail -5 gpurun_out/gp_err.txt
done
SCRG_BENCH_NOCOLL=1 python bench.py --cpu-seconds 0 --gather-format edits 2>/dev/null | show "edits, no collective "
