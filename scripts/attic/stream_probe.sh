#!/bin/bash
# Pipeline depth / stream priorities / hardware queues: headline rate per combination.  usage: scripts/stream_probe.sh
show() { python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$1', round(j['value']/1e6,2), 'M pairs/s', round(j['ms_per_step'],3), 'ms/step')"; }
run() { # name, streams, prios, max_hw_queues
  if [ -n "$4" ]; then export GPU_MAX_HW_QUEUES=$4; else unset GPU_MAX_HW_QUEUES; fi
  SCRG_BENCH_PRIOS=$3 python bench.py --cpu-seconds 0 --streams $2 --no-build 2>/dev/null | show "$1"
}
run "4 streams 1,-1,0,1 (default)       " 4 1,-1,0,1
run "4 streams 0,0,0,0                  " 4 0,0,0,0
run "4 streams 0,0,0,0 hwq 8            " 4 0,0,0,0 8
run "4 streams 1,-1,0,1 hwq 8           " 4 1,-1,0,1 8
run "6 streams 1,-1,0,1,-1,0 hwq 8      " 6 1,-1,0,1,-1,0 8
run "6 streams 0 x6 hwq 8               " 6 0,0,0,0,0,0 8
run "5 streams 1,-1,0,1,-1 hwq 8        " 5 1,-1,0,1,-1 8
run "8 streams 0 x8 hwq 8               " 8 0,0,0,0,0,0,0,0 8
run "6 streams 1,-1,0,1,-1,0            " 6 1,-1,0,1,-1,0
run "3 streams 1,-1,0                   " 3 1,-1,0
