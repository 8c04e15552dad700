#!/bin/bash
# usage (GPU box, repo root): scripts/r06_gate_sweep.sh [short]  > lines of JSON
# Where short streams end: scripts/decode_timing.py over read lengths, one launch each, with the decoder forced
# (SCRG_DEC_KERNEL=plain: the one-pair-per-lane loop without staging; quad; lane; lane + SCRG_DEC_SORT=1: longest stream first,
# what the lane kernel did for every launch before this sweep) and left to the
# library (auto).  `short`: the lengths around DEC_PLAIN_BELOW only.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
if [ "$1" = short ]; then
  specs=("150 2000000 illumina" "150 2000000 ont" "200 1500000 ont" "250 1200000 ont" "300 1000000 ont" "1000 300000 illumina")
else
  specs=("150 2000000 illumina" "300 1000000 ont" "500 600000 ont" "750 400000 ont" "1000 300000 ont" "1500 200000 ont" "2000 150000 ont" "4000 75000 ont" "10000 100000 ont")
fi
for spec in "${specs[@]}"; do
  set -- $spec
  for k in auto plain quad lane lane-sorted; do
    unset SCRG_DEC_KERNEL SCRG_DEC_SORT
    case $k in auto) ;; lane-sorted) export SCRG_DEC_KERNEL=lane SCRG_DEC_SORT=1 ;; *) export SCRG_DEC_KERNEL=$k ;; esac
    timeout 200 python3 scripts/decode_timing.py --pairs $2 --read-len $1 --profile $3 --slots 1 --busy 20 2>/dev/null < /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'read_len': $1, 'profile': '$3', 'pairs': $2, 'decoder': '$k', 'stream_bytes_per_pair': round(d['stream_bytes_per_pair'], 1), 'runs_per_pair': round(d['runs_per_pair'], 1), 'decode_ms': round(d['slots_1']['decode_ms'], 4), 'count_only_ms': round(d['slots_1']['count_only_ms'], 4)}))"
  done
done
