#!/bin/bash
# per-phase shader cycles per window round (bench.py --stats) for A/B builds in ab_libs/; WPC, ABL, ROWS env vars
for v in "$@"; do SCRG_LIB=$PWD/ab_libs/lib_$v.so python3 bench.py --cpu-seconds 0 --stats --steps 2 --waves-per-cu ${WPC:-4} --lds-rows ${ROWS:-13} --pairs 50000 --ablate ${ABL:-0} 2>&1 | grep -v amdgpu | python3 -c "
import sys,json,ast
for l in sys.stdin:
    if l.startswith('stats'): d=ast.literal_eval(l.split(':',1)[1].strip()); print('$v', 'dc/step', round(d['cyc_per_round_dc']/d['steps_per_round']), 'tbloop/macro', round(d['cyc_per_round_tb_loop']/d['macro_per_round']), 'tbloop/round', round(d['cyc_per_round_tb_loop']), 'tb/round', round(d['cyc_per_round_tb']), 'dc/round', round(d['cyc_per_round_dc']), 'setup', round(d['cyc_per_round_setup']), 'fetch', round(d['cyc_per_round_fetch']), 'steps', round(d['steps_per_round'],1), 'macro', round(d['macro_per_round'],1))
"; done
