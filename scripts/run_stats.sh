for v in "$@"; do SCRG_LIB=$PWD/ab_libs/lib_$v.so python3 bench.py --cpu-seconds 0 --stats --steps 2 --waves-per-cu ${WPC:-4} --pairs 50000 --ablate ${ABL:-0} 2>&1 | grep -v amdgpu | python3 -c "
import sys,json,ast
for l in sys.stdin:
    if l.startswith('stats'): d=ast.literal_eval(l.split(':',1)[1].strip()); print('$v', 'dc/step', round(d['cyc_per_round_dc']/d['steps_per_round']), 'tb/macro', round(d['cyc_per_round_tb']/d['macro_per_round']), 'tbloop/round', round(d['cyc_per_round_tb_loop']), 'tb/round', round(d['cyc_per_round_tb']), 'setup', round(d['cyc_per_round_setup']), 'steps', round(d['steps_per_round'],1))
"; done
