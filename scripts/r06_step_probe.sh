# usage (GPU box): what the decode of the N > 1 step costs, piece by piece (one-rank RCCL group, root fixed, one slot)
root=${GRAFT_REPO_ROOT:-$(pwd)}
export SCRG_BENCH_FORCE_GATHER=1
run() { echo "$1: $(env $2 python3 $root/bench.py --no-build --cpu-seconds 0 --pairs 100000 --steps 40 --warmup 8 --other-configs off $3 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step %.3f  without_decode %.3f' % (d['ms_per_step'], 1e5 / ((d.get('gather_without_decode') or {}).get('value', 0) or 1) * 1e3))")"; }
run "depth 4               " "SCRG_X=0" "--gather-depth 4"
run "depth 8               " "SCRG_X=0" "--gather-depth 8"
run "depth 12              " "SCRG_X=0" "--gather-depth 12"
run "depth 8 no collective " "SCRG_BENCH_NOCOLL=1" "--gather-depth 8"
run "depth 8, N=8 rotate   " "SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=8" "--gather-depth 8"
run "depth 16, N=8 rotate  " "SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=8" "--gather-depth 16"
run "depth 8, N=4 rotate   " "SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=4" "--gather-depth 8"
run "depth 8, N=2 rotate   " "SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=2" "--gather-depth 8"
