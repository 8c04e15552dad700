#!/bin/bash
# usage (GPU box, repo root): scripts/root_load_probe.sh  > lines of JSON
# What a rank of an N-GPU job does per step, on ONE GPU: the N > 1 step of bench.py on a one-rank RCCL group
# (SCRG_BENCH_FORCE_GATHER=1: align PAIRS pairs as edit streams, compaction into the step's slot, the gather — which on one rank
# has nothing to transfer — and the root's decode launch covering N slots, SCRG_GATHER_SIMULATE_SENDERS=N).  The same workload
# size for every N (round 5 mixed 100 k and 125 k).  The first line is the reference of the same box and minute: bench.py's plain
# N = 1 step (align + compaction into a dense run array, no gather), same number of steps.
# Root fixed: ms_per_step is rank 0's step, the other ranks only align, so the job runs at N x pairs / that time as long as the
# links keep up.  Root rotating (SCRG_GATHER_SIMULATE_ROTATE=1, bench.py's default): the one rank decodes its N slots only every
# N-th step — ms_per_step is then the step of EVERY rank.  steps = a multiple of N (40): every timed region holds whole rotations.
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
pairs=${PAIRS:-100000}
python3 $root/bench.py --no-build --cpu-seconds 0 --pairs $pairs --steps 40 --warmup 8 --other-configs off --host-api off --sustained-steps 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'reference': 'plain N = 1 step of this box (no gather)', 'pairs': $pairs, 'ms_per_step': d['ms_per_step'], 'M_pairs_per_s': d['value'] / 1e6}))"
export SCRG_BENCH_FORCE_GATHER=1
for rot in 0 1; do
for n in 1 2 4 8; do
  [ $rot = 1 ] && [ $n = 1 ] && continue
  SCRG_GATHER_SIMULATE_ROTATE=$rot SCRG_GATHER_SIMULATE_SENDERS=$n python3 $root/bench.py --no-build --cpu-seconds 0 --pairs $pairs --steps 40 --warmup 8 --other-configs off 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
n, pairs, rot = $n, $pairs, $rot
print(json.dumps({'root': 'rotates (every rank decodes one step in N)' if rot else 'rank 0 (decodes every step)', 'simulated_ranks': n, 'pairs_per_rank': pairs,
                  ('every_ranks_ms_per_step' if rot else 'root_ms_per_step'): d['ms_per_step'], 'projected_job_M_pairs_per_s': n * pairs / d['ms_per_step'] / 1e3,
                  'without_decode_M_pairs_per_s_per_rank': (d.get('gather_without_decode') or {}).get('value', 0) / 1e6}))"
done; done
