#!/bin/bash
# usage (GPU box, repo root): scripts/root_load_probe.sh
# What the ROOT of an N-GPU job does per step, on one GPU: the N > 1 step of bench.py on a one-rank RCCL group
# (SCRG_BENCH_FORCE_GATHER=1: align 125 k pairs as edit streams, compaction, the gather onto itself) with the root's decode
# launch covering N slots (SCRG_GATHER_SIMULATE_SENDERS=N).  ms_per_step is the step time of rank 0; the other ranks only
# align, so the job runs at N x pairs / that time as long as the links keep up (51 GB/s of 77 per link).
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
# With a ROTATING root (bench.py --gather-root rotate) every rank is the root of one step in N: SCRG_GATHER_SIMULATE_ROTATE=1 makes
# the one rank decode its N slots only every N-th step — ms_per_step is then the step of EVERY rank, and the job runs at N x pairs /
# that time.  steps = a multiple of N (40), so that every timed region holds whole rotations.
export SCRG_BENCH_FORCE_GATHER=1
for rot in 0 1; do
for n in 1 2 4 8; do
  [ $rot = 1 ] && [ $n = 1 ] && continue
  pairs=100000; [ $n = 8 ] && pairs=125000
  SCRG_GATHER_SIMULATE_ROTATE=$rot SCRG_GATHER_SIMULATE_SENDERS=$n python3 $root/bench.py --no-build --cpu-seconds 0 --pairs $pairs --steps 40 --warmup 8 --other-configs off 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
n, pairs, rot = $n, $pairs, $rot
print(json.dumps({'root': 'rotates (every rank decodes one step in N)' if rot else 'rank 0 (decodes every step)', 'simulated_ranks': n, 'pairs_per_rank': pairs,
                  ('every_ranks_ms_per_step' if rot else 'root_ms_per_step'): d['ms_per_step'], 'projected_job_M_pairs_per_s': n * pairs / d['ms_per_step'] / 1e3,
                  'without_decode_M_pairs_per_s_per_rank': (d.get('gather_without_decode') or {}).get('value', 0) / 1e6}))"
done; done
