#!/bin/bash
# usage (GPU box, repo root): scripts/root_load_probe.sh
# What the ROOT of an N-GPU job does per step, on one GPU: the N > 1 step of bench.py on a one-rank RCCL group
# (SCRG_BENCH_FORCE_GATHER=1: align 125 k pairs as edit streams, compaction, the gather onto itself) with the root's decode
# launch covering N slots (SCRG_GATHER_SIMULATE_SENDERS=N).  ms_per_step is the step time of rank 0; the other ranks only
# align, so the job runs at N x pairs / that time as long as the links keep up (51 GB/s of 77 per link).
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
export SCRG_BENCH_FORCE_GATHER=1
for n in 1 2 4 8; do
  pairs=100000; [ $n = 8 ] && pairs=125000
  SCRG_GATHER_SIMULATE_SENDERS=$n python3 $root/bench.py --no-build --cpu-seconds 0 --pairs $pairs --other-configs off 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
n, pairs = $n, $pairs
print(json.dumps({'simulated_ranks': n, 'pairs_per_rank': pairs, 'root_ms_per_step': d['ms_per_step'], 'projected_job_M_pairs_per_s': n * pairs / d['ms_per_step'] / 1e3,
                  'without_decode_M_pairs_per_s_per_rank': (d.get('gather_without_decode') or {}).get('value', 0) / 1e6,
                  'scaling_efficiency_vs_single_gpu_52M': n * pairs / d['ms_per_step'] / 1e3 / (n * 52.0)}))"
done
