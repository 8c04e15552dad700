#!/bin/bash
# usage (GPU box, repo root): scripts/r06_chain_probe.sh
# What the scan + compaction behind every align launch cost the 20-step timed region: the headline as it is against the same
# steps without them (SCRG_BENCH_PROBE=no-chain: the results stay in the pairs' slices), without the scan (no-scan: the offsets of
# the same batch from an earlier step) and without the compaction (no-compact); nothing is checked in these; interleaved.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for v in chain no-chain no-scan no-compact; do
  if [ $v = chain ]; then unset SCRG_BENCH_PROBE; else export SCRG_BENCH_PROBE=$v; fi
  python3 bench.py --no-build --headline-only --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', 'value', round(d['value']/1e6,2), 'ms_per_step', round(d['ms_per_step'],3))"
done; done
