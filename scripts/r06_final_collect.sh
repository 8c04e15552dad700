#!/bin/bash
# usage (GPU box, repo root; ab_libs/lib_stats.so built first: scripts/ab.sh build stats -DSCRG_STATS): scripts/r06_final_collect.sh
# Everything profiles/r06_* is made from, on ONE box and the shipped library: the GPU test suite, scripts/collect_profiles.sh r06
# (kernel statistics of the bench's legs, PMC passes), scripts/headline_trace.sh, scripts/root_load_probe.sh and
# scripts/r06_chain_probe.sh.  Afterwards, in the container: python scripts/make_profiles.py r06 r06; python scripts/mix_roof.py >
# profiles/r06_mix_roof.json; the trace, root-load and chain-probe files are copied from gpurun_out/ (profiles/README.md).
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
scripts/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
scripts/headline_trace.sh > gpurun_out/headline_trace.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
scripts/root_load_probe.sh > gpurun_out/r06_root_load_final.jsonl 2>/dev/null
scripts/r06_chain_probe.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_probe_final.txt
cat gpurun_out/r06_chain_probe_final.txt | tail -4
cat gpurun_out/headline_trace/summary.json | head -20
