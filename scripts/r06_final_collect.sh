#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
scripts/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
scripts/headline_trace.sh > gpurun_out/headline_trace.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
scripts/root_load_probe.sh > gpurun_out/r06_root_load_final.jsonl 2>/dev/null
scripts/r06_chain_probe.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_probe_final.txt
cat gpurun_out/r06_chain_probe_final.txt | tail -4
cat gpurun_out/headline_trace/summary.json | head -20
