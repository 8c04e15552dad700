# the last collection pass of round 4 (after the decoder change): everything of collect_profiles.sh + the root-load table
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
scripts/collect_profiles.sh r04 2>&1 | tail -3
scripts/root_load_probe.sh > gpurun_out/more_r04/root_load.jsonl 2> gpurun_out/more_r04/root_load.err
cat gpurun_out/more_r04/root_load.jsonl
