#!/bin/bash
# usage (GPU box, repo root): scripts/collect_more.sh <tag>  -> gpurun_out/more_<tag>/
# The measurements of a round that are not part of scripts/collect_profiles.sh: the knob sweep (reference's CSV columns), what
# the ROOT of an N-GPU job does per step (scripts/root_load_probe.sh), the other one-pair-per-lane kernels' counters
# (scripts/wide_pmc.sh), the table-in-parts kernel next to the kernel it replaces, the split kernel's launch times, and a
# trace of the host entry points (rocprofv3 --kernel-trace --memory-copy-trace --stats of one scrg_align_pairs call series).
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/more_$tag; mkdir -p $out
cd $root
python3 -c "import scrooge_amd; scrooge_amd.build_library()" || exit 1
python3 scripts/sweep.py $out/knob_sweep.csv 100000 10000 > $out/sweep.log 2>&1
scripts/root_load_probe.sh > $out/root_load.jsonl 2> $out/root_load.err
python3 scripts/parts_sweep.py > $out/parts_sweep.txt 2>&1
python3 scripts/split_time.py 25000 > $out/split_time.txt 2>&1; python3 scripts/split_time.py 50000 >> $out/split_time.txt 2>&1; python3 scripts/split_time.py 100000 >> $out/split_time.txt 2>&1
scripts/wide_pmc.sh > $out/wide_pmc.txt 2>&1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
for n in 20000 100000; do
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/host_$n -o prof --output-format csv -- \
      python3 $root/scripts/host_timing_probe.py $n 1 6 > $out/host_$n.log 2> $out/host_$n.err)
done
ls $out
