# after a change of the decoder only: the decode and gather parts of scripts/collect_profiles.sh + two passes of the root-load table
tag=r04
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
for part in gather decode; do rm -rf gpurun_out/prof_${tag}_$part; mkdir -p gpurun_out/prof_${tag}_$part; done
(cd /tmp && export TMPDIR=/tmp && export SCRG_BENCH_FORCE_GATHER=1 && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_gather -o prof --output-format csv -- \
    python3 $root/bench.py --no-build --cpu-seconds 0 > $root/gpurun_out/prof_${tag}_gather/bench.json 2> $root/gpurun_out/prof_${tag}_gather/bench.err)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_decode -o prof --output-format csv -- \
    python3 $root/scripts/decode_timing.py > $root/gpurun_out/prof_${tag}_decode/decode_timing.json 2> $root/gpurun_out/prof_${tag}_decode/err.txt)
unset SCRG_LIB
mkdir -p gpurun_out/more_r04
for pass in 1 2; do scripts/root_load_probe.sh > gpurun_out/more_r04/root_load_$pass.jsonl 2> gpurun_out/more_r04/root_load.err; cat gpurun_out/more_r04/root_load_$pass.jsonl | python3 -c "
import json,sys
print([(json.loads(l)['simulated_ranks'], round(json.loads(l)['root_ms_per_step'],2)) for l in sys.stdin if l.startswith('{')])"; done
cat gpurun_out/prof_${tag}_decode/decode_timing.json | cut -c1-900
