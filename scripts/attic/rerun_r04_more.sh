root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/more_r04; mkdir -p $out
cd $root
python3 scripts/sweep.py $out/knob_sweep.csv 100000 10000 > $out/sweep.log 2>&1
export SCRG_LIB=$root/scrooge_amd/libscrooge_amd.so
for n in 20000 100000; do
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/host_$n -o prof --output-format csv -- \
      python3 $root/scripts/host_timing_probe.py $n 1 6 > $out/host_$n.log 2> $out/host_$n.err)
  tail -3 $out/host_$n.err
done
ls $out/host_20000
