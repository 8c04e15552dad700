cd $GRAFT_REPO_ROOT
for P in 1 2 4 8 16 3; do
  echo "probe $P" >> gpurun_out/r06_dec_probe.txt
  SCRG_LIB=$GRAFT_REPO_ROOT/ab_libs/lib_qdprobe_$P.so SCRG_DEC_NOCHECK=1 SCRG_DEC_KERNEL=quad python3 scripts/decode_timing.py --slots 2 >> gpurun_out/r06_dec_probe.txt 2>&1
  SCRG_LIB=$GRAFT_REPO_ROOT/ab_libs/lib_qdprobe_$P.so SCRG_DEC_NOCHECK=1 SCRG_DEC_KERNEL=quad python3 scripts/decode_timing.py --slots 2 --read-len 1000 >> gpurun_out/r06_dec_probe.txt 2>&1
done
echo "as is" >> gpurun_out/r06_dec_probe.txt
SCRG_DEC_KERNEL=quad python3 scripts/decode_timing.py --slots 2 >> gpurun_out/r06_dec_probe.txt 2>&1
SCRG_DEC_KERNEL=quad python3 scripts/decode_timing.py --slots 2 --read-len 1000 >> gpurun_out/r06_dec_probe.txt 2>&1
