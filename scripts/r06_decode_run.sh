# usage (GPU box): scripts/r06_decode_run.sh  — the decoder tests, then scripts/decode_timing.py under SCRG_DEC_KERNEL=quad|wave|lane
# (10 kb reads, one and eight slots) and quad|wave on 1 kb reads -> gpurun_out/r06_dec_timing_*.json (scripts/r06_make_decode_profile.py
# turns them into profiles/r06_decode_timing.json)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "edit_stream_round_trip or decode_large_launch or decode_two_runs or test_encode" 2>&1 | tail -15 > gpurun_out/r06_dec_tests.txt
for k in quad wave lane; do
  SCRG_DEC_KERNEL=$k python3 scripts/decode_timing.py > gpurun_out/r06_dec_timing_$k.json 2> gpurun_out/r06_dec_timing_$k.err
done
SCRG_DEC_KERNEL=quad python3 scripts/decode_timing.py --read-len 1000 --slots 4 > gpurun_out/r06_dec_timing_quad_1k.json 2>> gpurun_out/r06_dec_timing_quad.err
SCRG_DEC_KERNEL=wave python3 scripts/decode_timing.py --read-len 1000 --slots 4 > gpurun_out/r06_dec_timing_wave_1k.json 2>> gpurun_out/r06_dec_timing_wave.err
tail -5 gpurun_out/r06_dec_tests.txt; cat gpurun_out/r06_dec_timing_*.json
