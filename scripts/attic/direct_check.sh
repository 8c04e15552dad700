python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "written_in_place or host_pipeline or config3_full" 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for n in 100000 20000; do for o in 0 1; do
 python scripts/host_timing_probe.py $n $o 5 2>&1 | grep "^call" ; done; done
echo "--- staged"
for n in 100000 20000; do SCRG_HOST_STAGED=1 python scripts/host_timing_probe.py $n 1 5 2>&1 | grep "^call"; done
