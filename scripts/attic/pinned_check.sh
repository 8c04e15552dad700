./scripts/ubench/d2h_rate | grep -v "^ *[0-9.]* MB ->"
python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "grow_while or host_pipeline or config3_full" 2>&1 | tail -3
for n in 100000 20000; do python scripts/host_timing_probe.py $n 1 4 2>&1 | grep "^call" ; done
python scripts/host_timing_probe.py 100000 0 3 2>&1 | grep "^call"
