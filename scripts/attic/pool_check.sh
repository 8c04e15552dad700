python -m pytest tests -m gpu -x -q -k "host or pairs or mapping or multi" 2>&1 | tail -3
python tests/tools/host_stress.py 2>&1 | tail -3
python bench.py --no-build --cpu-seconds 0 --other-configs off > gpurun_out/pool_bench.json 2> gpurun_out/pool_bench.err; echo bench rc=$?
python - <<'P'
import json
d=json.load(open('gpurun_out/pool_bench.json'))
for sz in ('100000_pairs','20000_pairs'):
    h=d['host_api']['pairwise'][sz]
    print(sz,{k:(round(v['first_call_s']*1e3,1),round(v['steady_best_s']*1e3,1),round(v['steady_median_s']*1e3,1)) for k,v in h.items()})
P
