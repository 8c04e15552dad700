mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-build > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r4f/bench.err
python - <<'P'
import json
j=json.loads(open("gpurun_out/r4f/bench.json").readline())
print(j["value"], j["kernel_ms"], j["serial"]["value"], j["sustained"]["value"], j["parity"]["runs_bit_exact"], len(j["parity"]["other_batches"]))
print({k: (round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].items() if k in ("achieved","frac","frac_at_step_rate","frac_sustained","window_rounds_per_launch","window_rounds_source")})
h=j["host_api"]
for k in ("100000_pairs","20000_pairs"): print(k, {v: round(h["pairwise"][k][v]["steady_best_s"]*1e3,2) for v in h["pairwise"][k]})
m=h["read_mapping_configs2"]; print({k: round(m[k]["steady_best_s"]*1e3,2) for k in m if isinstance(m[k],dict)})
for oc in j["other_configs"]: print(oc["workload"][:60], round(oc["value"]/1e6,2), oc["parity"]["runs_bit_exact"])
P
