"""stdin: output of `bench.py --stats`; prints the per-round cycle counters and the headline numbers on two lines."""
import sys, ast, json
for l in sys.stdin:
    if l.startswith("stats"):
        d = ast.literal_eval(l.split(":", 1)[1].strip())
        print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()
               if k.startswith("cyc_per") or k.startswith("diag") or k in ("rounds", "steps_per_round", "macro_per_round")})
    elif l.startswith("{"):
        j = json.loads(l)
        print("rows", j["config"]["lds_rows"], "waves", j["config"]["launch"]["n_waves"], "kernel_ms", round(j["kernel_ms"], 3),
              "kernel pairs/s", round(j["kernel_pairs_per_s_per_gpu"]), "value", round(j["value"]))
