#!/bin/bash
# usage (GPU box, repo root): scripts/r06_gate_probe.sh
# The decoder's choice made on the device (dec_sample_long): the decoder tests, then scripts/decode_timing.py on
#  - 10 kb streams (the headline's slot; the gate must cost nothing measurable),
#  - 150 bp reads' streams in a buffer sized at 128 bytes per pair: the library's own choice against either kernel forced.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 600 python3 -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "decode or decoders or gather or edit" 2>&1 | tail -3
timeout 300 python3 scripts/decode_timing.py > gpurun_out/r06_gate_10kb.json 2> gpurun_out/r06_gate.err < /dev/null
for k in auto lane quad; do
  if [ $k = auto ]; then unset SCRG_DEC_KERNEL; else export SCRG_DEC_KERNEL=$k; fi
  timeout 300 python3 scripts/decode_timing.py --pairs 2000000 --read-len 150 --profile illumina --slots 1 --buffer-per-pair 128 > gpurun_out/r06_gate_150bp_$k.json 2>> gpurun_out/r06_gate.err < /dev/null
done
unset SCRG_DEC_KERNEL
python3 - <<'PY' < /dev/null
import json
for f in ("10kb", "150bp_auto", "150bp_lane", "150bp_quad"):
    try:
        d = json.load(open("gpurun_out/r06_gate_%s.json" % f))
        print(f, {k: (v if not isinstance(v, dict) else {"decode_ms": round(v["decode_ms"], 4), "count_only_ms": round(v["count_only_ms"], 4)}) for k, v in d.items() if k.startswith("slots_") or k in ("stream_bytes_per_pair", "buffer_bytes_per_pair")})
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 gpurun_out/r06_gate.err
