#!/bin/bash
# usage (GPU box, repo root): scripts/kernel_ab.sh <name>...   — value / sustained / serial of several builds of the library
# (ab_libs/lib_<name>.so, scripts/ab.sh build), interleaved, three passes; then, if ab_libs/lib_stats.so exists, the per-phase
# cycle counters of the align kernel (one stream, 100 k pairs) and its SQ counters (scripts/pmc_run.sh, the shipped library).
for rep in 1 2 3; do for v in "$@"; do
  SCRG_LIB=$PWD/ab_libs/lib_$v.so python3 bench.py --no-build --cpu-seconds 0 --other-configs off --host-api off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v value %.2f M  sustained %.2f M  serial %.2f M (kernel %.3f ms)' % (d['value']/1e6, d['sustained']['value']/1e6, d['serial']['value']/1e6, d['serial']['kernel_ms']))"
done; done
if [ -f ab_libs/lib_stats.so ]; then
  SCRG_LIB=$PWD/ab_libs/lib_stats.so python3 bench.py --no-build --stats --cpu-seconds 0 --steps 3 2>&1 >/dev/null | grep "stats(last launch)" | python3 -c "
import sys, ast
for l in sys.stdin:
    st = ast.literal_eval(l.split('stats(last launch):',1)[1].strip())
    print({k: round(v, 1) if isinstance(v, float) else v for k, v in st.items() if k.startswith('cyc_per_round') or k in ('rounds', 'shader_clock_ghz')})"
fi
