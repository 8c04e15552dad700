#!/bin/bash
# usage (GPU box, repo root; needs ab_libs/lib_ablate.so = scripts/ab.sh build ablate -DSCRG_ABLATE):  scripts/ablate_probe.sh
# Kernel time and per-phase cycles of the align kernel (one stream, 100 k x 10 kb pairs) with parts switched off — results are
# wrong by design: 0 nothing off, 4 no second traceback pass (no runs, hence no flush), 16 no stores, 8 no walk, 2 no table.
for abl in 0 16 4 12 2; do
  SCRG_LIB=$PWD/ab_libs/lib_ablate.so python3 bench.py --no-build --stats --ablate $abl --cpu-seconds 0 --steps 3 --other-configs off --host-api off 2> /tmp/abl.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ablate $abl kernel_ms %.3f' % d['kernel_ms'])"
  grep "stats(last launch)" /tmp/abl.err | python3 -c "
import sys, ast
for l in sys.stdin:
    st = ast.literal_eval(l.split('stats(last launch):',1)[1].strip())
    print('   ', {k.replace('cyc_per_round_',''): round(v) for k, v in st.items() if k.startswith('cyc_per_round')})"
done
