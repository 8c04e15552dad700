#!/bin/bash
# Per-phase shader cycles per window round and wavefront life times of the align kernel (bench.py --stats, one stream):
#   scripts/run_stats.sh [pairs, default 262144 = 16 wavefronts per CU]
n=${1:-262144}
python3 bench.py --no-build --stats --pairs $n --cpu-seconds 0 --steps 3 2> /tmp/scrg_stats.txt | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('kernel_ms', j['kernel_ms'])"
python3 scripts/wave_life.py /tmp/scrg_stats.txt $(( (n + 63) / 64 < 4096 ? (n + 63) / 64 : 4096 ))
