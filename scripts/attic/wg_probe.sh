#!/bin/bash
# Wavefronts per workgroup of genasm_lane_kernel (4 = default, flag 128 = 2, flag 64 = 1): pipelined, one stream, one
# launch that fills the GPU.  usage: scripts/wg_probe.sh
show() { python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$1', round(j['value']/1e6,2), 'M pairs/s', round(j['ms_per_step'],3), 'ms/step, kernel', round(j['kernel_ms'],3), 'ms')"; }
for ab in 0 128 64; do
export SCRG_BENCH_DEBUG_FLAGS=$ab
python bench.py --cpu-seconds 0 --no-build 2>/dev/null | show "pipelined  flag $ab"
python bench.py --cpu-seconds 0 --no-build --serial 2>/dev/null | show "one stream flag $ab"
python bench.py --cpu-seconds 0 --no-build --serial --pairs 262144 --steps 6 2>/dev/null | show "262144/launch flag $ab"
done
