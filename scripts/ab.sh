#!/bin/bash
# Kernel A/B helper.  Local:   scripts/ab.sh build <name> [extra hipcc flags...]   -> ab_libs/lib_<name>.so  (the whole library, current sources)
#                     GPU box: scripts/ab.sh run <name>...                          -> pairs/s of each build, interleaved three times
# Builds are picked up through SCRG_LIB (scrooge_amd/api.py); BENCH_ARGS adds bench.py arguments.
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/ab_libs
if [ "$1" = build ]; then
  name=$2; shift; shift
  cd $root/scrooge_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -shared \
     -o $root/ab_libs/lib_$name.so genasm_kernels.hip genasm_kernel_multiword.hip genasm_lane_kernel.hip genasm_lane_mw_kernel.hip genasm_lane_wide_kernel.hip genasm_lane_parts_kernel.hip edit_stream_kernels.hip edit_stream_decode_kernel.hip host_path_kernels.hip \
     seq_kernels.hip scrg_api.cpp scrg_host.cpp scrg_io.cpp -lpthread 2>&1 | grep -E "error" -A3
  ls -la $root/ab_libs/lib_$name.so
else
  shift
  for rep in 1 2 3; do for v in "$@"; do
    SCRG_LIB=$root/ab_libs/lib_$v.so python3 $root/bench.py --no-build --cpu-seconds 0 --steps 20 ${BENCH_ARGS} | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'] / 1e6, 2), 'M pairs/s, kernel', round(d['kernel_ms'], 3), 'ms')"
  done; done
fi
