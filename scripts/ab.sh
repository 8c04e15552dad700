#!/bin/bash
# Kernel A/B helper.  Local:  scripts/ab.sh build <name> [file.hip]   -> gpurun_ab/lib_<name>.so
#                     GPU box: scripts/ab.sh run <name>...              -> pairs/s of each build, interleaved twice
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/ab_libs
if [ "$1" = build ]; then
  src=${3:-$root/scrooge_amd/csrc/genasm_kernels.hip}
  cp $src /tmp/ab_$2.hip
  cd $root/scrooge_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$root/scrooge_amd/csrc \
     -o $root/ab_libs/lib_$2.so -x hip /tmp/ab_$2.hip genasm_kernel_multiword.hip seq_kernels.hip -x hip scrg_api.cpp scrg_io.cpp -lpthread 2>&1 | grep -E "error" -A3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$root/scrooge_amd/csrc -c -x hip /tmp/ab_$2.hip -o /tmp/ab.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | grep -A2 "kernelILi8E" | grep -E "VGPRs|Scratch" | sed 's/.*remark: *//;s/\[-Rpass.*//' | tr '\n' ' '; echo " <- $2"
else
  shift
  for rep in 1 2; do for v in "$@"; do
    SCRG_LIB=$root/ab_libs/lib_$v.so python3 $root/bench.py --cpu-seconds 0 --steps 3 ${BENCH_ARGS} | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['value']), round(d['kernel_ms'],3))"
  done; done
fi
