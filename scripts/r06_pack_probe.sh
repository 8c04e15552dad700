#!/bin/bash
# usage (GPU box, repo root): scripts/r06_pack_probe.sh
# The packer's tests, bench.py's headline and pack-inclusive rates (value_incl_pack) on the built library, and the packer's own
# kernel time (kernel trace of the same command).  Every step under its own timeout.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 300 python3 -m pytest tests/test_gpu_scale.py -x -q -m gpu -rs -k "group_packer" 2>&1 | tail -3
timeout 300 python3 bench.py --no-build --steps 20 --warmup 5 --cpu-seconds 1 --other-configs off --host-api off 2>/dev/null | tail -1 > gpurun_out/r06_pack_bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/r06_pack_bench.json'))
print({k: d[k] for k in d if 'value' in k or k=='ms_per_step'})" < /dev/null
rm -rf gpurun_out/prof_pack
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_pack -o pack --output-format csv -- \
   python3 $root/bench.py --no-build --steps 20 --warmup 5 --cpu-seconds 1 --other-configs off --host-api off > /dev/null 2>&1 < /dev/null)
f=$(find gpurun_out/prof_pack -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r06_kernel_stats_incl_pack.csv; head -8 "$f" | cut -c1-160; fi
find gpurun_out/prof_pack -type f ! -name '*stats.csv' -delete
