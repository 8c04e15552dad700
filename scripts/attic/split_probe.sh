#!/bin/bash
# The default kernel as one wavefront per 64 pairs (reserved[0] = 1024) and with a window's work split over two (512), and the
# library's own choice (0), on the bench workload: pipelined value, sustained, one launch at a time.  usage (GPU box): scripts/split_probe.sh
for f in 1024 512 0; do
  SCRG_BENCH_DEBUG_FLAGS=$f python bench.py --no-build --other-configs off --host-api off --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('flags $f', 'value %.2f M' % (j['value']/1e6), 'sustained %.2f M' % (j['sustained']['value']/1e6), 'serial %.2f M (kernel %.3f ms)' % (j['serial']['value']/1e6, j['serial']['kernel_ms']), 'edits step %.2f M' % (j['edit_stream_step']['value']/1e6))"
done
