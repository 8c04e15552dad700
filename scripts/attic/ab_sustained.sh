# value / sustained / serial of several library builds (ab_libs/lib_<name>.so), interleaved, two passes: scripts/attic/ab_sustained.sh name...
for rep in 1 2; do for v in "$@"; do
  SCRG_LIB=$PWD/ab_libs/lib_$v.so python bench.py --no-build --cpu-seconds 0 --other-configs off --host-api off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v value %.2f M  sustained %.2f M  serial %.2f M (kernel %.3f ms)' % (d['value']/1e6, d['sustained']['value']/1e6, d['serial']['value']/1e6, d['serial']['kernel_ms']))"
done; done
