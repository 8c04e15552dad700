# value / sustained by pipeline depth (bench.py --streams), two passes
for rep in 1 2; do for s in 2 3 4 5 6; do
  python bench.py --no-build --cpu-seconds 0 --other-configs off --host-api off --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('streams $s value %.2f M  sustained %.2f M  serial %.2f M' % (d['value']/1e6, d['sustained']['value']/1e6, d['serial']['value']/1e6))"
done; done
