for pol in default passive; do
  if [ $pol = passive ]; then export OMP_WAIT_POLICY=passive; fi
  python bench.py --no-build --other-configs off --host-api on --sustained-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); h=j['host_api']['pairwise']
for k in ('100000_pairs','20000_pairs'):
    print('$pol', k, {v: round(h[k][v]['steady_best_s']*1e3,2) for v in h[k]})"
done
