for mode in direct staged direct staged; do
  if [ $mode = staged ]; then export SCRG_HOST_STAGED=1; else unset SCRG_HOST_STAGED; fi
  python bench.py --no-build --cpu-seconds 3 > gpurun_out/db_$mode.json 2>/dev/null
  python - $mode <<'P'
import json,sys
d=json.load(open('gpurun_out/db_%s.json' % sys.argv[1]))
h=d['host_api']
out=[sys.argv[1]]
for sz in ('100000_pairs','20000_pairs'):
    for k,v in h['pairwise'][sz].items():
        out.append('%s/%s %.1f|%.1f' % (sz[:3],k[:6],v['steady_best_s']*1e3,v['steady_median_s']*1e3))
for k,v in h['read_mapping_configs2'].items():
    if isinstance(v,dict): out.append('map/%s %.1f|%.1f' % (k[:14],v['steady_best_s']*1e3,v['steady_median_s']*1e3))
print('  '.join(out))
P
done
