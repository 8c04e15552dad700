# decode time by number of slots for several builds: scripts/attic/dec_ab_slots.sh name...
for v in "$@"; do for sl in 2 4 8; do
  SCRG_LIB=$PWD/ab_libs/lib_$v.so python scripts/decode_timing.py --slots $sl --reps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', {k:(round(v['count_only_ms'],3),round(v['decode_ms'],3)) for k,v in d.items() if k.startswith('slots_')})"
done; done
