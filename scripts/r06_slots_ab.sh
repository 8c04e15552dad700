# usage (GPU box): scripts/r06_slots_ab.sh   — the N > 1 step with 16-bit against 32-bit ring slots in decode_edits_quad_kernel.
# ab_libs/lib_slots32.so = the library with edit_stream_decode_kernel.hip as of commit 5f3bd08~1 ("Quad decoder: every run written
# once ...": 32-bit slots, 17 KB of LDS per workgroup), built in the container with scripts/ab.sh-style hipcc; result in
# profiles/r06_root_load.json ("ring_slots_ab"): no difference in the step.
root=${GRAFT_REPO_ROOT:-$(pwd)}
export SCRG_BENCH_FORCE_GATHER=1
run() { echo "$1: $(env $2 python3 $root/bench.py --no-build --cpu-seconds 0 --pairs 100000 --steps 40 --warmup 8 --other-configs off $3 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step %.3f' % d['ms_per_step'])")"; }
for i in 1 2; do
run "16-bit slots N=1 fixed " "SCRG_X=0" ""
run "32-bit slots N=1 fixed " "SCRG_LIB=$root/ab_libs/lib_slots32.so" ""
run "16-bit slots N=8 rotate" "SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=8" ""
run "32-bit slots N=8 rotate" "SCRG_LIB=$root/ab_libs/lib_slots32.so SCRG_GATHER_SIMULATE_ROTATE=1 SCRG_GATHER_SIMULATE_SENDERS=8" ""
run "16-bit slots N=8 fixed " "SCRG_GATHER_SIMULATE_SENDERS=8" ""
run "32-bit slots N=8 fixed " "SCRG_LIB=$root/ab_libs/lib_slots32.so SCRG_GATHER_SIMULATE_SENDERS=8" ""
done
