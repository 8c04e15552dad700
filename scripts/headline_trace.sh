#!/bin/bash
# usage (GPU box, repo root): scripts/headline_trace.sh [steps=20]
# The step rate of the headline from a KERNEL TRACE, not from the bench's own clock: rocprofv3 --kernel-trace (no counters) of
# `bench.py --headline-only` — set-up, warm-up and the timed region, nothing else — so that the last `steps` launches of
# genasm_lane_kernel<false> in the trace ARE the timed ones.  Writes gpurun_out/headline_trace/{timed_region.csv,summary.json}:
# every kernel that ran between the first timed launch's begin and the last one's end (begin / end timestamps in ns), and the
# sum of the timed launches' durations, the wall time they span, and the overlap factor (sum / wall).
steps=${1:-20}
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$root'); import scrooge_amd; scrooge_amd.build_library()" || exit 1
out=$root/gpurun_out/headline_trace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/raw -o trace --output-format csv -- python3 $root/bench.py --no-build --headline-only --steps $steps --warmup 2 > $out/bench.log 2> $out/bench.err
python3 - $out $steps <<'P'
import csv, glob, json, sys
d, steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/raw/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
align = [r for r in rows if "genasm_lane_kernel" in r["Kernel_Name"] and ("<false>" in r["Kernel_Name"] or "ILb0" in r["Kernel_Name"])]
timed = align[-steps:]
t0, t1 = int(timed[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in timed)
inside = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1 + 2_000_000]
with open(d + "/timed_region.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "begin_ns_since_first_timed_launch", "end_ns", "duration_ns", "stream_or_queue"])
    for r in inside:
        w.writerow([r["Kernel_Name"][:80], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0,
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Queue_Id", "")])
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
comp = [r for r in inside if "compact_runs" in r["Kernel_Name"]]
line = [l for l in open(d + "/bench.log") if l.startswith('{"metric"')]
bench = json.loads(line[-1]) if line else {}
s = {"command": "rocprofv3 --kernel-trace -- python3 bench.py --no-build --headline-only --steps %d --warmup 2" % steps,
     "timed_launches": len(timed), "align_launches_in_trace": len(align),
     "sum_of_timed_align_durations_ms": sum(dur) / 1e6, "mean_align_duration_ms": sum(dur) / len(dur) / 1e6,
     "min_align_duration_ms": min(dur) / 1e6, "max_align_duration_ms": max(dur) / 1e6,
     "wall_first_begin_to_last_end_ms": (t1 - t0) / 1e6, "overlap_factor_sum_over_wall": sum(dur) / (t1 - t0),
     "ms_per_step_from_the_trace": (t1 - t0) / 1e6 / steps,
     "begin_to_begin_ms_mean": (int(timed[-1]["Start_Timestamp"]) - t0) / 1e6 / max(1, steps - 1),
     "compact_runs_kernels_inside": len(comp), "compact_runs_mean_ms": (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in comp) / max(1, len(comp))) / 1e6,
     "bench_ms_per_step_same_run": bench.get("ms_per_step"), "bench_value_same_run": bench.get("value")}
json.dump(s, open(d + "/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
P
