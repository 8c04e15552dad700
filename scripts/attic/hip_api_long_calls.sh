# which HIP API calls of a host-entry-point run take long (rocprofv3 --hip-trace; no counters): args = a python script and its arguments
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/hiptrace
rm -rf $out; mkdir -p $out
rocprofv3 --hip-trace --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1
f=$(find $out -name "*hip_api_trace.csv" | head -1)
echo "trace: $f"
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
long_ = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"]) - t0, r.get("Thread_Id")) for r in rows]
long_.sort(reverse=True)
for d, fn, at, th in long_[:40]:
    print("%9.3f ms  %-28s at %10.3f ms  thread %s" % (d / 1e6, fn, at / 1e6, th))
P
grep "^rep" $out/run.log
