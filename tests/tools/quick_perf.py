"""Ad-hoc throughput probe through the host-pointer API (kernel_ns only)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import scrooge_amd
from scrooge_amd import synth
from oracle.pyoracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
prof = sys.argv[3] if len(sys.argv) > 3 else "ont"
t0 = time.time()
t, q = synth.make_pairs(256, L, prof, seed=42)
reps = (n + 255) // 256
T, Q = t * reps, q * reps
print("gen %.1fs, pairs=%d" % (time.time() - t0, len(T)))
a = scrooge_amd.Aligner(0)
o = Oracle()
eds, cig, st, ns = o.align(t, q, threads=8)
print("oracle 8thr: %.0f pairs/s; per pair:" % (256 / ns * 1e9), {k: v / 256 for k, v in st.items()})
for g in (8, 4, 16, 64):
    for rows in (16,):
        for wpc in (4, 8, 9):
            try:
                r = a.align_pairs(T, Q, lanes_per_pair=g, lds_rows=rows, waves_per_cu=wpc)
            except Exception as e:
                print(g, rows, wpc, "ERR", e); continue
            ok = all(x.edit_distance == e and x.cigar == c for x, e, c in zip(r[:256], eds, cig))
            k = a.last_timing["kernel_ns"]
            print("G=%2d rows=%d wpc=%d  kernel %.2f ms  %.0f pairs/s  parity=%s  launch=%s" % (
                g, rows, wpc, k / 1e6, len(T) / k * 1e9, ok, a.query_launch(lanes_per_pair=g, lds_rows=rows, waves_per_cu=wpc)))
