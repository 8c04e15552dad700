"""One steady-state scrg_align_pairs call of 20 k x 10 kb pairs for a rocprofv3 kernel + memory-copy trace."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrooge_amd
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t, q = synth.make_pairs(2000, 10000, "ont", seed=42)
T, Q = t * (n // 2000), q * (n // 2000)
a = scrooge_amd.Aligner(0)
for rep in range(3):
    t0 = time.time(); r = a.align_pairs(T, Q, arrays=True, outputs=1); print("rep", rep, time.time() - t0, a.last_timing["total_ns"] / 1e6, "ms", file=sys.stderr)
