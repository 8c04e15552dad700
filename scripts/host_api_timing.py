"""End-to-end timing of the host-pointer API (PCIe-inclusive), for DESIGN.md."""
import sys, time
sys.path.insert(0, ".")
import scrooge_amd
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t, q = synth.make_pairs(2000, 10000, "ont", seed=42)
T, Q = t * (n // 2000), q * (n // 2000)
a = scrooge_amd.Aligner(0)
a.align_pairs(T[:2000], Q[:2000])      # warm (allocations)
for rep in range(2):
    t0 = time.time(); r = a.align_pairs(T, Q); dt = time.time() - t0
    tm = a.last_timing
    print("pairs=%d  python wall %.3fs | library total %.3fs  pack(H2D+2bit) %.3fs  kernel %.4fs -> end-to-end %.0f pairs/s, kernel-only %.0f pairs/s"
          % (len(T), dt, tm["total_ns"] / 1e9, tm["pack_ns"] / 1e9, tm["kernel_ns"] / 1e9, len(T) / (tm["total_ns"] / 1e9), len(T) / (tm["kernel_ns"] / 1e9)))
