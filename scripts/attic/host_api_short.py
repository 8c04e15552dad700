"""End-to-end timing of scrg_align_pairs on short reads (BASELINE configs[0] shape through the host API, PCIe included).
usage: python scripts/host_api_short.py [pairs=1000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrooge_amd
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
t, q = synth.make_pairs(20000, 150, "illumina", seed=42)
T, Q = t * (n // 20000), q * (n // 20000)
a = scrooge_amd.Aligner(0)
a.align_pairs(T[:20000], Q[:20000])
for outputs, name in ((0, "runs+text"), (1, "text only"), (2, "runs only")):
    for rep in range(3):
        r = a.align_pairs(T, Q, arrays=True, outputs=outputs)
        tm = a.last_timing
    print("%-9s pairs=%d x 150 bp: library total %.2f ms -> end-to-end %.1f M pairs/s (kernels %.2f ms)" % (
        name, len(T), tm["total_ns"] / 1e6, len(T) / (tm["total_ns"] / 1e9) / 1e6, tm["kernel_ns"] / 1e6))
