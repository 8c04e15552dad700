"""Steady-state resident read-mapping calls (1 M x 150 bp reads x 4 candidates on a 100 Mbp genome) for a rocprofv3
kernel + memory-copy trace: python3 scripts/mapping_trace_probe.py [reads] [outputs: 0 runs + text, 1 text, 2 runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scrooge_amd
from scrooge_amd import synth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
outputs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = 100_000_000
rng = np.random.Generator(np.random.PCG64(42))
gcodes = rng.integers(0, 4, G, dtype=np.uint8)
genome = synth.BASES[gcodes].tobytes()
starts = rng.integers(0, G - 400, n_reads)
codes = gcodes[starts[:, None] + np.arange(150)[None, :]]
ascii_reads = synth.BASES[codes]
reads = [ascii_reads[r].tobytes() for r in range(n_reads)]
cands = np.stack([starts, np.maximum(0, starts - 2), starts + 3, rng.integers(0, G - 10, n_reads)], axis=1).tolist()
a = scrooge_amd.Aligner(0)
a.set_genome(genome)
for rep in range(3):
    t0 = time.time()
    r = a.align_mapping(None, reads, cands, arrays=True, outputs=outputs)
    print("rep", rep, "outputs", outputs, time.time() - t0, a.last_timing["total_ns"] / 1e6, "ms", "%.1f M pairs/s" % (4 * n_reads / a.last_timing["total_ns"] * 1e3), file=sys.stderr)
