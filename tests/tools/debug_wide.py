"""Debug aid: first difference between the HIP path and the oracle at a given W/O (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scrooge_amd
from scrooge_amd import synth
from oracle.pyoracle import Oracle
w, o = int(sys.argv[1]), int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
outputs = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t, q = synth.make_pairs(64, L, "ont", seed=3)
eds, cigars, _, _ = Oracle().align(t, q, W=w, O=o)
a = scrooge_amd.Aligner(0)
res = a.align_pairs(t, q, W=w, O=o, outputs=outputs)
nbad = 0
for k, (r, e, c) in enumerate(zip(res, eds, cigars)):
    if r.edit_distance != e or r.cigar != c:
        nbad += 1
        if nbad <= 3:
            i = next((i for i in range(min(len(c), len(r.cigar))) if c[i] != r.cigar[i]), min(len(c), len(r.cigar)))
            print("pair", k, "ed", r.edit_distance, e, "len", len(r.cigar), len(c), "first diff at", i)
            print("  got ", r.cigar[max(0, i - 60):i + 40])
            print("  want", c[max(0, i - 60):i + 40])
print("bad", nbad, "of", len(t))
