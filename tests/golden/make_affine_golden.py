#!/usr/bin/env python3
"""Regenerate tests/golden/affine_scores.json from the reference's own affine re-scorer.

Runs only in the build container: needs oracle/_ref/libbaseline_ref.so, which `make -C oracle` links from the
unmodified src/cpu_baseline.cpp (get_alignment_score, :694-725) behind oracle/ref_score_driver.cpp.  The fixture is
data: CIGAR strings (the reference CPU path's own alignments of the other golden files, plus hand-made corner cases:
adjacent insertion/deletion runs, split runs of one kind, the empty string, long counts), cost sets (the baseline driver's
default 2,4,4,2, src/cpu_baseline.cpp:883, and others), and the score the reference returned."""
import ctypes as C
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
COSTS = [(2, 4, 4, 2), (1, 1, 1, 1), (0, 1, 0, 1), (5, 4, 10, 1), (1, 3, 0, 2), (0, 0, 7, 0)]


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbaseline_ref.so"))
    lib.ref_alignment_score.restype = C.c_longlong
    lib.ref_alignment_score.argtypes = [C.c_char_p] + [C.c_int] * 4
    cigars = ["", "10=", "4=1X4=", "3=2I3=", "2I3D", "3D2I", "2I1=3D", "2I1X3D", "31I13I16=2I2I", "1I1D1I1D1I1D", "5X", "1=1X1=1X",
              "100000=", "65535I", "70000D3=", "12=3X4=1I1I1I7=2D2D9="]
    for f in sorted(glob.glob(os.path.join(HERE, "pairs_w*.json")) + glob.glob(os.path.join(HERE, "mapping_w*.json"))):
        d = json.load(open(f))
        found = [c["cigar"] for c in d["cases"]] if "cases" in d else d["cigar"]
        cigars += [c for c in found if len(c) <= 400][::5]
    rng = np.random.Generator(np.random.PCG64(7))
    for _ in range(60):                    # random run lists: every kind may follow every kind, itself included
        k = int(rng.integers(1, 30))
        cigars.append("".join("%d%s" % (int(rng.integers(1, 40)), "=XID"[int(rng.integers(0, 4))]) for _ in range(k)))
    cigars = sorted(set(cigars), key=lambda c: (len(c), c))
    cases = []
    for i, c in enumerate(cigars):
        for costs in (COSTS if i % 4 == 0 or len(c) < 40 else COSTS[:2]):
            cases.append({"cigar": c, "costs": list(costs), "score": int(lib.ref_alignment_score(c.encode(), *costs))})
    with open(os.path.join(HERE, "affine_scores.json"), "w") as f:
        json.dump({"source": "get_alignment_score, src/cpu_baseline.cpp:694-725", "cases": cases}, f, separators=(",", ":"))
        f.write("\n")
    print(len(cigars), "cigars,", len(cases), "cases")


if __name__ == "__main__":
    main()
