#!/usr/bin/env python3
"""Fixtures for O = 0 (no window overlap) from the UNMODIFIED reference CPU path built with its own -DCLI_KNOBS -DCLI_W=.. -DCLI_O=0
(src/genasm_cpu.cpp:22-35; O = 0 is the special case of :104-110: TB_LIMIT = W, and the reference's O sweep reaches it for
--override_W < 32, scripts/profile.py:92-93).  Runs only in the build container (oracle/_ref/libgenasm_ref_w*_o0.so, `make -C
oracle ref`); the fixtures are data: inputs and the (edit distance, CIGAR) the reference returned."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Reference  # noqa: E402
from scrooge_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for W in (16, 24, 40, 64, 128):
        r = Reference(W, 0)
        rng = np.random.Generator(np.random.PCG64(77 * W))
        L = 300 if W <= 64 else 900
        T, Q = synth.make_pairs(25, L, "ont", seed=W * 13)
        t2, q2 = synth.make_pairs(10, 150, "illumina", seed=W + 5)
        t3, q3 = synth.make_pairs(8, 400, "pacbio15", seed=W + 9)
        T, Q = T + t2 + t3, Q + q2 + q3
        for _ in range(40):
            T.append(synth.random_seq(int(rng.integers(0, 200)), rng))
            Q.append(synth.random_seq(int(rng.integers(0, 200)), rng))
        T += [b"", b"ACGT", b"ACGTACGTAC" * 30]
        Q += [b"ACGTAC", b"", b"ACGTACGTAC" * 30]
        eds, cigs, _ = r.align(T, Q)
        with open(os.path.join(HERE, "pairs_w%d_o0.json" % W), "w") as f:
            json.dump({"W": W, "O": 0, "cases": [{"group": "no overlap", "text": t.decode(), "read": q.decode(), "ed": e, "cigar": c}
                                                 for t, q, e, c in zip(T, Q, eds, cigs)]}, f, separators=(",", ":"))
            f.write("\n")
        print("W=%d O=0 cases:" % W, len(T), "mean ed", sum(eds) / len(eds))


if __name__ == "__main__":
    main()
