"""Deterministic synthetic read/text pairs shaped like the reference's datasets.

The reference's long-read sets are PBSIM2 simulations (DATASETS.md:51: 10 kb
reads, accuracy 0.95, sub:ins:del = 6:50:54); BASELINE.json's headline config is
10 kb reads with ONT-like errors (10 %, 23:31:46).  Text handed to the aligner
is the read's source segment plus 15 % slack, the same slack the reference
gives its pairwise baselines (src/cpu_baseline.cpp:144, 337).

Host generator (numpy) for tests and CPU samples; ``device_pairs`` builds the
same kind of data with torch on the GPU for full-size bench batches.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

PROFILES = {
    # name: (error rate, (sub, ins, del) ratio)
    "illumina": (0.01, (90, 5, 5)),
    "ont": (0.10, (23, 31, 46)),
    "pacbio": (0.05, (6, 50, 54)),
    "pacbio15": (0.15, (6, 50, 54)),
    "uniform": (0.10, (1, 1, 1)),
}


def mutate(src, err, ratio, rng):
    """Apply i.i.d. per-base errors to a code array (values 0..3)."""
    n = src.shape[0]
    r = np.asarray(ratio, dtype=np.float64)
    r = r / r.sum()
    u = rng.random(n)
    kind = np.zeros(n, dtype=np.int8)               # 0 keep, 1 sub, 2 ins, 3 del
    kind[u < err] = 1 + np.searchsorted(np.cumsum(r), rng.random(int((u < err).sum())), side="right").clip(0, 2)
    out = src.copy()
    sub = kind == 1
    out[sub] = (out[sub] + rng.integers(1, 4, int(sub.sum()))) & 3
    emit = np.ones(n, dtype=np.int64)
    emit[kind == 2] = 2
    emit[kind == 3] = 0
    pos = np.cumsum(emit) - emit
    total = int(emit.sum())
    res = np.empty(total, dtype=np.uint8)
    keep = emit > 0
    # an insertion emits a random base, then the source base
    ins = kind == 2
    res[pos[ins]] = rng.integers(0, 4, int(ins.sum()))
    res[pos[keep] + (emit[keep] - 1)] = out[keep]
    return res


def make_pair(read_len, err, ratio, rng, slack=0.15):
    """-> (text codes, read codes); read has exactly read_len bases."""
    need = int(read_len * (1.0 + slack) + 0.999999)
    src_len = max(need, int(read_len * 1.08) + 64)
    while True:
        src = rng.integers(0, 4, src_len, dtype=np.uint8)
        read = mutate(src, err, ratio, rng)
        if read.shape[0] >= read_len:
            return src[:need], read[:read_len]
        src_len *= 2


def make_pairs(n, read_len, profile="ont", seed=42, slack=0.15, err=None, ratio=None):
    """-> (texts, reads) as lists of ASCII bytes."""
    e, r = PROFILES[profile]
    err = e if err is None else err
    ratio = r if ratio is None else ratio
    rng = np.random.Generator(np.random.PCG64(seed))
    texts, reads = [], []
    for _ in range(n):
        t, q = make_pair(read_len, err, ratio, rng, slack)
        texts.append(BASES[t].tobytes())
        reads.append(BASES[q].tobytes())
    return texts, reads


def random_seq(n, rng):
    return BASES[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()
