#!/usr/bin/env python3
"""Launch time of the align kernel at a few W / O settings on the bench workload (GPU box): scripts/wide_ab.py [W O ...]
(SCRG_LIB selects the build, as in scripts/ab.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import scrooge_amd
    from scrooge_amd import synth
    import bench
    pts = [int(x) for x in sys.argv[1:]] or [64, 2, 64, 16, 96, 49, 128, 65]
    dev = torch.device("cuda", 0)
    al = scrooge_amd.Aligner(0)
    al.set_stream(torch.cuda.current_stream().cuda_stream)
    n, L = 100000, 10000
    err, ratio = synth.PROFILES["ont"]
    # texts that END with their reads (text = the read's source segment, no slack): the last window of every pair is short
    rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
    G = scrooge_amd.api.GROUP
    row_words = tw + rw
    seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    al.pack_planar_groups(rows.view(-1), n, row_words, seq, bad)
    del rows
    cap = (2 * L + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    first = (idx // G) * row_words * G + idx % G
    desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L),
                        idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    kw = dict(text_stride_words=G, read_stride_words=G)
    runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
    ed = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    for W, O in zip(pts[0::2], pts[1::2]):
        for _ in range(3):
            al.align_device(n, seq, desc, runs, ed, cnt, st, W=W, O=O, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            al.align_device(n, seq, desc, runs, ed, cnt, st, W=W, O=O, **kw)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print("W=%d O=%d: %.3f ms per launch of %d pairs = %.2f M pairs/s (mean edit distance %.1f)" % (W, O, ms, n, n / ms / 1e3, float(ed.double().mean())))


if __name__ == "__main__":
    main()
