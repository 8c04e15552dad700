"""genasm_lane_wide_kernel (32 <= W-O <= 63, W <= 128) next to the kernel it replaces (table in HBM, reserved[0] = 256)
at the bench workload.  usage: python scripts/wide_timing.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0); al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
G = scrooge_amd.api.GROUP
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
row_words = tw + rw
seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
al.pack_planar_groups(rows_a.view(-1), n, row_words, seq, bad); del rows_a
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
first = (idx // G) * row_words * G + idx % G
desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev); nr = torch.empty(n, dtype=torch.int32, device=dev); st = torch.empty(n, dtype=torch.int32, device=dev)
cfgs = [(64, 33, 0), (64, 2, 0), (64, 2, 256), (64, 16, 0), (64, 32, 0), (64, 32, 256), (80, 41, 0), (96, 49, 0), (112, 57, 0), (128, 65, 0), (128, 65, 256), (128, 96, 0)]
for W, O, sw in cfgs:
    for wpc in (0, 12) if sw == 0 and (W, O) != (64, 33) else (0,):
        p = al.make_params(W=W, O=O, waves_per_cu=wpc, text_stride_words=G, read_stride_words=G)
        p.reserved[0] = sw
        keep = al.params; al.params = p
        try:
            for edits in (False, True):
                best = 1e9
                for rep in range(3):
                    if edits: al.align_device_edits(n, seq, desc, runs, ed, nr, st)
                    else: al.align_device(n, seq, desc, runs, ed, nr, st)
                    best = min(best, al.last_kernel_ms())
                print("W=%d O=%d %s waves/cu=%d %s: %.2f ms  %.2f M pairs/s  mean ed %.1f" % (
                    W, O, "mw(HBM)" if sw else "default", al.resolved_params().waves_per_cu, "edits" if edits else "runs ", best, n / best / 1e3, float(ed.double().mean())), flush=True)
        finally:
            al.params = keep
