"""Kernel time of one launch of the default table's two kernels (reserved[0] = 512: a window's work on two wavefronts, 1024: on one)
on the bench workload; no result checks (for experiment builds).  usage: python scripts/split_time.py [pairs=100000]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 10000
dev = torch.device("cuda", 0)
scrooge_amd.build_library(variant="select")          # the kernel selection switches exist in the test build only
al = scrooge_amd.Aligner(0, variant="select"); al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
G = scrooge_amd.api.GROUP
seq = torch.zeros((n + G - 1) // G * G * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
al.pack_planar_groups(rows_a.view(-1), n, tw + rw, seq, bad)
first = (idx // G) * (tw + rw) * G + idx % G
desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
del rows_a
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev); nr = torch.empty(n, dtype=torch.int32, device=dev); st = torch.empty(n, dtype=torch.int32, device=dev)
for sw in (512, 1024, 512, 1024):
    p = al.make_params(text_stride_words=G, read_stride_words=G)
    p.reserved[0] = sw
    al.params = p
    ms = []
    for rep in range(4):
        al.align_device(n, seq, desc, runs, ed, nr, st)
        ms.append(al.last_kernel_ms())
    print("flags", sw, "kernel ms", ["%.3f" % m for m in ms], "mean ed %.1f" % float(ed.double().mean()), flush=True)
