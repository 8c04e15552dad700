"""Kernel time of single launches at given W/O on the bench workload, with checksums of the results (for comparing two builds of
the library through SCRG_LIB).  usage: python scripts/attic/wo_time.py W,O [W,O ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth
n, L = 100000, 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0); al.set_stream(0)
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
for wo in sys.argv[1:]:
    W, O = (int(v) for v in wo.split(","))
    al.params = al.make_params(W=W, O=O, text_stride_words=G, read_stride_words=G)
    ms = []
    for rep in range(4):
        runs.zero_()
        al.align_device(n, seq, desc, runs, ed, nr, st)
        ms.append(al.last_kernel_ms())
    torch.cuda.synchronize()
    chk = int(runs.view(torch.int64).sum().item()) & 0xffffffffffff
    print("W=%d O=%d kernel ms %s  ed sum %d  runs sum %d  checksum %x" % (W, O, ["%.2f" % m for m in ms], int(ed.sum()), int(nr.sum()), chk), flush=True)
