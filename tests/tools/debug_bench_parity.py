import sys, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch, numpy as np
import scrooge_amd, bench
from scrooge_amd import synth
from oracle.pyoracle import Oracle
n = int(sys.argv[1]); L = 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0); al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
row_words = tw + rw
seq = torch.zeros(n * row_words + 4, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
al.pack_planar(rows.view(-1), seq, bad); torch.cuda.synchronize()
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
desc = torch.stack([idx * row_words * 32, torch.full_like(idx, text_len), (idx * row_words + tw) * 32,
                    torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.full((n,), -1, dtype=torch.int64, device=dev)
nr = torch.full((n,), -1, dtype=torch.int32, device=dev)
st = torch.full((n,), -1, dtype=torch.int32, device=dev)
al.align_device(n, seq, desc, runs, ed, nr, st); torch.cuda.synchronize()
h = rows.cpu().numpy()
T = [h[i, :text_len].tobytes() for i in range(n)]
Q = [h[i, tw * 32: tw * 32 + L].tobytes() for i in range(n)]
e, c, _, ns = Oracle().align(T, Q, threads=16)
edh = ed.cpu().numpy(); nrh = nr.cpu().numpy()
badi = np.nonzero(edh != np.array(e))[0]
print("n", n, "ed mismatches", len(badi), badi[:20], "status max", int(st.max()))
if len(badi):
    i = int(badi[0]); print(i, edh[i], e[i], nrh[i], c[i][:100])
    seg = runs[2 * i * cap: 2 * (i * cap + nrh[i])].cpu().numpy()
    print("".join("%d%s" % (seg[2 * j], chr(seg[2 * j + 1])) for j in range(min(40, nrh[i]))))
