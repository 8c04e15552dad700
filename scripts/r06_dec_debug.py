#!/usr/bin/env python3
"""Debug aid (GPU box): decode a batch with SCRG_DEC_KERNEL=quad and report where its runs differ from scrg_compact_runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import scrooge_amd
from scrooge_amd import synth

os.environ["SCRG_DEC_KERNEL"] = "quad"
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0)
al.set_stream(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n_ = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t, q = synth.make_pairs(n_, L, "pacbio15", seed=78)
n = len(t)
tw, rw = (max(len(x) for x in t) + 31) // 32, (L + 31) // 32
rows = np.zeros((n, (tw + rw) * 32), dtype=np.uint8)
for k in range(n):
    rows[k, :len(t[k])] = np.frombuffer(t[k], dtype=np.uint8)
    rows[k, tw * 32: tw * 32 + len(q[k])] = np.frombuffer(q[k], dtype=np.uint8)
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
al.pack_planar(torch.from_numpy(rows).to(dev).view(-1), seq, bad)
desc = torch.stack([idx * (tw + rw) * 32, torch.tensor([len(x) for x in t], device=dev), (idx * (tw + rw) + tw) * 32,
                    torch.tensor([len(x) for x in q], device=dev), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
runs = torch.zeros(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev)
nr = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
al.align_device(n, seq, desc, runs, ed, nr, st)
stream = torch.zeros(n * (L // 2 + 64) + 64, dtype=torch.uint8, device=dev)
s_off = torch.empty(n, dtype=torch.int64, device=dev)
s_len = torch.empty(n, dtype=torch.int32, device=dev)
tot = torch.empty(2, dtype=torch.int64, device=dev)
al.encode_edit_stream(n, desc, runs, nr, stream, s_off, s_len, tot)
cnt = nr.to(torch.int64)
total = int(cnt.sum().item())
off = torch.cumsum(cnt, 0) - cnt
dense = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
back = torch.zeros(total * 2 + 8, dtype=torch.uint8, device=dev)
nbad = torch.zeros(1, dtype=torch.int32, device=dev)
al.compact_runs(n, desc, runs, nr, off, dense)
al.decode_edit_stream(n, stream, s_off, s_len, desc.view(-1)[3:], 6, off, back, nr, nbad)
torch.cuda.synchronize()
d = dense.cpu().numpy().view(np.uint16)[:total]
b = back.cpu().numpy().view(np.uint16)[:total]
offs = off.cpu().numpy()
lens = s_len.cpu().numpy()
sb = stream.cpu().numpy()
so = s_off.cpu().numpy()
bad_at = np.nonzero(d != b)[0]
print("nbad", int(nbad.item()), "total runs", total, "mismatching runs", len(bad_at))
for g in bad_at[:25]:
    p = int(np.searchsorted(offs, g, side="right") - 1)
    r = int(g - offs[p])
    print("run G=%d pair %d r=%d (pair runs %d, g0=%d h=%d, stream len %d off %d) want %04x got %04x" % (g, p, r, int(cnt[p]), offs[p], offs[p] & 7, lens[p], so[p], d[g], b[g]))
if len(bad_at):
    g = int(bad_at[0]); p = int(np.searchsorted(offs, g, side="right") - 1)
    print("stream of pair", p, ":", bytes(sb[so[p]:so[p] + lens[p]]).hex())
    print("want runs:", [hex(v) for v in d[offs[p]:offs[p] + int(cnt[p])][:80]])
    print("got  runs:", [hex(v) for v in b[offs[p]:offs[p] + int(cnt[p])][:80]])
