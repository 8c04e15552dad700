"""genasm_lane_mw_kernel (W-O > 31, W > 64) at the bench workload, next to the GenASM-row kernel with multi-word
entries.  usage: python scripts/mw_sweep.py"""
import sys
sys.path.insert(0, ".")
import torch
import scrooge_amd, bench
from scrooge_amd import synth
n, L = 100000, 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0); al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
al.pack_planar(rows_a.view(-1), seq, bad); del rows_a
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
desc = torch.stack([idx * (tw + rw) * 32, torch.full_like(idx, text_len), (idx * (tw + rw) + tw) * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev); nr = torch.empty(n, dtype=torch.int32, device=dev); st = torch.empty(n, dtype=torch.int32, device=dev)
for O in (2, 16, 32):
    kw = dict(W=64, O=O)
    for rep in range(2):
        al.align_device(n, seq, desc, runs, ed, nr, st, **kw)
        ms = al.last_kernel_ms()
    print(64, O, "waves/cu", al.resolved_params(**kw).waves_per_cu, "%.2f ms" % ms, "%.2f M pairs/s" % (n / ms / 1e3), "mean ed %.1f" % float(ed.double().mean()), flush=True)
for W, O, g, wpc in [(80,41,0,0),(96,49,0,0),(128,65,0,0),(128,65,0,4),(128,65,0,12),(128,65,0,16),(128,65,32,0),(160,81,0,0),(192,97,0,0),(256,129,0,0),(256,129,0,4),(256,129,0,12),(256,129,32,0),(128,20,0,0),(256,1,0,0)]:
    kw = dict(W=W, O=O, lanes_per_pair=g, waves_per_cu=wpc)
    for rep in range(2):
        al.align_device(n, seq, desc, runs, ed, nr, st, **kw)
        ms = al.last_kernel_ms()
    rp = al.resolved_params(**kw)
    print(W, O, "lanes", rp.lanes_per_pair, "waves/cu", rp.waves_per_cu, "%.2f ms" % ms, "%.2f M pairs/s" % (n / ms / 1e3), "mean ed %.1f" % float(ed.double().mean()), flush=True)
