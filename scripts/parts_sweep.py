"""genasm_lane_parts_kernel (64 <= W-O <= 127: the table in registers, in parts of 16 columns) next to the kernel it replaces
for these W/O (genasm_lane_mw_kernel, table in HBM: reserved[0] = 256) on the bench workload, single launches.
usage: python scripts/parts_sweep.py [pairs=100000] [layout=linear|groups]"""
import json, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
groups = len(sys.argv) > 2 and sys.argv[2] == "groups"
L = 10000
dev = torch.device("cuda", 0)
scrooge_amd.build_library(variant="select")          # the kernel selection switches exist in the test build only
al = scrooge_amd.Aligner(0, variant="select"); al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
kw0 = {}
if groups:
    G = scrooge_amd.api.GROUP
    seq = torch.zeros((n + G - 1) // G * G * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
    al.pack_planar_groups(rows_a.view(-1), n, tw + rw, seq, bad)
    first = (idx // G) * (tw + rw) * G + idx % G
    desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    kw0 = dict(text_stride_words=G, read_stride_words=G)
else:
    seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
    al.pack_planar(rows_a.view(-1), seq, bad)
    desc = torch.stack([idx * (tw + rw) * 32, torch.full_like(idx, text_len), (idx * (tw + rw) + tw) * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
del rows_a
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev); nr = torch.empty(n, dtype=torch.int32, device=dev); st = torch.empty(n, dtype=torch.int32, device=dev)
out = []
for W, O in [(160, 81), (192, 97), (224, 113), (256, 129), (128, 20), (128, 1), (200, 100), (256, 200), (192, 150), (256, 224)]:
    res = {}
    for name, sw in (("parts", 0), ("table_in_hbm", 256)):
        p = al.make_params(W=W, O=O, **kw0)
        p.reserved[0] = sw
        keep, al.params = al.params, p
        try:
            for rep in range(3):
                al.align_device(n, seq, desc, runs, ed, nr, st)
                ms = al.last_kernel_ms()
            res[name] = {"ms": ms, "M_pairs_per_s": n / ms / 1e3, "mean_ed": float(ed.double().mean()), "runs": int(nr.sum().item()), "status_max": int(st.max().item())}
        finally:
            al.params = keep
    res["same_results"] = res["parts"]["mean_ed"] == res["table_in_hbm"]["mean_ed"] and res["parts"]["runs"] == res["table_in_hbm"]["runs"]
    print(W, O, json.dumps(res), flush=True)
