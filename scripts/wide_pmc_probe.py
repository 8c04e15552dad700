"""Three launches of the one-pair-per-lane aligner at a given W/O on the bench workload, for rocprofv3 --pmc
(scripts/wide_pmc.sh: the shipped library), and — run with a -DSCRG_STATS build, not under the profiler — the window rounds the kernel counted: python3 scripts/wide_pmc_probe.py W O [pairs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth
W, O = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
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
p = al.make_params(W=W, O=O, text_stride_words=G, read_stride_words=G)
stats = bool(scrooge_amd.load_library().scrg_build_flags() & 1)         # the kernels' own counters: a -DSCRG_STATS build only
if stats:
    p.reserved[1] = 1
al.params = p
ms = []
for rep in range(3):
    al.align_device(n, seq, desc, runs, ed, nr, st)
    ms.append(al.last_kernel_ms())
rounds = al.debug_stats_lane()["rounds"] if stats else None
print(json.dumps({"W": W, "O": O, "pairs": n, "read_len": L, "kernel_ms": ms, "window_rounds_per_launch": rounds,
                  "windows_per_pair": (rounds * 64 / n) if rounds else None, "mean_edit_distance": float(ed.double().mean())}))
