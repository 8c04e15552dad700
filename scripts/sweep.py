"""Knob sweep in the shape of the reference's scripts/profile.py GPU sweeps (:130-248): one CSV row per
configuration with the reference's column names where a knob has an equivalent here.

    python scripts/sweep.py out.csv [pairs] [read_len]

W and O are runtime parameters here (the reference recompiles per configuration, profile.py:131-142);
supported range 2 <= W <= 256, 1 <= O < W (one pair per lane: W-O <= 31 genasm_lane_kernel, 32 <= W-O <= 63 and W <= 128
genasm_lane_wide_kernel — table in registers, built in two halves —, 64 <= W-O <= 127 genasm_lane_parts_kernel — table in registers, in parts of 16 columns —, beyond that genasm_lane_mw_kernel, table in HBM).  "threadblocks/sm" = persistent wavefronts per CU,
"used smem per threadblock (B)" = LDS bytes per wavefront; SENE/DENT/ET are always on (they do not
change results)."""
import csv, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import scrooge_amd, bench
from scrooge_amd import synth

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0)
al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
al.pack_planar(rows_a.view(-1), seq, bad)
del rows_a
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
desc = torch.stack([idx * (tw + rw) * 32, torch.full_like(idx, text_len), (idx * (tw + rw) + tw) * 32,
                    torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev)
nr = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
name = torch.cuda.get_device_name(0)
configs = [(64, 33, 0, 0, 0)]                                                                    # library defaults (one pair per lane, 16 waves/CU)
configs += [(W, min(W // 2 + 1, W - 1), 0, 0, 0) for W in (16, 24, 32, 40, 48, 56)]        # W sweep, O = W/2+1 (profile.py:78): lane kernel
configs += [(64, O, 0, 0, 0) for O in (36, 40, 48, 56, 60)]                                    # O sweep at W=64 (profile.py:88-100): lane kernel
configs += [(64, O, 0, 0, 0) for O in (2, 16, 32)]                                              # small overlaps (W-O > 31): genasm_lane_wide_kernel (64-bit rows, table in registers in two halves)
configs += [(64, O, 8, 13, 3) for O in (2, 16, 32)]                                             # the same on the GenASM-row kernel (G=8, WIDE storage)
configs += [(W, W // 2 + 1, 0, 0, 0) for W in (80, 96, 112, 128, 160, 192, 224, 256)]         # W sweep past one word (profile.py:180-185): genasm_lane_wide_kernel up to W=128, genasm_lane_parts_kernel beyond
configs += [(128, 20, 0, 0, 0), (256, 1, 0, 0, 0)]                                               # rows of two and four words
configs += [(W, W // 2 + 1, 32, 0, 0) for W in (96, 128, 192, 256)]                              # the same on the GenASM-row kernel with multi-word entries
configs += [(128, 65, 64, 0, 0), (256, 129, 64, 0, 0), (256, 129, 32, 20, 0)]
configs += [(64, 33, g, 13, w) for g, w in ((64, 16), (32, 16), (16, 16), (8, 11), (4, 6))]    # the GenASM-row lane mappings
configs += [(64, 33, 1, 0, w) for w in (4, 8, 12, 16)]                                          # one pair per lane: waves per CU
for _ in range(4):                     # (warm-up: the first launches of a process run at a lower clock)
    al.align_device(n, seq, desc, runs, ed, nr, st)
torch.cuda.synchronize()
with open(out, "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["W", "O", "sene", "dent", "early termination", "threadblocks/sm", "lanes per pair", "lds rows",
                 "arch", "gpu", "sm count", "used smem per threadblock (B)", "pairs", "read length",
                 "mean edit distance", "throughput (aligns/s)"])
    for W, O, g, r, w in configs:
        kw = dict(W=W, O=O, lanes_per_pair=g, lds_rows=r, waves_per_cu=w)
        geo = al.query_launch(**kw)
        rp = al.resolved_params(**kw)
        g, r = rp.lanes_per_pair, rp.lds_rows
        for rep in range(2):
            al.align_device(n, seq, desc, runs, ed, nr, st, **kw)
            ms = al.last_kernel_ms()
        assert int(st.max()) == 0
        wr.writerow([W, O, True, True, True, geo["n_waves"] // geo["n_cus"], g, r, "gfx950", name, geo["n_cus"],
                     geo["lds_bytes"], n, L, round(float(ed.double().mean()), 2), round(n / (ms * 1e-3))])
        f.flush()
print(open(out).read())
