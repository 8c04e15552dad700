"""Stand-alone duration of the result kernels on the bench workload: run compaction, packed compaction, edit-stream
encoding, edit-stream decoding.  usage: python scripts/encode_probe.py [pairs] [read_len]"""
import sys
sys.path.insert(0, ".")
import torch
import scrooge_amd, bench
from scrooge_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
al = scrooge_amd.Aligner(0)
al.set_stream(0)
err, ratio = synth.PROFILES["ont"]
rows_a, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
G = scrooge_amd.api.GROUP
row_words = tw + rw
seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
al.pack_planar_groups(rows_a.view(-1), n, row_words, seq, bad)
del rows_a
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
first = (idx // G) * row_words * G + idx % G
desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32,
                    torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
al.params.text_stride_words = al.params.read_stride_words = G
runs = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
ed = torch.empty(n, dtype=torch.int64, device=dev)
nr = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
al.align_device(n, seq, desc, runs, ed, nr, st)
cnt = nr.to(torch.int64)
total = int(cnt.sum())
off = torch.cumsum(cnt, 0) - cnt
dense = torch.empty(total * 2 + 8, dtype=torch.uint8, device=dev)
packed = torch.empty(total + 16, dtype=torch.uint8, device=dev)
stream = torch.empty(int(ed.sum()) + n * (L >> 6) + 4 * n + 64, dtype=torch.uint8, device=dev)
s_off = torch.empty(n, dtype=torch.int64, device=dev)
s_len = torch.empty(n, dtype=torch.int32, device=dev)
tot = torch.empty(2, dtype=torch.int64, device=dev)
nbad = torch.zeros(1, dtype=torch.int32, device=dev)


def timed(name, f, reps=5):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    print("%-28s %.3f ms" % (name, a.elapsed_time(b) / reps))


for name, f in (("align_device (runs)", lambda: al.align_device(n, seq, desc, runs, ed, nr, st)),
                ("align_device_edits", lambda: al.align_device_edits(n, seq, desc, runs, ed, s_len, st)),
                ("align_device (runs)", lambda: al.align_device(n, seq, desc, runs, ed, nr, st))):
    ms = []
    for _ in range(4):
        f()
        ms.append(al.last_kernel_ms())
    print("%-28s %s ms" % (name, " ".join("%.3f" % x for x in ms)))
timed("compact_runs", lambda: al.compact_runs(n, desc, runs, nr, off, dense))
timed("compact_runs_packed", lambda: al.compact_runs_packed(n, desc, runs, nr, off, packed))
timed("encode_edit_stream", lambda: al.encode_edit_stream(n, desc, runs, nr, stream, s_off, s_len, tot))
print("stream bytes", int(tot[0]), "per pair", int(tot[0]) / n, "runs per pair", total / n)
back = torch.empty_like(dense)
timed("decode_edit_stream", lambda: al.decode_edit_stream(n, stream, s_off, s_len, desc.view(-1)[3:], 6, off, back, nr, nbad), reps=2)
print("decoded == compacted:", bool(torch.equal(back[: 2 * total], dense[: 2 * total])), "bad", int(nbad))
# shader cycles per window round and phase, both output formats (kernel counters)
al.params.reserved[1] = 1
for name, f in (("runs", lambda: al.align_device(n, seq, desc, runs, ed, nr, st)),
                ("edits", lambda: al.align_device_edits(n, seq, desc, runs, ed, s_len, st))):
    f()
    s = al.debug_stats_lane()
    r = max(1, s["rounds"])
    print(name, {k[7:]: round(s[k] / r, 1) for k in ("cycles_fetch", "cycles_setup", "cycles_table", "cycles_pass1", "cycles_traceback")})
