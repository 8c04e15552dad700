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


timed("compact_runs", lambda: al.compact_runs(n, desc, runs, nr, off, dense))
timed("compact_runs_packed", lambda: al.compact_runs_packed(n, desc, runs, nr, off, packed))
timed("encode_edit_stream", lambda: al.encode_edit_stream(n, desc, runs, nr, stream, s_off, s_len, tot))
print("stream bytes", int(tot[0]), "per pair", int(tot[0]) / n, "runs per pair", total / n)
back = torch.empty_like(dense)
timed("decode_edit_stream", lambda: al.decode_edit_stream(n, stream, s_off, s_len, desc.view(-1)[3:], 6, off, back, nr, nbad), reps=2)
print("decoded == compacted:", bool(torch.equal(back[: 2 * total], dense[: 2 * total])), "bad", int(nbad))
