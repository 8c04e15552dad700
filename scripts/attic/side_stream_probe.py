"""Does a kernel with multi-wave workgroups on another stream get CU time while the pipelined align launches
keep the GPU full?  (Stand-in for RCCL's gather kernels in bench.py --gpus N.)  Runs the bench's two-lane
step loop and, on a third stream, device-to-device copies of 0.43 GB; prints each copy's duration."""
import sys, time
sys.path.insert(0, ".")
import torch, bench, scrooge_amd
from scrooge_amd import synth
dev = torch.device("cuda", 0)
n, L = 100000, 10000
err, ratio = synth.PROFILES["ont"]
rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
als = [scrooge_amd.Aligner(0), scrooge_amd.Aligner(0)]
seq = torch.zeros(n * (tw + rw) + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
als[0].set_stream(torch.cuda.current_stream().cuda_stream)
als[0].pack_planar(rows.view(-1), seq, bad); del rows
cap = (2 * L + 8 + 15) // 16 * 16
idx = torch.arange(n, dtype=torch.int64, device=dev)
desc = torch.stack([idx * (tw + rw) * 32, torch.full_like(idx, text_len), (idx * (tw + rw) + tw) * 32,
                    torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
outs = [dict(runs=torch.empty(n * cap * 2, dtype=torch.uint8, device=dev), ed=torch.empty(n, dtype=torch.int64, device=dev),
             nr=torch.empty(n, dtype=torch.int32, device=dev), st=torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(2)]
streams = [torch.cuda.current_stream(), torch.cuda.Stream()] if "default0" in sys.argv else ([torch.cuda.Stream(), torch.cuda.Stream(priority=-1)] if "prio" in sys.argv else [torch.cuda.Stream(), torch.cuda.Stream()])
mode = sys.argv[1] if len(sys.argv) > 1 else "copy"      # copy | high (copy on a high-priority stream) | kernel (elementwise kernel) | none
side = torch.cuda.Stream(priority=-1) if mode == "high" else torch.cuda.Stream()
src = torch.empty(430_000_000, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
for a, s in zip(als, streams): a.set_stream(s.cuda_stream)
def copies(k):
    ev = []
    with torch.cuda.stream(side):
        for _ in range(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if mode == "kernel":
                dst.add_(1)
            else:
                dst.copy_(src)
            b.record(); ev.append((a, b))
    return ev
ev0 = copies(5); torch.cuda.synchronize()
print("idle GPU: copy ms", [round(a.elapsed_time(b), 2) for a, b in ev0])
t0 = time.perf_counter()
K = 16
evs = []
done = []
first = torch.cuda.Event(enable_timing=True); first.record()
for k in range(K):
    o = outs[k % 2]
    with torch.cuda.stream(streams[k % 2]):
        als[k % 2].align_device(n, seq, desc, o["runs"], o["ed"], o["nr"], o["st"])
        e = torch.cuda.Event(enable_timing=True); e.record(); done.append(e)
    if k >= 2 and mode != "none": evs += copies(1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("under pipelined align: copy ms", [round(a.elapsed_time(b), 2) for a, b in evs])
print("align steps: %.2f ms/step" % (dt / K * 1e3))
print("completion times (ms):", [round(first.elapsed_time(e), 1) for e in done])
