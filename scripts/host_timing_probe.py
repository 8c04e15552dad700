"""The host entry points' own timeline (SCRG_HOST_TIMING=1: stage times per chunk on stderr) for scrg_align_pairs on unique
synthetic pairs held in one host array.   usage: python scripts/host_timing_probe.py [pairs=20000] [outputs=1] [reps=3]"""
import os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
os.environ["SCRG_HOST_TIMING"] = "1"
import numpy as np, torch
import scrooge_amd, bench
from scrooge_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
outputs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
err, ratio = synth.PROFILES["ont"]
rows, tw, rw, text_len = bench.device_pairs(torch, n, 10000, err, ratio, 42, dev)
rows = rows.cpu().numpy()
a = scrooge_amd.Aligner(0)
for rep in range(reps):
    print("---- call %d" % rep, file=sys.stderr)
    t0 = time.time()
    r = a.align_pairs_rows(rows, 0, text_len, tw * 32, 10000, outputs=outputs)
    print("call %d: library total %.3f ms, pack (thread time) %.3f ms, python wall %.3f ms" %
          (rep, a.last_timing["total_ns"] / 1e6, a.last_timing["pack_ns"] / 1e6, (time.time() - t0) * 1e3))
