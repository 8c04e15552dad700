"""mixed-length batch of bench.py's other_configs at several wavefront counts per CU.  usage: python scripts/attic/mixed_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scrooge_amd, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
streams = [torch.cuda.ExternalStream(scrooge_amd.api.create_stream(0, pr), device=dev) for pr in (1, -1, 0, 1)]
for wpc in (0, 4, 5, 6, 8, 3):
    r = bench.run_other_config(torch, scrooge_amd, dev, 0, streams, "mixed", 100000, 20000, "ont", 10, 2, 0, 59, 16, len_range=(2000, 20000), waves_per_cu=wpc)
    print("waves_per_cu", wpc, "%.2f M pairs/s" % (r["value"] / 1e6), "%.3f ms/step" % r["ms_per_step"], flush=True)
