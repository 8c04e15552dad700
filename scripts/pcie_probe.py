"""D2H / H2D rates of pinned copies by size (GPU box): python3 scripts/pcie_probe.py"""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (1, 2, 4, 8, 16, 64, 256):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        reps = max(3, 256 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s %4d MB: %.3f ms  %.1f GB/s" % (name, mb, dt * 1e3, n / dt / 1e9))
