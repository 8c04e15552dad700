"""End-to-end timing of the host-pointer API (PCIe-inclusive), for DESIGN.md.
usage: python scripts/host_api_timing.py [pairs=20000] [devices=1]"""
import sys, time
sys.path.insert(0, ".")
import ctypes as C
import scrooge_amd
from scrooge_amd import synth, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_dev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t, q = synth.make_pairs(2000, 10000, "ont", seed=42)
T, Q = t * (n // 2000), q * (n // 2000)
a = scrooge_amd.Aligner(0)
a.align_pairs(T[:2000], Q[:2000])      # warm (allocations)
for outputs, name in ((0, "runs+text"), (1, "text only"), (2, "runs only")):
    for rep in range(3):
        t0 = time.time(); r = a.align_pairs(T, Q, arrays=True, outputs=outputs); dt = time.time() - t0
        tm = a.last_timing
    print("%-9s pairs=%d  python wall %.3fs | library total %.4fs  host pack %.4fs (thread time)  kernels %.4fs (sum) -> end-to-end %.2f M pairs/s"
          % (name, len(T), dt, tm["total_ns"] / 1e9, tm["pack_ns"] / 1e9, tm["kernel_ns"] / 1e9, len(T) / (tm["total_ns"] / 1e9) / 1e6))
if n_dev > 1:
    lib = api.load_library()
    tp = (C.c_char_p * n)(*T); qp = (C.c_char_p * n)(*Q)
    tl = (C.c_uint64 * n)(*[len(x) for x in T]); ql = (C.c_uint64 * n)(*[len(x) for x in Q])
    devs = (C.c_int32 * n_dev)(*([0] * n_dev))
    for rep in range(3):
        res = C.POINTER(api.Result)()
        t0 = time.time()
        st = lib.scrg_align_pairs_multi(devs, n_dev, None, n, tp, tl, qp, ql, C.byref(res))
        dt = time.time() - t0
        assert st == 0, lib.scrg_multi_last_error()
        tot = res.contents.total_ns
        lib.scrg_result_free(res)
    print("multi x%d (same GPU) pairs=%d library total %.4fs -> %.2f M pairs/s" % (n_dev, n, tot / 1e9, n / (tot / 1e9) / 1e6))
