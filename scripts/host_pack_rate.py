"""How fast the host packer (scrg_pack_planar_host: the AVX2 code of the host entry points) turns ASCII into 2 bits per base on this
box, by the number of worker PROCESSES (each packs its own slice, writing to ordinary memory): the rate a host entry point's packing
stage can reach at best.   usage: python scripts/host_pack_rate.py [MB per worker=32]"""
import ctypes as C, multiprocessing as mp, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def worker(mb, start, go, q):
    os.environ["SCRG_NO_TORCH_PRELOAD"] = "1"
    from scrooge_amd import api
    lib = api.load_library()
    n = mb << 20
    rng = np.random.Generator(np.random.PCG64(os.getpid()))
    buf = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
    out = np.zeros(n // 32 + 8, dtype=np.uint64)
    lib.scrg_pack_planar_host(C.c_void_p(buf.ctypes.data), n, C.c_void_p(out.ctypes.data), 1, n // 32)     # warm
    start.wait()
    go.wait()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        lib.scrg_pack_planar_host(C.c_void_p(buf.ctypes.data), n, C.c_void_p(out.ctypes.data), 1, n // 32)
        best = min(best, time.perf_counter() - t0)
    q.put(best)

if __name__ == "__main__":
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for nt in (1, 2, 4, 8, 16, 32):
        start, go, q = mp.Barrier(nt + 1), mp.Barrier(nt + 1), mp.Queue()
        ps = [mp.Process(target=worker, args=(mb, start, go, q)) for _ in range(nt)]
        for p in ps: p.start()
        start.wait(); go.wait()
        ts = [q.get() for _ in ps]
        for p in ps: p.join()
        print("%2d workers x %d MB: slowest %.2f ms -> %.1f GB/s of ASCII in total (%.1f GB/s per worker)" %
              (nt, mb, max(ts) * 1e3, nt * (mb << 20) / max(ts) / 1e9, (mb << 20) / max(ts) / 1e9), flush=True)
