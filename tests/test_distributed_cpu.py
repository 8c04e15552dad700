"""world_size-2 gloo test of the N>1 path: shard plan, per-rank alignment,
variable-length gather to rank 0, restoration of input order.  The per-rank
aligner here is a stand-in backed by the CPU oracle (tests only): what is under
test is the sharding/gather logic, which is the same code the GPU ranks run
with backend nccl (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scrooge_amd import distributed as sd
from scrooge_amd import synth


class OracleAligner:
    def __init__(self):
        from oracle.pyoracle import Oracle
        self.o = Oracle()

    def align_pairs(self, texts, queries, **kw):
        if not texts:
            return []
        eds, cigars, _, _ = self.o.align(texts, queries, W=kw.get("W", 64), O=kw.get("O", 33))
        return list(zip(cigars, eds))


def _batch():
    rng = np.random.Generator(np.random.PCG64(5))
    T, Q = [], []
    for L in [300, 40, 0, 1200, 77, 5, 640, 0, 90, 333, 1000]:
        t, q = synth.make_pairs(1, max(L, 1), "ont", seed=L + 3)
        T.append(t[0])
        Q.append(q[0][:L])
    return T, Q


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, Q = _batch()
        res = sd.align_pairs_sharded(OracleAligner(), T, Q, dst=0)
        if rank == 0:
            q.put(res)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def _worker_fixed(rank, world, port, q, depth=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5
        counts = torch.tensor([rank + 1, 0, 3, 2 * rank, 1], dtype=torch.int32)
        total = int(counts.sum())
        g = sd.ResultGather(n, total, torch.device("cpu"), dst=0, depth=depth)
        assert g.DEPTH == (depth or 2)
        ed = torch.arange(n, dtype=torch.int64) * (rank + 1)
        for k in range(5):        # multi-buffered, reusable across steps
            g.finish(k)
            g.send_runs[k % g.DEPTH][: 2 * total] = torch.arange(2 * total, dtype=torch.uint8) + 10 * rank + k
            g.start(k, ed + k, counts)
        g.finish_all()
        if rank == 0:
            out = []
            for r in range(world):
                e, c, b = g.results(4, r)
                e3, _, b3 = g.results(3, r)
                out.append((e.tolist(), c.tolist(), b.tolist(), e3.tolist(), b3.tolist()))
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("depth", [None, 4])            # 4: one buffer set per pipelined step of bench.py
def test_fixed_size_result_gather(depth):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fixed, args=(r, 2, port, q, depth)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        counts = [r + 1, 0, 3, 2 * r, 1]
        total = sum(counts)
        assert out[r][0] == [k * (r + 1) + 4 for k in range(5)]
        assert out[r][1] == counts
        assert out[r][2] == [(x + 10 * r + 4) % 256 for x in range(2 * total)]
        assert out[r][3] == [k * (r + 1) + 3 for k in range(5)]            # the other buffer still holds step 3
        assert out[r][4] == [(x + 10 * r + 3) % 256 for x in range(2 * total)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_plan_is_a_balanced_partition():
    lens = [5, 100, 7, 100, 3, 50, 50, 1]
    plan = sd.shard_plan(lens, 3)
    allidx = sorted(int(i) for p in plan for i in p)
    assert allidx == list(range(len(lens)))
    sums = [sum(lens[int(i)] for i in p) for p in plan]
    assert max(sums) - min(sums) <= max(lens)
    assert lens[int(plan[0][0])] == 100      # longest first


@pytest.mark.timeout(180)
def test_two_rank_gather_restores_input_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    T, Q = _batch()
    want = OracleAligner().align_pairs(T, Q)
    assert res == want


def test_bench_generator_shapes_on_cpu():
    """bench.py's on-device generator (run here on CPU tensors): fixed read length, zero-padded
    32-byte slots, and an error profile the oracle confirms (ONT 10 % -> ~0.1 edits per base)."""
    import bench
    from oracle.pyoracle import Oracle
    n, L = 24, 600
    err, ratio = synth.PROFILES["ont"]
    rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 7, torch.device("cpu"), chunk=16)
    assert rows.shape == (n, (tw + rw) * 32) and text_len == 690
    h = rows.numpy()
    texts = [h[i, :text_len].tobytes() for i in range(n)]
    reads = [h[i, tw * 32: tw * 32 + L].tobytes() for i in range(n)]
    assert all(set(t) <= set(b"ACGT") for t in texts + reads)
    assert (h[:, text_len:tw * 32] == 0).all() and (h[:, tw * 32 + L:] == 0).all()
    eds, _, _, _ = Oracle().align(texts, reads)
    assert 0.06 * L < np.mean(eds) < 0.14 * L


def _worker_edits(rank, world, port, q, ordered):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from scrooge_amd import api, distributed as sd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank "aligned" its own three reads; CIGARs (W=64, O=33) -> edit streams with the host encoder
        cigars = [["5=1X4=", "31=31=9=", "2I3="], ["1D30=1=1X29=11=", "", "31=1I30=9="]][rank]
        streams = [api.cigar_to_edit_stream(c) for c in cigars]
        n = 3
        total = sum((len(s) + 3) // 4 * 4 for s in streams)
        import re
        n_runs = [len(re.findall(r"[=XID]", c)) for c in cigars]
        g = sd.EditStreamGather(n, total, torch.device("cpu"), dst=0, depth=2, ordered=ordered,
                                total_runs=sum(n_runs) if ordered else None)
        assert g.with_counts == ordered and g.run_totals == ([8, 10] if ordered else [0, 0])
        order = [0, 1, 2] if ordered else [2, 0, 1]          # not ordered: the streams sit anywhere, offsets travel
        for k in range(3):
            g.finish(k)
            v = g.buffers(k)
            at = 0
            for i in order:
                s = streams[i]
                v["off"][i], v["len"][i] = at, len(s)
                if ordered:
                    v["cnt"][i] = n_runs[i]              # the run counts travel with ordered streams (one-pass decode on the root)
                v["stream"][at: at + len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8) if s else torch.zeros(0, dtype=torch.uint8)
                at += (len(s) + 3) // 4 * 4
            ed = torch.tensor([sum(1 for b in s if b >> 6) + k for s in streams], dtype=torch.int64)
            g.start(k, ed)
        g.finish_all()
        if rank == 0:
            out = []
            for r in range(world):
                v = g.results(2, r)
                raw = bytes(v["stream"].tolist())
                got = [raw[o: o + l] for o, l in zip(v["off"].tolist(), v["len"].tolist())]
                out.append((v["ed"].tolist(), got, g.totals[r], g.wire, v["cnt"].tolist() if ordered else None,
                            (g.o_len, g.o_cnt, g.head)))
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("ordered", [True, False])
def test_edit_stream_gather(ordered):
    """EditStreamGather: one fixed-size collective per step carrying scores, stream lengths (and offsets, when the
    streams are not in pair order) and the edit streams of every rank; what arrives on rank 0 decodes (host decoder)
    to each rank's CIGARs."""
    from scrooge_amd import api
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edits, args=(r, 2, port, q, ordered)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cigars = [["5=1X4=", "31=31=9=", "2I3="], ["1D30=1=1X29=11=", "", "31=1I30=9="]]
    read_len = [[10, 71, 5], [72, 0, 71]]
    for r in range(2):
        eds, streams, total, wire, cnt, parts = out[r]
        assert [api.edit_stream_to_cigar(s, L) for s, L in zip(streams, read_len[r])] == cigars[r]
        assert eds == [sum(1 for b in s if b >> 6) + 2 for s in streams]
        # every part of a slot starts at a multiple of 64 bytes: edit distances, lengths, run counts (or offsets), streams
        assert parts == (64, 128, 192) and wire == 192 + 64
        if ordered:
            import re
            assert cnt == [len(re.findall(r"[=XID]", c)) for c in cigars[r]]


def _worker_rotate(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from scrooge_amd import distributed as sd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 2
        g = sd.EditStreamGather(n, 8, torch.device("cpu"), dst="rotate", depth=2)
        seen = {}
        for k in range(5):
            g.finish(k)
            v = g.buffers(k)
            v["len"][:] = torch.tensor([3, 1], dtype=torch.int32)
            v["stream"][:8] = torch.tensor([10 * rank + k, 1, 2, 0, 99, 0, 0, 0], dtype=torch.uint8)
            g.start(k, torch.tensor([100 * rank + k, 7], dtype=torch.int64))
            assert g.root_of(k) == k % world
        g.finish_all()
        for k in (3, 4):                                   # the two steps whose buffers are still there
            if g.root_of(k) == rank:
                seen[k] = [(g.results(k, r)["ed"].tolist(), g.results(k, r)["stream"][:5].tolist(), g.results(k, r)["off"].tolist())
                           for r in range(world)]
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_edit_stream_gather_rotating_root():
    """dst="rotate": step k lands on rank k mod world — every rank receives, every rank sends."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rotate, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert list(got[0].keys()) == [4] and list(got[1].keys()) == [3]          # step 4 -> rank 0, step 3 -> rank 1
    for rank, k in ((0, 4), (1, 3)):
        for r in range(2):
            ed, stream, off = got[rank][k][r]
            assert ed == [100 * r + k, 7] and stream == [10 * r + k, 1, 2, 0, 99] and off == [0, 4]


def test_bench_self_launch_relays_the_ranks_exit_code():
    """`python bench.py --gpus 2` without a launcher (no WORLD_SIZE): the parent builds, then starts its two ranks as a fresh
    torch.distributed.run child — it never touches the GPU itself — and exits with the child's code.  Without a GPU the ranks
    refuse to run ("the HIP path has no CPU fallback"): what must come back is THEIR failure, promptly, not a hang and not the
    old "WORLD_SIZE != --gpus" refusal of the parent."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run (covered by tests/test_gpu_scale.py::test_bench_two_ranks_dry_run)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "640",
                          "--read-len", "500", "--cpu-seconds", "0"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr and "WORLD_SIZE" not in out.stderr.split("needs a GPU")[0][-300:]
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]


def test_bench_self_launch_refuses_more_ranks_than_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("SCRG_BENCH_DRYRUN", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--no-build"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "this node has" in out.stderr


def test_bench_rank_watchdog_ends_a_stalled_rank():
    """--deadline: a rank that is still running when the deadline passes (here: one that never returns, SCRG_BENCH_TEST_STALL; on
    hardware: one stalled in a collective) says which phase it was in and exits with code 124 by itself — no re-exec, no new
    GPU work."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_TEST_STALL="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-build", "--deadline", "3"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 124 and time.time() - t0 < 60
    assert "still running after --deadline 3 s (phase: test stall" in out.stderr


def test_bench_self_launch_ends_a_stalled_job():
    """`python bench.py --gpus 2` whose ranks stall: with the ranks' own watchdogs the job ends with their code; with the
    watchdogs off (a rank wedged so hard that no thread of it runs) the PARENT ends the child's whole process group at
    deadline + 30 s and exits with 124 — never by re-executing anything that touched a GPU (it starts a fresh child per
    attempt, and by default there is one attempt)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1", SCRG_BENCH_TEST_STALL="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-build", "--deadline", "5"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and time.time() - t0 < 120
    assert "still running after --deadline 5 s" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    # the ranks' watchdogs off: the parent's limit (deadline + 30 s) ends the whole process group
    env["SCRG_BENCH_TEST_STALL"] = "no-watchdog"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-build", "--deadline", "2"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 124 and 30 < time.time() - t0 < 150
    assert "did not finish within --deadline 2 s (+30): ending its process group" in out.stderr


# ---------------------------------------------------------------------------------------------------------------------
# Rehearsal at world sizes 4 and 8 (round 6): the first execution of this logic on eight ranks must not be the driver's run
# on hardware.  Same classes, same calls as bench.py's N > 1 step; gloo through the host.

def _spawn(world, target, extra=(), timeout=170):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=timeout) for _ in range(world)]
    finally:
        for p in procs:
            p.join(60)
    assert [p.exitcode for p in procs] == [0] * world
    return dict(got)


def _rank_streams(rank, k, n):
    """What rank `rank` "aligned" in step k: n streams of lengths that differ by rank, step and pair (some empty); the bytes say
    who wrote them."""
    lens = [(7 * rank + 3 * k + 5 * i) % 23 for i in range(n)]
    return [bytes(((rank << 4) | ((k + i + j) & 15)) for j in range(l)) for i, l in enumerate(lens)]


def _worker_wide(rank, world, port, q, dst, depth, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5
        cap = n * 24
        g = sd.EditStreamGather(n, cap, torch.device("cpu"), dst=dst, depth=depth, ordered=True, total_runs=n * 40)
        assert g.DEPTH == depth
        g.prime()                                           # (as bench.py does before anything is timed: a full-size exchange with every root)
        assert g.p2p == (os.environ.get("SCRG_GATHER_COLLECTIVE", "p2p") != "gather")
        seen = {}

        def look(k):
            if g.root_of(k) != rank:
                return
            per = []
            for r in range(world):
                v = g.results(k, r)
                raw = bytes(v["stream"].tolist())
                per.append((v["ed"].tolist(), [raw[o: o + l] for o, l in zip(v["off"].tolist(), v["len"].tolist())], v["cnt"].tolist()))
            seen[k] = per

        for k in range(steps):
            if k >= depth:
                g.finish(k)                                 # the buffers of step k - depth are free again ...
            else:
                g.finish(k)
            v = g.buffers(k)
            at = 0
            for i, s in enumerate(_rank_streams(rank, k, n)):
                v["len"][i] = len(s)
                v["cnt"][i] = len(s) + rank
                if s:
                    v["stream"][at: at + len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8)
                at += (len(s) + 3) // 4 * 4
            g.start(k, torch.tensor([1000 * rank + 10 * k + i for i in range(n)], dtype=torch.int64))
            if k >= depth - 1:
                g.finish(k + 1)                             # ... which is when step k - depth + 1's results are complete and still there
                look(k - depth + 1)
        g.finish_all()
        for k in range(max(0, steps - depth + 1), steps):
            look(k)
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("mode", ["p2p", "gather"])
@pytest.mark.parametrize("world,dst,depth", [(8, "rotate", 2), (8, "rotate", 4), (8, 0, 4), (4, "rotate", 4), (4, 0, 2)])
def test_edit_stream_gather_at_four_and_eight_ranks(world, dst, depth, mode, monkeypatch):
    """EditStreamGather as bench.py drives it — `depth` steps in flight, root fixed or rotating — on 4 and 8 ranks: every step's
    root ends up with every rank's scores, stream lengths, run counts and stream bytes of THAT step (ragged, some empty), looked
    at when the step's buffers are the oldest still alive, exactly as the decode of bench.py does."""
    # (mode: the step as point-to-point sends with the root's own slot produced in place — the default — and as one dist.gather)
    if mode == "gather" and (world, depth) not in ((8, 4), (4, 2)):
        pytest.skip("the collective form: two of the five shapes")
    monkeypatch.setenv("SCRG_GATHER_COLLECTIVE", mode)
    steps = 2 * world + 3
    got = _spawn(world, _worker_wide, (dst, depth, steps))
    n = 5
    checked = 0
    for k in range(steps):
        root = k % world if dst == "rotate" else 0
        for rank in range(world):
            if rank != root:
                assert k not in got[rank], (rank, k)
        per = got[root][k]
        for r in range(world):
            ed, streams, cnt = per[r]
            want = _rank_streams(r, k, n)
            assert ed == [1000 * r + 10 * k + i for i in range(n)] and streams == want and cnt == [len(s) + r for s in want], (k, r)
            checked += 1
    assert checked == steps * world


def _worker_sharded_wide(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, Q = _batch()
        T, Q = T * 3, Q * 3                                  # 33 pairs over 8 ranks: shards of 5 and 4, ragged lengths, empty reads
        res = sd.align_pairs_sharded(OracleAligner(), T, Q, dst=0)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_eight_rank_gather_restores_input_order():
    got = _spawn(8, _worker_sharded_wide)
    T, Q = _batch()
    want = OracleAligner().align_pairs(T * 3, Q * 3)
    assert got[0] == want and all(got[r] is None for r in range(1, 8))


def test_shard_plan_unequal_at_eight():
    rng = np.random.Generator(np.random.PCG64(3))
    lens = rng.integers(0, 20000, 1003).tolist()
    plan = sd.shard_plan(lens, 8)
    assert sorted(int(i) for p in plan for i in p) == list(range(1003))
    sizes = [len(p) for p in plan]
    assert max(sizes) - min(sizes) <= 1                       # round-robin after the length sort
    sums = [sum(lens[int(i)] for i in p) for p in plan]
    assert max(sums) - min(sums) <= max(lens)


@pytest.mark.timeout(300)
def test_bench_eight_ranks_stalled_job_ends_within_the_deadline():
    """`python bench.py --gpus 8` whose eight ranks stall: every rank's watchdog ends it with code 124 and the job ends well inside
    deadline + 30 s — on eight ranks as on two (the driver's 8-GPU run has a budget; a hang must not consume it)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRG_BENCH_DRYRUN="1", SCRG_BENCH_TEST_STALL="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--no-build", "--deadline", "6"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=280)
    assert out.returncode != 0 and time.time() - t0 < 150
    assert out.stderr.count("still running after --deadline 6 s") >= 1
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
