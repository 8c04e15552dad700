"""Multi-GPU sharding of a batch of independent pairs.

The path shards embarrassingly: pairs share nothing (src/genasm_cpu.cpp:451-455),
so there is no data-path collective.  One process per GPU aligns its shard; the
only exchange is the result gather to rank 0 (edit distances + CIGARs), done
with torch.distributed — backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.  The reference itself is single-GPU (GPU_ID 0,
src/genasm_gpu.cu:67); this file is new functionality, not a port.

Partitioning follows what the reference's callers do for load balance
(src/tests.cu:375-377): order pairs by read length, longest first, then deal
them round-robin so every rank sees the same length mix.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def shard_plan(read_lens, world):
    """-> list (per rank) of int64 index arrays into the input batch."""
    order = np.argsort(-np.asarray(read_lens, dtype=np.int64), kind="stable")
    return [order[r::world] for r in range(world)]


def _device_for_backend(device=None):
    if device is not None:
        return device
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_varlen(payload, dst=0, group=None):
    """Gather 1-D tensors of different lengths to `dst`.

    RCCL has no gatherv: lengths are all-gathered first, every rank pads to the
    maximum and one fixed-size gather moves the data.  Returns the list of
    trimmed tensors on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=payload.dtype, device=payload.device)
    send[: payload.numel()] = payload
    recv = [torch.empty(cap, dtype=payload.dtype, device=payload.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][: sizes[r]] for r in range(world)]


def align_pairs_sharded(aligner, texts, queries, dst=0, group=None, device=None, **params):
    """Every rank passes the full batch; rank r aligns plan[r] with `aligner`
    (anything with the Aligner.align_pairs signature) and the results are
    gathered to `dst`, which returns the alignments in input order
    (list of (cigar, edit_distance)); other ranks return None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = _device_for_backend(device)
    plan = shard_plan([len(q) for q in queries], world)
    mine = plan[rank]
    local = aligner.align_pairs([texts[i] for i in mine], [queries[i] for i in mine], **params)

    ed = torch.tensor([a[1] for a in local], dtype=torch.int64, device=device)
    blob = b"\n".join(a[0].encode() for a in local)
    cig = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device) if blob else \
        torch.zeros(0, dtype=torch.uint8, device=device)

    eds = gather_varlen(ed, dst=dst, group=group)
    cigs = gather_varlen(cig, dst=dst, group=group)
    if rank != dst:
        return None
    out = [None] * len(texts)
    for r in range(world):
        e = eds[r].cpu().tolist()
        c = bytes(cigs[r].cpu().numpy().tobytes()).decode().split("\n") if len(plan[r]) else []
        if len(plan[r]) and len(c) != len(plan[r]):
            raise RuntimeError("rank %d returned %d CIGARs for %d pairs" % (r, len(c), len(plan[r])))
        for k, i in enumerate(plan[r]):
            out[int(i)] = (c[k], e[k])
    return out


class ResultGather:
    """Fixed-size, multi-buffered gather of a batch's device results to rank `dst`
    (bench.py: the data, hence every size, is the same each step, so buffers are allocated once
    and no size exchange happens inside the timed region).

    Payload per rank: int64 edit distances [n], int32 run counts [n] and the dense runs, padded to the
    largest total over ranks.  With `packed=True` the runs travel as ONE byte each (scrg_compact_runs_packed:
    op << 6 | count, valid for W-O <= 63) and rank `dst` restores scrg_run pairs with scrg_unpack_runs once a
    step's gather has landed — half the bytes on the xGMI links into `dst`, which is what bounds N > 1 once a
    rank produces runs faster than a link carries them.  `start(k, ...)` enqueues the collectives
    asynchronously (RCCL runs them on its own stream, after the work already enqueued on the current
    stream), so the gather of step k overlaps the align kernels of the following steps; `finish(k)` makes the
    current stream wait for step k's gather (and unpacking) before its buffers are reused."""

    DEPTH = 2

    def __init__(self, n_pairs, total_runs, device, dst=0, group=None, depth=None, packed=False):
        self.group, self.dst = group, dst
        if depth is not None:
            self.DEPTH = max(1, int(depth))        # buffers in flight: one per pipelined step
        self.packed = bool(packed)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        t = torch.tensor([int(total_runs)], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(sizes, t, group=group)
        self.totals = [int(x.item()) for x in sizes]
        self.cap = max(max(self.totals), 1)
        d = self.DEPTH
        wire = (self.cap * (1 if self.packed else 2) + 7) // 8 * 8          # bytes on the wire per rank and step
        self.send_runs = [torch.zeros(wire, dtype=torch.uint8, device=device) for _ in range(d)]
        self.send_ed = [torch.zeros(n_pairs, dtype=torch.int64, device=device) for _ in range(d)]
        self.send_cnt = [torch.zeros(n_pairs, dtype=torch.int32, device=device) for _ in range(d)]
        self.recv_runs = self.recv_ed = self.recv_cnt = [None] * d
        self.runs16 = [None] * d                   # packed: the restored scrg_run bytes on dst
        self._unpacker = None
        if self.rank == dst:
            mk = lambda shape, dt: [[torch.empty(shape, dtype=dt, device=device) for _ in range(self.world)]
                                    for _ in range(d)]
            self.recv_runs = mk(wire, torch.uint8)
            self.recv_ed = mk(n_pairs, torch.int64)
            self.recv_cnt = mk(n_pairs, torch.int32)
            if self.packed:
                from . import api
                self.runs16 = mk(self.cap * 2 + 8, torch.uint8)
                # unpacking runs on its own stream and handle, ordered after the collective and before buffer reuse
                self._unpack_stream = torch.cuda.Stream(device=device)
                self._unpacker = api.Aligner(torch.device(device).index or 0)
                self._unpacker.set_stream(self._unpack_stream.cuda_stream)
        self.pending = [None] * d
        self.landed = [False] * d                  # a step's data is in recv_* but not unpacked yet
        # gloo cannot gather device tensors: stage through the host (the multi-rank dry run of bench.py on a
        # box with one GPU, SCRG_BENCH_DRYRUN=1; never the measured configuration)
        self.host_stage = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"

    def start(self, k, ed, n_runs):
        """`self.send_runs[k % DEPTH]` must already hold this rank's dense runs of step k (packed bytes if
        `packed`, scrg_run pairs otherwise)."""
        b = k % self.DEPTH
        self.send_ed[b].copy_(ed)
        self.send_cnt[b].copy_(n_runs)
        if self.host_stage:
            torch.cuda.current_stream().synchronize()
            for send, recv in ((self.send_ed[b], self.recv_ed[b]), (self.send_cnt[b], self.recv_cnt[b]),
                               (self.send_runs[b], self.recv_runs[b])):
                host = [torch.empty(send.shape, dtype=send.dtype) for _ in range(self.world)] if self.rank == self.dst else None
                dist.gather(send.cpu(), host, dst=self.dst, group=self.group)
                if host is not None:
                    for r in range(self.world):
                        recv[r].copy_(host[r])
            self.landed[b] = True
            return
        self.pending[b] = [
            dist.gather(self.send_ed[b], self.recv_ed[b], dst=self.dst, group=self.group, async_op=True),
            dist.gather(self.send_cnt[b], self.recv_cnt[b], dst=self.dst, group=self.group, async_op=True),
            dist.gather(self.send_runs[b], self.recv_runs[b], dst=self.dst, group=self.group, async_op=True)]
        self.landed[b] = True

    def finish(self, k):
        b = k % self.DEPTH
        if self.pending[b]:
            for w in self.pending[b]:
                w.wait()
            self.pending[b] = None
        if self.landed[b]:
            self.landed[b] = False
            if self._unpacker is not None:
                cur = torch.cuda.current_stream()
                self._unpack_stream.wait_stream(cur)           # (cur already waits for the collectives)
                with torch.cuda.stream(self._unpack_stream):
                    for r in range(self.world):
                        self._unpacker.unpack_runs(self.totals[r], self.recv_runs[b][r], self.runs16[b][r])
                cur.wait_stream(self._unpack_stream)

    def finish_all(self):
        for b in range(self.DEPTH):
            self.finish(b)

    def results(self, k, r):
        """(ed, counts, scrg_run bytes) of rank r for step k, on dst (after finish(k))."""
        b = k % self.DEPTH
        runs = self.runs16[b][r] if self.packed else self.recv_runs[b][r]
        return self.recv_ed[b][r], self.recv_cnt[b][r], runs[: 2 * self.totals[r]]


class EditStreamGather:
    """Gather of a batch's device results to rank `dst` with the CIGARs as EDIT STREAMS (one byte per edit and per window end, ~1.3 KB
    for a 10 kb read at 10 % error instead of 4.3 KB of scrg_run pairs), so that what a rank produces per second fits
    the one xGMI link it has to `dst`, and their DECODING back to scrg_run pairs on `dst` — what the
    reference delivers is CIGARs on the receiving side (src/genasm_gpu.cu:955-968), so the step is not finished before
    the runs exist there.

    One buffer per rank and step, ONE collective per step:
        ordered (streams in pair order at 4-byte aligned offsets — what scrg_align_device_edits + compaction gives):
            [ int32 edit distance x n | int32 stream length x n | int32 run count x n | streams ]       12 bytes per pair
        not ordered (scrg_encode_edit_stream places the streams in no particular order):
            [ int32 edit distance x n | int32 stream length x n | int64 stream offset x n | streams ]   16 bytes per pair
    (every part starts at a multiple of 64 bytes).  `buffers(k)` hands out the views the producer writes into ("len",
    "cnt", "stream", and "off" — a scratch tensor when the offsets do not travel); `start(k, ed)` adds the scores and
    enqueues the gather asynchronously (it overlaps the align kernels of the following steps); `finish(k)` makes the
    current stream wait for it — and for the decoding of what it brought — before the buffers are reused.

    Receiving side.  The slots of all ranks are ONE tensor per buffer set, so `decode_all(k, ...)` restores the runs of
    every rank's pairs with ONE scrg_decode_edit_stream launch (world x n pairs: enough wavefronts to fill the GPU) into
    one dense array per buffer set: run offsets are a prefix sum over the run counts that travelled with the streams,
    so there is no counting pass and no host synchronisation.  `start(k, ed, decode=(aligner, read_len, stride))` enqueues
    that on a stream of its own, ordered after the collective: inside whatever region the caller is timing.
    `results(k, r)` gives views of rank r's slot, `decoded(k)` the dense runs with their offsets and counts, and
    `decode(...)` (two passes, one slot) remains for streams that came without counts.  Sizes are exchanged once: the
    data, hence every size, is the same each step in bench.py.

    `dst` may be "rotate": step k is gathered to rank k mod world, so that every rank receives (and decodes) one step in
    `world`; every rank then holds receive buffers, and `results` / `decoded` of step k are valid on `root_of(k)`."""

    def __init__(self, n_pairs, stream_bytes, device, dst=0, group=None, depth=2, ordered=True, total_runs=None):
        self.group, self.n = group, int(n_pairs)
        self.rotate = dst == "rotate"
        self.dst = 0 if self.rotate else int(dst)
        self.DEPTH = max(1, int(depth))
        self.ordered = bool(ordered)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        t = torch.tensor([int(stream_bytes), int(total_runs or 0), int(n_pairs)], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(sizes, t, group=group)
        self.totals = [int(x[0].item()) for x in sizes]
        self.run_totals = [int(x[1].item()) for x in sizes]
        if any(int(x[2].item()) != self.n for x in sizes):
            raise ValueError("EditStreamGather: every rank must bring the same number of pairs (pad the last shard)")
        self.with_counts = self.ordered and total_runs is not None
        self.cap = (max(max(self.totals), 8) + 63) // 64 * 64
        n = self.n
        r64 = lambda x: (x + 63) // 64 * 64
        self.o_len = r64(4 * n)                               # byte offsets of the parts of a slot
        self.o_cnt = self.o_off = self.o_len + r64(4 * n)
        self.head = self.o_cnt + (r64(4 * n) if self.ordered else r64(8 * n))      # bytes of scalars in front of the streams
        self.wire = self.head + self.cap                     # bytes per rank and step on the link
        d = self.DEPTH
        self.send = [torch.zeros(self.wire, dtype=torch.uint8, device=device) for _ in range(d)]
        self.total = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(d)]
        self.off_scratch = [torch.zeros(n, dtype=torch.int64, device=device) for _ in range(d)] if self.ordered else None
        self.recv_all = [None] * d
        self.recv = [None] * d
        self.is_cuda = torch.device(device).type == "cuda"
        self.holds_results = self.rotate or self.rank == self.dst
        if self.holds_results:
            self.recv_all = [torch.zeros(self.world * self.wire + 64, dtype=torch.uint8, device=device) for _ in range(d)]
            self.recv = [[ra[r * self.wire: (r + 1) * self.wire] for r in range(self.world)] for ra in self.recv_all]
        self.pending = [None] * d
        self.host_stage = dist.get_backend(group) == "gloo" and self.is_cuda
        # How a step travels.  "p2p" (default): the step's root produces its own slot IN PLACE (buffers(k) hands out views of its
        # receive buffer) and posts one receive per peer, every other rank posts one send: nothing of the root's own 130 MB per step
        # goes through a collective — ncclGather copies the root's contribution to itself with a copy kernel, which on a one-rank
        # group is the whole "transfer" and cost 0.25 ms per step (scripts/r06_step_probe.sh).  "gather": one dist.gather per step,
        # as until round 5 (SCRG_GATHER_COLLECTIVE=gather; also what the host-staged gloo dry run of bench.py uses).
        self.p2p = os.environ.get("SCRG_GATHER_COLLECTIVE", "p2p") != "gather" and not self.host_stage
        # decoding on the receiving rank: one dense array, offsets, counts and an error counter per buffer set
        self.dense = [None] * d
        self.dec = [None] * d
        self.dec_event = [None] * d
        self.dec_streams = {}                 # one decode stream per buffer set: the decodes of consecutive steps overlap

    def root_of(self, k):
        """The rank step k is gathered to."""
        return k % self.world if self.rotate else self.dst

    def prime(self):
        """One full-size gather to every root that will be used, outside any timed region: RCCL sets up its
        point-to-point connections (per pair of ranks, and channel by channel as message sizes ask for them) on first
        use, and with a rotating root the first `world` steps would otherwise each pay for seven new ones."""
        if self.host_stage or os.environ.get("SCRG_BENCH_NOCOLL") == "1":
            return
        if self.p2p and self.world > 1:
            # (a rank on which the point-to-point form raises — a backend without grouped send / receive — takes every rank back to
            # the one-collective form: the verdict is agreed on with an all_reduce, outside any timed region)
            ok = 1
            try:
                for dst in (range(self.world) if self.rotate else [self.dst]):
                    for w in self._post(0, dst):
                        w.wait()
            except Exception as e:                            # noqa: BLE001
                ok = 0
                print("EditStreamGather: point-to-point form failed on rank %d (%s: %s); using dist.gather" % (self.rank, type(e).__name__, e),
                      file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if int(flag.item()) == 1:
                if self.send[0].is_cuda:
                    torch.cuda.synchronize()
                return
            self.p2p = False
        for dst in (range(self.world) if self.rotate else [self.dst]):
            if not self.p2p:
                dist.gather(self.send[0], self.recv[0] if self.rank == dst else None, dst=dst, group=self.group)
        if self.send[0].is_cuda:
            torch.cuda.synchronize()

    def _views(self, buf):
        n = self.n
        v = {"ed": buf[: 4 * n].view(torch.int32), "len": buf[self.o_len: self.o_len + 4 * n].view(torch.int32),
             "stream": buf[self.head:]}
        if self.ordered:
            v["cnt"] = buf[self.o_cnt: self.o_cnt + 4 * n].view(torch.int32)
        else:
            v["off"] = buf[self.o_off: self.o_off + 8 * n].view(torch.int64)
        return v

    def _post(self, b, dst):
        """The point-to-point form of the gather of buffer set b to rank dst: -> the work handles (none in a one-rank group)."""
        if self.rank == dst:
            ops = [dist.P2POp(dist.irecv, self.recv[b][r], r, self.group, b) for r in range(self.world) if r != dst]
        else:
            ops = [dist.P2POp(dist.isend, self.send[b], dst, self.group, b)]
        return dist.batch_isend_irecv(ops) if ops else []

    def _slot(self, k):
        """Where this rank's contribution to step k is produced: its send buffer, or — as the step's root in p2p mode — its own
        slot of the receive buffer."""
        b = k % self.DEPTH
        return self.recv[b][self.rank] if (self.p2p and self.rank == self.root_of(k)) else self.send[b]

    def buffers(self, k):
        b = k % self.DEPTH
        v = self._views(self._slot(k))
        v["total"] = self.total[b]
        if self.ordered:
            v["off"] = self.off_scratch[b]
        return v

    def start(self, k, ed, decode=None):
        """The streams, their lengths and run counts (or offsets, if they travel) of step k must already be in
        `buffers(k)` (enqueued on the current stream).  decode = (aligner or a list of aligners — one per buffer set, so
        that the decode launches of consecutive steps run side by side —, read_len tensor, read_len_stride, params dict):
        the root also enqueues the decoding of all slots (decode_all), ordered after the collective.  read_len holds
        world * n lengths in slot order when the stride is 1 (unequal shards: a padding pair has length 0), one length
        for all when it is 0."""
        b = k % self.DEPTH
        dst = self.root_of(k)
        self._views(self._slot(k))["ed"].copy_(ed)           # (int64 -> int32)
        if self.host_stage:
            torch.cuda.current_stream().synchronize()
            host = [torch.empty(self.wire, dtype=torch.uint8) for _ in range(self.world)] if self.rank == dst else None
            dist.gather(self.send[b].cpu(), host, dst=dst, group=self.group)
            if host is not None:
                for r in range(self.world):
                    self.recv[b][r].copy_(host[r])
        elif os.environ.get("SCRG_BENCH_NOCOLL") == "1":      # experiment: everything but the collective
            pass
        elif self.p2p:
            self.pending[b] = self._post(b, dst) or None
        else:
            self.pending[b] = dist.gather(self.send[b], self.recv[b] if self.rank == dst else None, dst=dst, group=self.group,
                                          async_op=True)
        if decode is not None and self.rank == dst:
            self.decode_all(k, *decode)

    def _wait(self, b):
        if self.pending[b] is not None:
            for w in (self.pending[b] if isinstance(self.pending[b], (list, tuple)) else [self.pending[b]]):
                w.wait()                                      # the CURRENT stream waits for the collective
            self.pending[b] = None

    def finish(self, k):
        b = k % self.DEPTH
        self._wait(b)
        if self.dec_event[b] is not None:                     # ... and for the decoding of what it brought
            if self.is_cuda:
                torch.cuda.current_stream().wait_event(self.dec_event[b])
            self.dec_event[b] = None

    def finish_all(self):
        for b in range(self.DEPTH):
            self.finish(b)

    def results(self, k, r):
        """Views (ed, len, cnt, off, stream) of what rank r sent for step k, on root_of(k) (after finish(k)); `off` is
        re-derived from the lengths when the streams are ordered."""
        v = self._views(self.recv[k % self.DEPTH][r])
        if self.ordered:
            r4 = (v["len"].to(torch.int64) + 3) & -4
            v["off"] = torch.cumsum(r4, 0) - r4
        return v

    def decode_all(self, k, aligner, read_len, read_len_stride, params=None):
        """Restores the scrg_run pairs of EVERY rank's pairs of step k on its root with one launch (needs the run
        counts on the wire: ordered streams built with total_runs).  read_len: int64 device tensor for the world * n
        pairs in slot order, with its stride (0: one length for all) — see scrg_decode_edit_stream.  Enqueued on the
        gather's own decode stream, ordered after the collective of step k; `decoded(k)` gives the result."""
        if not self.with_counts:
            raise ValueError("decode_all needs ordered streams with run counts (total_runs=...)")
        b = k % self.DEPTH
        n, W = self.n, self.world
        # measurement aid (scripts/root_load_probe.sh): a one-rank group decodes its slot `sim` times in the one launch, i.e.
        # does per step what the root of a `sim`-rank job does (the same stream bytes are read, `sim` dense copies written)
        sim = int(os.environ.get("SCRG_GATHER_SIMULATE_SENDERS", "1")) if W == 1 else 1
        # ... and with SCRG_GATHER_SIMULATE_ROTATE=1 only every `sim`-th step: what EVERY rank of a `sim`-rank job does when the
        # root rotates (it is the root of one step in `sim`)
        if sim > 1 and os.environ.get("SCRG_GATHER_SIMULATE_ROTATE") == "1" and k % sim != 0:
            self._wait(b)
            return
        if self.dense[b] is None:
            cap_runs = sum(self.run_totals) * sim
            self.dense[b] = torch.zeros(cap_runs * 2 + 64, dtype=torch.uint8, device=self.device)
            self.dec[b] = {"bad": torch.zeros(1, dtype=torch.int32, device=self.device)}
        # `aligner` may be a list of handles, one per buffer set: a decode launch is bound by its longest streams (a lane
        # walks its stream sequentially — in the lane-per-pair decoder), so the launches of consecutive steps run side by side, each on the stream and
        # with the handle (sort workspace) of its buffer set
        if isinstance(aligner, (list, tuple)):
            sb = b % len(aligner)
            aligner = aligner[sb]
        else:
            sb = 0
        if self.is_cuda and sb not in self.dec_streams:
            self.dec_streams[sb] = torch.cuda.Stream(device=self.device)
        dec_stream = self.dec_streams.get(sb)
        cur = torch.cuda.current_stream() if self.is_cuda else None
        ctx = torch.cuda.stream(dec_stream) if self.is_cuda else _NullCtx()
        if self.is_cuda:
            dec_stream.wait_stream(cur)
        with ctx:
            self._wait(b)                                     # the decode stream waits for the collective, nobody else
            slots = self.recv_all[b][: W * self.wire].view(W, self.wire)
            ln = slots[:, self.o_len: self.o_len + 4 * n].contiguous().view(torch.int32).reshape(-1)
            cnt = slots[:, self.o_cnt: self.o_cnt + 4 * n].contiguous().view(torch.int32).reshape(-1)
            r4 = ((ln.to(torch.int64) + 3) & -4).view(W, n)
            off = torch.cumsum(r4, 1) - r4 + (torch.arange(W, dtype=torch.int64, device=self.device) * self.wire + self.head).view(W, 1)
            off = off.reshape(-1)
            if sim > 1:
                ln, cnt, off = ln.repeat(sim), cnt.repeat(sim), off.repeat(sim)
            c64 = cnt.to(torch.int64)
            doff = torch.cumsum(c64, 0) - c64
            d = self.dec[b]
            d.update(len=ln, cnt=cnt, off=off, run_off=doff)
            d["bad"].zero_()
            # the caller's handle enqueues on the decode stream for this one call only: whatever it aligned on before, it
            # aligns on afterwards (a caller may pass the handle it also aligns with)
            saved = aligner.stream_setting if self.is_cuda else None
            if self.is_cuda:
                aligner.set_stream(dec_stream.cuda_stream)
            try:
                if os.environ.get("SCRG_GATHER_PROBE") != "no-decode-kernel":      # (measurement aid: everything of the step but the decode launch)
                    aligner.decode_edit_stream(W * n * sim, self.recv_all[b], d["off"], ln, read_len, read_len_stride, doff, self.dense[b],
                                               cnt, d["bad"], **(params or {}))
            finally:
                if self.is_cuda:
                    aligner.restore_stream(saved)
            if self.is_cuda:
                self.dec_event[b] = torch.cuda.Event()
                self.dec_event[b].record(dec_stream)

    def decoded(self, k):
        """-> dict(runs: uint8 [2 * total runs], run_off: int64 [world * n], cnt: int32 [world * n], bad: int32 [1]) of
        step k on its root, pairs in slot order (rank 0's n pairs, rank 1's, ...).  Valid after finish(k)."""
        b = k % self.DEPTH
        if self.dec[b] is None:
            return None
        d = dict(self.dec[b])
        d["runs"] = self.dense[b]
        if self.is_cuda:
            # the tensors were made (and are rewritten, DEPTH steps later) on the decode stream; whoever reads them on the
            # current stream must be known to the caching allocator
            cur = torch.cuda.current_stream()
            for v in d.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        return d

    def decode(self, aligner, k, r, read_len, read_len_stride, **params):
        """scrg_run bytes of rank r's pairs of step k on dst -> (dense uint8 tensor, run counts int32, run offsets
        int64, number of pairs whose stream is not an alignment of a read of that length).  `read_len`: int64 device
        tensor, see scrg_decode_edit_stream.  Two passes: count, then decode (for streams that travel without counts)."""
        v = self.results(k, r)
        dev = v["len"].device
        cnt = torch.zeros(self.n, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        sync = torch.cuda.synchronize if cnt.is_cuda else (lambda: None)     # the handle's stream need not be torch's
        sync()
        aligner.decode_edit_stream(self.n, v["stream"], v["off"], v["len"], read_len, read_len_stride, None, None,
                                   cnt, bad, **params)
        sync()
        if int(bad.item()):
            return None, cnt, None, int(bad.item())
        cnt64 = cnt.to(torch.int64)
        off = torch.cumsum(cnt64, 0) - cnt64
        dense = torch.zeros(int(cnt64.sum().item()) * 2 + 64, dtype=torch.uint8, device=dev)
        sync()
        aligner.decode_edit_stream(self.n, v["stream"], v["off"], v["len"], read_len, read_len_stride, off, dense,
                                   cnt, bad, **params)
        sync()
        return dense, cnt, off, int(bad.item())


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
