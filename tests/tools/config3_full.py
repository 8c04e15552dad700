"""BASELINE configs[3] at its stated size on the one GPU there is: 1 M x 10 kb ONT-error pairs as EIGHT SHARDS of 125 000
(every shard its own seed = what rank r of an 8-GPU job generates), each through the real N > 1 step — the align kernel
writing edit streams + run counts (scrg_align_device_edits), compaction of the streams into the wire buffer, the RCCL
gather (a one-rank group: the collective runs, onto itself) — into slot r of ONE eight-slot receive buffer, which the
root's decoder (scrg_decode_edit_stream, ONE launch over all 8 x 125 000 pairs, as EditStreamGather.decode_all issues it)
turns into one dense scrg_run array: what rank 0 of the 8-GPU job holds when the clock stops.

Held to, for ALL 1 000 000 pairs (no sampling):
  * the size-independent properties of src/tests.cu:27-169 (tests/cigar_check.py: validate_batch — ops, counts, the read
    consumed exactly, the text not overrun, '=' / 'X' against the sequences, edit distance == non-match columns);
  * edit distance == number of edit bytes of the pair's stream; run count on the wire == runs decoded;
  * the decoded runs == the runs the RUNS kernel (scrg_align_device + compaction) writes for the same shard, byte for byte;
and `--ref-pairs` pairs spread evenly over all eight shards run for run against the reference CPU path itself
(oracle/_ref/libgenasm_ref.so = the unmodified src/genasm_cpu.cpp; the restatement where that build is absent).

With --multi the same 1 M pairs also go through ONE scrg_align_pairs_multi call with eight logical devices (host
pointers in, runs out: chunk k to device k mod 8, every device copying its own chunks back) and must give the same
edit distances and runs as the device path above.

Prints one JSON line; exit code 0 only if every check held.  Reference: src/genasm_cpu.cpp:411-438 (the window chain, ~330
windows per pair here), src/tests.cu:375-377 (length sort; all reads have one length here)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=125000, help="pairs per shard")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--profile", default="ont")
    ap.add_argument("--ref-pairs", type=int, default=20000, help="pairs compared run for run with the reference CPU path")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--multi", action="store_true", help="also one scrg_align_pairs_multi call over all pairs, eight logical devices")
    ap.add_argument("--no-rccl", action="store_true", help="skip the collective (debugging without RCCL)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import bench
    import scrooge_amd
    from scrooge_amd import synth
    from scrooge_amd.distributed import EditStreamGather
    from tests.cigar_check import validate_batch

    scrooge_amd.build_library()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if not args.no_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    S, n, L = args.shards, args.pairs, args.read_len
    err, ratio = synth.PROFILES[args.profile]
    G = scrooge_amd.api.GROUP
    cap = (2 * L + 8 + 15) // 16 * 16
    al = scrooge_amd.Aligner(0)
    al.set_stream(torch.cuda.current_stream().cuda_stream)
    kw = dict(text_stride_words=G, read_stride_words=G)
    out = {"workload": "BASELINE configs[3]: %d x %d bp %s-error pairs as %d shards of %d on one GPU" % (S * n, L, args.profile, S, n),
           "checks": {}}
    t_start = time.time()
    rows_all, ed_all, cnt_all, dense_runs_kernel = [], [], [], []
    recv_slots = None
    gather = None
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    slices = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
    ed = torch.empty(n, dtype=torch.int64, device=dev)
    ln = torch.empty(n, dtype=torch.int32, device=dev)
    rc = torch.empty(n, dtype=torch.int32, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    align_ms = []
    for s in range(S):
        rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, args.seed + 1000 * s, dev)      # (bench.py: rank r's seed)
        row_words = tw + rw
        seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
        al.pack_planar_groups(rows.view(-1), n, row_words, seq, bad)
        first = (idx // G) * row_words * G + idx % G
        desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L),
                            idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        # (1) the runs kernel: the formulation the decoded slot is compared with
        al.align_device(n, seq, desc, slices, ed, rc, st, **kw)
        c64 = rc.to(torch.int64)
        off = torch.cumsum(c64, 0) - c64
        dense = torch.empty(int(c64.sum().item()) * 2 + 64, dtype=torch.uint8, device=dev)
        al.compact_runs(n, desc, slices, rc, off, dense)
        torch.cuda.synchronize()
        assert int(st.max().item()) == 0 and int(bad.item()) == 0
        ed_runs_kernel, cnt_runs_kernel = ed.clone(), rc.clone()
        # (2) the N > 1 step: edit streams + run counts, compaction into the wire buffer, the collective
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        al.align_device_edits(n, seq, desc, slices, ed, ln, st, rc, **kw)
        e1.record()
        torch.cuda.synchronize()
        align_ms.append(e0.elapsed_time(e1))
        assert int(st.max().item()) == 0
        assert torch.equal(ed, ed_runs_kernel) and torch.equal(rc, cnt_runs_kernel), "shard %d: the two kernels disagree on scores / run counts" % s
        r4 = (ln.to(torch.int64) + 3) & -4
        boff = torch.cumsum(r4, 0) - r4
        if gather is None:
            # sized once, generously (what bench.py exchanges once): 1.25 x this shard's stream bytes and run total
            stream_bytes = int(r4.sum().item()) * 5 // 4
            if args.no_rccl:
                class _Local:            # the wire layout without a process group (debugging)
                    pass
                gather = _Local()
                r64 = lambda x: (x + 63) // 64 * 64
                gather.o_len, gather.o_cnt = r64(4 * n), 2 * r64(4 * n)
                gather.head = 3 * r64(4 * n)
                gather.wire = gather.head + r64(stream_bytes)
                gather.send = [torch.zeros(gather.wire, dtype=torch.uint8, device=dev)]
            else:
                gather = EditStreamGather(n, stream_bytes, dev, dst=0, depth=2, ordered=True, total_runs=int(c64.sum().item()) * 5 // 4)
                gather.prime()
            recv_slots = torch.zeros(S * gather.wire + 64, dtype=torch.uint8, device=dev)
        assert int(r4.sum().item()) <= gather.wire - gather.head, "stream bytes exceed the wire buffer"
        if args.no_rccl:
            buf = gather.send[0]
            buf[gather.o_len: gather.o_len + 4 * n].view(torch.int32).copy_(ln)
            buf[gather.o_cnt: gather.o_cnt + 4 * n].view(torch.int32).copy_(rc)
            al.compact_runs(n, desc, slices, (r4 >> 1).to(torch.int32), boff >> 1, buf[gather.head:])
            buf[: 4 * n].view(torch.int32).copy_(ed)
            torch.cuda.synchronize()
            recv_slots[s * gather.wire: (s + 1) * gather.wire].copy_(buf)
        else:
            g = gather.buffers(s)
            g["len"].copy_(ln)
            g["cnt"].copy_(rc)
            al.compact_runs(n, desc, slices, (r4 >> 1).to(torch.int32), boff >> 1, g["stream"])
            gather.start(s, ed)                                   # the RCCL gather of step s (asynchronous)
            gather.finish(s)                                      # the current stream waits for it
            recv_slots[s * gather.wire: (s + 1) * gather.wire].copy_(gather.recv[s % gather.DEPTH][0])
        torch.cuda.synchronize()
        rows_all.append((rows, tw, text_len))
        ed_all.append(ed_runs_kernel)
        cnt_all.append(cnt_runs_kernel)
        dense_runs_kernel.append((dense, off))
        del seq, desc
    out["align_edits_kernel_ms_per_shard"] = align_ms
    # ---- the root's decode: ONE launch over all S slots (EditStreamGather.decode_all's arithmetic, S slots) ----
    slots = recv_slots[: S * gather.wire].view(S, gather.wire)
    ln_w = slots[:, gather.o_len: gather.o_len + 4 * n].contiguous().view(torch.int32).reshape(-1)
    cnt_w = slots[:, gather.o_cnt: gather.o_cnt + 4 * n].contiguous().view(torch.int32).reshape(-1)
    ed_w = slots[:, : 4 * n].contiguous().view(torch.int32).reshape(-1).to(torch.int64)
    r4 = ((ln_w.to(torch.int64) + 3) & -4).view(S, n)
    soff = (torch.cumsum(r4, 1) - r4 + (torch.arange(S, dtype=torch.int64, device=dev) * gather.wire + gather.head).view(S, 1)).reshape(-1)
    c64 = cnt_w.to(torch.int64)
    doff = torch.cumsum(c64, 0) - c64
    total_runs = int(c64.sum().item())
    dense_all = torch.zeros(total_runs * 2 + 64, dtype=torch.uint8, device=dev)
    nbad = torch.zeros(1, dtype=torch.int32, device=dev)
    rl = torch.tensor([L], dtype=torch.int64, device=dev)
    cnt_dec = cnt_w.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    al.decode_edit_stream(S * n, recv_slots, soff, ln_w, rl, 0, doff, dense_all, cnt_dec, nbad, **kw)
    e1.record()
    torch.cuda.synchronize()
    out["decode_all_slots_ms"] = e0.elapsed_time(e1)
    out["total_runs"] = total_runs
    ck = out["checks"]
    ck["streams_decode"] = int(nbad.item()) == 0
    ck["run_counts_on_the_wire_are_the_decoded_counts"] = bool(torch.equal(cnt_dec, cnt_w))
    # edit distance == number of edit bytes of the stream (one byte per edit; the bytes with op 0 are window ends)
    is_edit = (recv_slots >= 64).to(torch.int32)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(is_edit, 0, dtype=torch.int64)])
    del is_edit
    ck["edit_distance_is_the_streams_edit_count"] = bool(torch.equal(csum[soff + ln_w.to(torch.int64)] - csum[soff], ed_w))
    del csum
    ck["gathered_scores_are_the_kernels"] = bool(torch.equal(ed_w, torch.cat(ed_all)) and torch.equal(cnt_w, torch.cat(cnt_all)))
    # decoded runs == the runs kernel's, shard by shard, byte for byte; properties of every pair
    same, n_bad_pairs = True, 0
    for s in range(S):
        dense, off = dense_runs_kernel[s]
        a0 = int(doff[s * n].item())
        tr = int(c64[s * n: (s + 1) * n].sum().item())
        same = same and bool(torch.equal(dense_all[2 * a0: 2 * (a0 + tr)], dense[: 2 * tr]))
        rows, tw, text_len = rows_all[s]
        badp = validate_batch(torch, rows, 0, text_len, tw * 32, L, dense_all, doff[s * n: (s + 1) * n], cnt_dec[s * n: (s + 1) * n],
                              ed_w[s * n: (s + 1) * n], chunk_pairs=max(256, 125000000 // (L + L // 8)))
        n_bad_pairs += int(badp.numel())
    ck["decoded_runs_equal_the_runs_kernels"] = same
    ck["properties_hold_for_every_pair"] = n_bad_pairs == 0
    out["pairs_violating_a_property"] = n_bad_pairs
    out["mean_edit_distance"] = float(ed_w.double().mean().item())
    # ---- the reference CPU path on pairs spread over all shards ----
    from oracle.pyoracle import Oracle, Reference
    per = max(1, args.ref_pairs // S)
    stride = max(1, n // per)
    cores = bench.usable_cores()
    ref_ok, ref_n, against = True, 0, None
    for s in range(S):
        rows, tw, text_len = rows_all[s]
        pick = torch.arange(per, device=dev) * stride
        sample = rows[pick].cpu().numpy()
        if Reference.available():
            e_cpu, off_cpu, runs_cpu, _ = Reference().align_rows(sample, 0, text_len, tw * 32, L, threads=cores)
            against = "reference genasm_cpu.cpp (oracle/_ref)"
        else:
            e_cpu, off_cpu, runs_cpu, _, _ = Oracle(allow_compile=False).align_rows(sample, 0, text_len, tw * 32, L, threads=cores)
            against = "oracle/liboracle.so (restatement)"
        gp = pick + s * n
        cnt = c64[gp].cpu().numpy().astype(np.uint64)
        off_gpu = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt, dtype=np.uint64)])
        seg = torch.repeat_interleave(torch.arange(per, device=dev), c64[gp])
        within = torch.arange(int(off_gpu[per]), device=dev) - torch.from_numpy(off_gpu[:per].astype(np.int64)).to(dev)[seg]
        src = (doff[gp][seg] + within) * 2
        runs_gpu = torch.stack([dense_all[src], dense_all[src + 1]], dim=1).cpu().numpy()
        ok = bool((ed_w[gp].cpu().numpy() == e_cpu).all() and (off_gpu == off_cpu).all() and np.array_equal(runs_gpu, runs_cpu))
        ref_ok = ref_ok and ok
        ref_n += per
    ck["run_for_run_with_the_reference"] = ref_ok
    out["reference_pairs"] = {"checked": ref_n, "spread": "every %d-th pair of each of the %d shards" % (stride, S), "against": against}
    # ---- the same pairs through ONE scrg_align_pairs_multi call, eight logical devices ----
    if args.multi:
        t0 = time.time()
        tw, text_len = rows_all[0][1], rows_all[0][2]
        host = np.empty((S * n, rows_all[0][0].shape[1]), dtype=np.uint8)
        for s in range(S):
            host[s * n: (s + 1) * n] = rows_all[s][0].cpu().numpy()
        t1 = time.time()
        res = al.align_pairs_rows(host, 0, text_len, tw * 32, L, devices=[0] * S, outputs=2)        # SCRG_OUT_RUNS
        t2 = time.time()
        ed_h = torch.from_numpy(res["edit_distance"]).to(dev)
        ro = res["run_offset"].astype(np.int64)
        ck["multi_edit_distances_equal"] = bool(torch.equal(ed_h, ed_w))
        ck["multi_run_counts_equal"] = bool(np.array_equal(np.diff(ro), c64.cpu().numpy()))
        runs_h = torch.from_numpy(res["runs"].reshape(-1)).to(dev)
        ck["multi_runs_equal"] = bool(runs_h.numel() == 2 * total_runs and torch.equal(runs_h, dense_all[: 2 * total_runs]))
        ck["multi_no_overflow"] = bool((res["status"] == 0).all())
        out["multi"] = {"devices": [0] * S, "call_seconds": t2 - t1, "staging_seconds": t1 - t0,
                        "library_total_ms": al.last_timing["total_ns"] / 1e6, "pairs_per_s_pcie_inclusive": S * n / max(1e-9, al.last_timing["total_ns"] * 1e-9)}
        del host, res
    out["seconds"] = time.time() - t_start
    out["ok"] = all(ck.values())
    print(json.dumps(out))
    if not args.no_rccl:
        dist.destroy_process_group()
    al.close()
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
