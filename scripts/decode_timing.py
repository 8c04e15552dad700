#!/usr/bin/env python3
"""Times scrg_decode_edit_stream on the bench workload (GPU box): one slot of `--pairs` pairs and `--slots` slots in one
launch (what the gathering rank of an N-GPU job decodes per step), count-only and the one-pass decode into a dense array,
and checks the decoded runs against the runs the align kernel writes itself.

    python3 scripts/decode_timing.py [--pairs 100000] [--read-len 10000] [--slots 8] [--reps 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--profile", default="ont")
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--busy", type=int, default=60, help="align launches enqueued in front of every timed series (clock warm-up)")
    ap.add_argument("--probe", action="store_true", help="library built with -DSCRG_DEC_PROBE: print the kernel's cycle counters")
    ap.add_argument("--buffer-per-pair", type=int, default=0, help="make the stream buffer at least this many bytes per pair (a capacity-sized buffer)")
    ap.add_argument("--W", type=int, default=64)
    ap.add_argument("--O", type=int, default=33)
    args = ap.parse_args()
    import torch
    import scrooge_amd
    from scrooge_amd import synth
    import bench

    scrooge_amd.build_library()
    dev = torch.device("cuda", 0)
    al = scrooge_amd.Aligner(0, W=args.W, O=args.O)
    al.set_stream(torch.cuda.current_stream().cuda_stream)
    n, L = args.pairs, args.read_len
    err, ratio = synth.PROFILES[args.profile]
    rows, tw, rw, text_len = bench.device_pairs(torch, n, L, err, ratio, 42, dev)
    G = scrooge_amd.api.GROUP
    row_words = tw + rw
    seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    al.pack_planar_groups(rows.view(-1), n, row_words, seq, bad)
    del rows
    cap = (2 * L + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    first = (idx // G) * row_words * G + idx % G
    desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L),
                        idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    kw = dict(text_stride_words=G, read_stride_words=G)
    slices = torch.empty(n * cap * 2, dtype=torch.uint8, device=dev)
    ed = torch.empty(n, dtype=torch.int64, device=dev)
    ln = torch.empty(n, dtype=torch.int32, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    # reference: the runs kernel's own output, compacted
    al.align_device(n, seq, desc, slices, ed, cnt, st, **kw)
    c64 = cnt.to(torch.int64)
    roff = torch.cumsum(c64, 0) - c64
    total_runs = int(c64.sum().item())
    want = torch.zeros(total_runs * 2 + 64, dtype=torch.uint8, device=dev)
    al.compact_runs(n, desc, slices, cnt, roff, want)
    cnt_runs = cnt.clone()
    # streams + run counts from the edit-stream kernel, gathered in pair order at 4-byte aligned offsets
    cnt2 = torch.empty(n, dtype=torch.int32, device=dev)
    al.align_device_edits(n, seq, desc, slices, ed, ln, st, cnt2, **kw)
    torch.cuda.synchronize()
    assert torch.equal(cnt2, cnt_runs), "run counts of the edit-stream kernel differ from the runs kernel's"
    r4 = (ln.to(torch.int64) + 3) & -4
    boff = torch.cumsum(r4, 0) - r4
    sbytes = int(r4.sum().item())
    slot_bytes = (sbytes + 63) // 64 * 64
    S = args.slots
    stream = torch.zeros(max(S * slot_bytes, S * n * args.buffer_per_pair) + 64, dtype=torch.uint8, device=dev)
    al.compact_runs(n, desc, slices, (r4 >> 1).to(torch.int32), boff >> 1, stream)
    for k in range(1, S):
        stream[k * slot_bytes: k * slot_bytes + sbytes].copy_(stream[:sbytes])
    off_all = torch.cat([boff + k * slot_bytes for k in range(S)])
    len_all = ln.repeat(S)
    cnt_all = cnt_runs.repeat(S)
    c64a = cnt_all.to(torch.int64)
    doff_all = torch.cumsum(c64a, 0) - c64a
    dense = torch.zeros(S * total_runs * 2 + 64, dtype=torch.uint8, device=dev)
    rl = torch.tensor([L], dtype=torch.int64, device=dev)
    nbad = torch.zeros(32, dtype=torch.int32, device=dev)        # ([2..13]: counters of a -DSCRG_DEC_PROBE build)
    out_cnt = torch.empty(S * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        # the clocks of an idle GPU take milliseconds to come up: keep it busy with align launches right up to the timed launches
        for _ in range(args.busy):
            al.align_device(n, seq, desc, slices, ed, cnt, st, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.reps

    nocheck = bool(os.environ.get("SCRG_DEC_NOCHECK"))          # (probe builds with parts of the kernel switched off)
    res = {"pairs_per_slot": n, "read_len": L, "slots": S, "stream_bytes_per_pair": sbytes / n, "runs_per_pair": total_runs / n,
           "buffer_bytes_per_pair": stream.numel() / (S * n), "SCRG_DEC_KERNEL": os.environ.get("SCRG_DEC_KERNEL")}
    for slots in sorted(set([1, S])):
        m = slots * n
        t_count = timed(lambda: al.decode_edit_stream(m, stream, off_all, len_all, rl, 0, None, None, out_cnt, nbad, **kw))
        assert nocheck or (torch.equal(out_cnt[:m], cnt_all[:m]) and int(nbad[0].item()) == 0)
        t_dec = timed(lambda: al.decode_edit_stream(m, stream, off_all, len_all, rl, 0, doff_all, dense, cnt_all, nbad, **kw))
        torch.cuda.synchronize()
        assert nocheck or int(nbad[0].item()) == 0
        for k in range(slots):
            assert nocheck or torch.equal(dense[2 * k * total_runs: 2 * (k + 1) * total_runs], want[: 2 * total_runs]), "slot %d differs" % k
        gb = (slots * (sbytes + 2.0 * total_runs)) / 1e9
        res["slots_%d" % slots] = {"count_only_ms": t_count, "decode_ms": t_dec, "decode_ms_per_slot": t_dec / slots,
                                   "decode_M_pairs_per_s": m / t_dec / 1e3, "algorithmic_GB": gb, "GB_per_s": gb / (t_dec * 1e-3)}
    if args.probe:
        for store in (False, True):
            nbad.zero_()
            if store:
                al.decode_edit_stream(n, stream, off_all, len_all, rl, 0, doff_all, dense, cnt_all, nbad, **kw)
            else:
                al.decode_edit_stream(n, stream, off_all, len_all, rl, 0, None, None, out_cnt, nbad, **kw)
            torch.cuda.synchronize()
            v = nbad[2:22].view(torch.int64).cpu().tolist()
            w = (n + 63) // 64
            first = (1 << 62) - v[7]
            res["probe_store" if store else "probe_count"] = {
                "waves": w, "wave_life_us": v[5] / w / 100.0, "shader_clock_ghz": v[4] / max(1, v[5]) * 0.1,
                "last_start_us": (v[6] - first) / 100.0, "first_end_us": (((1 << 62) - v[9]) - first) / 100.0,
                "last_end_us": (v[8] - first) / 100.0,
                "iterations_per_wave": v[0] / w, "steps_per_wave": 16 * v[0] / w,        # (an iteration = the 16 bytes of a block)
                "cycles_per_step": v[1] / max(1, 16 * v[0]), "flush_cycles_per_iteration": v[2] / max(1, v[0]),
                "input_cycles_per_iteration": v[3] / max(1, v[0]), "loop_cycles_per_wave": v[4] / w}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
