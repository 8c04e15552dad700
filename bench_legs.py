"""bench_legs.py — the legs `bench.py` runs AFTER its timed region (and the synthetic-batch generator they share with it): the
other BASELINE configurations on one GPU (`run_other_config`, `run_mapping_config`), and the host-pointer entry points with PCIe
included (`pcie_probe`, `host_call_stats`, `run_host_pairs`).  Nothing here is part of `value`; `bench.py` re-exports every name,
so `bench.device_pairs` etc. keep working."""
import time

import numpy as np


def device_pairs(torch, n, read_len, err, ratio, seed, device, slack=0.15, chunk=8192):
    """Synthetic pairs built on the GPU: uniform random text; read = text prefix with
    i.i.d. per-base errors split sub:ins:del by `ratio` (scrooge_amd.synth.mutate, on device).

    Returns (ascii uint8 [n, row_bytes], text_words, read_words, text_len): every row holds
    the text in a 32-byte-aligned slot followed by the read in a 32-byte-aligned slot,
    zero padded — the staging layout scrg_pack_planar expects."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    text_len = int(read_len * (1.0 + slack) + 0.999999)
    src_len = max(text_len, int(read_len * 1.08) + 64)
    tw, rw = (text_len + 31) // 32, (read_len + 31) // 32
    out = torch.zeros((n, (tw + rw) * 32), dtype=torch.uint8, device=device)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    r = torch.tensor(ratio, dtype=torch.float64)
    r = (r / r.sum()).cumsum(0)
    for b0 in range(0, n, chunk):
        b = min(chunk, n - b0)
        src = torch.randint(0, 4, (b, src_len), generator=g, device=device, dtype=torch.uint8)
        u = torch.rand((b, src_len), generator=g, device=device)
        v = torch.rand((b, src_len), generator=g, device=device)
        is_err = u < err
        sub = is_err & (v < float(r[0]))
        ins = is_err & (v >= float(r[0])) & (v < float(r[1]))
        dele = is_err & (v >= float(r[1]))
        base = src.clone()
        delta = torch.randint(1, 4, (b, src_len), generator=g, device=device, dtype=torch.uint8)
        base = torch.where(sub, (base + delta) & 3, base)
        emit = torch.ones((b, src_len), dtype=torch.int64, device=device)
        emit[ins] = 2
        emit[dele] = 0
        pos = emit.cumsum(1) - emit
        read = torch.zeros((b, read_len + 1), dtype=torch.uint8, device=device)   # last column = dump
        # inserted random base first, then the (possibly substituted) source base
        ins_base = torch.randint(0, 4, (b, src_len), generator=g, device=device, dtype=torch.uint8)
        p_ins = torch.where(ins, pos, torch.full_like(pos, read_len)).clamp_(max=read_len)
        read.scatter_(1, p_ins, ins_base)
        p_keep = torch.where(emit > 0, pos + emit - 1, torch.full_like(pos, read_len)).clamp_(max=read_len)
        read.scatter_(1, p_keep, base)
        total = emit.sum(1)
        assert int(total.min()) >= read_len, "source segment too short for the requested read length"
        out[b0:b0 + b, :text_len] = lut[src[:, :text_len].long()]
        out[b0:b0 + b, tw * 32: tw * 32 + read_len] = lut[read[:, :read_len].long()]
        del src, u, v, emit, pos, read, p_ins, p_keep, base, delta, ins_base
    return out, tw, rw, text_len



def run_other_config(torch, scrooge_amd, device, local_rank, streams, name, n, L, profile, steps, warmup, check_pairs, seed, cores, W=64, O=33,
                     len_range=None, waves_per_cu=0):
    """One more BASELINE configuration of the unstructured interface, measured the same way as the headline (pairs generated
    and packed on the GPU, lane-interleaved layout, steps = align kernel + run compaction rotating over the streams) after
    the timed region, with `check_pairs` pairs of the last step compared, runs and all, with the CPU checker.

    len_range = (lo, hi): a MIXED-LENGTH batch — read lengths uniform in [lo, hi] (L = hi is the slot size), every text its
    read's source segment + 15 %; the pairs are issued longest read first, as the reference's callers sort them
    (src/tests.cu:375-377) and as the host entry points do, so that the 64 pairs of a wavefront have similar lengths; the
    checked pairs are spread over the whole batch."""
    from scrooge_amd import synth
    err, ratio = synth.PROFILES[profile]
    rows, tw, rw, text_len = device_pairs(torch, n, L, err, ratio, seed, device, chunk=max(256, min(8192, (1 << 28) // (L + 64))))
    G = scrooge_amd.api.GROUP
    row_words = tw + rw
    seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=device)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    als = [scrooge_amd.Aligner(local_rank) for _ in streams]
    for a_, st_ in zip(als, streams):
        a_.set_stream(st_.cuda_stream)
    for st_ in streams:
        st_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(streams[0]):
        als[0].pack_planar_groups(rows.view(-1), n, row_words, seq, bad)
    idx = torch.arange(n, dtype=torch.int64, device=device)
    if len_range:
        g = torch.Generator(device=device)
        g.manual_seed(seed + 1)
        rl = torch.sort(torch.randint(len_range[0], len_range[1] + 1, (n,), generator=g, device=device, dtype=torch.int64), descending=True).values
        tl = torch.clamp((rl * 115 + 99) // 100, max=text_len)        # the read's source segment + 15 %
        pick = (torch.arange(check_pairs, device=device) * (n // max(1, check_pairs))) if check_pairs else None
    else:
        rl, tl = torch.full_like(idx, L), torch.full_like(idx, text_len)
        pick = torch.arange(check_pairs, device=device) if check_pairs else None
    sample = rows[pick].cpu().numpy() if check_pairs else None
    torch.cuda.synchronize()
    del rows
    torch.cuda.empty_cache()
    assert int(bad.item()) == 0
    cap = (2 * L + 8 + 15) // 16 * 16
    first = (idx // G) * row_words * G + idx % G
    desc = torch.stack([first * 32, tl, (first + tw * G) * 32, rl, idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    kw = dict(text_stride_words=G, read_stride_words=G, W=W, O=O)
    if waves_per_cu:
        kw["waves_per_cu"] = waves_per_cu
    outs = [dict(runs=torch.empty(n * cap * 2, dtype=torch.uint8, device=device), ed=torch.empty(n, dtype=torch.int64, device=device),
                 n_runs=torch.empty(n, dtype=torch.int32, device=device), status=torch.empty(n, dtype=torch.int32, device=device))
            for _ in streams]
    with torch.cuda.stream(streams[0]):
        als[0].align_device(n, seq, desc, outs[0]["runs"], outs[0]["ed"], outs[0]["n_runs"], outs[0]["status"], **kw)
    torch.cuda.synchronize()
    assert int(outs[0]["status"].max().item()) == 0
    total_runs = int(outs[0]["n_runs"].sum().item())
    denses = [torch.empty(max(total_runs, 8) * 2, dtype=torch.uint8, device=device) for _ in streams]

    def one(j):
        b = j % len(streams)
        o = outs[b]
        with torch.cuda.stream(streams[b]):
            als[b].align_device(n, seq, desc, o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
            c64 = o["n_runs"].to(torch.int64)
            als[b].compact_runs(n, desc, o["runs"], o["n_runs"], torch.cumsum(c64, 0) - c64, denses[b])

    for j in range(warmup):
        one(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        one(warmup + j)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = (warmup + steps - 1) % len(streams)
    res = {"workload": name, "pairs": n, "read_len": L, "error_profile": profile, "W": W, "O": O, "steps": steps, "value": n * steps / dt, "unit": "pairs/s",
           "ms_per_step": dt / steps * 1e3, "runs_per_pair": total_runs / n,
           "step": "align kernel + run compaction, steps rotate over %d streams; measured after the timed region of the headline" % len(streams)}
    if len_range:
        res["read_len"] = {"min": int(rl.min().item()), "max": int(rl.max().item()), "mean": float(rl.double().mean().item()),
                           "order": "longest read first (src/tests.cu:375-377)"}
        res["bases_per_s"] = float(rl.sum().item()) * steps / dt
    if check_pairs:
        from oracle.pyoracle import Oracle, Reference
        k = check_pairs
        vl = dict(text_lens=tl[pick].cpu().numpy(), read_lens=rl[pick].cpu().numpy()) if len_range else {}
        if Reference.available(W, O):
            e_cpu, off_cpu, runs_cpu, ns = Reference(W, O).align_rows(sample, 0, text_len, tw * 32, L, threads=cores, **vl)
            against = "reference genasm_cpu.cpp (oracle/_ref%s)" % ("" if (W, O) == (64, 33) else ", built with -DCLI_W=%d -DCLI_O=%d" % (W, O))
        else:
            e_cpu, off_cpu, runs_cpu, _, ns = Oracle(allow_compile=False).align_rows(sample, 0, text_len, tw * 32, L, W=W, O=O, threads=cores, **vl)
            against = "oracle/liboracle.so (restatement)"
        o = outs[last]
        cnt_all = o["n_runs"].to(torch.int64)
        off_all = torch.cumsum(cnt_all, 0) - cnt_all
        cnt = cnt_all[pick].cpu().numpy().astype(np.uint64)
        off_gpu = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt, dtype=np.uint64)])
        if len_range:
            # the checked pairs are spread over the batch: gather their runs from the dense array
            starts = off_all[pick]
            seg = torch.repeat_interleave(torch.arange(k, device=device), cnt_all[pick])
            within = torch.arange(int(off_gpu[k]), device=device) - torch.from_numpy(off_gpu[:k].astype(np.int64)).to(device)[seg]
            src = (starts[seg] + within) * 2
            d_ = denses[last]
            runs_gpu = torch.stack([d_[src], d_[src + 1]], dim=1).cpu().numpy()
        else:
            runs_gpu = denses[last][: 2 * int(off_gpu[k])].cpu().numpy().reshape(-1, 2)
        ok = bool((o["ed"][pick].cpu().numpy() == e_cpu).all() and (off_gpu == off_cpu).all() and np.array_equal(runs_gpu, runs_cpu))
        res["parity"] = {"checked_pairs": k, "runs_bit_exact": ok, "against": against}
        res["cpu_pairs_per_s"] = k / (ns * 1e-9)
        res["cpu_threads"] = cores
        assert ok, "other_configs: GPU result differs from the CPU checker (%s)" % name
    for a_ in als:
        a_.close()
    del outs, denses, seq, desc
    torch.cuda.empty_cache()
    return res


def run_mapping_config(torch, scrooge_amd, device, local_rank, streams, genome_len, n_reads, steps, warmup, check_reads, seed, cores, host_api=None):
    """BASELINE configs[2], the read-mapping interface through the device-pointer layer: one synthetic chromosome packed once
    (contiguous), n_reads x 150 bp Illumina-like reads x 4 candidates each (true locus, two shifted by 1-3 bases, one random
    locus), the text of a candidate = the genome suffix from its start (src/genasm_cpu.cpp:512-514); reads in lane-interleaved
    groups.  Kernel + run compaction per step, like the headline; the candidates of the first `check_reads` reads are
    compared, runs and all, with the CPU checker."""
    G = scrooge_amd.api.GROUP
    L, n_c = 150, 4
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    gcodes = torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)
    gw = (genome_len + 31) // 32
    g_ascii = torch.zeros(gw * 32, dtype=torch.uint8, device=device)
    g_ascii[:genome_len] = lut[gcodes.long()]
    starts = torch.randint(0, genome_len - 400, (n_reads,), generator=g, device=device, dtype=torch.int64)
    # reads: 0.9 % substitutions, and one deletion or one insertion in 7.5 % of the reads each (1 % errors, 90:5:5)
    j = torch.arange(L, device=device, dtype=torch.int64).view(1, L)
    u = torch.rand((n_reads, 3), generator=g, device=device)
    pos = torch.randint(1, L - 1, (n_reads, 1), generator=g, device=device, dtype=torch.int64)
    is_del = (u[:, 0:1] < 0.075)
    is_ins = (u[:, 1:2] < 0.075) & ~is_del
    src = j + (is_del & (j >= pos)).long() - (is_ins & (j > pos)).long()
    codes = gcodes[(starts.view(-1, 1) + src).view(-1)].view(n_reads, L)
    sub = torch.rand((n_reads, L), generator=g, device=device) < 0.009
    codes = torch.where(sub, (codes + torch.randint(1, 4, (n_reads, L), generator=g, device=device, dtype=torch.uint8)) & 3, codes)
    codes = torch.where(is_ins & (j == pos), torch.randint(0, 4, (n_reads, L), generator=g, device=device, dtype=torch.uint8), codes)
    del src, sub
    sh = torch.randint(1, 4, (n_reads, 2), generator=g, device=device, dtype=torch.int64)
    cand = torch.stack([starts, (starts - sh[:, 0]).clamp_(min=0), starts + sh[:, 1],
                        torch.randint(0, genome_len - 10, (n_reads,), generator=g, device=device, dtype=torch.int64)], dim=1)     # [n_reads, 4]
    n = n_reads * n_c
    rw = (L + 31) // 32
    r_ascii = torch.zeros((n_reads, rw * 32), dtype=torch.uint8, device=device)
    r_ascii[:, :L] = lut[codes.long()]
    del codes
    pair_rows = r_ascii.repeat_interleave(n_c, dim=0)               # a row per pair: the read of candidate k of read r
    n_groups = (n + G - 1) // G
    seq = torch.zeros(gw + n_groups * G * rw + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=device)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    als = [scrooge_amd.Aligner(local_rank) for _ in streams]
    for a_, st_ in zip(als, streams):
        a_.set_stream(st_.cuda_stream)
        st_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(streams[0]):
        als[0].pack_planar(g_ascii, seq[:gw], bad)
        als[0].pack_planar_groups(pair_rows.view(-1), n, rw, seq[gw:], bad)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    del pair_rows
    cap = (2 * L + 8 + 15) // 16 * 16
    idx = torch.arange(n, dtype=torch.int64, device=device)
    first = gw + (idx // G) * rw * G + idx % G
    cs = cand.view(-1)
    desc = torch.stack([cs, genome_len - cs, first * 32, torch.full_like(idx, L), idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
    kw = dict(text_stride_words=1, read_stride_words=G)
    outs = [dict(runs=torch.empty(n * cap * 2, dtype=torch.uint8, device=device), ed=torch.empty(n, dtype=torch.int64, device=device),
                 n_runs=torch.empty(n, dtype=torch.int32, device=device), status=torch.empty(n, dtype=torch.int32, device=device))
            for _ in streams]
    with torch.cuda.stream(streams[0]):
        als[0].align_device(n, seq, desc, outs[0]["runs"], outs[0]["ed"], outs[0]["n_runs"], outs[0]["status"], **kw)
    torch.cuda.synchronize()
    assert int(outs[0]["status"].max().item()) == 0
    total_runs = int(outs[0]["n_runs"].sum().item())
    denses = [torch.empty(max(total_runs, 8) * 2, dtype=torch.uint8, device=device) for _ in streams]

    def one(k_):
        b = k_ % len(streams)
        o = outs[b]
        with torch.cuda.stream(streams[b]):
            als[b].align_device(n, seq, desc, o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
            c64 = o["n_runs"].to(torch.int64)
            als[b].compact_runs(n, desc, o["runs"], o["n_runs"], torch.cumsum(c64, 0) - c64, denses[b])

    for k_ in range(warmup):
        one(k_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k_ in range(steps):
        one(warmup + k_)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = (warmup + steps - 1) % len(streams)
    res = {"workload": "read-mapping interface: %d Mbp chromosome, %d x 150 bp reads x 4 candidates (BASELINE configs[2])" % (genome_len // 1000000, n_reads),
           "pairs": n, "steps": steps, "value": n * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3,
           "mean_edit_distance_true_locus": float(outs[last]["ed"][0::4][:100000].float().mean().item()),
           "step": "align kernel + run compaction over the genome packed once and the reads in lane-interleaved groups, steps rotate over %d streams; "
                   "measured after the timed region of the headline (device-pointer layer: host staging and text rendering not included)" % len(streams)}
    if check_reads:
        from oracle.pyoracle import Oracle, Reference
        k = check_reads * n_c
        TL = 400                                   # a candidate's text runs to the end of the genome; the alignment of a 150 bp read stays far inside 400
        tpos = cand[:check_reads].reshape(-1, 1) + torch.arange(TL, device=device, dtype=torch.int64).view(1, TL)
        trows = lut[gcodes[tpos.clamp_(max=genome_len - 1).view(-1)].long()].view(k, TL)
        rows = torch.zeros((k, 416 + rw * 32), dtype=torch.uint8, device=device)
        rows[:, :TL] = trows
        rows[:, 416: 416 + rw * 32] = r_ascii[:check_reads].repeat_interleave(n_c, dim=0)
        sample = rows.cpu().numpy()
        if Reference.available():
            e_cpu, off_cpu, runs_cpu, ns = Reference().align_rows(sample, 0, TL, 416, L, threads=cores)
            against = "reference genasm_cpu.cpp (oracle/_ref), pairwise overload on 400-base text prefixes"
        else:
            e_cpu, off_cpu, runs_cpu, _, ns = Oracle(allow_compile=False).align_rows(sample, 0, TL, 416, L, threads=cores)
            against = "oracle/liboracle.so (restatement) on 400-base text prefixes"
        o = outs[last]
        cnt = o["n_runs"][:k].cpu().numpy().astype(np.uint64)
        off_gpu = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt, dtype=np.uint64)])
        runs_gpu = denses[last][: 2 * int(off_gpu[k])].cpu().numpy().reshape(-1, 2)
        ok = bool((o["ed"][:k].cpu().numpy() == e_cpu).all() and (off_gpu == off_cpu).all() and np.array_equal(runs_gpu, runs_cpu))
        res["parity"] = {"checked_pairs": k, "runs_bit_exact": ok, "against": against}
        res["cpu_pairs_per_s"] = k / (ns * 1e-9)
        res["cpu_threads"] = cores
        assert ok, "other_configs: GPU result differs from the CPU checker (read mapping)"
    if host_api is not None:
        # The library surface for this configuration (never `value`): scrg_align_mapping / scrg_align_mapping_resident with the
        # genome, the reads and the candidate lists in host memory, results in host arrays — genasm_gpu::align_all(Genome_t&,
        # vector<Read_t>&), src/genasm_gpu.cu:890-980 — compared with what the device-pointer path produced above.
        pcie = host_api
        genome_h = g_ascii[:genome_len].cpu().numpy()
        reads_h = r_ascii.cpu().numpy()
        cs_h = cand.reshape(-1).cpu().numpy().astype(np.uint64)
        co_h = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(n_c)
        o = outs[last]
        ed_ref = o["ed"].cpu().numpy()
        cnt_ref = o["n_runs"].cpu().numpy().astype(np.uint64)
        off_ref = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt_ref, dtype=np.uint64)])
        runs_ref = denses[last][: 2 * int(off_ref[n])].cpu().numpy().reshape(-1, 2)
        ha = scrooge_amd.Aligner(local_rank)
        h2d = n_reads * (rw * 8) + n * 16
        hres = {"entry_points": "scrg_genome_set + scrg_align_mapping_resident / scrg_align_mapping (host pointers in, host arrays out; "
                                "PCIe-inclusive; library clock scrg_result.total_ns)",
                "reference": "genasm_gpu::align_all(Genome_t&, vector<Read_t>&), src/genasm_gpu.cu:890-980"}
        t_ = time.perf_counter()
        ha.set_genome_array(genome_h)
        hres["genome_set_s"] = time.perf_counter() - t_

        def check(res_, outputs):
            ok_ = bool((res_["edit_distance"] == ed_ref).all())
            if outputs != 1:
                ok_ = ok_ and bool((res_["run_offset"] == off_ref).all()) and bool(np.array_equal(res_["runs"], runs_ref))
            if outputs != 2:
                co_, txt = res_["cigar_offset"], res_["cigar_text"]
                k0 = 0
                for i in range(256):
                    c_ = int(cnt_ref[i])
                    want = "".join("%d%s" % (runs_ref[k0 + j, 0], chr(runs_ref[k0 + j, 1])) for j in range(c_))
                    k0 += c_
                    ok_ = ok_ and txt[int(co_[i]): int(co_[i + 1]) - 1].decode() == want
            return ok_

        for name_, outputs in (("resident_runs_and_text", 0), ("resident_text_only", 1), ("resident_runs_only", 2)):
            def call(outputs=outputs):
                r_ = ha.align_mapping_rows(None, reads_h, L, co_h, cs_h, outputs=outputs)
                return r_, ha.last_timing["total_ns"]
            r_, st_ = host_call_stats(n, h2d, call)
            st_["pcie_bound_s"] = max(h2d / (pcie[0] * 1e9), st_["d2h_bytes"] / (pcie[1] * 1e9))
            st_["frac_of_pcie_bound"] = st_["pcie_bound_s"] / st_["steady_best_s"]
            st_["identical_to_device_path"] = check(r_, outputs)
            assert st_["identical_to_device_path"], "host API (read mapping, %s) differs from the device-pointer path" % name_
            hres[name_] = st_
            del r_
        ha.clear_genome()

        def call_staged():
            r_ = ha.align_mapping_rows(genome_h, reads_h, L, co_h, cs_h, outputs=0)
            return r_, ha.last_timing["total_ns"]
        r_, st_ = host_call_stats(n, h2d + (genome_len + 3) // 4, call_staged, reps=3)
        st_["identical_to_device_path"] = check(r_, 0)
        assert st_["identical_to_device_path"]
        hres["genome_staged_per_call_runs_and_text"] = st_
        del r_
        ha.close()
        res["host_api"] = hres
    for a_ in als:
        a_.close()
    return res


def pcie_probe(torch, device, mb=256):
    """-> (H2D GB/s, D2H GB/s) of one pinned copy of `mb` MB each way (HIP events): the bound a host-pointer call is held to."""
    nbytes = mb << 20
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=device)
    out = []
    for src, dst in ((h, d), (d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        dst.copy_(src, non_blocking=True)
        b_.record()
        torch.cuda.synchronize()
        out.append(nbytes / (a_.elapsed_time(b_) * 1e-3) / 1e9)
    return out[0], out[1]


def host_call_stats(n_pairs, h2d_bytes, call, reps=6):
    """Times `call()` (-> result dict of Aligner._collect_arrays, library clock in .last_timing via the closure) `reps` times:
    the first call of a size allocates buffers and result arrays, the rest are the steady state (best and median)."""
    times, res = [], None
    for _ in range(reps):
        res, total_ns = call()
        times.append(total_ns * 1e-9)
    n = n_pairs
    total_runs = int(res["run_offset"][n]) if len(res["run_offset"]) else 0
    total_text = int(res["cigar_offset"][n]) if len(res["cigar_offset"]) else 0
    d2h = (12 if total_text else 8) * n + 2 * total_runs + total_text       # per pair: edit distance, run count (+ text length) as 32 bits each
    steady = sorted(times[1:])
    best, med = steady[0], steady[len(steady) // 2]
    return res, {"first_call_s": times[0], "steady_best_s": best, "steady_median_s": med,
                 "pairs_per_s": n / best, "pairs_per_s_median": n / med, "first_call_pairs_per_s": n / times[0],
                 "d2h_bytes": d2h, "h2d_bytes": h2d_bytes, "d2h_gbs": d2h / best / 1e9, "h2d_gbs": h2d_bytes / best / 1e9}


def run_host_pairs(torch, scrooge_amd, local_rank, rows, tw, text_len, L, dev_ed, dev_cnt, dev_dense, pcie, sizes=(100000, 20000)):
    """The library surface itself (never `value`): scrg_align_pairs — host strings in, edit distances + CIGAR runs + CIGAR
    text out, what the reference's genasm_gpu::align_all(texts, queries) does (src/genasm_gpu.cu:982-1065) — on the pairs of
    one of the timed batches (`rows`: the batch's ASCII in host memory), PCIe included.  Its results are compared with what
    the device-pointer path produced for the same pairs in the timed region (which the CPU leg has checked)."""
    out = {"entry_point": "scrg_align_pairs (host pointers in, host arrays out; PCIe-inclusive; library clock scrg_result.total_ns)",
           "reference": "genasm_gpu::align_all(texts, queries), src/genasm_gpu.cu:982-1065 (kernel vs end to end: README.md:103-108)"}
    ha = scrooge_amd.Aligner(local_rank)
    ed_ref = dev_ed.cpu().numpy()
    cnt_ref = dev_cnt.cpu().numpy().astype(np.uint64)
    for n_ in sizes:
        n_ = min(n_, rows.shape[0])
        sub = rows[:n_]
        h2d = n_ * (((text_len + 31) // 32 + (L + 31) // 32) * 8 + 8)
        legs = {}
        for name, outputs in (("runs_and_text", 0), ("text_only", 1), ("runs_only", 2)):
            def call(outputs=outputs):
                r_ = ha.align_pairs_rows(sub, 0, text_len, tw * 32, L, outputs=outputs)
                return r_, ha.last_timing["total_ns"]
            res, st = host_call_stats(n_, h2d, call)
            st["pcie_bound_s"] = max(h2d / (pcie[0] * 1e9), st["d2h_bytes"] / (pcie[1] * 1e9))
            st["frac_of_pcie_bound"] = st["pcie_bound_s"] / st["steady_best_s"]
            ok = bool((res["edit_distance"] == ed_ref[:n_]).all())
            if outputs != 1:
                off = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt_ref[:n_], dtype=np.uint64)])
                ok = ok and bool((res["run_offset"] == off).all())
                ok = ok and bool(np.array_equal(res["runs"], dev_dense[: 2 * int(off[n_])].cpu().numpy().reshape(-1, 2)))
            if outputs != 2:
                # the text is the "%d%c" rendering of the runs: total length and the first CIGARs, letter for letter
                co, txt = res["cigar_offset"], res["cigar_text"]
                dd = dev_dense[: 2 * int(cnt_ref[:64].sum())].cpu().numpy().reshape(-1, 2)
                k0 = 0
                for i in range(min(64, n_)):
                    c_ = int(cnt_ref[i])
                    want = "".join("%d%s" % (dd[k0 + j, 0], chr(dd[k0 + j, 1])) for j in range(c_))
                    k0 += c_
                    ok = ok and txt[int(co[i]): int(co[i + 1]) - 1].decode() == want
            st["identical_to_device_path"] = ok
            assert ok, "host API result differs from the device-pointer path (%s, %d pairs)" % (name, n_)
            legs[name] = st
            del res
        out["%d_pairs" % n_] = legs
    ha.close()
    return out


