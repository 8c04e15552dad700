#!/usr/bin/env python3
"""bench.py — aligned pairs/s of the GenASM-DC + GenASM-TB hot path on MI355X.

One "step" = one pass of the align kernel over one batch of synthetic
PBSIM2-shaped pairs that is already packed and resident in HBM, followed by
the run compaction and (N > 1) the RCCL gather of edit distances + CIGARs
to the step's root, which decodes them to runs inside the timed region.
Workload at N=1 = BASELINE.json configs[1]: 100k x 10 kb ONT-error pairs,
W=64, O=33.  N > 1 is weak scaling: a step is N x --pairs pairs (125k per GPU
at N = 8 = configs[3], 1 M pairs over 8 GPUs), sharded with no data-path
collective other than the result gather.  The root of step k is rank k mod N
(--gather-root rotate, the default: every GPU decodes one step in N and every
xGMI link carries the same) or always rank 0 (--gather-root 0; rank 0 then
aligns a smaller share of the step, --root-share).  After
the timed region an N > 1 run measures the other policies, the gather alone and
every rank's own rate in the same line (`diagnose`, `per_gpu_value`), and a
rank that stalls is ended by --deadline.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus 8 --steps 5 --warmup 1          # starts its own 8 ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 1

Plain `python bench.py --gpus N` with N > 1 (no WORLD_SIZE in the environment) builds the libraries and then starts
`python -m torch.distributed.run ... bench.py --no-build ...` as a fresh CHILD process — before this process has made
any HIP call, and never by exec — relays the child's output and exits with its code.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T int32 lane-ops/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=0,
                    help="pairs per GPU and step (default: 100000 = BASELINE configs[1]; 125000 at --gpus 8 = configs[3], 1 M pairs over 8 GPUs)")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--profile", default="ont", help="error profile (scrooge_amd.synth.PROFILES)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per pair (0 = library default)")
    ap.add_argument("--lds-rows", type=int, default=0)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--layout", default="auto", choices=["auto", "linear", "groups"],
                    help="sequence layout in HBM: linear = every sequence contiguous; groups = lane-interleaved groups of 64 "
                         "pairs (scrg_pack_planar_groups), what the one-pair-per-lane kernel reads best; auto = groups for that kernel")
    ap.add_argument("--gather-format", default="edits", choices=["edits", "edits-from-runs", "packed", "runs"],
                    help="N > 1: how CIGARs travel to rank 0 — edit streams, 1 byte per edit and per window end, written by the align kernel "
                         "itself (default; W-O <= 31) or encoded from its runs (edits-from-runs: any W/O), packed runs "
                         "(1 byte per run, restored on rank 0 inside the timed region) or scrg_run pairs")
    ap.add_argument("--gather-root", default="rotate", choices=["rotate", "0"],
                    help="N > 1, edit streams: the rank a step's results are gathered to and decoded on — step k to rank k mod N "
                         "(default: every GPU receives and decodes one step in N, every xGMI link carries the same, no rank is slower "
                         "than the others; projected 81 %% of N x one GPU at N = 8, profiles/r05_root_load.json) or always rank 0 "
                         "(SURVEY.md §8e's literal gather; rank 0 then also decodes every step: 45 %% with equal shards, ~65-73 %% with "
                         "--root-share auto).  Either way `diagnose` measures both after the timed region")
    ap.add_argument("--root-share", default="auto",
                    help="N > 1 with --gather-root 0 and the decode on: what rank 0 aligns itself, as a fraction of an equal share "
                         "(T / N pairs of the step's T = N x --pairs); the other ranks split the rest.  Rank 0 also decodes all N "
                         "slots (~0.2 of the align work per pair), so an equal split makes it the slowest rank.  'auto' = "
                         "max(0, 1 - 0.2 (N - 1)): 0.8 at N = 2, 0.4 at N = 4, 0 at N = 8 (a pure collector); 'equal' = 1")
    ap.add_argument("--no-decode", action="store_true",
                    help="N > 1, edit streams: leave the gathered CIGARs as edit streams on the root (default: the root restores "
                         "scrg_run pairs for every rank's pairs inside the timed region, scrg_decode_edit_stream)")
    ap.add_argument("--sustained-steps", type=int, default=600,
                    help="N = 1: after the timed region, time this many more pipelined steps for the 'sustained' field (0 = skip)")
    ap.add_argument("--streams", type=int, default=4, help="pipeline depth: consecutive steps rotate over this many streams/handles")
    ap.add_argument("--gather-depth", type=int, default=8,
                    help="N > 1: buffer sets of the gather (a step's send / receive buffers are reused this many steps later: by then its "
                         "transfer and, on its root, the decoding must be over; at 4 — the number of align lanes — a step waited for the "
                         "transfer of the step four before it)")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="after the timed region, also measure BASELINE configs[0] (4 M x 150 bp), configs[2] (read mapping, 1 M reads x 4 "
                         "candidates on a 100 Mbp chromosome) and configs[4] (40 k x 50 kb PacBio-error pairs), each with an oracle-checked sample; "
                         "auto = only in the default single-GPU run of the headline workload")
    ap.add_argument("--host-api", default="auto", choices=["auto", "on", "off"],
                    help="after the timed region, also time the host-pointer entry points (scrg_align_pairs on one of the timed batches, "
                         "scrg_align_mapping_resident on configs[2]; PCIe-inclusive, reported as `host_api`, never `value`); auto = only in "
                         "the default single-GPU run of the headline workload")
    ap.add_argument("--diagnose", default="auto", choices=["auto", "on", "off"],
                    help="N > 1 (edit streams, decode on): after the timed region, measure in the same run what the open design questions "
                         "of the multi-GPU step need — the same steps under the other policies (root 0 with equal shards / with the "
                         "'auto' root share, rotating root), without the decode, the gather alone (achieved GB/s per peer into rank 0, all "
                         "peers at once and one at a time) and every rank's own N=1-equivalent rate — reported as `diagnose` and "
                         "`per_gpu_value`; auto = on when N > 1")
    ap.add_argument("--diagnose-budget", type=float, default=240.0,
                    help="N > 1: seconds the diagnostics may take; after that (or if they raise) rank 0 prints its line with the error in "
                         "`diagnose` and all ranks leave with exit code 0")
    ap.add_argument("--deadline", type=float, default=1500.0,
                    help="hard limit in seconds for the whole run (0 = none): a rank that is still running then — e.g. stalled in a "
                         "collective — prints where it was and exits with code 124 (the launcher tears the other ranks down); the "
                         "self-launching parent (`python bench.py --gpus N`) kills its child's process group at deadline + 30 s and, with "
                         "--attempts > 1, starts a FRESH child (never a re-exec of a process that touched the GPU)")
    ap.add_argument("--attempts", type=int, default=1, help="self-launch only: fresh children to try when one hits the deadline")
    ap.add_argument("--no-build", action="store_true",
                    help="never rebuild the library (profiling: nothing may fork a compiler under rocprofv3)")
    ap.add_argument("--stats", action="store_true", help="profiling only: print kernel round/step counters")
    ap.add_argument("--ablate", type=int, default=0, help="profiling only: 1 = no TB table reads, 2 = no DC sweep")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiling: nothing but set-up, warm-up and the timed region (no CPU leg, no serial / sustained / inclusive / "
                         "edit-stream / other-config / host-API legs): the LAST --steps launches of the align kernel in a rocprofv3 "
                         "kernel trace of this command are the timed ones (scripts/headline_trace.sh)")
    ap.add_argument("--serial", action="store_true",
                    help="one stream: every step waits for the previous one to finish (default: consecutive steps "
                         "alternate between two streams and two handles, so the next step's wavefronts fill the "
                         "GPU while the previous step's last pairs finish)")
    args = ap.parse_args()
    if args.headline_only:
        args.cpu_seconds, args.sustained_steps, args.other_configs, args.host_api = 0.0, 0, "off", "off"
    return args


def usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


from bench_legs import (device_pairs, host_call_stats, pcie_probe, run_host_pairs, run_mapping_config,      # noqa: E402,F401
                        run_other_config)


def root_share_plan(world, nominal, mode):
    """-> (pairs rank 0 aligns, pairs every other rank aligns = the buffer size of all ranks) for a step of world x nominal
    pairs.  mode: 'auto' = max(0, 1 - 0.2 (world - 1)) of an equal share for rank 0 (it also decodes all slots, ~0.2 of the
    align work per pair: DESIGN.md section 4), or that fraction as a number; whole groups of 64 pairs; the step never has
    fewer than world x nominal pairs."""
    frac = max(0.0, 1.0 - 0.2 * (world - 1)) if mode == "auto" else min(1.0, max(0.0, float(mode)))
    total = world * nominal
    n0 = int(nominal * frac + 1e-6) // 64 * 64
    n_other = (-(-(total - n0) // (world - 1)) + 63) // 64 * 64
    return n0, n_other


def kernel_sources_digest():
    """sha256 of the sources the one-pair-per-lane align kernel is made of: the PMC instruction count in profiles/ is only
    used for the roofline figure if it was measured on exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("genasm_lane_kernel.hip", "genasm_device.h", "genasm_kernels.h"):
        with open(os.path.join(ROOT, "scrooge_amd", "csrc", f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def pmc_instruction_count():
    """-> (dict from the newest profiles/r*_pmc_sq_summary.json that was measured on the current kernel sources, source note),
    else (None, why).  The dict has valu_instructions_per_window_round, SQ_INSTS_VALU and window_rounds_per_launch of the
    profiled launch and the workload it was (pairs, read_len, profile, seed)."""
    import glob
    want = kernel_sources_digest()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_summary.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("kernel_sources_sha256") == want and pj.get("valu_instructions_per_window_round"):
            return pj, "SQ_INSTS_VALU / window rounds, rocprofv3 --pmc, %s (same kernel sources: sha256 %s...)" % (os.path.basename(f), want[:12])
        stale = stale or os.path.basename(f)
    return None, ("no PMC summary for the current kernel sources (sha256 %s...; newest file: %s): re-run scripts/collect_profiles.sh" % (want[:12], stale))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: build both libraries here, then run the N ranks as a fresh
    child process tree (torch.distributed.run, one process per GPU) and pass its exit code on.  This process never touches
    the GPU (no HIP call, no torch.cuda.is_available()): nothing that holds a device forks or execs anything."""
    import socket
    import subprocess
    import scrooge_amd
    if not args.no_build:
        scrooge_amd.build_library()
        if args.cpu_seconds > 0:
            from oracle.pyoracle import build as build_oracle
            build_oracle()
    if os.environ.get("SCRG_BENCH_DRYRUN") != "1":
        import torch
        have = torch.cuda.device_count()           # (counting devices does not initialise the GPU)
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this node has %d GPU(s)" % (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--no-build"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv + ["--no-build"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    import signal
    rc = 124
    for attempt in range(max(1, args.attempts)):
        # a fresh child per attempt, in its own process group (so that a stalled job can be ended as a whole); stdout / stderr
        # inherited: rank 0's JSON line is the child's
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True)
        try:
            rc = proc.wait(timeout=(args.deadline + 30.0) if args.deadline > 0 else None)
            break
        except subprocess.TimeoutExpired:
            print("bench.py --gpus %d: attempt %d did not finish within --deadline %.0f s (+30): ending its process group"
                  % (args.gpus, attempt + 1, args.deadline), file=sys.stderr)
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    proc.wait(timeout=20)
                    break
                except subprocess.TimeoutExpired:
                    continue
            rc = 124
        except KeyboardInterrupt:
            os.killpg(proc.pid, signal.SIGTERM)
            rc = proc.wait()
            break
    raise SystemExit(rc)


PHASE = ["start"]


def set_phase(text):
    PHASE[0] = text


def arm_deadline(seconds, rank):
    """In-rank watchdog (also under an explicit launcher, where no parent of ours exists): a daemon timer that ends THIS process
    with code 124 — no re-exec, no new GPU work — when the run is still going after `seconds`, saying which phase it was in
    (a rank stalled in a collective never returns to Python, so the main thread cannot do this itself)."""
    if seconds <= 0:
        return None
    import threading

    def fire():
        print("bench.py: rank %d still running after --deadline %.0f s (phase: %s): exiting with code 124" % (rank, seconds, PHASE[0]),
              file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


STEP_TEXT = {
    "local": "align kernel + run compaction",
    "edits": "align kernel writing every CIGAR as an edit stream + its run count (scrg_align_device_edits: one byte per edit and per window end, a "
             "lossless encoding that carries the window ends) + compaction of the streams + RCCL gather of scores, run "
             "counts and streams to the step's root (config.gather.root), one collective and one buffer set per pipelined step "
             "(overlaps the next kernels) + on the root, INSIDE the timed region, scrg_decode_edit_stream of every rank's slot "
             "(one launch for all N x pairs) into one dense scrg_run array: the step ends with the CIGAR runs of all pairs on the root",
    "edits-from-runs": "align kernel (runs) + edit-stream encoding (scrg_encode_edit_stream) + RCCL gather of scores and "
                       "streams to the step's root (config.gather.root), one collective and one buffer set per pipelined step; the "
                       "root keeps the streams, their decoding is checked for every rank's slot after the timed region",
    "packed": "align kernel + run compaction to one byte per run + RCCL gather of scores and runs to rank 0 (one buffer set "
              "per pipelined step); rank 0 restores scrg_run pairs inside the timed region",
    "runs": "align kernel + run compaction + RCCL gather of scores and scrg_run pairs to rank 0 (one buffer set per pipelined step)",
}


def main():
    args = parse()
    if not args.pairs:
        args.pairs = 125000 if args.gpus == 8 else 100000
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                   # (does not return)
    import torch
    import torch.distributed as dist

    import scrooge_amd
    from scrooge_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    stall = os.environ.get("SCRG_BENCH_TEST_STALL")      # test only (tests/test_distributed_cpu.py): a rank that never comes back
    watchdog = None if stall == "no-watchdog" else arm_deadline(args.deadline, rank)
    if stall and (stall != "rank1" or rank == 1):
        set_phase("test stall (SCRG_BENCH_TEST_STALL)")
        while True:
            time.sleep(1.0)
    import datetime
    pg_timeout = datetime.timedelta(seconds=max(60.0, min(args.deadline if args.deadline > 0 else 1800.0, 1800.0)))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): for N > 1 launch with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`"
                         % (world, args.gpus))
    # build (if stale) before anything initialises the GPU: a compiler must never be forked from a process that
    # holds the device, least of all under a profiler
    if not args.no_build:
        scrooge_amd.build_library()
        if args.cpu_seconds > 0 and rank == 0:
            from oracle.pyoracle import build as build_oracle      # the checker of the CPU leg: compiled now, only loaded later
            build_oracle()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dryrun = os.environ.get("SCRG_BENCH_DRYRUN") == "1"    # test only: all ranks on GPU 0, gloo through the host
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # test only: run the gather path on the real RCCL backend with this one rank (tests/test_gpu_scale.py)
    force_gather = world == 1 and os.environ.get("SCRG_BENCH_FORCE_GATHER") == "1"
    dist_on = world > 1 or force_gather
    if force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device, timeout=pg_timeout)
    elif world > 1:
        set_phase("process group set-up")
        if dryrun:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)    # nccl == RCCL on ROCm

    al = scrooge_amd.Aligner(local_rank)
    al.set_stream(torch.cuda.current_stream().cuda_stream)
    stats_pipelined = args.stats and os.environ.get("SCRG_BENCH_STATS_PIPELINED") == "1"     # counters of one launch among overlapping ones
    n_lanes = 1 if (args.serial or (args.stats and not stats_pipelined) or args.ablate) else max(1, args.streams)      # software pipeline depth over streams
    gather_depth = max(2, n_lanes, args.gather_depth)
    kw = {}
    if args.lanes:
        kw["lanes_per_pair"] = args.lanes
    if args.lds_rows:
        kw["lds_rows"] = args.lds_rows
    if args.waves_per_cu:
        kw["waves_per_cu"] = args.waves_per_cu
    p = al.resolved_params(**kw)
    geom = al.query_launch(**kw)
    # the kernels' own counters and the scheduling / ablation switches exist in profiling builds only (-DSCRG_STATS,
    # -DSCRG_ABLATE: scripts/ab.sh, SCRG_LIB=ab_libs/lib_stats.so); the shipped library has none of that code
    build_flags = int(scrooge_amd.load_library().scrg_build_flags())
    have_stats = bool(build_flags & 1)
    if args.stats and not have_stats:
        raise SystemExit("--stats needs a library built with -DSCRG_STATS (scripts/ab.sh build stats -DSCRG_STATS; "
                         "SCRG_LIB=ab_libs/lib_stats.so): the shipped kernels carry no counters")
    if args.stats:
        al.params.reserved[1] = 1
    if os.environ.get("SCRG_BENCH_DEBUG_FLAGS"):      # experiment knob: switches of scrg_params.reserved[0] that leave the results intact (test / profiling builds only: SCRG_LIB=ab_libs/lib_select.so)
        al.params.reserved[0] = int(os.environ["SCRG_BENCH_DEBUG_FLAGS"])
    if args.ablate:
        al.params.reserved[0] = args.ablate     # results are wrong by design; parity checks are skipped
        args.cpu_seconds = 0
        try:
            al.resolved_params()
        except scrooge_amd.ScroogeError:
            raise SystemExit("--ablate needs a library built with -DSCRG_ABLATE (scripts/ab.sh build ablate -DSCRG_ABLATE; "
                             "SCRG_LIB=ab_libs/lib_ablate.so): the shipped library has no ablation code paths")

    # ---------------- synthetic batches, generated and packed on the GPU: ONE PER PIPELINE LANE ----------------
    # Consecutive steps are independent batches: step k aligns batch k mod n_lanes (own seed, own sequence array, own
    # descriptors and output buffers), so no launch in flight reads what another one reads.
    n = args.pairs
    L = args.read_len
    # Unequal shards (--root-share): every rank's buffers hold n pairs, n_real of them are real (the rest are empty reads,
    # which the kernel retires at once and the gather carries as empty streams); rank 0 has n0, the others n.
    nominal = n
    share_on = (world > 1 and args.gather_root == "0" and args.gather_format == "edits" and not args.no_decode
                and args.root_share != "equal" and not args.lanes)
    n0 = n
    if share_on:
        n0, n = root_share_plan(world, nominal, args.root_share)
        if n0 >= n:
            share_on, n0, n = False, nominal, nominal
    n_real = n0 if (share_on and rank == 0) else n
    pairs_per_step_all = (n0 + (world - 1) * n) if share_on else world * n
    err, ratio = synth.PROFILES[args.profile]
    t_gen = time.time()
    groups = args.layout == "groups" or (args.layout == "auto" and p.lanes_per_pair == 1)
    cap = (2 * L + 8 + 15) // 16 * 16                 # runs per pair slice (genasm_gpu.cu:906-911), 32-byte pieces
    idx = torch.arange(n, dtype=torch.int64, device=device)
    sidx = idx * 0 if os.environ.get("SCRG_BENCH_SAMESEQ") else idx     # experiment only: every pair reads pair 0's sequences
    same_batch = os.environ.get("SCRG_BENCH_SAMEBATCH") == "1"           # experiment only: every lane aligns lane 0's batch (rounds 1-3)
    keep_ascii = rank == 0 and world == 1 and args.cpu_seconds > 0        # the CPU leg checks every lane's batch
    seqs, descs, descs_full, ascii_keep = [], [], [], []
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    for b_ in range(n_lanes):
        if b_ and same_batch:
            seqs.append(seqs[0]); descs.append(descs[0]); descs_full.append(descs_full[0]); ascii_keep.append(ascii_keep[0])
            continue
        ascii_rows, tw, rw, text_len = device_pairs(torch, n, L, err, ratio, args.seed + 1000 * rank + 7919 * b_, device)
        row_words = tw + rw
        n_words = n * row_words
        if groups:
            # lane-interleaved groups of 64 pairs: word w of pair p at ((p // 64) * row_words + w) * 64 + p % 64
            G = scrooge_amd.api.GROUP
            seq = torch.zeros((n + G - 1) // G * G * row_words + scrooge_amd.api.SEQ_PAD_WORDS_GROUPS, dtype=torch.int64, device=device)
            al.pack_planar_groups(ascii_rows.view(-1), n, row_words, seq, bad)
            first = (sidx // G) * row_words * G + sidx % G            # word index of the text's first word
            desc = torch.stack([first * 32, torch.full_like(idx, text_len), (first + tw * G) * 32, torch.full_like(idx, L),
                                idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
            kw["text_stride_words"] = kw["read_stride_words"] = G
        else:
            seq = torch.zeros(n_words + scrooge_amd.api.SEQ_PAD_WORDS, dtype=torch.int64, device=device)
            al.pack_planar(ascii_rows.view(-1), seq, bad)
            desc = torch.stack([sidx * row_words * 32, torch.full_like(idx, text_len),
                                (sidx * row_words + tw) * 32, torch.full_like(idx, L),
                                idx * cap, torch.full_like(idx, cap)], dim=1).contiguous()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0
        desc_full = desc
        if n_real < n:
            desc = desc.clone()
            desc[n_real:, 1] = 0                                  # text_len
            desc[n_real:, 3] = 0                                  # read_len
        seqs.append(seq); descs.append(desc); descs_full.append(desc_full)
        ascii_keep.append(ascii_rows if keep_ascii else None)    # (stays on the GPU until the CPU leg takes it, one lane at a time)
        del ascii_rows
    seq, desc, desc_full = seqs[0], descs[0], descs_full[0]
    seq_total_gb = sum(seqs[b_].numel() for b_ in range(n_lanes) if not (same_batch and b_)) * 8 / 1e9
    torch.cuda.empty_cache()
    # read lengths of all N x n pairs in slot order, for the root's decoder (one length for all unless shards are unequal)
    if share_on:
        rl_all = torch.full((world * n,), L, dtype=torch.int64, device=device)
        rl_all[n0:n] = 0
    # result buffers, one set per pipeline lane (scrg_run = 2 bytes)
    outs = [dict(runs=torch.empty(n * cap * 2, dtype=torch.uint8, device=device),
                 ed=torch.empty(n, dtype=torch.int64, device=device),
                 n_runs=torch.empty(n, dtype=torch.int32, device=device),
                 status=torch.empty(n, dtype=torch.int32, device=device)) for _ in range(n_lanes)]
    runs, ed, n_runs, status = (outs[0][k] for k in ("runs", "ed", "n_runs", "status"))
    sample_cap = n
    gen_s = time.time() - t_gen

    # one untimed pass per lane fixes the (deterministic) output sizes
    totals = []
    for b_ in range(n_lanes):
        o = outs[b_]
        al.align_device(n, seqs[b_], descs[b_], o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
        torch.cuda.synchronize()
        assert args.ablate or int(o["status"].max().item()) == 0, "CIGAR slice overflow"
        totals.append(int(o["n_runs"].sum().item()))
    total_runs = totals[0]
    denses = [torch.empty(max(t_, 8) * 2, dtype=torch.uint8, device=device) for t_ in totals]
    gather = None
    decode_on = False
    decode_args = None
    gather_format = args.gather_format
    if gather_format == "packed" and p.W - p.O > 63:
        gather_format = "runs"                       # packed runs hold counts up to 63
    stream_bytes = None
    if gather_format == "edits" and p.lanes_per_pair != 1:
        gather_format = "edits-from-runs"            # only the one-pair-per-lane kernels write edit streams themselves
    edits = gather_format in ("edits", "edits-from-runs")
    if dist_on and edits:
        # CIGARs travel as edit streams (one byte per edit and per window end); rank 0 keeps them in that form
        from scrooge_amd.distributed import EditStreamGather
        # (sizes are exchanged once: the largest over the lanes' batches; what a step really sends travels in its lengths)
        stream_bytes = 0
        for b_ in range(n_lanes):
            o = outs[b_]
            if gather_format == "edits":
                t_len = torch.empty(n, dtype=torch.int32, device=device)
                t_cnt = torch.empty(n, dtype=torch.int32, device=device)
                al.align_device_edits(n, seqs[b_], descs[b_], o["runs"], o["ed"], t_len, o["status"], t_cnt, **kw)
                torch.cuda.synchronize()
                assert int(o["status"].max().item()) == 0
                assert torch.equal(t_cnt, o["n_runs"]), "run counts of the edit-stream kernel differ from the runs kernel's"
                stream_bytes = max(stream_bytes, int(((t_len.to(torch.int64) + 3) // 4 * 4).sum().item()))
                al.align_device(n, seqs[b_], descs[b_], o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
                torch.cuda.synchronize()
                del t_cnt
            else:
                # (include/scrooge_amd.h: a byte per edit and per window, streams start at multiples of 4)
                bound = int(o["ed"].sum().item()) + n * (2 * (L + L // 2) // (p.W - p.O) + L // 63 + 12) + 64
                tmp = torch.empty(bound, dtype=torch.uint8, device=device)
                t_off = torch.empty(n, dtype=torch.int64, device=device)
                t_len = torch.empty(n, dtype=torch.int32, device=device)
                t_tot = torch.zeros(2, dtype=torch.int64, device=device)
                al.encode_edit_stream(n, descs[b_], o["runs"], o["n_runs"], tmp, t_off, t_len, t_tot, W=p.W, O=p.O)
                torch.cuda.synchronize()
                assert int(t_tot[1].item()) == 0
                stream_bytes = max(stream_bytes, int(t_tot[0].item()))
                del tmp, t_off
            del t_len
        gather = EditStreamGather(n, stream_bytes, device, dst="rotate" if args.gather_root == "rotate" else 0,
                                  depth=gather_depth, ordered=gather_format == "edits",
                                  total_runs=max(totals) if gather_format == "edits" else None)
        gather.prime()                               # (set-up: connections to every root exist before anything is timed)
        # the root's decoder: a handle of its own (its stream is the gather's decode stream), one read length for all pairs
        decode_on = gather_format == "edits" and not args.no_decode
        if decode_on:
            # (one handle per buffer set of the gather: the decode launches of consecutive steps run side by side)
            decoders = [scrooge_amd.Aligner(local_rank) for _ in range(gather_depth)]
            for d_ in decoders:
                d_.params = al.params
            decode_args = (decoders, rl_all, 1, dict(kw)) if share_on else (decoders, torch.tensor([L], dtype=torch.int64, device=device), 0, dict(kw))
    elif dist_on:
        from scrooge_amd.distributed import ResultGather
        packed_gather = gather_format == "packed"    # runs travel as one byte each; rank 0 restores scrg_run pairs
        gather = ResultGather(n, max(totals), device, dst=0, depth=max(2, n_lanes), packed=packed_gather)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    # Consecutive steps are independent batches (each lane has its own): step k runs on stream k % n_lanes with its own
    # library handle (work queue, spill area), input batch and output buffers: its persistent wavefronts start as soon as
    # the previous step's wavefronts begin to retire, instead of after its last pair has finished.
    aligners = [al] + [scrooge_amd.Aligner(local_rank) for _ in range(n_lanes - 1)]
    for extra in aligners[1:]:
        extra.params = al.params
    # Lane streams of different priorities (low, high; everything else, RCCL included, runs at normal priority):
    # HIP gives each priority its own hardware queues, whereas two streams of one priority can share a queue,
    # and kernels in one queue never overlap (scripts/side_stream_probe.py).
    PRIOS = [int(x) for x in os.environ.get("SCRG_BENCH_PRIOS", "1,-1,0,1,-1,0").split(",")]      # experiment knob
    if n_lanes > 1:
        streams = [torch.cuda.ExternalStream(scrooge_amd.api.create_stream(local_rank, pr), device=device) for pr in (PRIOS[:n_lanes])]
    else:
        streams = [torch.cuda.current_stream()]
    setup_stream = torch.cuda.current_stream()
    for st_ in streams:
        if st_ != setup_stream:
            st_.wait_stream(setup_stream)
    for a_, st_ in zip(aligners, streams):
        a_.set_stream(st_.cuda_stream)

    # what a step does is read from `cur`, so that the diagnostics after the timed region can run the same steps under another
    # policy (another gather object / root, other shards, no decode, no collective at all)
    cur = {"fmt": gather_format if dist_on else "local", "gather": gather, "descs": descs, "decode_args": decode_args,
           "denses": denses}

    def step(k=None):
        j = step.count
        step.count += 1
        b = j % n_lanes
        o = outs[b]
        fmt, gather_, descs_ = cur["fmt"], cur["gather"], cur["descs"]
        with torch.cuda.stream(streams[b]):
            if fmt == "edits":
                # The align kernel writes every pair's CIGAR as an edit stream into the pair's slice; the slices are
                # gathered into the step's send buffer (4-byte aligned, pair order) and one RCCL collective takes
                # scores + streams to rank 0 over xGMI; one buffer set per pipelined step, so the gather of this step
                # overlaps the next steps' align kernels
                gather_.finish(j)                       # buffers of step j-DEPTH are free again (gathered, and decoded on their root)
                g = gather_.buffers(j)
                if k is not None:
                    ev[k][0].record()
                aligners[b].align_device_edits(n, seqs[b], descs_[b], o["runs"], o["ed"], o["n_runs"], o["status"], g["cnt"], **kw)
                if k is not None:
                    ev[k][1].record()
                r4 = (o["n_runs"].to(torch.int64) + 3) & -4
                boff = torch.cumsum(r4, 0) - r4
                g["len"].copy_(o["n_runs"])
                aligners[b].compact_runs(n, descs_[b], o["runs"], (r4 >> 1).to(torch.int32), boff >> 1, g["stream"])
                gather_.start(j, o["ed"], decode=cur["decode_args"] if step.decode else None)
                return
            if k is not None:
                ev[k][0].record()
            aligners[b].align_device(n, seqs[b], descs_[b], o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
            if k is not None:
                ev[k][1].record()
            if fmt == "edits-from-runs":
                # the same with the streams encoded from the kernel's runs (any W/O, any kernel)
                gather_.finish(j)
                g = gather_.buffers(j)
                aligners[b].encode_edit_stream(n, descs_[b], o["runs"], o["n_runs"], g["stream"], g["off"], g["len"], g["total"], W=p.W, O=p.O)
                gather_.start(j, o["ed"])
                return
            probe = os.environ.get("SCRG_BENCH_PROBE") if args.headline_only else None      # measuring aids (scripts/r06_chain_probe.sh; --headline-only: nothing is checked there); the line's metric says INVALID
            if probe == "no-chain":                                    # the step without its scan and compaction
                return
            if probe == "no-scan" and b in step.cached_off:            # the offsets of the same batch from an earlier step
                aligners[b].compact_runs(n, descs_[b], o["runs"], o["n_runs"], step.cached_off[b], cur["denses"][b])
                return
            cnt64 = o["n_runs"].to(torch.int64)
            dense_off = torch.cumsum(cnt64, 0) - cnt64
            if probe == "no-scan":
                step.cached_off[b] = dense_off
            if probe == "no-compact":                                  # the scan alone
                return
            if fmt in ("packed", "runs"):
                # the same with the runs themselves (--gather-format runs | packed)
                gather_.finish(j)                       # buffers of step j-DEPTH are free again
                if packed_gather:
                    aligners[b].compact_runs_packed(n, descs_[b], o["runs"], o["n_runs"], dense_off, gather_.send_runs[j % gather_.DEPTH], **kw)
                else:
                    aligners[b].compact_runs(n, descs_[b], o["runs"], o["n_runs"], dense_off, gather_.send_runs[j % gather_.DEPTH])
                gather_.start(j, o["ed"], o["n_runs"])
            else:
                aligners[b].compact_runs(n, descs_[b], o["runs"], o["n_runs"], dense_off, cur["denses"][b])

    step.count = 0
    step.cached_off = {}
    step.decode = decode_on
    set_phase("warm-up steps")
    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.finish_all()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    set_phase("timed region (%d steps)" % args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    t_enqueued = time.perf_counter() - t0
    if gather is not None:
        gather.finish_all()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    set_phase("checks after the timed region")
    last = (step.count - 1) % n_lanes
    # the rank that holds the last step's gathered results checks them (rank 0 unless the root rotates)
    check_rank = gather.root_of(step.count - 1) if (gather is not None and edits) else 0
    if dist_on and gather_format == "edits" and rank == check_rank:
        # (the slices of the last step hold edit streams: make the runs for the checks below, outside the timed region)
        with torch.cuda.stream(streams[last]):
            aligners[last].align_device(n, seqs[last], descs[last], outs[last]["runs"], outs[last]["ed"], outs[last]["n_runs"], outs[last]["status"], **kw)
        torch.cuda.synchronize()
    ed, n_runs, dense = outs[last]["ed"], outs[last]["n_runs"], denses[last]
    total_runs = totals[last]                           # (of the batch the last step aligned)
    gather_check = None
    if gather is not None and rank == check_rank and os.environ.get("SCRG_BENCH_NOCOLL") != "1":
        # outside the timed region: what the root holds for itself after the last step's gather (scores, counts and the
        # runs restored from the wire format) must be what its own kernel produced
        cnt64 = n_runs.to(torch.int64)
        dense_off = torch.cumsum(cnt64, 0) - cnt64
        torch.cuda.synchronize()                        # (torch's stream is not the handle's)
        aligners[last].compact_runs(n, descs[last], outs[last]["runs"], n_runs, dense_off, dense)
        torch.cuda.synchronize()
        if edits:
            # every rank's slot must decode (scrg_decode_edit_stream) into runs for reads of this length, with as many
            # edits as the gathered edit distance; the root's own slot must be, run for run, what its kernel produced
            rl = torch.tensor([L], dtype=torch.int64, device=device)
            gather_check = True
            # what the timed region itself produced on this root: the runs of ALL ranks' pairs of the last step, one launch
            dec = gather.decoded(step.count - 1) if decode_on else None
            if dec is not None and int(dec["bad"].item()) != 0:
                print("decode on rank %d: %d pairs did not decode" % (rank, int(dec["bad"].item())), file=sys.stderr)
                gather_check = False
            for r in range(world):
                v = gather.results(step.count - 1, r)
                torch.cuda.synchronize()
                if share_on:
                    runs_g, cnt_g, off_g, n_bad = gather.decode(aligners[last], step.count - 1, r, rl_all[r * n: (r + 1) * n], 1, **kw)
                else:
                    runs_g, cnt_g, off_g, n_bad = gather.decode(aligners[last], step.count - 1, r, rl, 0, **kw)
                torch.cuda.synchronize()
                is_edit = (v["stream"][: gather.totals[r]] >= 64).to(torch.int64)
                csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(is_edit, 0)])
                n_edits = csum[v["off"] + v["len"].to(torch.int64)] - csum[v["off"]]
                checks = {"streams_decode": n_bad == 0, "edits_equal_distance": bool(torch.equal(n_edits, v["ed"].to(torch.int64)))}
                if r == rank:
                    checks.update(own_scores=bool(torch.equal(v["ed"].to(torch.int64), ed) and torch.equal(cnt_g, n_runs)),
                                  own_runs=bool(runs_g is not None and torch.equal(runs_g[: 2 * total_runs], dense[: 2 * total_runs])))
                if dec is not None and runs_g is not None:
                    # the slot's part of the one-launch decode == the two-pass decode of the slot alone
                    a0 = int(dec["run_off"][r * n].item())
                    tr = int(cnt_g.to(torch.int64).sum().item())
                    checks["decoded_in_timed_region"] = bool(torch.equal(dec["cnt"][r * n: (r + 1) * n], cnt_g) and
                                                             torch.equal(dec["runs"][2 * a0: 2 * (a0 + tr)], runs_g[: 2 * tr]))
                if not all(checks.values()):
                    print("gather check on rank %d, slot of rank %d: %s (undecodable pairs: %d)" % (rank, r, checks, n_bad), file=sys.stderr)
                    gather_check = False
        else:
            ed_g, cnt_g, runs_g = gather.results(step.count - 1, 0)        # (the buffers are sized for the largest batch of the lanes)
            gather_check = bool(torch.equal(ed_g, ed) and torch.equal(cnt_g, n_runs) and torch.equal(runs_g[: 2 * total_runs], dense[: 2 * total_runs]))
    if dist_on and world > 1 and gather is not None:
        # the verdict travels to rank 0, which prints the line
        flag = torch.tensor([1 if gather_check in (True, None) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = bool(int(flag.item()))
    assert gather_check in (True, None), "gathered results differ from the local ones"
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    # Reference point outside the timed region (N > 1): the same steps with the results LEFT as edit streams on the root
    # (no decoding) — what the decoding costs the job is the difference to `value`.  Never the headline.
    streams_only = None
    if dist_on and decode_on:
        step.decode = False
        gather.finish_all()
        dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step()
        gather.finish_all()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tso = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=device)
        dist.all_reduce(tso, op=dist.ReduceOp.MAX)
        streams_only = {"value": pairs_per_step_all * args.steps / float(tso.item()), "unit": "pairs/s",
                        "ms_per_step": float(tso.item()) / args.steps * 1e3,
                        "note": "the same steps without the root's decoding (gathered CIGARs stay edit streams); measured after the timed region"}
        step.decode = decode_on
    # ---------------- N > 1: the diagnostics of the multi-GPU step, after everything that is timed ----------------
    # One run on a multi-GPU node should answer the open design questions at once: root 0 against a rotating
    # root, equal shards against the root-share plan, what the decode costs, what the links carry, what every GPU does alone.
    # They run LAST, on every rank, when rank 0 has its whole line ready (guarded_diagnostics, at the end of main): whatever happens
    # in them — an exception, a collective that never returns on links nobody has used yet — the line is still printed, with the
    # error in `diagnose`.
    diagnose = None
    per_gpu_value = None
    want_diag = (dist_on and gather_format == "edits" and decode_on and p.lanes_per_pair == 1 and
                 (args.diagnose == "on" or (args.diagnose == "auto" and world > 1)))

    def run_diagnostics():
        set_phase("diagnostics: sizes of the full batches")
        fault = os.environ.get("SCRG_BENCH_TEST_DIAG")    # test only (tests/test_gpu_scale.py): diagnostics that raise / never return
        if fault == "raise" and rank == world - 1:
            raise RuntimeError("test fault (SCRG_BENCH_TEST_DIAG)")
        if fault == "hang":
            while True:
                time.sleep(1.0)
        K = max(2, args.steps)
        gather.finish_all()
        torch.cuda.synchronize()
        # stream bytes / run totals of the FULL batches (every pair real): what any policy may have to carry
        sb_full, rt_full = 0, 0
        t_len = torch.empty(n, dtype=torch.int32, device=device)
        t_cnt = torch.empty(n, dtype=torch.int32, device=device)
        for b_ in range(n_lanes):
            o = outs[b_]
            with torch.cuda.stream(streams[b_]):
                aligners[b_].align_device_edits(n, seqs[b_], descs_full[b_], o["runs"], o["ed"], t_len, o["status"], t_cnt, **kw)
            torch.cuda.synchronize()
            sb_full = max(sb_full, int(((t_len.to(torch.int64) + 3) // 4 * 4).sum().item()))
            rt_full = max(rt_full, int(t_cnt.to(torch.int64).sum().item()))
        del t_len, t_cnt
        dense_local = [torch.empty(max(rt_full, 8) * 2, dtype=torch.uint8, device=device) for _ in range(n_lanes)]

        def all_max(x):
            t_ = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())

        def timed_steps(pairs_step, after=None):
            """warm-up, barrier, K steps, barrier: job pairs/s under whatever `cur` says (max over ranks of the time)"""
            for _ in range(2):
                step()
            if cur["gather"] is not None:
                cur["gather"].finish_all()
            dist.barrier()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(K):
                step()
            if cur["gather"] is not None:
                cur["gather"].finish_all()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            d_ = all_max(time.perf_counter() - t_)
            return {"value": pairs_step * K / d_, "unit": "pairs/s", "ms_per_step": d_ / K * 1e3, "pairs_per_step_all_gpus": pairs_step, "steps": K}

        def policy(root, shards):
            """the same steps with the results gathered to `root` (0 | "rotate") and the step's pairs split `shards`
            ("equal": every rank aligns --pairs; "auto": root_share_plan) — None if this run's buffers cannot hold it"""
            scaled = None
            if shards == "equal":
                real0 = real_other = nominal
            else:
                real0, real_other = root_share_plan(world, nominal, "auto")
                if real_other > n:
                    # the buffers of this run hold n pairs per rank (they were sized for another policy): the same split at the
                    # size that fits — the other ranks align n pairs, rank 0 the same fraction of theirs as in the full plan
                    scaled = n / real_other
                    real0, real_other = int(real0 * scaled) // 64 * 64, n // 64 * 64
            mine = real0 if rank == 0 else real_other
            descs_p = []
            for b_ in range(n_lanes):
                d_ = descs_full[b_].clone()
                d_[mine:, 1] = 0
                d_[mine:, 3] = 0
                descs_p.append(d_)
            rl_p = torch.zeros((world, n), dtype=torch.int64, device=device)
            rl_p[0, :real0] = L
            rl_p[1:, :real_other] = L
            g_ = EditStreamGather(n, sb_full, device, dst=root, depth=gather_depth, ordered=True, total_runs=rt_full)
            g_.prime()
            keep = dict(cur)
            cur.update(fmt="edits", gather=g_, descs=descs_p, decode_args=(decoders, rl_p.reshape(-1), 1, dict(kw)))
            step.decode = True
            try:
                res_ = timed_steps(real0 + (world - 1) * real_other)
                # the decode of the last step on its root must have found every stream an alignment of a read of its length
                flag = 1
                if rank == g_.root_of(step.count - 1):
                    flag = 1 if int(g_.decoded(step.count - 1)["bad"].item()) == 0 else 0
                ft = torch.tensor([flag], dtype=torch.int32, device=device)
                dist.all_reduce(ft, op=dist.ReduceOp.MIN)
                res_["every_slot_decoded"] = bool(int(ft.item()))
                res_["shards"] = {"rank_0": real0, "other_ranks": real_other}
                if scaled is not None:
                    res_["shards"]["note"] = ("the root-share plan of %d pairs per step scaled by %.3f to the %d pairs per rank this run's "
                                              "buffers hold: same split, smaller step" % (world * nominal, scaled, n))
                res_["root"] = "rank 0" if root == 0 else "step k to rank k mod N"
            finally:
                cur.clear()
                cur.update(keep)
                step.decode = decode_on
                del g_, descs_p, rl_p
                torch.cuda.empty_cache()
            return res_

        diagnose = {"steps": K, "note": "measured after the timed region, same pipeline, same batches; `value` of the line is the configured policy"}
        set_phase("diagnostics: root 0, equal shards")
        diagnose["root0_equal_shards"] = policy(0, "equal")
        set_phase("diagnostics: root 0, root-share plan")
        diagnose["root0_auto_shards"] = policy(0, "auto")
        set_phase("diagnostics: rotating root")
        diagnose["rotating_root_equal_shards"] = policy("rotate", "equal")
        diagnose["gather_without_decode"] = streams_only
        # ---- the links: the gather alone (nothing else on the GPUs), all peers at once, then one peer at a time ----
        set_phase("diagnostics: the gather alone")
        gather.finish_all()
        host_stage = gather.host_stage
        reps = 2 if host_stage else 10
        dist.barrier()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for k_ in range(reps):
            b_ = k_ % gather.DEPTH
            if host_stage:
                hs = [torch.empty(gather.wire, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
                dist.gather(gather.send[b_].cpu(), hs, dst=0)
            else:
                dist.gather(gather.send[b_], gather.recv[b_] if rank == 0 else None, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        d_ = all_max(time.perf_counter() - t_)
        links = {"bytes_per_rank_and_gather": gather.wire, "gathers": reps, "ms_per_gather": d_ / reps * 1e3,
                 "GBs_per_peer_all_at_once": gather.wire * reps / d_ / 1e9,
                 "GBs_into_rank_0": (world - 1) * gather.wire * reps / d_ / 1e9,
                 "backend": dist.get_backend() + (" (dry run: staged through the host, not a link figure)" if host_stage else "")}
        set_phase("diagnostics: one peer at a time")
        one = []
        for r_ in range(1, world):
            # (one untimed message first: RCCL sets up the point-to-point connection of a pair of ranks on first use)
            for timed_ in (False, True):
                dist.barrier()
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                if rank == r_:
                    for k_ in range(reps if timed_ else 1):
                        dist.send(gather.send[0].cpu() if host_stage else gather.send[0], dst=0)
                elif rank == 0:
                    buf_ = torch.empty(gather.wire, dtype=torch.uint8) if host_stage else gather.recv[0][r_]
                    for k_ in range(reps if timed_ else 1):
                        dist.recv(buf_, src=r_)
                torch.cuda.synchronize()
            dist.barrier()
            d1 = all_max(time.perf_counter() - t_)
            one.append({"peer": r_, "GBs": gather.wire * reps / d1 / 1e9, "ms_per_message": d1 / reps * 1e3})
        links["one_peer_at_a_time"] = one
        links["xgmi_link_peak_GBs"] = 76.8            # MI355X_MICROARCH.md: 7 links x 153.6 GB/s bidirectional per GPU = 76.8 GB/s per direction and peer
        diagnose["links"] = links
        # ---- every rank alone: the N = 1 step (align kernel + run compaction, no collective), all ranks at the same time ----
        set_phase("diagnostics: every rank's own step")
        descs_eq = []
        for b_ in range(n_lanes):
            d_ = descs_full[b_].clone()
            d_[nominal:, 1] = 0
            d_[nominal:, 3] = 0
            descs_eq.append(d_)
        keep = dict(cur)
        cur.update(fmt="local", gather=None, descs=descs_eq, denses=dense_local)
        try:
            for _ in range(2):
                step()
            dist.barrier()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(K):
                step()
            torch.cuda.synchronize()
            mine_rate = nominal * K / (time.perf_counter() - t_)
        finally:
            cur.clear()
            cur.update(keep)
        rates = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(rates, torch.tensor([mine_rate], dtype=torch.float64, device=device))
        rates = [float(x.item()) for x in rates]
        per_gpu_value = {"per_rank": rates, "min": min(rates), "mean": sum(rates) / world, "max": max(rates), "unit": "pairs/s",
                         "pairs_per_rank_and_step": nominal, "steps": K,
                         "note": "the N = 1 step of BENCH (align kernel + run compaction, same pipeline, no collective), all ranks at the "
                                 "same time: what N x this would be with a free gather"}
        diagnose["efficiency_vs_per_gpu_value"] = {k_: (v_["value"] / (world * per_gpu_value["mean"]) if isinstance(v_, dict) and "value" in v_ else None)
                                                   for k_, v_ in diagnose.items() if k_.endswith("_shards")}
        diagnose["efficiency_vs_per_gpu_value"]["configured"] = pairs_per_step_all * args.steps / dt / (world * per_gpu_value["mean"])
        del dense_local, descs_eq
        torch.cuda.empty_cache()
        set_phase("after the diagnostics")
        return diagnose, per_gpu_value

    def guarded_diagnostics(line):
        """Runs the diagnostics under a time budget.  line: rank 0's finished JSON object (None on the other ranks).
        -> (diagnose, per_gpu_value); if they raise or do not finish within --diagnose-budget seconds, rank 0 prints its line
        with the error in `diagnose` and every rank leaves with exit code 0: the headline of a multi-GPU run is never lost to
        its diagnostics (a rank stuck in a collective never returns to Python: a timer thread does this)."""
        import threading
        finished = threading.Event()

        def give_up(msg):
            if line is not None:
                line["diagnose"] = {"error": msg}
                print(json.dumps(line), flush=True)
            print("bench.py: rank %d: %s" % (rank, msg), file=sys.stderr, flush=True)
            os._exit(0)

        def expire():
            if not finished.is_set():
                give_up("the diagnostics did not finish within --diagnose-budget %.0f s (phase: %s); the rest of the line is complete"
                        % (args.diagnose_budget, PHASE[0]))

        timer = threading.Timer(args.diagnose_budget, expire)
        timer.daemon = True
        timer.start()
        try:
            res = run_diagnostics()
        except Exception as e:          # (the other ranks may now wait in a collective this rank never enters: their timers end them)
            give_up("the diagnostics raised %s: %s (phase: %s); the rest of the line is complete" % (type(e).__name__, e, PHASE[0]))
        finished.set()
        timer.cancel()
        return res

    # reference point outside the timed region: the same step on ONE stream (no overlap between launches)
    serial = None
    if n_lanes > 1 and not dist_on and not args.stats and not args.headline_only:
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        o = outs[0]
        torch.cuda.synchronize()
        ts = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for a_, b_ in sev:
                a_.record()
                aligners[0].align_device(n, seq, desc, o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
                b_.record()
                cnt64 = o["n_runs"].to(torch.int64)
                aligners[0].compact_runs(n, desc, o["runs"], o["n_runs"], torch.cumsum(cnt64, 0) - cnt64, denses[0])
        torch.cuda.synchronize()
        serial = {"ms_per_step": (time.perf_counter() - ts) / 3 * 1e3,
                  "kernel_ms": sum(a_.elapsed_time(b_) for a_, b_ in sev) / 3}
        serial["value"] = n / (serial["ms_per_step"] * 1e-3)
    # Reference point outside the timed region: the same pipelined step over many more steps.  K timed steps contain
    # one pipeline fill and one drain — a 10 kb pair is ~334 dependent window rounds, so the last launch cannot finish
    # sooner than ~2.5 ms after it starts, whatever the rate of the steps before it; at K = 20 that is ~10 % of the
    # timed region.
    sustained = None
    if n_lanes > 1 and not dist_on and not args.stats and args.sustained_steps > 0:
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.sustained_steps):
            step()
        torch.cuda.synchronize()
        sdt = time.perf_counter() - ts
        sustained = {"steps": args.sustained_steps, "value": n * args.sustained_steps / sdt, "unit": "pairs/s",
                     "ms_per_step": sdt / args.sustained_steps * 1e3}
        last = (step.count - 1) % n_lanes
        ed, n_runs, dense = outs[last]["ed"], outs[last]["n_runs"], denses[last]
    # Reference point outside the timed region: the same pipelined step INCLUDING the packing of the sequences — ASCII in HBM ->
    # runs: scrg_pack_planar_groups of the batch's 21.5 KB per pair, then the align kernel and the compaction, every step.  The
    # headline excludes the packing, as the reference's timed region does (src/genasm_gpu.cu:939-944: its 2-bit conversion runs
    # before the clock starts); a caller whose sequences arrive as text pays this rate.
    incl_pack = None
    if (n_lanes > 1 and not dist_on and not args.stats and not args.ablate and groups and not same_batch
            and all(a_ is not None for a_ in ascii_keep)):
        row_words_ = tw + rw

        def step_incl_pack():
            j = step.count
            step.count += 1
            b = j % n_lanes
            o = outs[b]
            with torch.cuda.stream(streams[b]):
                aligners[b].pack_planar_groups(ascii_keep[b].view(-1), n, row_words_, seqs[b], bad)
                aligners[b].align_device(n, seqs[b], descs[b], o["runs"], o["ed"], o["n_runs"], o["status"], **kw)
                cnt64 = o["n_runs"].to(torch.int64)
                aligners[b].compact_runs(n, descs[b], o["runs"], o["n_runs"], torch.cumsum(cnt64, 0) - cnt64, denses[b])
        for _ in range(n_lanes):
            step_incl_pack()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step_incl_pack()
        torch.cuda.synchronize()
        idt = time.perf_counter() - ts
        assert int(bad.item()) == 0
        incl_pack = {"value": n * args.steps / idt, "unit": "pairs/s", "steps": args.steps, "ms_per_step": idt / args.steps * 1e3,
                     "includes": "ASCII in HBM -> lane-interleaved 2-bit planes (scrg_pack_planar_groups, %.1f KB of text per pair) + align kernel + "
                                 "run compaction, pipelined over the same %d streams as the headline" % (row_words_ * 32 / 1e3, n_lanes)}
        last = (step.count - 1) % n_lanes
        ed, n_runs, dense = outs[last]["ed"], outs[last]["n_runs"], denses[last]
    # duration of one align launch for the roofline line: events around a launch that has the GPU to itself.
    # (Events around a pipelined launch also contain the time its wavefronts wait for the previous launch's
    # to retire; rocprofv3 timestamps start at the first dispatch and agree with the stand-alone figure.)
    events_ms = kernel_ms
    if serial is not None:
        kernel_ms = serial["kernel_ms"]

    if args.stats and p.lanes_per_pair == 1:
        st = al.debug_stats_lane()
        r = max(1, st["rounds"])
        for k in ("fetch", "setup", "table", "pass1", "traceback"):
            st["cyc_per_round_" + k] = st["cycles_" + k] / r
        # shader clock while the kernel ran: cycles the wavefronts counted / their life time on the 100 MHz wall clock
        busy = st["cycles_fetch"] + st["cycles_setup"] + st["cycles_table"] + st["cycles_traceback"]
        st["shader_clock_ghz"] = busy / max(1, st["life_ticks_sum"]) * 0.1
        print("stats(last launch):", st, file=sys.stderr)
    elif args.stats:
        st = al.debug_stats()
        st["steps_per_round"] = st["dc_steps"] / max(1, st["rounds"])
        st["macro_per_round"] = st["tb_macro_steps"] / max(1, st["rounds"])
        for k in ("fetch", "setup", "dc", "tb", "tb_loop"):
            st["cyc_per_round_" + k] = st["cycles_" + k] / max(1, st["rounds"])
        old_rounds = max(1, st["rounds"] - st["diag_rounds"])
        for k in ("dc", "tb"):
            st["cyc_per_old_round_" + k] = st["cycles_" + k] / old_rounds
            st["cyc_per_diag_round_" + k] = st["cycles_diag_" + k] / max(1, st["diag_rounds"])
        print("stats(last launch):", st, file=sys.stderr)
    if rank != 0:
        if want_diag:
            guarded_diagnostics(None)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---------------- work model for the roofline line ----------------
    runs_per_pair = totals[0] / n                 # (lane 0's batch: the one the stand-alone launches below and `serial` align)
    # CPU leg: bounded sample, also the source of dc_cells / text_used per pair and a parity check
    cpu = None
    dc_cells = tb_steps = text_used = windows = None
    parity = None
    host_api = None
    pcie = None
    default_run = (world == 1 and not dist_on and n == 100000 and L == 10000 and args.profile == "ont" and not args.serial
                   and not args.stats and not args.ablate and not args.lanes)
    want_host_api = keep_ascii and (args.host_api == "on" or (args.host_api == "auto" and default_run))
    if keep_ascii:
        from oracle.pyoracle import Oracle, Reference
        r_off = tw * 32                       # a row = text slot, then read slot (device_pairs)
        cores = usable_cores()
        orc = Oracle(allow_compile=False)          # (built before the GPU was initialised, or prebuilt: never compile from here)
        use_ref = Reference.available() and (p.W, p.O) == (64, 33)
        against = "reference genasm_cpu.cpp (oracle/_ref)" if use_ref else "oracle/liboracle.so (restatement)"

        def cpu_align(rows_, threads_):
            if use_ref:
                return Reference().align_rows(rows_, 0, text_len, r_off, L, threads=threads_)
            e_, off_, runs_, _, ns_ = orc.align_rows(rows_, 0, text_len, r_off, L, W=p.W, O=p.O, threads=threads_)
            return e_, off_, runs_, ns_

        def compare(lane, m_, e_cpu, off_cpu, runs_cpu):
            """the GPU results the timed region (or the steps after it) left in this lane's buffers against the CPU's, the
            first m_ pairs: edit distances, run offsets and the runs themselves, array against array ({count, op} byte pairs,
            the reference's CIGARs parsed in oracle/ref_driver.cpp)"""
            o_ = outs[lane]
            ed_all = o_["ed"][:m_].cpu().numpy()
            cnt_gpu = o_["n_runs"][:m_].cpu().numpy().astype(np.uint64)
            off_gpu = np.concatenate([np.zeros(1, np.uint64), np.cumsum(cnt_gpu, dtype=np.uint64)])
            runs_gpu = denses[lane][: 2 * int(off_gpu[m_])].cpu().numpy().reshape(-1, 2)
            ok_ed = bool((ed_all == e_cpu).all())
            ok_off = bool((off_gpu == off_cpu).all())
            ok_runs = ok_off and bool(np.array_equal(runs_gpu, runs_cpu))
            return ok_ed, ok_off, ok_runs, int(off_cpu[m_])

        # the lane of the last step first: work counters, the CPU baseline, the full comparison
        rows = ascii_keep[last].cpu().numpy()
        # work counters (dc_cells, tb_steps, windows, text used) from this repo's restatement, on a small sample
        cal = min(sample_cap, max(2 * cores, 64))
        _, _, _, st, ns = orc.align_rows(rows[:cal], 0, text_len, r_off, L, W=p.W, O=p.O, threads=cores)
        dc_cells, tb_steps, text_used = (st["dc_cells"] / cal, st["tb_steps"] / cal, st["text_used"] / cal)
        windows = st["windows"] / cal
        rate = cal / (ns * 1e-9)
        m = int(min(sample_cap, max(cal, rate * args.cpu_seconds)))
        # the CPU baseline: the reference itself (oracle/_ref, default knobs only) or the restatement, all usable cores
        e_cpu, off_cpu, runs_cpu, ns = cpu_align(rows[:m], cores)
        cpu = {"value": m / (ns * 1e-9), "unit": "pairs/s", "cores": cores,
               "kind": "reference" if use_ref else "port",
               "sample": "%s %d of the %d pairs of one of the %d batches of this workload, kernel-only time (%s), %d OpenMP threads"
                         % ("all" if m == n else "the first", m, n, n_lanes,
                            "genasm_cpu.cpp:589-591 via oracle/_ref" if use_ref else "oracle/liboracle.so", cores)}
        # the same checker on ONE thread (SURVEY.md §8d asks for both), about two seconds' worth of pairs
        m1 = int(min(m, max(8, cpu["value"] / max(1, cores) * 2.0)))
        _, _, _, ns1 = cpu_align(rows[:m1], 1)
        cpu["single_thread"] = {"value": m1 / (ns1 * 1e-9), "unit": "pairs/s", "sample_pairs": m1}
        ok_ed, ok_off, ok_runs, n_cmp = compare(last, m, e_cpu, off_cpu, runs_cpu)
        parity = {"checked_pairs": m, "of_pairs": n, "edit_distances_equal": ok_ed, "run_counts_equal": ok_off,
                  "runs_bit_exact": ok_runs, "runs_compared": n_cmp, "against": against, "batch": "pipeline lane %d (the last step's)" % last}
        assert ok_ed and ok_runs, "GPU result differs from the CPU checker on the bench batch: %s" % parity
        del runs_cpu, e_cpu, off_cpu
        if want_host_api:
            # the library surface on this very batch (its ASCII is in host memory now), compared with the results just checked
            pcie = pcie_probe(torch, device)
            host_api = {"pcie_probe_gbs": {"h2d": pcie[0], "d2h": pcie[1], "note": "one pinned 256 MB copy each way, HIP events"},
                        "pairwise": run_host_pairs(torch, scrooge_amd, local_rank, rows, tw, text_len, L, outs[last]["ed"], outs[last]["n_runs"],
                                                   denses[last], pcie)}
        del rows
        ascii_keep[last] = None
        # the other lanes' batches (each lane aligns its own): what their last steps left, against the CPU path as well
        others = []
        for lane in range(n_lanes):
            if lane == last or ascii_keep[lane] is None or same_batch:
                continue
            rows = ascii_keep[lane].cpu().numpy()
            ascii_keep[lane] = None
            e_cpu, off_cpu, runs_cpu, _ = cpu_align(rows[:m], cores)
            ok_ed, ok_off, ok_runs, n_cmp = compare(lane, m, e_cpu, off_cpu, runs_cpu)
            others.append({"batch": "pipeline lane %d" % lane, "checked_pairs": m, "edit_distances_equal": ok_ed,
                           "run_counts_equal": ok_off, "runs_bit_exact": ok_runs, "runs_compared": n_cmp})
            assert ok_ed and ok_runs, "GPU result differs from the CPU checker on the batch of pipeline lane %d" % lane
            del rows, runs_cpu, e_cpu, off_cpu
        parity["other_batches"] = others
        parity["pairs_checked_all_batches"] = m * (1 + len(others))
    ascii_keep = None
    torch.cuda.empty_cache()

    # The N > 1 step writes CIGARs as edit streams (a different, lossless output format).  For a like-for-like
    # scaling figure the same step — align kernel with edit-stream output + compaction of the streams, pipelined
    # over the same streams, no collective — is timed here on one GPU, after everything above (it reuses the slices).
    edit_stream_step = None
    if not dist_on and n_lanes > 1 and not args.ablate and not args.stats and p.lanes_per_pair == 1 and not args.headline_only:
        lens = [torch.empty(n, dtype=torch.int32, device=device) for _ in range(n_lanes)]
        sbytes_lane = []
        for b_ in range(n_lanes):
            aligners[0].align_device_edits(n, seqs[b_], descs[b_], outs[b_]["runs"], outs[b_]["ed"], lens[b_], outs[b_]["status"], **kw)
            torch.cuda.synchronize()
            sbytes_lane.append(int(((lens[b_].to(torch.int64) + 3) & -4).sum().item()))
        sbytes = sum(sbytes_lane) / n_lanes
        sdense = [torch.empty(sb_ + 8, dtype=torch.uint8, device=device) for sb_ in sbytes_lane]

        def edit_step(j):
            b = j % n_lanes
            o = outs[b]
            with torch.cuda.stream(streams[b]):
                aligners[b].align_device_edits(n, seqs[b], descs[b], o["runs"], o["ed"], lens[b], o["status"], **kw)
                r4 = (lens[b].to(torch.int64) + 3) & -4
                boff = torch.cumsum(r4, 0) - r4
                aligners[b].compact_runs(n, descs[b], o["runs"], (r4 >> 1).to(torch.int32), boff >> 1, sdense[b])

        for j in range(args.warmup):
            edit_step(j)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for j in range(args.steps):
            edit_step(j)
        torch.cuda.synchronize()
        es_dt = time.perf_counter() - ts
        edit_stream_step = {"value": n * args.steps / es_dt, "unit": "pairs/s", "ms_per_step": es_dt / args.steps * 1e3,
                            "stream_bytes_per_pair": sbytes / n,
                            "note": "the step of the N > 1 runs without the collective: scrg_align_device_edits + compaction "
                                    "of the streams, same pipeline; measured after the timed region"}
        # the receiving side of the same step: one slot's streams back to scrg_run pairs (what the root of an N-GPU job does
        # for every rank's slot, all slots in one launch), checked against the runs kernel's output of the timed region
        with torch.cuda.stream(streams[0]):
            # (lane 0's batch: its runs of the timed region are in denses[0], their counts in outs[0]["n_runs"])
            n_runs0, tr0 = outs[0]["n_runs"], totals[0]
            aligners[0].align_device_edits(n, seqs[0], descs[0], outs[0]["runs"], outs[0]["ed"], lens[0], outs[0]["status"], **kw)
            r4 = (lens[0].to(torch.int64) + 3) & -4
            boff = torch.cumsum(r4, 0) - r4
            aligners[0].compact_runs(n, descs[0], outs[0]["runs"], (r4 >> 1).to(torch.int32), boff >> 1, sdense[0])
            cnt64 = n_runs0.to(torch.int64)
            doff = torch.cumsum(cnt64, 0) - cnt64
            back = torch.zeros(tr0 * 2 + 64, dtype=torch.uint8, device=device)
            nbad = torch.zeros(1, dtype=torch.int32, device=device)
            rl1 = torch.tensor([L], dtype=torch.int64, device=device)
            dev_ = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            aligners[0].decode_edit_stream(n, sdense[0], boff, lens[0], rl1, 0, doff, back, n_runs0, nbad, **kw)
            dev_[0].record()
            for _ in range(5):
                aligners[0].decode_edit_stream(n, sdense[0], boff, lens[0], rl1, 0, doff, back, n_runs0, nbad, **kw)
            dev_[1].record()
        torch.cuda.synchronize()
        dec_ok = int(nbad.item()) == 0 and bool(torch.equal(back[: 2 * tr0], denses[0][: 2 * tr0]))
        assert dec_ok, "decoded edit streams differ from the runs of the timed region"
        edit_stream_step["decode_ms_per_slot"] = dev_[0].elapsed_time(dev_[1]) / 5
        edit_stream_step["decode_note"] = ("scrg_decode_edit_stream of this step's %d streams into the dense scrg_run array, one launch "
                                           "alone on the GPU; all %d pairs' runs identical to the timed region's" % (n, n))
        del sdense, lens, back

    # window rounds of one launch (one round = one window of each of a wavefront's 64 pairs), from the kernel's own
    # counters: one more launch with the counters switched on, after everything that is timed
    rounds_live = None
    if have_stats and p.lanes_per_pair == 1 and not args.ablate and os.environ.get("SCRG_BENCH_NO_STATS_LAUNCH") != "1":      # (PMC passes: only plain launches)
        keep = al.params.reserved[1]
        al.params.reserved[1] = 1
        with torch.cuda.stream(streams[0]):
            aligners[0].align_device(n, seq, desc_full, outs[0]["runs"], outs[0]["ed"], outs[0]["n_runs"], outs[0]["status"], **kw)
        torch.cuda.synchronize()
        rounds_live = aligners[0].debug_stats_lane()["rounds"]
        al.params.reserved[1] = keep
        if n_real < n:
            # rank 0 aligns fewer pairs than it has buffers for: the roofline figures below are those of a full launch
            # (n pairs), taken here, after everything that is timed
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[0]):
                a_.record()
                aligners[0].align_device(n, seq, desc_full, outs[0]["runs"], outs[0]["ed"], outs[0]["n_runs"], outs[0]["status"], **kw)
                b_.record()
            torch.cuda.synchronize()
            kernel_ms = a_.elapsed_time(b_)
            runs_per_pair = float(outs[0]["n_runs"].to(torch.int64).sum().item()) / n

    # ---------------- the other single-GPU configurations of BASELINE.json, after everything that is timed ----------------
    other_configs = None
    want_other = args.other_configs == "on" or (args.other_configs == "auto" and default_run)
    if want_other:
        del outs, denses, seq, desc, desc_full, seqs, descs, descs_full, runs, ed, n_runs, status, dense
        torch.cuda.empty_cache()
        cores_ = usable_cores()
        chk = args.cpu_seconds > 0
        other_configs = [
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "unstructured pairwise: 4 M x 150 bp Illumina-like pairs (BASELINE configs[0] shape at GPU scale)",
                             4000000, 150, "illumina", 6, 2, 20000 if chk else 0, args.seed + 11, cores_),
            run_mapping_config(torch, scrooge_amd, device, local_rank, streams, 100000000, 1000000, 6, 2, 5000 if chk else 0, args.seed + 12, cores_,
                               host_api=pcie if want_host_api else None),
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "long-read stress: 40 k x 50 kb PacBio-error (15 %) pairs, the single-GPU share of BASELINE configs[4]",
                             40000, 50000, "pacbio15", 6, 2, 1500 if chk else 0, args.seed + 13, cores_),
            # the workload of the reference's one published absolute number (README.md:103-108, DATASETS.md:51: PBSIM2 PacBio
            # P6C4, 10 kb reads, accuracy 0.95, sub:ins:del 6:50:54 — 25 004 aligns/s kernel-only on an RTX 3060): SURVEY §8d "Cfg2b"
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "Cfg2b: 100 k x 10 kb PacBio-error pairs (5 %, sub:ins:del 6:50:54; the workload of the reference's README.md:103-108 figure)",
                             100000, 10000, "pacbio", 10, 2, 2000 if chk else 0, args.seed + 16, cores_),
            # a mixed-length batch of the headline's error profile: reads of 2 to 20 kb, issued longest first
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "mixed lengths: 100 k ONT-error pairs, read lengths uniform in 2 .. 20 kb, issued longest read first, 4 persistent "
                             "wavefronts per CU (scrg_params.waves_per_cu: with fewer wavefronts than groups of 64 pairs the work queue hands the "
                             "short pairs at the end of the order to the lanes that finish first)",
                             100000, 20000, "ont", 10, 2, 2000 if chk else 0, args.seed + 17, cores_, len_range=(2000, 20000), waves_per_cu=4),
            # two points of the reference's knob sweeps (scripts/profile.py:88-100 small overlaps, :180-185 W > 64) on the headline
            # workload: 32 <= W-O <= 63 runs on genasm_lane_wide_kernel (table in registers, built in two halves)
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "knob sweep point W=64 O=2: 100 k x 10 kb ONT-error pairs", 100000, 10000, "ont", 10, 2, 2000 if chk else 0,
                             args.seed + 14, cores_, W=64, O=2),
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "knob sweep point W=128 O=65: 100 k x 10 kb ONT-error pairs", 100000, 10000, "ont", 10, 2, 2000 if chk else 0,
                             args.seed + 15, cores_, W=128, O=65),
            # ... and two of its large-window points (:180-185; 64 <= W-O <= 127, vectors of 3 and 4 words): genasm_lane_parts_kernel
            # (table in registers, in parts of 16 columns re-swept from checkpoints)
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "knob sweep point W=192 O=97: 100 k x 10 kb ONT-error pairs", 100000, 10000, "ont", 6, 2, 1000 if chk else 0,
                             args.seed + 18, cores_, W=192, O=97),
            run_other_config(torch, scrooge_amd, device, local_rank, streams,
                             "knob sweep point W=256 O=129: 100 k x 10 kb ONT-error pairs", 100000, 10000, "ont", 6, 2, 1000 if chk else 0,
                             args.seed + 19, cores_, W=256, O=129),
        ]

    if host_api is not None and other_configs:
        for oc in other_configs:
            if "host_api" in oc:
                host_api["read_mapping_configs2"] = oc.pop("host_api")
    pairs_total = pairs_per_step_all * args.steps
    value = pairs_total / dt
    if text_used is None:
        text_used = L * 1.015
    bytes_per_pair = (L + 3) // 4 + int(text_used + 3) // 4 + 48 + 8 + 4 + 4 + 2 * runs_per_pair
    launch_bytes = bytes_per_pair * n
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf))
            if t.get("pairs") == n and t.get("read_len") == L and t.get("profile") == args.profile:
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # The binding roofline is VALU issue, not HBM (~9.4 KB and ~750 k issued lane-ops per pair).  achieved =
    # instructions per window round (SQ_INSTS_VALU / rounds, rocprofv3 --pmc, used only if it was measured on exactly this
    # kernel source) x window rounds of a launch (the kernel's own counter, this run) x 64 lanes / the launch's duration
    # (HIP events, this run); peak = 1024 SIMDs x 32 lanes per cycle x 2.4 GHz, i.e. every op priced as a full-rate op.
    kernel_name = "genasm_lane_kernel<false>" if p.lanes_per_pair == 1 else "genasm_align_kernel<%d, false>" % p.lanes_per_pair
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
           "algorithmic_bytes_per_pair": bytes_per_pair, "algorithmic_bytes_per_launch": launch_bytes,
           "note": "algorithmic bytes of a launch / its duration; `traffic` = HBM bytes per launch from the PMC counters (profiles/hbm_traffic.json)"}
    pmc, instr_src = (pmc_instruction_count() if p.lanes_per_pair == 1 else (None, "PMC summary exists for the one-pair-per-lane kernel only"))
    instr = float(pmc["valu_instructions_per_window_round"]) if pmc else None
    mix_roof = None          # what the chip sustains for THIS kernel's instruction mix (scripts/mix_roof.py from scripts/ubench/valu_rate.hip)
    try:
        import glob
        for mf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mix_roof.json")), reverse=True):
            mj = json.load(open(mf))
            if mj.get("kernel_sources_sha256") == kernel_sources_digest():
                mix_roof = {"T_lane_ops_per_s": mj["roof"]["4 waves/SIMD"]["T_lane_ops_per_s"], "at": "4 wavefronts per SIMD",
                            "source": "profiles/%s: static opcode mix of the kernel x per-class issue rates measured on this chip" % os.path.basename(mf)}
                break
    except Exception:
        pass
    # The instructions one launch issues.  The profiled launch IS this run's stand-alone launch when the workload is the one
    # the PMC passes ran (same pairs, read length, profile, seed: pipeline lane 0's batch): SQ_INSTS_VALU of that launch is
    # used as counted.  Otherwise: instructions per window round x the window rounds of this launch — counted by the kernel
    # itself in a -DSCRG_STATS build, else estimated from the CPU checker's window count (64 windows per round, + 0.5 % for
    # the lanes that wait at the end of a wavefront's life).
    lane_ops_launch, rounds, rounds_src = None, rounds_live, "the kernel's own counter (profiling build)"
    if pmc is not None:
        wl = pmc.get("workload") or {}
        same_wl = (wl.get("pairs") == n and wl.get("read_len") == L and wl.get("profile") == args.profile and wl.get("seed") == args.seed
                   and n_real == n and not same_batch)
        if same_wl and pmc.get("SQ_INSTS_VALU"):
            lane_ops_launch = float(pmc["SQ_INSTS_VALU"]) * 64.0
            rounds = rounds_live or pmc.get("window_rounds_per_launch")
            rounds_src = "SQ_INSTS_VALU of this very launch (same workload and seed), %s" % instr_src
        elif rounds_live:
            lane_ops_launch = instr * rounds_live * 64.0
        elif windows is not None:
            rounds = n * windows / 64.0 * 1.005
            rounds_src = "estimated: pairs x windows per pair (CPU checker) / 64 x 1.005"
            lane_ops_launch = instr * rounds * 64.0
    if lane_ops_launch is not None:
        a_valu = lane_ops_launch / (kernel_ms * 1e-3)
        roofline = {"bound": "valu-issue", "achieved": a_valu / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T int32 lane-ops/s",
                    "frac": a_valu / VALU_PEAK_LANE_OPS, "traffic": traffic, "kernel": kernel_name,
                    "valu_instructions_per_window_round": instr, "instruction_count_source": instr_src,
                    "window_rounds_per_launch": rounds, "window_rounds_source": rounds_src,
                    "issued_lane_ops_per_launch": lane_ops_launch, "kernel_ms": kernel_ms,
                    "frac_at_step_rate": lane_ops_launch / (dt / args.steps) / VALU_PEAK_LANE_OPS if not dist_on else None,
                    "frac_sustained": (lane_ops_launch / (sustained["ms_per_step"] * 1e-3) / VALU_PEAK_LANE_OPS) if sustained else None,
                    "mix_weighted_roof": mix_roof,
                    "note": "`achieved` counts the VALU instructions the kernel REALLY ISSUED (SQ_INSTS_VALU from rocprofv3 --pmc, x 64 lanes) "
                            "over the duration of a launch that has the GPU to itself (1.5 wavefronts per SIMD: cannot fill the issue slots); "
                            "frac_at_step_rate divides by ms_per_step of the pipelined timed region instead (2-3 launches share the SIMDs), "
                            "frac_sustained by the sustained step; peak prices every op as full-rate at 2.4 GHz — the mix-weighted issue roof "
                            "of this instruction mix is `mix_weighted_roof` (DESIGN.md §3.1).  SURVEY.md §8(d)'s op count (7 64-bit ops per "
                            "R[i][d] cell of genasm_cpu.cpp:247-251 = 14 lane-ops x dc_cells + the traceback) is `frac_reference_ops`: it reads "
                            "ABOVE 1 at the step rate, and that is not skipped work — the kernel does not run the reference's per-distance "
                            "recurrence; it carries the same table as Myers/Hyyro difference vectors (all 64 rows and every distance d of a "
                            "text column in 19 VALU instructions, no loop over d), which issues ~3.6x fewer lane-ops for the same R table, and "
                            "this same run compares every pair of every timed batch (edit distances, run counts, all runs) byte for byte "
                            "with the reference CPU path (`parity`)",
                    "hbm": hbm}
        if windows is not None and p.lanes_per_pair == 1 and p.W <= 64 and p.W - p.O <= 31:
            # USEFUL work of the formulation this kernel really runs, per pair and at each lane's own counts — not
            # the maximum over a wavefront's 64 lanes, not the instructions of idle lanes, fetches, set-up or stores:
            #   the table     19 VALU per text column x 64 columns per window (all 64 rows and every distance at once)
            #   pass 1        10 VALU per traceback column x (W-O) columns per window (the walk, branch-free)
            #   pass 2        20.5 VALU per event (41 per trip of two) x the pair's events: one per run that starts in a window =
            #                 the pair's runs (insertion runs and D / X / = runs are separate events)
            # windows per pair: the CPU checker's count on this batch; runs per pair: this launch's own output.
            USEFUL_TABLE, USEFUL_PASS1, USEFUL_PASS2 = 19.0 * 64.0, 10.0 * (p.W - p.O), 20.5
            useful_pair = windows * (USEFUL_TABLE + USEFUL_PASS1) + runs_per_pair * USEFUL_PASS2
            useful_launch = useful_pair * n_real
            roofline["useful_lane_ops"] = {
                "lane_ops_per_pair": useful_pair, "windows_per_pair": windows, "runs_per_pair": runs_per_pair,
                "formula": "windows x (19 x 64 table + 10 x (W-O) pass 1) + runs x 20.5 pass 2, per lane at its own counts (DESIGN.md §3.1)",
                "useful_over_issued": useful_launch / lane_ops_launch,
                "frac_useful": useful_launch / (kernel_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                "frac_useful_at_step_rate": (useful_launch / (dt / args.steps) / VALU_PEAK_LANE_OPS) if not dist_on else None,
                "frac_useful_sustained": (useful_launch / (sustained["ms_per_step"] * 1e-3) / VALU_PEAK_LANE_OPS) if sustained else None,
                "note": "what a regression shows in: `frac` (issued / peak) goes UP if the kernel is padded with instructions and does not "
                        "move if lanes idle more; `frac_useful` (this) only moves with time per pair; `frac_reference_ops` prices the "
                        "reference's per-distance recurrence, which this kernel does not run"}
            roofline["frac_useful"] = roofline["useful_lane_ops"]["frac_useful_at_step_rate"] or roofline["useful_lane_ops"]["frac_useful"]
        if dc_cells is not None:
            # what SURVEY.md §8(d) asks `achieved` to be computed from: 14 int32 lane-ops per GenASM-DC cell + ~13 per traceback step
            ref_lane_ops_pair = 14.0 * dc_cells + 13.0 * (tb_steps or 0)
            ref_launch = ref_lane_ops_pair * n_real
            roofline["reference_ops"] = {
                "lane_ops_per_pair": ref_lane_ops_pair, "formula": "14 x dc_cells + 13 x tb_steps (SURVEY.md §8d; counted by the CPU checker on this batch)",
                "issued_over_reference": lane_ops_launch / ref_launch,
                "frac_reference_ops": ref_launch / (kernel_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                "frac_reference_ops_at_step_rate": (ref_launch / (dt / args.steps) / VALU_PEAK_LANE_OPS) if not dist_on else None,
                "note": "the reference formulation's op count for the same pairs priced at this kernel's times: > 1 at the step rate because "
                        "the kernel issues issued_over_reference x as many lane-ops as that formulation would need, not because work is skipped"}
            roofline["frac_reference_ops"] = roofline["reference_ops"]["frac_reference_ops_at_step_rate"] or roofline["reference_ops"]["frac_reference_ops"]
    else:
        roofline = {"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T int32 lane-ops/s", "frac": None,
                    "traffic": traffic, "kernel": kernel_name, "window_rounds_per_launch": rounds_live,
                    "note": "no instruction count for this build: " + str(instr_src), "hbm": hbm}
    probe_env = os.environ.get("SCRG_BENCH_PROBE") if args.headline_only else None
    out = {
        # (a line made with a measuring aid that leaves work out of the step says so in its metric: it is not a measurement of the path)
        "metric": "aligned pairs/s (+ GCUPS) at W=64, 10kb reads; 1/2/4/8 MI355X" if not probe_env
                  else "INVALID AS A RESULT: ablation SCRG_BENCH_PROBE=%s (scripts/r06_chain_probe.sh)" % probe_env,
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "unstructured pairwise: %d x %d bp %s-error pairs per GPU and step (%s), W=%d O=%d"
                               % (nominal, L, args.profile,
                                  "BASELINE configs[3]: 1 M pairs over 8 GPUs, every step's results gathered to and decoded on its root" if (world == 8 and nominal == 125000)
                                  else ("BASELINE configs[1]" if (world == 1 and n == 100000 and L == 10000) else "%d GPU(s)" % world), p.W, p.O),
                   "pairs_per_gpu": nominal, "pairs_per_step_all_gpus": pairs_per_step_all,
                   "shards": ({"rank_0": n0, "other_ranks": n, "root_share": args.root_share,
                               "note": "rank 0 also decodes every rank's CIGARs (~0.2 of the align work per pair): it aligns a smaller share, "
                                       "the other ranks split the rest of the step's %d pairs" % (world * nominal)} if share_on else "equal"),
                   "read_len": L, "text_len": text_len, "error_profile": args.profile,
                   "W": p.W, "O": p.O, "lanes_per_pair": p.lanes_per_pair, "lds_rows": p.lds_rows,
                   "sequence_layout": "lane-interleaved groups of 64 pairs (scrg_pack_planar_groups)" if groups else "contiguous per sequence",
                   "launch": geom, "step": STEP_TEXT[gather_format if dist_on else "local"],
                   "backend": (dist.get_backend() if dist_on else None), "rccl_ranks": (dist.get_world_size() if dist_on else 1),
                   "gather": ({"format": gather_format, "root": ("step k to rank k mod N" if (edits and args.gather_root == "rotate" and world > 1) else "rank 0"),
                               "decoded_to_runs_inside_timed_region": bool(decode_on),
                               "bytes_per_rank_and_step": gather.wire if edits else None,
                               "stream_bytes_per_pair": (stream_bytes / n) if stream_bytes is not None else None} if dist_on else None),
                   "pipeline": ("consecutive steps rotate over %d streams, each with its own handle, work queue, output buffers and its OWN "
                                "INPUT BATCH (%s): a step's wavefronts start while the previous step's last pairs finish"
                                % (n_lanes, "experiment: all lanes share lane 0's batch" if same_batch else
                                   "different seeds; %.2f GB of packed sequence in total" % seq_total_gb)) if n_lanes > 1
                               else "one stream: a step starts after the previous one has finished"},
        "gcups": value * L * L / 1e9,           # EQUIVALENT full-matrix GCUPS (pairs/s x L x L, the reference's convention, scripts/profile.py:426-427)
        "gcups_note": "equivalent full L x L matrix cells (the reference's convention); the cells the windowed algorithm really computes are in bit_cell_gcups",
        "kernel_ms": kernel_ms,        # HIP events around one align launch that has the GPU to itself
        "kernel_pairs_per_s_per_gpu": n / (kernel_ms * 1e-3),
        "kernel_ms_events_in_timed_region": events_ms,   # pipelined launches: includes waiting for the previous launch's wavefronts to retire
        "serial": serial,              # the same step without overlap between launches, measured after the timed region
        "sustained": sustained,        # the same pipelined step over many more steps (fill and drain amortised), after the timed region
        "value_incl_pack": incl_pack,  # the same pipelined step with the packing of the sequences inside it (ASCII in HBM -> runs), after the timed region
        "other_configs": other_configs,   # BASELINE configs[0], [2], [4], Cfg2b, mixed lengths, two knob-sweep points on this GPU, measured after the timed region, each with an oracle-checked sample
        "host_api": host_api,             # the host-pointer entry points (PCIe-inclusive; never `value`), after the timed region
        "edit_stream_step": edit_stream_step,   # the N > 1 step (CIGARs as edit streams) on this one GPU, without the collective
        "gather_without_decode": streams_only,  # N > 1: the same steps with the gathered CIGARs left as edit streams (after the timed region)
        "per_gpu_value": per_gpu_value,         # N > 1: every rank's own N=1-equivalent step rate (compare with BENCH's `value`)
        "diagnose": diagnose,                   # N > 1: the same steps under the other policies, the gather alone, per-peer GB/s (after the timed region)
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity": parity,
        "gather_check": gather_check,
        "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,     # CPU time to enqueue a step (includes waiting for a free buffer set)
        "gen_seconds": gen_s,
    }
    if dc_cells is not None:
        ref_ops = 14 * dc_cells       # 7 64-bit logic ops per GenASM-DC cell = 14 int32 lane-ops (genasm_cpu.cpp:247-251)
        if p.lanes_per_pair == 1:
            # The lane kernel does not run the GenASM rows: per window it issues 64 text columns x 23 VALU
            # instructions for the table (all 64 pattern rows, every distance at once), and the traceback, setup and
            # queue code around it.  The instruction count per window round (one window of each of a wavefront's 64
            # pairs) is measured: SQ_INSTS_VALU / rounds in profiles/ (rocprofv3 --pmc); a lane-op = one lane of one
            # wave64 VALU instruction.  Peak = 1024 SIMDs x 32 lanes/cycle x 2.4 GHz (full-rate ops; the mix here has
            # ~25 % half-rate ops and the chip runs this kernel at ~2.0 GHz).
            instr = roofline.get("valu_instructions_per_window_round")          # None if profiles/ has no count for this kernel source
            lane_ops = instr * windows if instr else None   # per pair: one lane's share of every instruction of its windows' rounds
            out["valu"] = {"issued_lane_ops_per_pair": lane_ops, "valu_instructions_per_window_round": instr,
                           "windows_per_pair": windows, "achieved": (lane_ops * value / world) if lane_ops else None, "peak": VALU_PEAK_LANE_OPS,
                           "frac": (lane_ops * value / world / VALU_PEAK_LANE_OPS) if lane_ops else None, "unit": "int32 lane-ops/s",
                           "reference_formulation_lane_ops_per_pair": ref_ops,
                           "reference_formulation_equivalent_rate": ref_ops * value / world,
                           "dc_cells_per_pair": dc_cells, "tb_steps_per_pair": tb_steps,
                           "note": "per GPU, from the whole-step rate (the `roofline` object is the same figure for one launch from this "
                                   "run's own counters); 'reference_formulation_*' prices the same pairs at the 14 lane-ops per R[i][d] cell of "
                                   "genasm_cpu.cpp:247-251 (what round 1's kernel executed)"}
        else:
            out["valu"] = {"algorithmic_lane_ops_per_pair": ref_ops, "dc_cells_per_pair": dc_cells,
                           "tb_steps_per_pair": tb_steps,
                           "achieved": ref_ops * value / world, "peak": VALU_PEAK_LANE_OPS,
                           "frac": ref_ops * value / world / VALU_PEAK_LANE_OPS, "unit": "int32 lane-ops/s",
                           "note": "per GPU, from the whole-step rate"}
        out["bit_cell_gcups"] = value * dc_cells * 64 / 1e9
    if want_diag:
        out["diagnose"], out["per_gpu_value"] = guarded_diagnostics(out)
    print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
