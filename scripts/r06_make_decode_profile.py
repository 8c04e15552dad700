#!/usr/bin/env python3
"""gpurun_out/r06_dec_timing_*.json (scripts/r06_decode_run.sh), r06_dec_pmc_{sq,hbm}.json (scripts/decode_pmc.sh) -> profiles/r06_decode_timing.json"""
import hashlib, json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
def jl(p): return [json.loads(l) for l in open(root + p) if l.startswith('{')]
t = {k: jl('gpurun_out/r06_dec_timing_%s.json' % k)[0] for k in ('quad', 'wave', 'lane')}
t1k = {k: jl('gpurun_out/r06_dec_timing_%s_1k.json' % k)[0] for k in ('quad', 'wave')}
def slim(d): return {k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in d.items() if k.startswith('slots_')}
pmc = json.load(open(root + 'gpurun_out/r06_dec_pmc_sq.json'))
hbm = json.load(open(root + 'gpurun_out/r06_dec_pmc_hbm.json'))
src = hashlib.sha256(open(root + 'scrooge_amd/csrc/edit_stream_decode_kernel.hip', 'rb').read()).hexdigest()
trips = 507e3
out = {
 "what": "scripts/decode_timing.py on one MI355X (100 000 x 10 kb ONT pairs per slot, streams in format 2: 1 299 bytes and 2 141 runs per pair = 0.558 GB read + written per slot), scrg_decode_edit_stream alone on the GPU after 60 align launches (clock warm-up), 10 launches averaged; count only (d_dense = NULL) and the one-pass decode into the dense scrg_run array, every slot compared with the runs kernel's compacted output.  SCRG_DEC_KERNEL selects the kernel; the library's choice for these streams is decode_edits_quad_kernel.",
 "decode_kernel_source_sha256": src,
 "decode_edits_quad_kernel (round 6: one pair per wavefront, four bytes per lane, the library's choice for streams >= 64 bytes)": {"10kb": slim(t['quad']), "1kb_100k_pairs": slim(t1k['quad']),
    "roofline": {"bound": "hbm", "algorithmic_GB_per_slot": 0.558, "one_slot_GBs": round(t['quad']['slots_1']['GB_per_s'], 0), "frac_of_8TBs": round(t['quad']['slots_1']['GB_per_s'] / 8000, 3),
                 "traffic_from_counters_one_slot": {k: v["smallest_launches_mean"] for k, v in sorted(hbm.items())}, "traffic_note": "WRITE_SIZE / FETCH_SIZE in KB (scripts/decode_pmc.sh 8 hbm): 0.456 GB written = 1.06 x the 0.428 GB of runs; the streams (0.13 GB) mostly come from L2 / MALL, where the copy of the slot has just put them"}},
 "decode_edits_wave_kernel (round 5: one pair per wavefront, one byte per lane; SCRG_DEC_KERNEL=wave)": {"10kb": slim(t['wave']), "1kb_100k_pairs": slim(t1k['wave'])},
 "decode_edits_kernel (one pair per lane: the choice for streams < 64 bytes; SCRG_DEC_KERNEL=lane)": {"10kb": slim(t['lane'])},
 "steps_of_the_quad_kernel (one slot, decode ms)": {
   "first version: every byte ADDS to its run's 32-bit slot (ds_add_u32), slots zeroed behind the units": 0.248,
   "  probe builds of that version (profiles/r06_decode_probe_builds.txt): no global stores / no zeroing / plain writes for the adds / no adds / no flush": [0.225, 0.224, 0.174, 0.155, 0.177],
   "every run written once by its last byte (no atomics, no zeroing), one store instruction per trip, counted wait (vmcnt(1)) for the next trip's dwords, the next pair's numbers loaded a pair ahead": 0.171,
   "pair epilogue: the units a pair shares with its neighbours go out one run per lane (was: eight predicated stores by one lane); one scan for the placed characters": 0.1615,
   "carries between trips in vector registers (DPP wave_ror / wave_shr old operand), one compare for the store predicate: scalar instructions 33.9 M -> 26.8 M per slot": 0.1552,
   "16-bit ring slots (9 KB instead of 17 KB of LDS per workgroup, one ds_read_b128 per unit, no packing)": round(t['quad']['slots_1']['decode_ms'], 4)},
 "sq_counters_quad_kernel (scripts/decode_pmc.sh 8 sq: rocprofv3 --pmc, kernel trace only; the shipped kernel; one slot = 507 k trips of 256 bytes / eight slots)": {k: {"one_slot": v["smallest_launches_mean"], "eight_slots": v["largest_launches_mean"]} for k, v in sorted(pmc.items())},
 "per_trip_of_256_bytes (one-slot counters / 507 k trips, per-pair work included)": {"VALU": round(pmc["SQ_INSTS_VALU"]["smallest_launches_mean"] / trips, 1), "SALU": round(pmc["SQ_INSTS_SALU"]["smallest_launches_mean"] / trips, 1), "LDS": round(pmc["SQ_INSTS_LDS"]["smallest_launches_mean"] / trips, 1),
    "round_5_wave_kernel_per_256_bytes": {"VALU": 253.6, "SALU": 135.6, "VMEM stores": 10.8}},
 "round_5": "decode_edits_wave_kernel one slot 0.296 ms (1.77 TB/s = 0.22 of 8 TB/s), eight slots 2.26 ms (profiles/r05_decode_timing.json)"}
json.dump(out, open(root + 'profiles/r06_decode_timing.json', 'w'), indent=1)
print(out["per_trip_of_256_bytes (one-slot counters / 507 k trips, per-pair work included)"], {k: v["smallest_launches_mean"] for k, v in hbm.items()})
