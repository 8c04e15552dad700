#!/usr/bin/env python3
"""The VALU issue roof of genasm_lane_kernel for ITS instruction mix (build container; no GPU needed).

Static opcode histogram of the kernel's code (hipcc -S of genasm_lane_kernel.hip) weighted with the issue rates measured
on the MI355X by scripts/ubench/valu_rate.hip (profiles/r03_valu_issue_rates.txt: wall time of 4096 x 16 instructions per
wavefront at 1 / 2 / 4 wavefronts per SIMD on all 1024 SIMDs, i.e. under the clock the chip really sustains).

    python3 scripts/mix_roof.py            -> JSON on stdout (profiles/r03_mix_roof.json is a copy)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = os.path.join(ROOT, "profiles", "r03_valu_issue_rates.txt")
INSTS_PER_WAVE = 4096 * 16
PEAK = 256 * 4 * 32 * 2.4e9            # nominal: 1024 SIMDs x 32 lanes per cycle x 2.4 GHz

# ubench row that stands for an opcode (same encoding class / same unit)
CLASS_OF = [
    (r"v_bitop3_b32", "v_bitop3_b32"),
    (r"v_(and|or|xor|not|mov)_b32|v_(add|sub|subrev)_u32|v_ashrrev_i32|v_lshrrev_b32|v_mov_b64|v_bcnt", "v_and_b32"),
    (r"v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64", "v_lshlrev_b64"),
    (r"v_lshlrev_b32|v_lshl_add_u32|v_lshl_or_b32|v_add_lshl_u32|v_and_or_b32|v_or3_b32|v_add3_u32|v_alignbit_b32|v_perm_b32", "v_alignbit_b32"),
    (r"v_ffbh_u32|v_ffbl_b32|v_bfe_[ui]32|v_bfrev_b32|v_min_u32|v_max_u32|v_min3_u32|v_max_i32", "v_ffbh_u32"),
    (r"v_cmp_|v_cmpx_", "v_cmp_lt_i32"),
    (r"v_cndmask_b32", "v_cndmask_e64 sgpr"),
    (r"v_add_co_u32|v_addc_co_u32|v_subb|v_sub_co", "v_add_co_u32 vcc"),
    (r"v_readfirstlane|v_readlane|v_writelane|v_mbcnt", "v_mbcnt_lo"),
]


def rates():
    """-> {waves per SIMD: {ubench row: wavefront instructions per second of the whole GPU}}"""
    out, cur = {}, None
    for line in open(RATES):
        m = re.match(r"--- (\d+) wave", line)
        if m:
            cur = out.setdefault(int(m.group(1)), {})
            continue
        m = re.match(r"(\S.*?)\s+wall ([0-9.]+) ms", line)
        if m and cur is not None:
            wps = [k for k, v in out.items() if v is cur][0]
            cur[m.group(1).strip()] = INSTS_PER_WAVE * wps * 1024 / (float(m.group(2)) * 1e-3)
    return out


def histogram():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm,
                               os.path.join(ROOT, "scrooge_amd", "csrc", "genasm_lane_kernel.hip")], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    m = re.search(r"^_ZN4scrg18genasm_lane_kernelILb0EEEvNS_9AlignArgsE:(.*?)s_endpgm", text, re.S | re.M)
    h = collections.Counter()
    for line in m.group(1).splitlines():
        line = line.strip()
        if line.startswith("v_"):
            h[line.split()[0]] += 1
    return h


def main():
    r = rates()
    h = histogram()
    total = sum(h.values())
    classes = collections.Counter()
    unmatched = collections.Counter()
    for op, c in h.items():
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        for pat, row in CLASS_OF:
            if re.match(pat, base):
                classes[row] += c
                break
        else:
            unmatched[base] += c
            classes["v_alignbit_b32"] += c          # priced as the slow class
    sys.path.insert(0, ROOT)
    import bench
    out = {"kernel": "genasm_lane_kernel<false>", "kernel_sources_sha256": bench.kernel_sources_digest(), "static_valu_instructions": total,
           "class_share": {k: v / total for k, v in classes.most_common()},
           "unmatched_opcodes_priced_as_slow": dict(unmatched),
           "nominal_peak_T_lane_ops": PEAK / 1e12, "roof": {}}
    for wps, rr in sorted(r.items()):
        t = sum(share / rr[row] for row, share in out["class_share"].items())      # seconds per wavefront instruction of the mix
        out["roof"]["%d waves/SIMD" % wps] = {"T_lane_ops_per_s": 64.0 / t / 1e12, "frac_of_nominal_peak": 64.0 / t / PEAK,
                                             "full_rate_op_T_lane_ops_per_s": 64.0 * rr["v_and_b32"] / 1e12}
    out["note"] = ("issue rate of this kernel's static instruction mix at the per-class rates the chip sustains (wall time of the "
                   "micro-benchmark on all SIMDs, launch overhead and the clock the chip settles at included); the nominal peak prices "
                   "every instruction as a full-rate op at 2.4 GHz")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
