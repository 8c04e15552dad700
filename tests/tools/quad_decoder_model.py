"""A lane-by-lane model (numpy, 64 "lanes") of decode_edits_quad_kernel (scrooge_amd/csrc/edit_stream_decode_kernel.hip):
the edit-stream decoder that takes one pair per wavefront, FOUR stream bytes per lane (256 bytes per trip), classifies the
four bytes of a lane's dword side by side (SWAR), finds every run's index with one wavefront scan, and WRITES every run
once into a ring of 32-bit run slots in LDS: a stretch of matches by its byte, an edit run by its LAST byte (its length is
that byte's position in the run: a segmented count over the lane's bytes plus what the lane before hands over), the
ring leaves in aligned 16-byte units of eight runs.  Trips with 0x3F bytes or five equal edits in a row take the
per-byte path, where a joining byte ADDS 1 to its run's slot.

This file is TEST INFRASTRUCTURE: the arithmetic below is the kernel's, statement for statement, so that the CPU suite can
hold the formulation (fast path, the per-byte path for 0x3F bytes and long edit runs, the ring and its units, segments that
start anywhere in the dense array, capacities) to the format's definition (tests/test_edit_stream.py: py_decode) without a
GPU.  Nothing in scrooge_amd/ imports it."""
import numpy as np

U32 = np.uint32
RING = 1024                      # run slots (dwords) per wavefront
M32 = 0xFFFFFFFF
LETTERS = 0x44495800             # "\0XID", byte e


def _u(x):
    return np.asarray(x, dtype=np.uint64) & np.uint64(M32)


class Model:
    def __init__(self, dense_len):
        self.ring = np.full(RING, 0xBAD, dtype=np.uint64)      # 32-bit slots (LDS comes as it was left)
        self.dense = np.zeros(dense_len, dtype=np.uint16)     # scrg_run pairs as 16-bit words (count | letter << 8)
        self.written = np.zeros(dense_len, dtype=bool)
        self.fast_chunks = self.slow_chunks = 0

    # ---- the ring: every unit (8 runs, 16 bytes of the dense array, absolute index U) below u_lim leaves, and is zeroed
    def flush(self, g0, cap, uf, u_lim, n_final):
        """uf: first unit not yet written; u_lim: units below it are final; n_final: runs (relative) that exist (for the
        partial units at a pair's two ends).  Returns the new uf."""
        while uf < u_lim:
            for lane in range(64):
                U = uf + lane
                if U >= u_lim:
                    break
                base = (U * 8) & (RING - 1)
                words = self.ring[base:base + 8].copy()
                self.ring[base:base + 8] = 0xDEAD                 # (the kernel leaves them: every slot is written before it is read again)
                lo, hi = U * 8, U * 8 + 8
                whole = lo >= g0 and hi <= g0 + min(n_final, cap)
                for k in range(8):
                    G = lo + k
                    if g0 <= G < g0 + min(n_final, cap):
                        assert whole or True
                        assert not self.written[G], "a run stored twice"
                        self.dense[G] = int(words[k]) & 0xFFFF
                        self.written[G] = True
            uf = min(uf + 64, u_lim)
        return uf

    def write(self, slot_abs, value, pred):
        """ds_write_b32 of the lanes whose predicate holds (the kernel: the others write to a slot nobody reads)"""
        seen = set()
        for s, v, p in zip(np.atleast_1d(slot_abs), np.atleast_1d(value), np.atleast_1d(pred)):
            if p:
                assert int(s) not in seen, "two lanes write one slot in one instruction"
                seen.add(int(s))
                self.ring[int(s) & (RING - 1)] = np.uint64(int(v) & M32)

    def add(self, slot_abs, value):
        """ds_add_u32 of every lane (value 0: nothing)"""
        for s, v in zip(np.atleast_1d(slot_abs), np.atleast_1d(value)):
            if int(v):
                self.ring[int(s) & (RING - 1)] = (self.ring[int(s) & (RING - 1)] + np.uint64(int(v))) & np.uint64(M32)

    def decode_pair(self, stream, read_len, g0, cap, store=True):
        """-> (n_runs, clean)"""
        s = bytes(stream)
        n = len(s)
        lane = np.arange(64, dtype=np.uint64)
        base = 0                    # runs so far
        carry_x = 0                 # the dword before the chunk (its top byte is the byte before lane 0's first)
        carry_more = 0
        chain = 0                   # bytes in a row, up to the end of the chunk before, that join the edit run before them
        over = 0
        placed = 0                  # (the kernel: per lane, summed at the end)
        uf = g0 >> 3
        carry_rp = 0                # the length so far of the edit run that reaches the end of the trip before (0: none does)
        for c0 in range(0, n, 256):
            chunk = s[c0:c0 + 256].ljust(256, b"\0")            # bytes behind the stream: zeros
            x = _u(np.frombuffer(chunk, dtype="<u4"))
            px = np.concatenate(([carry_x], x[:-1])).astype(np.uint64)
            pv = _u((x << np.uint64(8)) | (px >> np.uint64(24)))
            T = x & np.uint64(0x3F3F3F3F)
            Em = _u(T + np.uint64(0x7F7F7F7F)) & np.uint64(0x80808080)
            OPB = x & np.uint64(0xC0C0C0C0)
            Ed = (OPB | _u(OPB << np.uint64(1))) & np.uint64(0x80808080)
            D = x ^ (pv & np.uint64(0xC0C0C0C0))
            nz = _u((D & np.uint64(0x7F7F7F7F)) + np.uint64(0x7F7F7F7F)) | D
            Hd = nz & Ed
            Jn = Ed & ~nz & np.uint64(M32)
            m63 = _u(T + np.uint64(0x41414141)) & np.uint64(0x80808080)
            More = m63 & ~Ed & np.uint64(M32)
            any_more = bool(More.any())
            all_join = bool((Jn == np.uint64(0x80808080)).any())
            if any_more or carry_more or all_join or chain >= 248:
                self.slow_chunks += 1
                # ---- the per-byte path: four sub-chunks of 64 bytes, one byte per lane
                for sub in range(4):
                    b = np.array([(int(x[sub * 16 + (l >> 2)]) >> (8 * (l & 3))) & 0xFF for l in range(64)], dtype=np.uint64)
                    pos = c0 + sub * 64 + np.arange(64)
                    pb = np.concatenate(([int(carry_x) >> 24 if sub == 0 else int(b_prev_last)], b[:-1])).astype(np.uint64)
                    e, ln = b >> np.uint64(6), b & np.uint64(63)
                    is_edit = b > 63
                    joins = is_edit & (b == (pb & np.uint64(0xC0)))
                    Mb = (b == 0x3F) & (pos < n)
                    t = ln.copy()
                    # the 0x3F lanes directly below each lane
                    cnt = np.zeros(64, dtype=np.uint64)
                    for l in range(64):
                        k = l - 1
                        c = 0
                        while k >= 0 and Mb[k]:
                            c += 1
                            k -= 1
                        t[l] = int(ln[l]) + 63 * c + (carry_more if k < 0 else 0)
                    t[Mb] = 0
                    t[pos >= n] = 0
                    trailing = 0
                    for l in range(63, -1, -1):
                        if Mb[l]:
                            trailing += 1
                        else:
                            break
                    carry_more = 63 * trailing + (carry_more if trailing == 64 else 0)
                    over |= int(np.bitwise_or.reduce(t))
                    Q = t != 0
                    H = is_edit & ~joins
                    # the chain of joining bytes
                    lead = 0
                    while lead < 64 and joins[lead]:
                        lead += 1
                    if lead == 64:
                        chain += 64
                    else:
                        if chain + lead >= 255:
                            over |= 0x100
                        chain = 0
                        l = 63
                        while l >= 0 and joins[l]:
                            chain += 1
                            l -= 1
                    if chain >= 255:
                        over |= 0x100
                    before = np.cumsum(Q.astype(np.int64) + H.astype(np.int64)) - (Q.astype(np.int64) + H.astype(np.int64))
                    eq_slot = g0 + base + before
                    ed_slot = g0 + base + before + Q + H - 1
                    letters = np.array([(LETTERS >> (8 * int(v))) & 0xFF for v in e], dtype=np.uint64)
                    if store:
                        self.write(eq_slot, np.uint64(0x3D00) | t, Q)
                        self.write(ed_slot, (letters << np.uint64(8)) | np.uint64(1), H)
                        self.add(ed_slot, np.where(joins, 1, 0))
                    placed += int(ln.sum()) + int((((np.uint64(6) >> e) & np.uint64(1))).sum())
                    base += int(Q.sum() + H.sum())
                    b_prev_last = int(b[63])
                carry_x = int(x[63])
                # what the next side-by-side trip's first lane is handed: the length so far of the edit run that reaches the trip's end
                carry_rp = (chain + 1) & 0xFF if int(b[63]) > 63 else 0
            else:
                self.fast_chunks += 1
                e7, h7 = Em >> np.uint64(7), Hd >> np.uint64(7)
                R = e7 + h7
                P = _u(R + (R << np.uint64(8)))
                P = _u(P + (P << np.uint64(16)))
                c = P >> np.uint64(24)
                incl = np.cumsum(c)
                idxA = (incl - c + np.uint64(g0 + base))
                # ---- an edit byte's position in its run: a segmented count over the lane's bytes, plus what the lane before hands over
                E1 = Ed >> np.uint64(7)
                Jm = Jn | _u(Jn - (Jn >> np.uint64(7)))                      # 0xFF in the bytes that join
                v1 = _u(((E1 & (Jm >> np.uint64(8))) << np.uint64(8)) + E1)
                f1 = Jm & _u(Jm << np.uint64(8))
                v2 = _u(((v1 & (f1 >> np.uint64(16))) << np.uint64(16)) + v1)
                g1 = Jm & (_u(Jm << np.uint64(8)) | np.uint64(0xFF))
                F = g1 & (_u(g1 << np.uint64(16)) | np.uint64(0xFFFF))       # bytes 0..k all join
                B = (v2 >> np.uint64(24)) * np.uint64(0x01010101)
                prB = np.concatenate(([carry_rp * 0x01010101], B[:-1])).astype(np.uint64)
                RP = _u(v2 + (F & prB))
                # an edit byte is the last of its run unless the byte after it joins (the last lane's last byte: as far as it knows)
                Jnext = np.concatenate((Jn[1:], [0])).astype(np.uint64)
                nJ = _u((Jnext << np.uint64(24)) | (Jn >> np.uint64(8)))
                Tl = Ed & ~nJ & np.uint64(M32)
                for k in range(4):
                    pex = (P >> np.uint64(8 * (k - 1))) & np.uint64(0xFF) if k else np.zeros(64, dtype=np.uint64)
                    pin = (P >> np.uint64(8 * k)) & np.uint64(0xFF)
                    t_k = (T >> np.uint64(8 * k)) & np.uint64(0xFF)
                    eq_k = (Em >> np.uint64(8 * k + 7)) & np.uint64(1)
                    tl_k = (Tl >> np.uint64(8 * k + 7)) & np.uint64(1)
                    e_k = (x >> np.uint64(8 * k + 6)) & np.uint64(3)
                    rp_k = (RP >> np.uint64(8 * k)) & np.uint64(0xFF)
                    let = np.array([(LETTERS >> (8 * int(v))) & 0xFF for v in e_k], dtype=np.uint64)
                    if store:
                        self.write(idxA + pex, np.uint64(0x3D00) | t_k, eq_k.astype(bool))
                        self.write(idxA + pin - np.uint64(1), (let << np.uint64(8)) | rp_k, tl_k.astype(bool))
                carry_rp = int(RP[63]) >> 24
                # read characters placed: the matches, and one per X or I
                xi = ((x >> np.uint64(1)) ^ x) & np.uint64(0x40404040)
                placed += sum(int(v).bit_count() for v in xi) + sum(sum(int(v).to_bytes(4, "little")) for v in T)
                base += int(incl[63])
                carry_x = int(x[63])
                # bytes at the chunk's end that join the run before them (lane 63 is never all four: that takes the other path)
                jn = int(Jn[63])
                chain = 0
                for k in (3, 2, 1):
                    if (jn >> (8 * k + 7)) & 1:
                        chain += 1
                    else:
                        break
            if store:
                # every unit below the last run (which the next chunk may still add to) is final
                u_lim = (g0 + base - 1) >> 3 if base > 0 else uf
                uf = self.flush(g0, cap, uf, max(u_lim, uf), base)
        if store:
            uf = self.flush(g0, cap, uf, (g0 + base + 7) >> 3, base)
        last = s[-1] if n else 0
        clean = carry_more == 0 and (last >> 6) == 0 and last != 0x3F and placed == read_len and (over >> 8) == 0
        return base, clean
