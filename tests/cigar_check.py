"""Independent property checker for (CIGAR, edit distance) results — the same
invariants as the reference's validateCigarString (src/tests.cu:27-169):
format (\\d+[=XID])*, no zero counts, the read is consumed exactly, the text
prefix stays inside the text, '='/'X' agree with the sequences, and the number
of edit operations equals the reported edit distance."""
import re

_TOKEN = re.compile(r"(\d+)([=XID])")


def parse(cigar):
    pos, runs = 0, []
    for m in _TOKEN.finditer(cigar):
        if m.start() != pos:
            raise ValueError("malformed CIGAR at %d" % pos)
        runs.append((int(m.group(1)), m.group(2)))
        pos = m.end()
    if pos != len(cigar):
        raise ValueError("malformed CIGAR tail")
    return runs


def validate(text, read, cigar, edit_distance):
    """Returns None if consistent, else a string describing the first violation."""
    text = text.upper() if isinstance(text, str) else text.decode().upper()
    read = read.upper() if isinstance(read, str) else read.decode().upper()
    try:
        runs = parse(cigar)
    except ValueError as e:
        return str(e)
    i = j = edits = 0
    for cnt, op in runs:
        if cnt == 0:
            return "zero-length run"
        if op in "=X":
            if i + cnt > len(text) or j + cnt > len(read):
                return "run past the end of a sequence"
            for k in range(cnt):
                same = text[i + k] == read[j + k]
                if same != (op == "="):
                    return "'%s' disagrees with sequences at text %d read %d" % (op, i + k, j + k)
            i += cnt
            j += cnt
        elif op == "I":
            if j + cnt > len(read):
                return "insertion past the end of the read"
            j += cnt
        else:
            if i + cnt > len(text):
                return "deletion past the end of the text"
            i += cnt
        if op != "=":
            edits += cnt
    if j != len(read):
        return "read not consumed exactly (%d of %d)" % (j, len(read))
    if edits != edit_distance:
        return "edit count %d != reported %d" % (edits, edit_distance)
    return None


def validate_batch(torch, rows, text_off, text_len, read_off, read_len, runs_u8, run_off, cnt, ed, chunk_pairs=8192):
    """The same invariants for a whole BATCH as tensor arithmetic (any torch device; no per-pair Python): every pair p has
    its text at rows[p, text_off : text_off + text_len[p]] and its read at rows[p, read_off : read_off + read_len[p]] (ASCII,
    one row per pair), its runs at runs_u8[2 * run_off[p] : 2 * (run_off[p] + cnt[p])] as (count, op) byte pairs
    (scrg_run = the reference's CigarEntry_t, src/util.hpp:43-46) and its reported edit distance in ed[p].  Checked per
    pair, as validateCigarString does (src/tests.cu:27-169): ops are '=XID', no zero counts, the read is consumed exactly,
    the text is not overrun, every '=' column has equal characters and every 'X' column different ones, and the number of
    non-match columns is the edit distance.  text_len / read_len: an int or an int64 tensor per pair.
    -> int64 tensor of the indices of the pairs that violate any of them (empty = all good)."""
    dev = rows.device
    n = int(cnt.numel())
    bad_pairs = []
    cnt = cnt.to(torch.int64)
    run_off = run_off.to(torch.int64)
    ed = ed.to(torch.int64)
    tl_all = text_len if torch.is_tensor(text_len) else torch.full((n,), int(text_len), dtype=torch.int64, device=dev)
    rl_all = read_len if torch.is_tensor(read_len) else torch.full((n,), int(read_len), dtype=torch.int64, device=dev)
    for p0 in range(0, n, chunk_pairs):
        p1 = min(n, p0 + chunk_pairs)
        k = p1 - p0
        c = cnt[p0:p1]
        R = int(c.sum().item())
        bad = torch.zeros(k, dtype=torch.bool, device=dev)
        tl, rl = tl_all[p0:p1], rl_all[p0:p1]
        if R == 0:
            bad |= (rl != 0) | (ed[p0:p1] != 0)
            bad_pairs.append(torch.nonzero(bad).view(-1) + p0)
            continue
        pair_of_run = torch.repeat_interleave(torch.arange(k, device=dev), c)
        first = torch.cumsum(c, 0) - c                                             # index of a pair's first run in this chunk
        src = run_off[p0:p1][pair_of_run] + (torch.arange(R, device=dev) - first[pair_of_run])
        length = runs_u8[2 * src].to(torch.int64)
        op = runs_u8[2 * src + 1]
        is_eq, is_x, is_i, is_d = op == 61, op == 88, op == 73, op == 68            # '=', 'X', 'I', 'D'
        bad.index_put_((pair_of_run[~(is_eq | is_x | is_i | is_d) | (length == 0)],), torch.tensor(True, device=dev))
        on_text = (is_eq | is_x | is_d).to(torch.int64) * length
        on_read = (is_eq | is_x | is_i).to(torch.int64) * length
        z = torch.zeros(k, dtype=torch.int64, device=dev)
        t_used = z.clone().index_add_(0, pair_of_run, on_text)
        r_used = z.clone().index_add_(0, pair_of_run, on_read)
        edits = z.clone().index_add_(0, pair_of_run, (~is_eq).to(torch.int64) * length)
        bad |= (r_used != rl) | (t_used > tl) | (edits != ed[p0:p1])
        # where every run starts in its pair's text and read
        ct, cr = torch.cumsum(on_text, 0) - on_text, torch.cumsum(on_read, 0) - on_read
        t_start, r_start = ct - ct[first][pair_of_run], cr - cr[first][pair_of_run]
        # the '=' / 'X' columns, one element each
        diag = is_eq | is_x
        dl = length * diag.to(torch.int64)
        Ccols = int(dl.sum().item())
        if Ccols:
            run_of_col = torch.repeat_interleave(torch.arange(R, device=dev), dl)
            col0 = torch.cumsum(dl, 0) - dl
            within = torch.arange(Ccols, device=dev) - col0[run_of_col]
            pc = pair_of_run[run_of_col]
            ti, ri = t_start[run_of_col] + within, r_start[run_of_col] + within
            inside = (ti < tl[pc]) & (ri < rl[pc])
            ti, ri = torch.where(inside, ti, torch.zeros_like(ti)), torch.where(inside, ri, torch.zeros_like(ri))
            tc = rows[pc + p0, text_off + ti] & 0xDF                                # (upper case)
            rc = rows[pc + p0, read_off + ri] & 0xDF
            wrong = ~inside | ((tc == rc) != is_eq[run_of_col])
            bad.index_put_((pc[wrong],), torch.tensor(True, device=dev))
        bad_pairs.append(torch.nonzero(bad).view(-1) + p0)
    return torch.cat(bad_pairs) if bad_pairs else torch.zeros(0, dtype=torch.int64, device=dev)
