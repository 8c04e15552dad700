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
