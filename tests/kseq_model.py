"""kseq_read (reference include/kseq.h:178-219) restated byte by byte: the model the FASTA/FASTQ reader of
the command-line tools (rowbowt_amd/csrc/fastx.hpp) is tested against.  -> (records seen by the caller's
`while (kseq_read(seq) >= 0)` loop, final return code)."""


def kseq_model(data: bytes):
    """kseq_read (reference include/kseq.h:178-219) restated byte by byte, for the CLI parser test."""
    recs, i, n, last = [], 0, len(data), 0
    def getc():
        nonlocal i
        if i >= n:
            return -1
        i += 1
        return data[i - 1]
    def getline():  # ks_getuntil2(KS_SEP_LINE): rest of line, without '\n'
        nonlocal i
        if i >= n:
            return None
        j = data.find(b"\n", i)
        if j < 0:
            j = n
        s = data[i:j]
        i = min(j + 1, n)
        return s
    while True:
        if last == 0:
            c = getc()
            while c >= 0 and c not in (62, 64):
                c = getc()
            if c < 0:
                return recs, -1
            last = c
        # name up to whitespace, rest of the line = comment
        j = i
        while j < n and not chr(data[j]).isspace():
            j += 1
        if j == i and i >= n:
            return recs, -1
        name = data[i:j]
        i = j
        if i < n and data[i] != 10:
            getline()
        elif i < n:
            i += 1
        seq = b""
        c = getc()
        while c >= 0 and c not in (62, 43, 64):
            if c != 10:
                seq += bytes([c]) + (getline() or b"")
                if len(seq) > 1 and seq.endswith(b"\r"):
                    seq = seq[:-1]
            c = getc()
        last = c if c in (62, 64) else 0
        if c != 43:
            recs.append((name, seq))
            if c < 0:
                return recs, -1
            continue
        c = getc()
        while c >= 0 and c != 10:
            c = getc()
        if c < 0:
            return recs, -2          # kseq.h:213; the caller's loop never sees this record (rb_align.cpp:176)
        qual = b""
        while True:
            l = getline()
            if l is None:
                break
            qual += l
            if len(qual) > 1 and qual.endswith(b"\r"):
                qual = qual[:-1]
            if len(qual) >= len(seq):
                break
        last = 0
        if len(qual) != len(seq):
            return recs, -2          # kseq.h:217
        recs.append((name, seq))
