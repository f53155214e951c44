// fastx_index.hpp -- FASTA/FASTQ front end of rb_align at GPU rate (next-row f2): instead of copying every record
// into strings (kseq_read, reference include/kseq.h:178-219; rb_align.cpp:176) the input stays where it is -- a
// memory-mapped file or a decompressed block -- and a scanner records WHERE each record's name and sequence lie.
// The search then reads the sequences straight from that buffer (rbg_find_range_spans), and the formatter reads the
// names from it.  The scanner is kseq_read's state machine at line granularity (memchr per line), so the records it
// reports -- and the point and kind of failure on malformed input -- are exactly kseq's (tests/test_fastx_host.py
// runs it against the byte-by-byte model on well-formed, truncated and adversarial inputs).  Records whose sequence
// is not one contiguous run of bytes (several lines, a trailing '\r') are rebuilt in a side arena.
//
// Large buffers are scanned by several threads: the buffer is cut into segments, every segment but the first starts
// at a GUESSED record boundary (a line starting with '@' whose next-but-one line starts with '+', or a line
// starting with '>'), and a guess is accepted only if the exact scan of the segment before it ends precisely there;
// otherwise that segment is scanned again from where the exact scan stands.  The result never depends on the cut.
#pragma once

#include <cctype>
#include <cstdint>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

namespace rbg_cli {

struct RecordSpans {              // records of one buffer, in input order
    std::vector<uint64_t> name_begin, seq_begin;   // offsets from the buffer's first byte (seq: modulo 2^64 when in the arena)
    std::vector<uint32_t> name_len, seq_len;
    std::vector<std::string> arena;                // rebuilt sequences (kept alive with the spans)
    size_t size() const { return seq_begin.size(); }
    void clear() { name_begin.clear(); seq_begin.clear(); name_len.clear(); seq_len.clear(); arena.clear(); }
    void append(RecordSpans &&o) {
        name_begin.insert(name_begin.end(), o.name_begin.begin(), o.name_begin.end());
        seq_begin.insert(seq_begin.end(), o.seq_begin.begin(), o.seq_begin.end());
        name_len.insert(name_len.end(), o.name_len.begin(), o.name_len.end());
        seq_len.insert(seq_len.end(), o.seq_len.begin(), o.seq_len.end());
        for (auto &s : o.arena) arena.push_back(std::move(s));   // (std::string moves keep heap buffers in place; short ones are re-pointed below)
    }
};

// kseq_read's state between calls (kseq.h: last_char)
struct ScanState {
    int last_char = 0;   // '>' or '@' already consumed as the next record's header character, else 0
};

enum ScanStop { kScanEnd = 0, kScanNeedMore = 1, kScanTruncQual = -2 };

// first '>' or '@' in [p, e), looked for a stretch at a time (a memchr for a character the file does not contain
// would otherwise run to the end of the buffer for every record)
inline const char *find_header_char(const char *p, const char *e) {
    while (p < e) {
        const size_t n = static_cast<size_t>(e - p) < 256 ? static_cast<size_t>(e - p) : 256;
        const char *a = static_cast<const char *>(memchr(p, '@', n));
        const char *b = static_cast<const char *>(memchr(p, '>', a ? static_cast<size_t>(a - p) : n));
        if (b) return b;
        if (a) return a;
        p += n;
    }
    return nullptr;
}

// Scans buf[pos, end) as kseq_read would, appending complete records to `out`.  `final` = the buffer ends the
// stream (otherwise a record that touches `end` is left for the next buffer: *resume = where it starts, with the
// state to resume in).  `limit`: stop before starting a record at or beyond this offset (segment scans).
// Returns kScanEnd (stream or limit reached; *resume = position reached), kScanNeedMore, or kScanTruncQual (-2, like
// kseq: records before the bad one are in `out`).
inline int scan_records(const char *buf, uint64_t pos, uint64_t end, bool final, uint64_t limit, ScanState &st, RecordSpans &out,
                        uint64_t *resume, ScanState *resume_state) {
    auto need_more = [&](uint64_t at, ScanState s) {
        *resume = at;
        *resume_state = s;
        return static_cast<int>(kScanNeedMore);
    };
    while (true) {
        const uint64_t rec_at = pos;
        const ScanState rec_state = st;
        // -- jump to the next header character, anywhere (kseq.h:183-187)
        if (st.last_char == 0) {
            const char *h = find_header_char(buf + pos, buf + end);
            if (!h) {
                if (!final) return need_more(end, st);   // nothing but skippable bytes so far: they can go
                *resume = end;
                *resume_state = st;
                return kScanEnd;
            }
            pos = static_cast<uint64_t>(h - buf);
            if (pos >= limit) {   // the next record belongs to the next segment
                *resume = pos;
                *resume_state = st;
                return kScanEnd;
            }
            ++pos;
        } else if (rec_at > limit) {   // the header character (already consumed, at rec_at - 1) opens the next segment;
                                       // rec_at == 0: a carried-over buffer that begins after it (cli_input.hpp, zlib path)
            *resume = rec_at;
            *resume_state = st;
            return kScanEnd;
        }
        // -- header line: name up to the first whitespace, the rest is the comment (kseq.h:189-190)
        const char *nl = pos < end ? static_cast<const char *>(memchr(buf + pos, '\n', end - pos)) : nullptr;
        if (!nl && !final) return need_more(rec_at, rec_state);
        const uint64_t line_end = nl ? static_cast<uint64_t>(nl - buf) : end;
        if (!nl && line_end == pos) {   // end of stream right after the header character: kseq returns -1, no record
            *resume = end;
            *resume_state = ScanState();
            return kScanEnd;
        }
        uint64_t ne = pos;
        while (ne < line_end && !isspace(static_cast<unsigned char>(buf[ne]))) ++ne;
        const uint64_t name_b = pos, name_l = ne - pos;
        pos = nl ? line_end + 1 : end;
        // -- sequence lines until a line that starts with '>', '+' or '@' (kseq.h:195-199)
        uint64_t seq_b = pos, seq_l = 0;
        bool contiguous = true;    // the sequence is exactly buf[seq_b, seq_b + seq_l)
        std::string rebuilt;
        int c = -1;
        uint64_t nlines = 0;
        while (true) {
            if (pos >= end) {
                if (!final) return need_more(rec_at, rec_state);
                c = -1;
                break;
            }
            c = static_cast<unsigned char>(buf[pos]);
            if (c == '>' || c == '+' || c == '@') { ++pos; break; }
            if (c == '\n') { ++pos; continue; }   // empty line
            const char *e = static_cast<const char *>(memchr(buf + pos, '\n', end - pos));
            if (!e && !final) return need_more(rec_at, rec_state);
            const uint64_t le = e ? static_cast<uint64_t>(e - buf) : end;
            if (nlines == 0) { seq_b = pos; seq_l = le - pos; }
            else {
                if (contiguous) { rebuilt.assign(buf + seq_b, seq_l); contiguous = false; }
                rebuilt.append(buf + pos, le - pos);
            }
            // kseq.h:141: a trailing '\r' goes when the string is longer than one byte (the accumulated sequence)
            if (contiguous) { if (seq_l > 1 && buf[seq_b + seq_l - 1] == '\r') --seq_l; }
            else if (rebuilt.size() > 1 && rebuilt.back() == '\r') rebuilt.pop_back();
            // (a '\r' dropped from a line that is followed by more lines is gone for good in kseq too)
            ++nlines;
            pos = e ? le + 1 : end;
        }
        const uint64_t final_len = contiguous ? seq_l : rebuilt.size();
        ScanState after;
        after.last_char = (c == '>' || c == '@') ? c : 0;
        if (c == '+') {
            // -- rest of the '+' line, then quality lines until they are as long as the sequence (kseq.h:212-217)
            const char *e = pos < end ? static_cast<const char *>(memchr(buf + pos, '\n', end - pos)) : nullptr;
            if (!e) {
                if (!final) return need_more(rec_at, rec_state);
                *resume = end;
                return kScanTruncQual;   // no quality string (kseq.h:213)
            }
            pos = static_cast<uint64_t>(e - buf) + 1;
            uint64_t qlen = 0;
            bool first = true;
            while (first || qlen < final_len) {   // at least one line (the call sits in kseq's loop condition)
                first = false;
                if (pos >= end) {
                    if (!final) return need_more(rec_at, rec_state);
                    break;
                }
                const char *qe = static_cast<const char *>(memchr(buf + pos, '\n', end - pos));
                if (!qe && !final) return need_more(rec_at, rec_state);
                const uint64_t le = qe ? static_cast<uint64_t>(qe - buf) : end;
                qlen += le - pos;
                if (qlen > 1 && le > pos && buf[le - 1] == '\r') --qlen;   // the '\r' rule, on the accumulated quality string
                pos = qe ? le + 1 : end;
            }
            if (qlen != final_len) {
                *resume = pos;
                return kScanTruncQual;
            }
            after.last_char = 0;
        }
        // -- the record
        out.name_begin.push_back(name_b);
        out.name_len.push_back(static_cast<uint32_t>(name_l));
        if (contiguous) {
            out.seq_begin.push_back(seq_b);
            out.seq_len.push_back(static_cast<uint32_t>(seq_l));
        } else {
            out.arena.push_back(std::move(rebuilt));
            out.seq_begin.push_back(~uint64_t(0));                 // fixed up by finish_arena()
            out.seq_len.push_back(static_cast<uint32_t>(out.arena.back().size()));
        }
        st = after;
        if (c < 0) {   // the stream ended inside / right after this record
            *resume = end;
            *resume_state = ScanState();
            return kScanEnd;
        }
    }
}

// arena records get their offset relative to `buf` (modulo 2^64: rbg_find_range_spans adds it back to the base)
inline void finish_arena(const char *buf, RecordSpans &r) {
    size_t a = 0;
    for (size_t i = 0; i < r.size(); ++i)
        if (r.seq_begin[i] == ~uint64_t(0))
            r.seq_begin[i] = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(r.arena[a++].data()) - reinterpret_cast<uintptr_t>(buf));
}

// a guessed record boundary at or after `from`: a line start that looks like a FASTQ or FASTA header
inline uint64_t guess_boundary(const char *buf, uint64_t from, uint64_t end) {
    uint64_t p = from;
    for (int tries = 0; tries < 64 && p < end; ++tries) {
        const char *nl = static_cast<const char *>(memchr(buf + p, '\n', end - p));
        if (!nl) return end;
        p = static_cast<uint64_t>(nl - buf) + 1;
        if (p >= end) return end;
        if (buf[p] == '>') return p;
        if (buf[p] == '@') {   // '@' also starts quality lines: ask for a '+' line two lines further down
            const char *l1 = static_cast<const char *>(memchr(buf + p, '\n', end - p));
            if (!l1) return end;
            const char *l2 = static_cast<const char *>(memchr(l1 + 1, '\n', end - (static_cast<uint64_t>(l1 - buf) + 1)));
            if (!l2) return end;
            if (static_cast<uint64_t>(l2 - buf) + 1 < end && l2[1] == '+') return p;
        }
    }
    return end;
}

// how many buffers went through the multi-threaded path (tests assert that it is taken)
inline std::atomic<uint64_t> &parallel_scans() {
    static std::atomic<uint64_t> n{0};
    return n;
}

// The whole buffer [pos, end) with `threads` workers; same contract as scan_records (limit = end).
inline int scan_records_parallel(const char *buf, uint64_t pos, uint64_t end, bool final, ScanState &st, RecordSpans &out, uint64_t *resume,
                                 ScanState *resume_state, unsigned threads, uint64_t min_segment = uint64_t(4) << 20) {
    const uint64_t span = end - pos;
    unsigned K = threads ? threads : 1;
    if (span / K < min_segment) K = static_cast<unsigned>(span / min_segment);
    // (a window that begins right after an already consumed '>' / '@' -- every FASTA window but the first -- is no
    // reason to go serial: segment 0 starts from `st`, and the stitch places the next header at `at - 1` then)
    if (K <= 1) {
        const int rc = scan_records(buf, pos, end, final, end, st, out, resume, resume_state);
        finish_arena(buf, out);
        return rc;
    }
    parallel_scans().fetch_add(1, std::memory_order_relaxed);
    std::vector<uint64_t> cut(K + 1);
    cut[0] = pos;
    cut[K] = end;
    for (unsigned k = 1; k < K; ++k) cut[k] = guess_boundary(buf, pos + span * k / K, end);
    struct Seg { RecordSpans recs; int rc = 0; uint64_t resume = 0; ScanState rstate, st; };
    std::vector<Seg> seg(K);
    std::vector<std::thread> th;
    auto work = [&](unsigned k) {
        Seg &s = seg[k];
        if (k > 0 && cut[k] >= cut[k + 1]) return;   // empty segment
        s.st = k == 0 ? st : ScanState();
        s.rc = scan_records(buf, cut[k], end, final, cut[k + 1], s.st, s.recs, &s.resume, &s.rstate);
    };
    for (unsigned k = 1; k < K; ++k) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    // stitch: a guessed segment counts only if the exact scan before it stopped precisely at its start
    uint64_t at = pos;
    ScanState cur = st;
    int rc = kScanEnd;
    for (unsigned k = 0; k < K; ++k) {
        Seg &s = seg[k];
        if (k > 0) {
            if (cut[k] >= cut[k + 1]) continue;                                 // empty segment
            const uint64_t eff = cur.last_char ? at - 1 : at;                   // where the next record's header character is
            const bool agrees = eff == cut[k] && (cur.last_char == 0 || buf[cut[k]] == cur.last_char);
            if (!agrees) {
                if (eff >= cut[k + 1]) continue;                                // the exact scan is already beyond this segment
                s.recs.clear();                                                 // wrong guess: scan the stretch again, exactly
                s.st = cur;
                s.rc = scan_records(buf, at, end, final, cut[k + 1], s.st, s.recs, &s.resume, &s.rstate);
            }
        }
        out.append(std::move(s.recs));
        rc = s.rc;
        at = s.resume;
        cur = s.rstate;
        if (rc != kScanEnd) break;   // need-more or a truncated quality string ends the buffer here
    }
    // the arena strings were moved: their data pointers are stable for heap strings but not for short ones -> fix up now
    finish_arena(buf, out);
    *resume = at;
    *resume_state = cur;
    st = cur;
    return rc;
}

}  // namespace rbg_cli
