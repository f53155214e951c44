// Host ingest shared by the command-line tools (next-row f2): a block-buffered FASTA/FASTQ reader
// with kseq_read's observable behaviour (reference include/kseq.h:178-219), appending straight into
// the C-ABI's packed batch layout.
#pragma once
#include <zlib.h>

#include <cctype>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

namespace rbg_cli {

// One batch of reads in the C-ABI's layout (names kept the same way)
struct PackedBatch {
    std::string names, seqs;
    std::vector<uint64_t> name_off{0}, off{0};
    size_t size() const { return off.size() - 1; }
    void clear() { names.clear(); seqs.clear(); name_off.assign(1, 0); off.assign(1, 0); }
};

// FASTA/FASTQ reader with kseq_read's observable behaviour (kseq.h:178-219), block-buffered over
// gzread and appending straight into a PackedBatch (no per-read strings)
class FastxReader {
   public:
    explicit FastxReader(gzFile fp) : fp_(fp), buf_(1 << 22) {}
    // 0 = record appended, -1 = EOF, -2 = truncated quality string, -3 = stream error.  On -2 / -3 the
    // batch is left exactly as it was at entry: kseq_read's caller never sees that record either
    // (`while ((err = kseq_read(seq)) >= 0)`, rb_align.cpp:176).
    int next(PackedBatch &b) {
        const size_t names0 = b.names.size(), seqs0 = b.seqs.size(), nrec0 = b.off.size();
        auto rollback = [&](int code) {
            b.names.resize(names0);
            b.seqs.resize(seqs0);
            b.name_off.resize(nrec0);
            b.off.resize(nrec0);
            return code;
        };
        int c;
        if (last_char_ == 0) {  // jump to the next header character (kseq.h:183-187)
            while ((c = getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) return c;
            last_char_ = c;
        }
        // name = up to the first whitespace (kseq.h:189: an error while reading it ends the call with -3,
        // end of file before any byte of it with -1); the rest of the line is the comment (kseq.h:190:
        // its return value is ignored, an error there surfaces in the sequence loop below)
        bool in_name = true, got_any = false;
        while ((c = getc()) >= 0 && c != '\n') {
            got_any = true;
            if (in_name && !isspace(c)) b.names.push_back(static_cast<char>(c));
            else in_name = false;
        }
        if (c == -3 && in_name) return rollback(-3);
        if (c == -1 && !got_any) return rollback(-1);
        b.name_off.push_back(b.names.size());
        // sequence lines until a line starting with '>', '+' or '@' (kseq.h:195-199)
        const size_t seq_begin = b.seqs.size();
        while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            b.seqs.push_back(static_cast<char>(c));
            read_line_into(b.seqs, seq_begin);
        }
        last_char_ = (c == '>' || c == '@') ? c : 0;
        const size_t seq_len = b.seqs.size() - seq_begin;
        b.off.push_back(b.seqs.size());
        // FASTA (kseq.h:207): the record read so far is returned even when the stream failed under it;
        // the error is reported by the NEXT call (ks_getc's sticky ks_err, kseq.h:70)
        if (c != '+') return 0;
        while ((c = getc()) >= 0 && c != '\n') {}  // rest of the '+' line
        if (c == -1) return rollback(-2);          // no quality string (kseq.h:213)
        size_t qlen = 0;
        qual_.clear();
        do {  // kseq.h:214: at least one line, then until the quality is as long as the sequence
            const size_t before = qual_.size();
            if (!read_line_into(qual_, 0)) break;
            qlen += qual_.size() - before;
        } while (qlen < seq_len);
        // kseq.h:215 tests `c == -3`, but c holds the loop's boolean there (operator precedence): a stream
        // error inside the quality string comes out as a length mismatch (-2) like a truncated one
        last_char_ = 0;
        return qlen == seq_len ? 0 : rollback(-2);
    }

   private:
    int getc() {
        if (err_) return -3;  // ks_err is sticky (kseq.h:70)
        if (begin_ >= end_) {
            if (eof_) return -1;
            const int got = gzread(fp_, buf_.data(), static_cast<unsigned>(buf_.size()));
            if (got < 0) { eof_ = true; err_ = true; return -3; }
            if (got == 0) { eof_ = true; return -1; }
            begin_ = 0;
            end_ = static_cast<size_t>(got);
        }
        return static_cast<unsigned char>(buf_[begin_++]);
    }
    // ks_getuntil2(KS_SEP_LINE, append): rest of the current line without '\n'; a trailing '\r' is
    // dropped when the string is longer than one byte (kseq.h:141).  false = nothing read at EOF.
    bool read_line_into(std::string &str, size_t str_begin) {
        bool got_any = false;
        while (true) {
            if (begin_ >= end_) {
                const int c = getc();
                if (c < 0) break;
                --begin_;
            }
            const char *p = buf_.data() + begin_;
            const char *nl = static_cast<const char *>(memchr(p, '\n', end_ - begin_));
            const size_t take = nl ? static_cast<size_t>(nl - p) : end_ - begin_;
            str.append(p, take);
            got_any = true;
            begin_ += take + (nl ? 1 : 0);
            if (nl) break;
        }
        if (str.size() - str_begin > 1 && str.back() == '\r') str.pop_back();
        return got_any;
    }
    gzFile fp_;
    std::vector<char> buf_;
    size_t begin_ = 0, end_ = 0;
    bool eof_ = false, err_ = false;
    int last_char_ = 0;
    std::string qual_;
};

inline void put_u64(std::string &out, uint64_t v) {   // two digits per division, one append (the tools print a number or three per read)
    static const char lut[] =
        "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
        "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    char tmp[20];
    char *const e = tmp + 20;
    char *q = e;
    while (v >= 100) {
        const uint64_t d = v / 100;
        const unsigned r = static_cast<unsigned>(v - d * 100);
        v = d;
        q -= 2;
        q[0] = lut[2 * r];
        q[1] = lut[2 * r + 1];
    }
    if (v >= 10) { q -= 2; q[0] = lut[2 * v]; q[1] = lut[2 * v + 1]; }
    else *--q = static_cast<char>('0' + v);
    out.append(q, static_cast<size_t>(e - q));
}

// Output text is produced through a raw pointer into the piece's string (grown in bulk, trimmed at the end) with a
// two-digits-at-a-time number writer: the formatter, not the GPU, sets the pace of this tool (150 ns per read with
// std::string::push_back per character; rb_align -s prints a number per location).
// A piece of output text: raw bytes with a capacity that survives clear(), grown WITHOUT initialising what is about to be
// overwritten (std::string::resize zero-fills: for a formatter that makes 10 GB of text per 10 M reads -- rb_align -s on
// the bench index -- that doubled the memory traffic).  Pieces are recycled from batch to batch by the tools.
struct TextBuf {
    std::unique_ptr<char[]> p;
    size_t cap = 0, len = 0;
    void clear() { len = 0; }
    const char *data() const { return p.get(); }
    size_t size() const { return len; }
    void reserve(size_t n) {
        if (n <= cap) return;
        std::unique_ptr<char[]> q(new char[n]);   // (default-initialised: no fill)
        if (len) std::memcpy(q.get(), p.get(), len);
        p.swap(q);
        cap = n;
    }
};
// the formatters' view of a piece: room(n) guarantees n writable bytes at the current end
struct FastOut {
    TextBuf &s;
    size_t len;
    explicit FastOut(TextBuf &b) : s(b), len(b.len) {}
    char *room(size_t n) {
        if (len + n > s.cap) { s.len = len; s.reserve(std::max(s.cap * 2, len + n + 65536)); }
        return s.p.get() + len;
    }
    void finish() { s.len = len; }
};

// Decimal text of v at p, no terminator; returns the end.  Four digits per table lookup: a 10 000-entry table of
// zero-padded quadruples, one 64-bit division by 10^8 and 32-bit arithmetic below it (positions and offsets are 8 to 11
// digits: two or three lookups instead of five divisions by 100 and a copy through a temporary).
struct Quads {
    char d[10000][4];
    Quads() {
        for (unsigned v = 0; v < 10000; ++v) {
            d[v][0] = static_cast<char>('0' + v / 1000);
            d[v][1] = static_cast<char>('0' + v / 100 % 10);
            d[v][2] = static_cast<char>('0' + v / 10 % 10);
            d[v][3] = static_cast<char>('0' + v % 10);
        }
    }
};
inline const Quads &quads() {
    static const Quads q;
    return q;
}
inline char *fmt_lead(char *p, unsigned v, const Quads &Q) {   // 0 <= v < 10000, no leading zeros (v == 0: "0")
    const unsigned skip = v >= 1000 ? 0u : v >= 100 ? 1u : v >= 10 ? 2u : 3u;
    std::memcpy(p, Q.d[v] + skip, 4);   // (copies 4 bytes, keeps 4 - skip: the caller's buffer has the slack)
    return p + (4 - skip);
}
inline char *fmt_u64(char *p, uint64_t v) {
    const Quads &Q = quads();
    if (v < 10000) return fmt_lead(p, static_cast<unsigned>(v), Q);
    if (v < 100000000ull) {
        const unsigned x = static_cast<unsigned>(v), hi = x / 10000, lo = x - hi * 10000;
        p = fmt_lead(p, hi, Q);
        std::memcpy(p, Q.d[lo], 4);
        return p + 4;
    }
    if (v < 1000000000000ull) {   // up to 12 digits
        const uint64_t top = v / 100000000ull;
        const unsigned rest = static_cast<unsigned>(v - top * 100000000ull), hi = rest / 10000, lo = rest - hi * 10000;
        p = fmt_lead(p, static_cast<unsigned>(top), Q);
        std::memcpy(p, Q.d[hi], 4);
        std::memcpy(p + 4, Q.d[lo], 4);
        return p + 8;
    }
    // 13 .. 20 digits: the leading part recursively (at most twice), then eight padded digits
    const uint64_t top = v / 100000000ull;
    const unsigned rest = static_cast<unsigned>(v - top * 100000000ull), hi = rest / 10000, lo = rest - hi * 10000;
    p = fmt_u64(p, top);
    std::memcpy(p, Q.d[hi], 4);
    std::memcpy(p + 4, Q.d[lo], 4);
    return p + 8;
}
inline char *fmt_lit(char *p, const char *lit, size_t n) { std::memcpy(p, lit, n); return p + n; }

}  // namespace rbg_cli
