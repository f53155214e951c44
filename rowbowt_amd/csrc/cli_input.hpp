// cli_input.hpp -- the input side shared by the command-line tools (rb_align, rb_markers): the FASTA/FASTQ file as a
// sequence of windows that each begin at a record boundary, scanned in place by fastx_index.hpp (memory-mapped plain
// files; gzip and pipes through zlib into a window-sized buffer).  kseq_read's observable behaviour (kseq.h:178-219)
// is the scanner's; see tests/test_fastx_host.py.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "fastx_index.hpp"

namespace rbg_cli {

// One stretch of the input with the records found in it.  The bytes stay where they are: in the mapping of a plain
// file, or in `own` (decompressed / piped input); names and sequences are offsets from `base`.
struct Window {
    const char *base = nullptr;
    std::vector<char> own;
    RecordSpans recs;
    size_t size() const { return recs.size(); }
};

// The input as a sequence of windows that each begin at a record boundary (fastx_index.hpp does the scanning).
class InputSource {
   public:
    ~InputSource() {
        if (map_) munmap(const_cast<char *>(map_), map_size_);
        if (gz_) gzclose(gz_);
    }
    bool open(const std::string &path, unsigned threads, uint64_t window_bytes, uint64_t min_segment = uint64_t(4) << 20) {
        threads_ = threads ? threads : 1;
        window_ = window_bytes;
        min_segment_ = min_segment ? min_segment : 1;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
        if (regular && sb.st_size >= 2 && pread(fd, magic, 2, 0) != 2) magic[0] = magic[1] = 0;
        if (regular && magic[0] == 0x1f && magic[1] == 0x8b && sb.st_size >= 28 && threads_ > 1) {
            // gzip: a BGZF file (bgzip, htslib: independent blocks of at most 64 KB that carry their own compressed size in
            // an extra field) is mapped and its blocks are inflated by the worker threads; any other gzip goes through zlib's
            // single stream below
            unsigned char hdr[18];
            if (pread(fd, hdr, 18, 0) == 18 && bgzf_header(hdr)) {
                map_size_ = static_cast<size_t>(sb.st_size);
                void *m = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, map_size_, MADV_SEQUENTIAL);
                    map_ = static_cast<const char *>(m);
                    bgzf_ = true;
                    ::close(fd);
                    return true;
                }
                map_size_ = 0;
            }
            // any other gzip: when more than one place in the file looks like the start of a gzip member (concatenated files: `cat a.gz b.gz`, lanes
            // of a sequencer written one after the other) the file is mapped and the members are inflated by the worker threads, speculatively
            // from every candidate and checked as a chain (next_members below); a single member stays zlib's single stream
            else {
                const size_t sz = static_cast<size_t>(sb.st_size);
                void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    const unsigned char *p = static_cast<const unsigned char *>(m);
                    std::vector<uint64_t> cand;
                    for (size_t i = 0; i + 10 <= sz;) {
                        const void *hit = std::memchr(p + i, 0x1f, sz - 10 - i + 1);
                        if (!hit) break;
                        i = static_cast<size_t>(static_cast<const unsigned char *>(hit) - p);
                        if (p[i + 1] == 0x8b && p[i + 2] == 8 && (p[i + 3] & 0xE0) == 0) cand.push_back(i);
                        ++i;
                    }
                    if (cand.size() >= 2 && cand[0] == 0) {
                        (void)madvise(m, sz, MADV_SEQUENTIAL);
                        map_ = static_cast<const char *>(m);
                        map_size_ = sz;
                        cand.push_back(sz);
                        cand_.swap(cand);
                        mgz_ = true;
                        ::close(fd);
                        return true;
                    }
                    munmap(m, sz);
                }
            }
        }
        if (regular && !(magic[0] == 0x1f && magic[1] == 0x8b)) {   // a plain file: map it, scan it in place
            map_size_ = static_cast<size_t>(sb.st_size);
            if (map_size_) {
                void *m = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { ::close(fd); return false; }
                (void)madvise(m, map_size_, MADV_SEQUENTIAL);
                map_ = static_cast<const char *>(m);
            }
            mapped_ = true;
            ::close(fd);
            return true;
        }
        ::close(fd);
        gz_ = gzopen(path.c_str(), "r");   // gzip, or anything that is not a regular file (zlib reads plain data through)
        if (!gz_) return false;
        gzbuffer(gz_, 1 << 20);
        return true;
    }
    // the next window; returns 0 while more input follows, -1 at its end, -2 / -3 like kseq_read (the window then
    // holds the records that came before the failure)
    int next(Window &w) {
        w.recs.clear();
        w.own.clear();
        uint64_t win = window_;
        while (true) {
            uint64_t resume = 0;
            rbg_cli::ScanState rstate;
            int rc;
            bool final;
            if (bgzf_) {
                // carry-over of the previous window's unfinished record, then the next blocks worth about `win` bytes
                w.own.assign(carry_.begin(), carry_.end());
                const size_t have = w.own.size();
                struct Blk { uint64_t in, in_len, out, out_len; };
                std::vector<Blk> blks;
                uint64_t total = 0;
                while (pos_ < map_size_ && total < win) {
                    const unsigned char *h = reinterpret_cast<const unsigned char *>(map_) + pos_;
                    if (map_size_ - pos_ < 28 || !bgzf_header(h)) { stream_error_ = true; break; }
                    const uint64_t xlen = h[10] | (static_cast<uint64_t>(h[11]) << 8);
                    const uint64_t bsz = (h[16] | (static_cast<uint64_t>(h[17]) << 8)) + 1;
                    if (bsz < 12 + xlen + 8 || bsz > map_size_ - pos_) { stream_error_ = true; break; }
                    const unsigned char *t = h + bsz - 4;
                    const uint64_t isz = t[0] | (static_cast<uint64_t>(t[1]) << 8) | (static_cast<uint64_t>(t[2]) << 16) | (static_cast<uint64_t>(t[3]) << 24);
                    if (isz > 65536) { stream_error_ = true; break; }   // BGZF blocks hold at most 64 KiB: a crafted ISIZE must not size the buffer
                    blks.push_back(Blk{pos_ + 12 + xlen, bsz - 12 - xlen - 8, total, isz});
                    total += isz;
                    pos_ += bsz;
                }
                w.own.resize(have + total);
                const unsigned T = std::max(1u, std::min<unsigned>(threads_, static_cast<unsigned>(blks.size() / 4 + 1)));
                std::vector<size_t> bad(T, SIZE_MAX);   // per worker: the first block of its share that did not inflate cleanly
                auto work = [&](unsigned t) {
                    z_stream zs;
                    std::memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = blks.size() * t / T; return; }
                    for (size_t i = blks.size() * t / T; i < blks.size() * (t + 1) / T; ++i) {
                        const Blk &b = blks[i];
                        (void)inflateReset(&zs);
                        zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(map_ + b.in));
                        zs.avail_in = static_cast<uInt>(b.in_len);
                        zs.next_out = reinterpret_cast<Bytef *>(w.own.data() + have + b.out);
                        zs.avail_out = static_cast<uInt>(b.out_len);
                        const int rc = inflate(&zs, Z_FINISH);
                        const unsigned char *tr = reinterpret_cast<const unsigned char *>(map_) + b.in + b.in_len;   // CRC32 | ISIZE
                        const uint32_t want = tr[0] | (static_cast<uint32_t>(tr[1]) << 8) | (static_cast<uint32_t>(tr[2]) << 16) | (static_cast<uint32_t>(tr[3]) << 24);
                        if (rc != Z_STREAM_END || zs.avail_out != 0 ||
                            static_cast<uint32_t>(crc32(0L, reinterpret_cast<const Bytef *>(w.own.data() + have + b.out), static_cast<uInt>(b.out_len))) != want) {
                            bad[t] = i;
                            break;
                        }
                    }
                    (void)inflateEnd(&zs);
                };
                {
                    std::vector<std::thread> th;
                    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
                    work(0);
                    for (auto &x : th) x.join();
                }
                {
                    // like gzread failing in the middle: what came before the first bad block still counts, the stream ends there (-3)
                    size_t first_bad = stream_error_ ? blks.size() : SIZE_MAX;   // (a bad header ended the walk: every block listed is good)
                    for (size_t v : bad) first_bad = std::min(first_bad, v);
                    if (first_bad != SIZE_MAX) {
                        stream_error_ = true;
                        w.own.resize(have + (first_bad < blks.size() ? blks[first_bad].out : total));
                    }
                }
                final = stream_error_ || pos_ >= map_size_;
                w.base = w.own.data();
                rc = rbg_cli::scan_records_parallel(w.base, 0, w.own.size(), final, st_, w.recs, &resume, &rstate, threads_, min_segment_);
                if (rc == rbg_cli::kScanTruncQual) return stream_error_ ? -3 : -2;
                carry_.assign(w.own.begin() + static_cast<std::ptrdiff_t>(resume), w.own.end());
                st_ = rstate;
                if (!final && w.recs.size() == 0) { win *= 2; continue; }
            } else if (mgz_) {
                w.own.assign(carry_.begin(), carry_.end());
                const bool more = next_members(w.own, win);
                final = stream_error_ || !more;
                w.base = w.own.data();
                rc = rbg_cli::scan_records_parallel(w.base, 0, w.own.size(), final, st_, w.recs, &resume, &rstate, threads_, min_segment_);
                if (rc == rbg_cli::kScanTruncQual) return stream_error_ ? -3 : -2;
                carry_.assign(w.own.begin() + static_cast<std::ptrdiff_t>(resume), w.own.end());
                st_ = rstate;
                if (!final && w.recs.size() == 0) { win *= 2; continue; }
            } else if (mapped_) {
                const uint64_t end = std::min<uint64_t>(map_size_, pos_ + win);
                final = end == map_size_;
                w.base = map_ ? map_ : "";
                rc = rbg_cli::scan_records_parallel(w.base, pos_, end, final, st_, w.recs, &resume, &rstate, threads_, min_segment_);
                if (rc == rbg_cli::kScanTruncQual) return -2;
                if (!(rc == rbg_cli::kScanEnd && final) && resume == pos_ && w.recs.size() == 0 && !final) { win *= 2; continue; }   // one record longer than the window
                pos_ = resume;
                st_ = rstate;
            } else {
                // carry-over of the previous window's unfinished record, then fresh bytes
                w.own.assign(carry_.begin(), carry_.end());
                const size_t have = w.own.size();
                w.own.resize(have + win);
                size_t got = 0;
                bool eof = false;
                while (got < win) {
                    const int r = gzread(gz_, w.own.data() + have + got, static_cast<unsigned>(std::min<uint64_t>(win - got, 1u << 30)));
                    if (r < 0) { stream_error_ = true; eof = true; break; }
                    if (r == 0) { eof = true; break; }
                    got += static_cast<size_t>(r);
                }
                w.own.resize(have + got);
                final = eof;
                w.base = w.own.data();
                rc = rbg_cli::scan_records_parallel(w.base, 0, w.own.size(), final, st_, w.recs, &resume, &rstate, threads_, min_segment_);
                if (rc == rbg_cli::kScanTruncQual) return stream_error_ ? -3 : -2;
                carry_.assign(w.own.begin() + static_cast<std::ptrdiff_t>(resume), w.own.end());
                st_ = rstate;
                if (!final && w.recs.size() == 0) { win *= 2; continue; }
            }
            if (final) return stream_error_ ? -3 : -1;
            return 0;
        }
    }

   private:
    // ---- multi-member gzip ------------------------------------------------------------------------------------------------
    // inflate ONE gzip member that starts at file offset `from`, reading no further than `limit`: 0 = the member ended (at *end), 1 = the
    // input ran out before it did (a candidate inside this member's data cut it short), -1 = not a valid member
    int inflate_member(uint64_t from, uint64_t limit, std::vector<char> &out, uint64_t *end) const {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return -1;   // gzip wrapper: header, CRC-32 and ISIZE are checked by zlib
        out.clear();
        out.resize(std::max<uint64_t>(uint64_t(64) << 10, (limit - from) * 4));
        size_t have = 0;
        uint64_t at = from;
        int res = 1;
        while (true) {
            const uint64_t in_now = std::min<uint64_t>(limit - at, uint64_t(1) << 30);
            zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(map_ + at));
            zs.avail_in = static_cast<uInt>(in_now);
            if (have == out.size()) out.resize(out.size() * 2);
            const size_t room = std::min<size_t>(out.size() - have, size_t(1) << 30);
            zs.next_out = reinterpret_cast<Bytef *>(out.data() + have);
            zs.avail_out = static_cast<uInt>(room);
            const int rc = inflate(&zs, Z_NO_FLUSH);
            have += room - zs.avail_out;
            at += in_now - zs.avail_in;
            if (rc == Z_STREAM_END) { res = 0; break; }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { res = -1; break; }
            if (at == limit && zs.avail_out != 0) { res = 1; break; }   // all input consumed, no end of stream
        }
        (void)inflateEnd(&zs);
        out.resize(have);
        *end = at;
        return res;
    }
    // appends the next members (about `win` bytes of text) to `dst`; false at the end of the input
    bool next_members(std::vector<char> &dst, uint64_t win) {
        if (ci_ + 1 >= cand_.size()) return false;
        // the candidates of this batch: compressed bytes of about a third of the window (text compresses three- to fourfold)
        size_t cj = ci_;
        while (cj + 1 < cand_.size() && (cand_[cj] - cand_[ci_]) * 3 < win && cj - ci_ < 4096) ++cj;
        if (cj == ci_) cj = ci_ + 1;
        const size_t nseg = cj - ci_;
        std::vector<std::vector<char>> out(nseg);
        std::vector<int> res(nseg, -1);
        std::vector<uint64_t> end(nseg, 0);
        const unsigned T = std::max(1u, std::min<unsigned>(threads_, static_cast<unsigned>(nseg)));
        auto work = [&](unsigned t) {
            for (size_t k = nseg * t / T; k < nseg * (t + 1) / T; ++k) res[k] = inflate_member(cand_[ci_ + k], cand_[ci_ + k + 1], out[k], &end[k]);
        };
        {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        // the chain: member k must end where candidate k + 1 starts.  A candidate inside a member's data cut that member short (res 1):
        // it is inflated again up to the candidate where it really ends, and the candidates it contains are skipped.
        size_t k = 0;
        while (k < nseg) {
            uint64_t e = end[k];
            int r = res[k];
            std::vector<char> again;
            const std::vector<char> *txt = &out[k];
            if (r == 1) {
                r = inflate_member(cand_[ci_ + k], map_size_, again, &e);
                txt = &again;
            }
            if (r != 0) { stream_error_ = true; ci_ = cand_.size(); return false; }   // like gzread failing: what came before counts
            dst.insert(dst.end(), txt->begin(), txt->end());
            // the candidate at which the next member starts
            size_t nxt = ci_ + k + 1;
            while (nxt < cand_.size() && cand_[nxt] < e) ++nxt;
            if (nxt >= cand_.size() || cand_[nxt] != e || cand_[nxt] == map_size_) {
                // the end of the file, or bytes that start no gzip member (zero padding included): zlib's gzread -- what the reference reads
                // through, kseq.h over gzFile -- takes them for trailing garbage and stops there as at the end of the file
                ci_ = cand_.size();
                return false;
            }
            if (nxt >= ci_ + nseg) { ci_ = nxt; return ci_ + 1 < cand_.size(); }
            k = nxt - ci_;
        }
        ci_ += nseg;
        return ci_ + 1 < cand_.size();
    }
    bool mgz_ = false;
    std::vector<uint64_t> cand_;   // file offsets that look like the start of a gzip member, then the file's size
    size_t ci_ = 0;
    // a gzip member header that is a BGZF block's: FEXTRA set, first extra subfield 'B' 'C' of two bytes (the block size - 1)
    static bool bgzf_header(const unsigned char *h) {
        return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && (h[10] | (h[11] << 8)) >= 6 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    bool bgzf_ = false;
    const char *map_ = nullptr;
    size_t map_size_ = 0;
    bool mapped_ = false;
    gzFile gz_ = nullptr;
    uint64_t pos_ = 0, window_ = uint64_t(256) << 20, min_segment_ = uint64_t(4) << 20;
    unsigned threads_ = 1;
    rbg_cli::ScanState st_;
    std::vector<char> carry_;
    bool stream_error_ = false;
};

}  // namespace rbg_cli
