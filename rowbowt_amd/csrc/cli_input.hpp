// cli_input.hpp -- the input side shared by the command-line tools (rb_align, rb_markers): the FASTA/FASTQ file as a
// sequence of windows that each begin at a record boundary, scanned in place by fastx_index.hpp (memory-mapped plain
// files; gzip and pipes through zlib into a window-sized buffer).  kseq_read's observable behaviour (kseq.h:178-219)
// is the scanner's; see tests/test_fastx_host.py.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
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
            // any other gzip: a file of SEVERAL members (concatenated files: `cat a.gz b.gz`, lanes of a sequencer written one after the
            // other) is mapped and its members are inflated by the worker threads, speculatively from every place that looks like a
            // member's start and checked as a chain (next_members below).  That mode is entered only on evidence: member 0 must END
            // exactly where a second member's header begins (checked here by inflating it, output discarded).  The bytes of a header
            // also turn up by chance inside deflate data, so a candidate alone proves nothing: an ordinary single-member file -- or
            // one whose first member is too long to be worth checking -- stays zlib's single stream, as before.
            else {
                const size_t sz = static_cast<size_t>(sb.st_size);
                void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    const unsigned char *p = static_cast<const unsigned char *>(m);
                    (void)madvise(m, sz, MADV_SEQUENTIAL);
                    std::vector<uint64_t> cand;
                    // speculative starts in [from, to): every place whose ten bytes pass the strict test
                    auto scan = [&](size_t from, size_t to, size_t at_most) {
                        for (size_t i = from; i < to && i + 18 <= sz && at_most; --at_most) {
                            const void *hit = std::memchr(p + i, 0x1f, std::min(to, sz - 17) - i);
                            if (!hit) break;
                            i = static_cast<size_t>(static_cast<const unsigned char *>(hit) - p);
                            if (member_header_strict(p + i)) cand.push_back(i);
                            else ++at_most;
                            ++i;
                        }
                    };
                    // member 0 must end where a second member's header begins, within the first kVerifyMax bytes: only then is the rest scanned
                    bool chain = false;
                    if (member_header_loose(p)) {
                        cand.push_back(0);
                        scan(1, static_cast<size_t>(std::min<uint64_t>(sz, kVerifyMax)), 1);
                        if (cand.size() >= 2) {
                            map_ = static_cast<const char *>(m);   // (inflate_member reads through map_)
                            map_size_ = sz;
                            MemberJob job;
                            inflate_member(0, cand[1], job, /*keep=*/false);
                            chain = job.res == 0 && job.at + 18 <= sz && member_header_loose(p + job.at);
                            if (chain && job.at != cand[1]) cand.insert(cand.begin() + 1, job.at);   // (a header only the loose test accepts)
                            map_ = nullptr;
                            map_size_ = 0;
                        }
                        if (chain) scan(static_cast<size_t>(cand.back()) + 1, sz, SIZE_MAX);
                    }
                    if (chain) {
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

    // how the input is read: "mapped", "bgzf", "members" (gzip members inflated in parallel) or "stream" (zlib's gzread)
    const char *mode() const { return bgzf_ ? "bgzf" : mgz_ ? "members" : mapped_ ? "mapped" : "stream"; }

   private:
    // ---- multi-member gzip ------------------------------------------------------------------------------------------------
    // What a gzip member's first ten bytes look like.  LOOSE is what zlib's gzread accepts (magic, deflate, no reserved flag): the test
    // for the place where a verified member ENDED.  STRICT adds what every writer in use produces -- XFL 0, 2 or 4 and a defined OS
    // byte -- and is the test for SPECULATIVE starts: the loose pattern alone occurs by chance once per 134 MB of deflate data.
    static bool member_header_loose(const unsigned char *h) { return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 0xE0) == 0; }
    static bool member_header_strict(const unsigned char *h) {
        return member_header_loose(h) && (h[8] == 0 || h[8] == 2 || h[8] == 4) && (h[9] <= 13 || h[9] == 255);
    }
    static constexpr uint64_t kVerifyMax = uint64_t(256) << 20;   // compressed bytes of member 0 the check at open() may inflate
    static constexpr size_t kMemberOutStart = size_t(1) << 20;    // first size of a member's text buffer (doubling from there)
    // One member being inflated: `zs` stays open while the input ran out before the member ended (res 1) so that the chain can go
    // on from `at` instead of starting over.  (A z_stream must not move once initialised: zlib keeps a pointer back to it.)
    struct MemberJob {
        std::unique_ptr<z_stream> zs;
        std::vector<char> out;   // the text so far
        uint64_t from = 0;       // file offset of the member's first byte
        uint64_t at = 0;         // file offset of the next compressed byte
        int res = -1;            // 0 = the member ended at `at`; 1 = the input ran out at the limit; -1 = not a valid member / corrupt
        ~MemberJob() { close(); }
        MemberJob() = default;
        MemberJob(MemberJob &&) = default;
        MemberJob &operator=(MemberJob &&o) {
            if (this != &o) { close(); zs = std::move(o.zs); out = std::move(o.out); from = o.from; at = o.at; res = o.res; }
            return *this;
        }
        void close() {
            if (zs) { (void)inflateEnd(zs.get()); zs.reset(); }
        }
    };
    // inflate ONE gzip member that starts at file offset `from`, reading no further than `limit`.  The text goes to job.out, which
    // starts small and doubles (never sized from the input's length); with keep == false it is discarded as it comes (the check of
    // open()).  A corrupt member leaves the text inflated before the damage in job.out.
    void inflate_member(uint64_t from, uint64_t limit, MemberJob &job, bool keep = true) const {
        job.close();
        job.out.clear();
        job.from = job.at = from;
        job.res = -1;
        job.zs.reset(new z_stream);
        std::memset(job.zs.get(), 0, sizeof(z_stream));
        if (inflateInit2(job.zs.get(), 15 + 16) != Z_OK) { job.zs.reset(); return; }   // gzip wrapper: header, CRC-32 and ISIZE are checked by zlib
        z_stream &zs = *job.zs;
        job.out.resize(std::min<uint64_t>(kMemberOutStart, std::max<uint64_t>(uint64_t(64) << 10, (limit - from) * 4)));
        size_t have = 0;
        while (true) {
            const uint64_t in_now = std::min<uint64_t>(limit - job.at, uint64_t(1) << 30);
            zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(map_ + job.at));
            zs.avail_in = static_cast<uInt>(in_now);
            if (!keep) have = 0;
            if (have == job.out.size()) job.out.resize(job.out.size() * 2);
            const size_t room = std::min<size_t>(job.out.size() - have, size_t(1) << 30);
            zs.next_out = reinterpret_cast<Bytef *>(job.out.data() + have);
            zs.avail_out = static_cast<uInt>(room);
            const int rc = inflate(&zs, Z_NO_FLUSH);
            have += room - zs.avail_out;
            job.at += in_now - zs.avail_in;
            if (rc == Z_STREAM_END) { job.res = 0; break; }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { job.res = -1; break; }
            if (job.at == limit && zs.avail_out != 0) { job.res = 1; break; }   // all input consumed, no end of stream
        }
        job.out.resize(keep ? have : 0);
        if (job.res != 1) job.close();
    }
    // the member in `big_` (cut short at a speculative start inside its data, or simply long) goes on into `dst`, at most `win` bytes of
    // text per call: 0 = it ended (at big_.at), 1 = the window is full, -1 = corrupt (the text before the damage is in dst), 2 = the file ends
    // inside the member (gzread hands out what there is and then reports the end of the file: so does this)
    int continue_big(std::vector<char> &dst, uint64_t win) {
        z_stream &zs = *big_.zs;
        const size_t have0 = dst.size();
        dst.resize(have0 + win);
        size_t have = have0;
        int res = 1;
        while (have < dst.size()) {
            const uint64_t in_now = std::min<uint64_t>(map_size_ - big_.at, uint64_t(1) << 30);
            zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(map_ + big_.at));
            zs.avail_in = static_cast<uInt>(in_now);
            const size_t room = std::min<size_t>(dst.size() - have, size_t(1) << 30);
            zs.next_out = reinterpret_cast<Bytef *>(dst.data() + have);
            zs.avail_out = static_cast<uInt>(room);
            const int rc = inflate(&zs, Z_NO_FLUSH);
            have += room - zs.avail_out;
            big_.at += in_now - zs.avail_in;
            if (rc == Z_STREAM_END) { res = 0; break; }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { res = -1; break; }
            if (big_.at == map_size_ && zs.avail_out != 0) { res = 2; break; }   // the file ends inside the member
        }
        dst.resize(have);
        if (res != 1) big_.close();
        return res;
    }
    // a verified member ended at file offset `e`: the candidate the chain goes on from; false = the end of the file, or bytes that start
    // no gzip member (zero padding included) -- zlib's gzread, what the reference reads through (kseq.h over gzFile), takes those for
    // trailing garbage and stops there as at the end of the file
    bool chain_from(uint64_t e, size_t from_cand) {
        size_t nxt = from_cand;
        while (nxt < cand_.size() && cand_[nxt] < e) ++nxt;
        if (e + 18 > map_size_ || !member_header_loose(reinterpret_cast<const unsigned char *>(map_) + e)) { ci_ = cand_.size(); return false; }
        if (nxt >= cand_.size() || cand_[nxt] != e) cand_.insert(cand_.begin() + static_cast<std::ptrdiff_t>(nxt), e);   // (only the loose test accepts it)
        ci_ = nxt;
        return ci_ + 1 < cand_.size();
    }
    // appends the next members (about `win` bytes of text) to `dst`; false at the end of the input
    bool next_members(std::vector<char> &dst, uint64_t win) {
        if (big_.zs) {
            const int r = continue_big(dst, win);
            if (r == 1) return true;
            if (r < 0) { stream_error_ = true; ci_ = cand_.size(); return false; }   // like gzread failing: what came before counts
            if (r == 2) { ci_ = cand_.size(); return false; }
            return chain_from(big_.at, big_cand_);
        }
        if (ci_ + 1 >= cand_.size()) return false;
        // the candidates of this batch: compressed bytes of about a third of the window (text compresses three- to fourfold).  A stretch
        // longer than two thirds of a window between two candidates is one long member: it is streamed window by window, never held whole.
        const uint64_t long_member = std::max<uint64_t>(win, kLongFloor) * 2 / 3;
        auto is_long = [&](size_t c) { return cand_[c + 1] - cand_[c] >= long_member; };
        if (is_long(ci_)) {
            inflate_member(cand_[ci_], cand_[ci_], big_);   // (opens the stream; no input yet: res 1)
            if (!big_.zs) { stream_error_ = true; ci_ = cand_.size(); return false; }
            big_cand_ = ci_ + 1;
            return next_members(dst, win);
        }
        size_t cj = ci_ + 1;   // members ci_ .. cj - 1
        while (cj + 1 < cand_.size() && (cand_[cj] - cand_[ci_]) * 3 < win && cj - ci_ < 4096 && !is_long(cj)) ++cj;
        const size_t nseg = cj - ci_;
        std::vector<MemberJob> job(nseg);
        const unsigned T = std::max(1u, std::min<unsigned>(threads_, static_cast<unsigned>(nseg)));
        auto work = [&](unsigned t) {
            for (size_t k = nseg * t / T; k < nseg * (t + 1) / T; ++k) inflate_member(cand_[ci_ + k], cand_[ci_ + k + 1], job[k]);
        };
        {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        // the chain: a member must end where the next one starts.  A candidate inside a member's data cut that member short (res 1): its
        // stream goes on from there, window by window (continue_big), and the candidates it contains are skipped.
        size_t k = 0;
        while (true) {
            MemberJob &j = job[k];
            dst.insert(dst.end(), j.out.begin(), j.out.end());   // (a corrupt member: the text before the damage, as gzread delivers it)
            if (j.res < 0) { stream_error_ = true; ci_ = cand_.size(); return false; }   // corrupt: gzread fails there (-3)
            if (j.res == 1 && j.at >= map_size_) { ci_ = cand_.size(); return false; }   // the file ends inside the member: gzread hands out what there is, then reports the end of the file
            if (j.res == 1) {
                j.out.clear();
                big_cand_ = ci_ + k + 1;
                big_ = std::move(j);
                return true;
            }
            size_t k2 = k + 1;
            while (k2 < nseg && job[k2].from < j.at) ++k2;
            if (k2 < nseg && job[k2].from == j.at) { k = k2; continue; }
            return chain_from(j.at, ci_ + k + 1);   // the next batch starts where this member ended (or the input ends there)
        }
    }
    static constexpr uint64_t kLongFloor = uint64_t(64) << 10;
    bool mgz_ = false;
    std::vector<uint64_t> cand_;   // file offsets that look like the start of a gzip member, then the file's size
    size_t ci_ = 0;
    MemberJob big_;          // the member being streamed across windows, if any
    size_t big_cand_ = 0;    // the candidates from here on may lie inside it
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
