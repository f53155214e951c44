// cli_input.hpp -- the input side shared by the command-line tools (rb_align, rb_markers): the FASTA/FASTQ file as a
// sequence of windows that each begin at a record boundary, scanned in place by fastx_index.hpp (memory-mapped plain
// files; gzip and pipes through zlib into a window-sized buffer).  kseq_read's observable behaviour (kseq.h:178-219)
// is the scanner's; see tests/test_fastx_host.py.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <string>
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
            if (mapped_) {
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
