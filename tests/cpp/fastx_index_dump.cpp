// Test harness (CPU only): runs rb_align's record scanner (rowbowt_amd/csrc/fastx_index.hpp) over a file the way
// rb_align does -- block by block with carry-over, every block cut into segments scanned by several threads -- and
// prints "name<TAB>seq" per record, then "rc=<code>" (-1 end of input, -2 truncated quality string).
// usage: fastx_index_dump <file> <block bytes> <threads> <min segment bytes>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../rowbowt_amd/csrc/fastx_index.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint64_t block = std::strtoull(argv[2], nullptr, 10), minseg = std::strtoull(argv[4], nullptr, 10);
    const unsigned threads = static_cast<unsigned>(std::atoi(argv[3]));
    uint64_t pos = 0;   // start of the unconsumed input
    uint64_t win = block;
    rbg_cli::ScanState st;
    int rc = -1;
    while (true) {
        const uint64_t end = std::min<uint64_t>(data.size(), pos + win);
        const bool final = end == data.size();
        rbg_cli::RecordSpans recs;
        uint64_t resume = pos;
        rbg_cli::ScanState rstate;
        // (the window [pos, end) is scanned in place: offsets are from data.data())
        const int r = rbg_cli::scan_records_parallel(data.data(), pos, end, final, st, recs, &resume, &rstate, threads, minseg);
        for (size_t i = 0; i < recs.size(); ++i) {
            std::fwrite(data.data() + recs.name_begin[i], 1, recs.name_len[i], stdout);
            std::fputc('\t', stdout);
            const char *sq = reinterpret_cast<const char *>(reinterpret_cast<uintptr_t>(data.data()) + recs.seq_begin[i]);
            std::fwrite(sq, 1, recs.seq_len[i], stdout);
            std::fputc('\n', stdout);
        }
        if (r == rbg_cli::kScanTruncQual) { rc = -2; break; }
        if (r == rbg_cli::kScanEnd && final) { rc = -1; break; }
        if (r == rbg_cli::kScanNeedMore || r == rbg_cli::kScanEnd) {
            if (resume == pos && recs.size() == 0) win *= 2;   // a record longer than the window
            else win = block;
            pos = resume;
            st = rstate;
            if (final && r == rbg_cli::kScanNeedMore) { std::printf("NEED MORE AT END\n"); return 3; }
        }
    }
    std::printf("rc=%d\n", rc);
    return 0;
}
