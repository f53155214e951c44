// Test harness (CPU only): the tools' input side as they use it (rowbowt_amd/csrc/cli_input.hpp: InputSource over a
// memory-mapped plain file, or through zlib for gzip and pipes) -- window after window, printing "name<TAB>seq" per
// record, then "rc=<code>" (-1 end of input, -2 truncated quality string, -3 stream error).
// usage: cli_input_dump <file> <window bytes> <threads> [<min bytes per scanning thread>]
// stderr: "windows=<n> parallel=<n> mode=<m>" -- how many windows there were, how many were scanned by more than one thread, and
// how the input was read (InputSource::mode)
#include <cstdio>
#include <cstdlib>

#include "../../rowbowt_amd/csrc/cli_input.hpp"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    rbg_cli::InputSource in;
    const unsigned long long minseg = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : (4ull << 20);
    if (!in.open(argv[1], static_cast<unsigned>(std::atoi(argv[3])), std::strtoull(argv[2], nullptr, 10), minseg)) {
        std::printf("invalid file\n");
        return 3;
    }
    rbg_cli::Window w;
    int rc;
    unsigned long long windows = 0;
    do {
        rc = in.next(w);
        ++windows;
        for (size_t i = 0; i < w.size(); ++i) {
            std::fwrite(w.base + w.recs.name_begin[i], 1, w.recs.name_len[i], stdout);
            std::fputc('\t', stdout);
            std::fwrite(w.base + w.recs.seq_begin[i], 1, w.recs.seq_len[i], stdout);
            std::fputc('\n', stdout);
        }
    } while (rc == 0);
    std::printf("rc=%d\n", rc);
    std::fprintf(stderr, "windows=%llu parallel=%llu mode=%s\n", windows, static_cast<unsigned long long>(rbg_cli::parallel_scans().load()), in.mode());
    return 0;
}
