// Test harness (CPU only): the tools' input side as they use it (rowbowt_amd/csrc/cli_input.hpp: InputSource over a
// memory-mapped plain file, or through zlib for gzip and pipes) -- window after window, printing "name<TAB>seq" per
// record, then "rc=<code>" (-1 end of input, -2 truncated quality string, -3 stream error).
// usage: cli_input_dump <file> <window bytes> <threads>
#include <cstdio>
#include <cstdlib>

#include "../../rowbowt_amd/csrc/cli_input.hpp"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    rbg_cli::InputSource in;
    if (!in.open(argv[1], static_cast<unsigned>(std::atoi(argv[3])), std::strtoull(argv[2], nullptr, 10))) {
        std::printf("invalid file\n");
        return 3;
    }
    rbg_cli::Window w;
    int rc;
    do {
        rc = in.next(w);
        for (size_t i = 0; i < w.size(); ++i) {
            std::fwrite(w.base + w.recs.name_begin[i], 1, w.recs.name_len[i], stdout);
            std::fputc('\t', stdout);
            std::fwrite(w.base + w.recs.seq_begin[i], 1, w.recs.seq_len[i], stdout);
            std::fputc('\n', stdout);
        }
    } while (rc == 0);
    std::printf("rc=%d\n", rc);
    return 0;
}
