// Test harness (CPU only): runs the command-line tools' FASTA/FASTQ reader (rowbowt_amd/csrc/fastx.hpp) over
// a file the way rb_align's loop does and prints what that loop would see: "name<TAB>seq" per record,
// then "rc=<final return code>".  Compared with tests/kseq_model.py (kseq.h:178-219 restated).
#include <cstdio>
#include <cstdlib>

#include "../../rowbowt_amd/csrc/fastx.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const size_t batch = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 3;  // small batches: rollback must not disturb earlier records
    gzFile fp = gzopen(argv[1], "r");
    if (!fp) return 2;
    rbg_cli::FastxReader reader(fp);
    rbg_cli::PackedBatch b;
    int e = 0;
    while (e == 0) {
        b.clear();
        while (b.size() < batch && (e = reader.next(b)) == 0) {}
        if (b.name_off.size() != b.off.size()) { std::printf("INCONSISTENT BATCH\n"); return 3; }
        for (size_t i = 0; i < b.size(); ++i) {
            std::fwrite(b.names.data() + b.name_off[i], 1, b.name_off[i + 1] - b.name_off[i], stdout);
            std::fputc('\t', stdout);
            std::fwrite(b.seqs.data() + b.off[i], 1, b.off[i + 1] - b.off[i], stdout);
            std::fputc('\n', stdout);
        }
    }
    std::printf("rc=%d\n", e);
    gzclose(fp);
    return 0;
}
