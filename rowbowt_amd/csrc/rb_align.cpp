// rb_align -- drop-in for the reference's `rb_align [-s] [-m] [-o pre] <index_prefix> <fastq>`
// (reference src/rb_align.cpp), with the per-read query loop (rb_align.cpp:176-178) replaced by
// batched calls into the MI355X engine (include/rbg.h via rowbowt_gpu.hpp).
//
// stdout is byte-for-byte what the reference prints (rb_align.cpp:118-145):
//   "<name> (<lo>,<hi>), count=<hi-lo+1>\n"
//   -s: "\tlocs: " { "<l>/<doc>:<off> " } "\n"
//   -m: "\tmarkers: " ( "no markers (consider building the marker array with a larger window size)"
//                      | { "<pos>/<allele> " } ) "\n"
// stderr keeps the reference's shape ("will load SA and DA", "<load_s> <query_s>").
// Input parsing follows kseq.h semantics (name = header up to the first whitespace; FASTA or FASTQ,
// plain or gzip; kseq_read's -2 "truncated quality string" error is reported the same way).
#include <getopt.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rowbowt_gpu.hpp"

namespace {

struct RbAlignArgs {  // rb_align.cpp:17-24
    std::string inpre, fastq_fname, outpre;
    int sam = 0, markers = 0;
    int device = 0;
    uint64_t batch = 1u << 20;
};

void print_help() {  // rb_align.cpp:26-35
    fprintf(stderr, "rb_align");
    fprintf(stderr, "Usage: rb_align [options] <index_prefix> <input_fastq_name>\n");
    fprintf(stderr, "    --output_prefix/-o <basename>    output prefix\n");
    fprintf(stderr, "    --markers/-m                     print markers\n");
    fprintf(stderr, "    --sam/-s                         print locations\n");
    fprintf(stderr, "    --gpu <n>                        HIP device ordinal (default 0)\n");
    fprintf(stderr, "    --batch <n>                      reads per GPU batch (default 1048576)\n");
    fprintf(stderr, "    <input_prefix>                   index prefix\n");
    fprintf(stderr, "    <input_fastq>                    input fastq\n");
}

RbAlignArgs parse_args(int argc, char **argv) {  // rb_align.cpp:37-84
    RbAlignArgs args;
    static struct option long_options[] = {{"output_prefix", required_argument, 0, 'o'},
                                           {"markers", no_argument, 0, 'm'},
                                           {"sam", no_argument, 0, 's'},
                                           {"gpu", required_argument, 0, 'g'},
                                           {"batch", required_argument, 0, 'b'},
                                           {0, 0, 0, 0}};
    int c, long_index = 0;
    while ((c = getopt_long(argc, argv, "o:smh", long_options, &long_index)) != -1) {
        switch (c) {
            case 'o': args.outpre = optarg; break;
            case 'h': print_help(); exit(0);
            case 's': args.sam = 1; break;
            case 'm': args.markers = 1; break;
            case 'g': args.device = atoi(optarg); break;
            case 'b': args.batch = strtoull(optarg, nullptr, 10); break;
            default: print_help(); exit(1);
        }
    }
    if (argc - optind < 2) {
        fprintf(stderr, "no argument provided\n");
        exit(1);
    }
    args.inpre = argv[optind++];
    args.fastq_fname = argv[optind++];
    if (args.outpre.empty()) args.outpre = args.inpre;
    if (args.batch == 0) args.batch = 1;
    return args;
}

// FASTA/FASTQ reader with kseq_read's observable behaviour (kseq.h:178-219)
class FastxReader {
   public:
    explicit FastxReader(gzFile fp) : fp_(fp) { buf_.resize(1 << 16); }
    // 0 = record read, -1 = EOF, -2 = truncated quality string
    int next(std::string &name, std::string &seq) {
        name.clear();
        seq.clear();
        if (!have_header_) {
            while (true) {
                if (!getline()) return -1;
                if (!line_.empty() && (line_[0] == '>' || line_[0] == '@')) break;
            }
        }
        have_header_ = false;
        const size_t ws = line_.find_first_of(" \t", 1);
        name = line_.substr(1, ws == std::string::npos ? std::string::npos : ws - 1);
        bool plus = false;
        while (getline()) {
            if (!line_.empty() && (line_[0] == '>' || line_[0] == '@')) { have_header_ = true; break; }
            if (!line_.empty() && line_[0] == '+') { plus = true; break; }
            for (char ch : line_)
                if (static_cast<unsigned char>(ch) > 32) seq.push_back(ch);  // kseq keeps isgraph() bytes
        }
        if (!plus) return 0;
        size_t qlen = 0;
        while (qlen < seq.size()) {
            if (!getline()) return -2;
            qlen += line_.size();
        }
        if (qlen != seq.size()) return -2;
        return 0;
    }

   private:
    bool getline() {
        line_.clear();
        while (true) {
            if (!gzgets(fp_, buf_.data(), static_cast<int>(buf_.size()))) return !line_.empty();
            const size_t len = strlen(buf_.data());
            line_.append(buf_.data(), len);
            if (len && buf_[len - 1] == '\n') break;
        }
        while (!line_.empty() && (line_.back() == '\n' || line_.back() == '\r')) line_.pop_back();
        return true;
    }
    gzFile fp_;
    std::vector<char> buf_;
    std::string line_;
    bool have_header_ = false;
};

// rb_report (rb_align.cpp:118-145) for a whole batch
void report_batch(const rbwt::RowBowt<> &rb, const RbAlignArgs &args, const std::vector<std::string> &names,
                  const std::vector<std::string> &seqs, std::string &out) {
    using RB = rbwt::RowBowt<>;
    const size_t N = seqs.size();
    std::vector<RB::LFData> lfs;
    std::vector<RB::range_t> ranges(N);
    if (args.sam) {  // rb_get_range(sa=true), rb_align.cpp:99-103
        rb.find_range_w_toehold_batch(seqs, lfs);
        for (size_t i = 0; i < N; ++i) ranges[i] = lfs[i].rn;
    } else {
        rb.find_range_batch(seqs, ranges);
    }
    std::vector<uint64_t> loc_off, locs, mk_off;
    std::vector<MarkerT> mk;
    if (args.sam) rb.locs_at_batch(lfs, static_cast<uint64_t>(-1), loc_off, locs);  // rb_align.cpp:125
    if (args.markers) rb.markers_at_batch(ranges, mk_off, mk);                       // rb_align.cpp:138
    char tmp[96];
    for (size_t i = 0; i < N; ++i) {
        out += names[i];
        snprintf(tmp, sizeof(tmp), " (%llu,%llu), count=%llu\n", (unsigned long long)ranges[i].first,
                 (unsigned long long)ranges[i].second, (unsigned long long)(ranges[i].second - ranges[i].first + 1));
        out += tmp;
        if (args.sam) {
            out += "\tlocs: ";
            for (uint64_t t = loc_off[i]; t < loc_off[i + 1]; ++t) {
                const auto x = rb.resolve_offset(locs[t]);
                snprintf(tmp, sizeof(tmp), "%llu/", (unsigned long long)locs[t]);
                out += tmp;
                out += x.first;
                snprintf(tmp, sizeof(tmp), ":%llu ", (unsigned long long)x.second);
                out += tmp;
            }
            out += "\n";
        }
        if (args.markers) {
            out += "\tmarkers: ";
            if (mk_off[i + 1] == mk_off[i]) out += "no markers (consider building the marker array with a larger window size)";
            for (uint64_t t = mk_off[i]; t < mk_off[i + 1]; ++t) {
                snprintf(tmp, sizeof(tmp), "%llu/%d ", (unsigned long long)get_pos(mk[t]), static_cast<int>(get_allele(mk[t])));
                out += tmp;
            }
            out += "\n";
        }
    }
}

}  // namespace

int main(int argc, char **argv) {
    const RbAlignArgs args = parse_args(argc, argv);
    auto start = std::chrono::high_resolution_clock::now();
    rbwt::LoadRbwtFlag flag = rbwt::LoadRbwtFlag::NONE;  // load_rbwt, rb_align.cpp:147-160
    if (args.sam) {
        std::cerr << "will load SA and DA" << std::endl;
        flag = flag | rbwt::LoadRbwtFlag::SA | rbwt::LoadRbwtFlag::DL;
    }
    if (args.markers) {
        std::cerr << "will load SA and DA" << std::endl;
        flag = flag | rbwt::LoadRbwtFlag::MA;
    }
    rbwt::RowBowt<> rb = rbwt::load_rowbowt<>(args.inpre, flag, args.device);
    auto stop = std::chrono::high_resolution_clock::now();
    const std::chrono::duration<double> index_load_time = stop - start;

    gzFile fq_fp = gzopen(args.fastq_fname.data(), "r");  // rb_align.cpp:169-173
    if (fq_fp == NULL) {
        fprintf(stderr, "invalid file\n");
        exit(1);
    }
    gzbuffer(fq_fp, 1 << 20);
    FastxReader reader(fq_fp);
    start = std::chrono::high_resolution_clock::now();
    std::vector<std::string> names, seqs;
    std::string name, seq, out;
    int err = 0;
    while (true) {
        names.clear();
        seqs.clear();
        while (names.size() < args.batch && (err = reader.next(name, seq)) == 0) {
            names.push_back(name);
            seqs.push_back(seq);
        }
        if (!names.empty()) {
            out.clear();
            report_batch(rb, args, names, seqs, out);
            fwrite(out.data(), 1, out.size(), stdout);
        }
        if (err != 0) break;
    }
    fflush(stdout);
    stop = std::chrono::high_resolution_clock::now();
    const std::chrono::duration<double> total_query_time = stop - start;
    gzclose(fq_fp);
    if (err == -2) {  // rb_align.cpp:182-191
        fprintf(stderr, "ERROR: truncated quality string\n");
        exit(1);
    }
    std::cerr << index_load_time.count() << " " << total_query_time.count() << std::endl;  // rb_align.cpp:192
    return 0;
}
