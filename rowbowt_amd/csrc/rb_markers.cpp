// rb_markers -- drop-in for the reference's marker genotyping tool
//   rb_markers [--wsize N] [--max-range N] [--min-range N] [--heuristic ...] <index_prefix> <fastq>
// (reference src/rb_markers.cpp) in its default seeding mode: get_markers_greedy_seeding without an
// ftab on the read and on its reverse complement (rb_markers.cpp:396-413).  The per-read work of the
// reference's thread pool (rb_markers.cpp:318-535) becomes one batched call into the MI355X engine
// (rbg_get_markers_greedy_seeding, include/rbg.h) over 2N sequences; what the reference's callback
// and worker do afterwards (sort/unique the markers, the seed filters, the text) runs on host
// threads, in read order -- the order the reference produces with --threads 1.
//
// One stdout line per seed (rb_markers.cpp:253-262):
//   "<name> <range_size> <+|-> <query_start> <query_len>" { " <seq>/<pos>/<allele>" | " ." } "\n"
//
// --ftab/-f loads <index_prefix>.ftab like the reference (LoadRbwtFlag::FT): the file must be the one
// `rb_build -f` makes for this index (checked), and seeding then goes through search_ftab
// (rowbowt.hpp:430-433, :454-464).
//
// Not carried over (each exits 1 with a message, like the reference does for --overlap):
//   --lmem      get_markers_lmems (rowbowt.hpp:341-404) prints debugging text per step and is O(m^2)
//   --fbb/-x    other string type, other index file
#include <getopt.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../include/rowbowt_gpu.hpp"
#include "fastx.hpp"
#include "cli_input.hpp"
#include <cstdlib>
#include "rbg_thread_team.hpp"

namespace {

using rbg_cli::FastxReader;
using rbg_cli::InputSource;
using rbg_cli::Window;

// reads [w0, w0 + n) of a window, where the scanner found them (names and sequences are spans of the window's buffer)
struct BatchView {
    const Window *w;
    size_t w0, n;
    size_t size() const { return n; }
    const char *name(size_t i) const { return w->base + w->recs.name_begin[w0 + i]; }
    size_t name_len(size_t i) const { return w->recs.name_len[w0 + i]; }
    const char *seq(size_t i) const { return w->base + w->recs.seq_begin[w0 + i]; }
    uint64_t seq_len(size_t i) const { return w->recs.seq_len[w0 + i]; }
};
using rbg_cli::put_u64;

struct RbMarkersArgs {  // rb_markers.cpp:22-40
    std::string inpre, fastq_fname;
    int ftab = 0, fbb = 0, overlap = 0, lmem = 0;
    uint64_t wsize = 19, max_range = 1000, min_range = 0;
    uint64_t threads = 8, max_tasks = 1024, read_len = 101, min_seed_len = 0;  // threads: host formatting workers (the reference's default of 1 is its search pool)
    int clear_conflicting = 0, clear_identical = 0, best_strand = 0, heuristic = 0;
    int device = 0;
    uint64_t batch = 1u << 18;   // (2^17 kept more batches per window in flight but paid the library call's fixed costs twice as often: 0.32 against 0.25 s per 2 M reads)
};

void print_help() {  // rb_markers.cpp:44-54
    fprintf(stderr, "rb_markers\n");
    fprintf(stderr, "Usage: rb_markers_only [options] <index_prefix> <input_fastq_name>\n");
    fprintf(stderr, "    --wsize            <int>         window size for performing marker queries along read\n");
    fprintf(stderr, "    --max-range        <int>         range-size upper threshold for performing marker queries\n");
    fprintf(stderr, "    --min-range        <int>         range-size lower threshold for reporting markers\n");
    fprintf(stderr, "    --threads          <int>         host threads formatting the output\n");
    fprintf(stderr, "    --heuristic [--best-strand-only] [--min-seed-length <int>] [--read-len <int>]\n");
    fprintf(stderr, "                [--clear-conflicting] [--clear-identical]\n");
    fprintf(stderr, "    --gpu <n>                        HIP device ordinal (default 0)\n");
    fprintf(stderr, "    --batch <n>                      reads per GPU batch (default 262144)\n");
    fprintf(stderr, "    <input_prefix>                   index prefix\n");
    fprintf(stderr, "    <input_fastq>                    input fastq\n");
}

RbMarkersArgs parse_args(int argc, char **argv) {  // rb_markers.cpp:56-134
    static RbMarkersArgs args;
    static struct option long_options[] = {{"wsize", required_argument, 0, 'w'},
                                           {"max-range", required_argument, 0, 'r'},
                                           {"min-range", required_argument, 0, 'm'},
                                           {"threads", required_argument, 0, 't'},
                                           {"max-tasks", required_argument, 0, 'u'},
                                           {"read-len", required_argument, 0, 'l'},
                                           {"fbb", no_argument, &args.fbb, 1},
                                           {"ftab", no_argument, &args.ftab, 1},
                                           {"overlap", no_argument, &args.overlap, 1},
                                           {"lmem", no_argument, &args.lmem, 1},
                                           {"heuristic", no_argument, &args.heuristic, 1},
                                           {"best-strand-only", no_argument, &args.best_strand, 1},
                                           {"min-seed-length", required_argument, 0, 'y'},
                                           {"clear-conflicting", no_argument, &args.clear_conflicting, 1},
                                           {"clear-identical", no_argument, &args.clear_identical, 1},
                                           {"gpu", required_argument, 0, 'G'},
                                           {"batch", required_argument, 0, 'B'},
                                           {0, 0, 0, 0}};
    int c, long_index = 0;
    // "o:" is accepted by the reference's optstring but has no case: it ends in the default branch
    while ((c = getopt_long(argc, argv, "o:w:r:hft:m:u:xl:y:", long_options, &long_index)) != -1) {
        switch (c) {
            case 0: break;
            case 'y': args.min_seed_len = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'l': args.read_len = static_cast<uint64_t>(std::atol(optarg)); break;
            case 't': args.threads = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'u': args.max_tasks = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'f': args.ftab = 1; break;
            case 'r': args.max_range = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'm': args.min_range = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'w': args.wsize = static_cast<uint64_t>(std::atol(optarg)); break;
            case 'h': print_help(); exit(0);
            case 'x': args.fbb = 1; break;
            case 'G': args.device = atoi(optarg); break;
            case 'B': args.batch = strtoull(optarg, nullptr, 10); break;
            default: print_help(); exit(1);
        }
    }
    if (args.overlap) {  // rb_markers.cpp:121-124
        fprintf(stderr, "overlapped seeds currently broken\n");
        exit(1);
    }
    if (args.lmem) {  // without an ftab the reference stops at rowbowt.hpp:346-349
        fprintf(stderr, args.ftab ? "rb_markers: --lmem is not built in this engine\n" : "ftab must be enabled!\n");
        exit(1);
    }
    if (args.fbb) {
        fprintf(stderr, "rb_markers: --fbb indexes are not supported by this engine\n");
        exit(1);
    }
    if (argc - optind < 2) {
        fprintf(stderr, "no argument provided\n");
        exit(1);
    }
    args.inpre = argv[optind++];
    args.fastq_fname = argv[optind++];
    if (args.batch == 0) args.batch = 1;
    if (args.threads == 0) args.threads = 1;
    return args;
}

// seqtk's nt->ACGT table as rb_markers.cpp:139-156 carries it: A/C/G/T in either case keep their
// base, 'N' and 'n' become 'A' (sic), every other byte becomes 'N'.
struct NtTable {
    uint8_t t[256];
    NtTable() {
        memset(t, 'N', sizeof t);
        t['A'] = t['a'] = 'A';
        t['C'] = t['c'] = 'C';
        t['G'] = t['g'] = 'G';
        t['T'] = t['t'] = 'T';
        t['N'] = t['n'] = 'A';
    }
};
const NtTable kNt;

// complement of the five letters the table above can produce (comp_tab, rb_markers.cpp:159-168)
inline char comp(char c) {
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        default: return c;
    }
}

bool marker_cmp(MarkerT a, MarkerT b) {  // rb_markers.cpp:228-236
    if (get_seq(a) != get_seq(b)) return get_seq(a) < get_seq(b);
    if (get_pos(a) != get_pos(b)) return get_pos(a) < get_pos(b);
    return get_allele(a) < get_allele(b);
}

enum class Strand { FWD, REV };

struct MarkerSeed {  // rb_markers.cpp:245-285
    Strand strand = Strand::FWD;
    uint64_t range_size = 0, query_start = 0, query_len = 0;
    std::vector<MarkerT> markers;

    // keeps only markers whose (seq,pos) no neighbour shares; assumes sorted (rb_markers.cpp:266-276).
    // pm starts as marker 0, so a leading marker at seq 0 / pos 0 is dropped too, as in the reference.
    void filter_identical_pos() {
        if (markers.empty()) return;
        std::vector<MarkerT> kept;
        MarkerT pm = 0;
        for (size_t i = 0; i < markers.size(); ++i) {
            const MarkerT m = markers[i];
            bool drop = get_seq(m) == get_seq(pm) && get_pos(m) == get_pos(pm);
            if (!drop) {
                pm = m;
                drop = i + 1 < markers.size() && get_seq(markers[i + 1]) == get_seq(m) && get_pos(markers[i + 1]) == get_pos(m);
            }
            if (!drop) kept.push_back(m);
        }
        markers.swap(kept);
    }
    void clear_if_conflicting(uint64_t read_len) {  // rb_markers.cpp:279-284
        if (markers.empty()) return;
        if (get_seq(markers.back()) != get_seq(markers.front()) || get_pos(markers.back()) - get_pos(markers.front()) >= read_len)
            markers.clear();
    }
};

void print_seed(rbg_cli::FastOut &out, const char *name, size_t name_len, const MarkerSeed &ms) {  // print_buf, :253-262
    using rbg_cli::fmt_lit;
    using rbg_cli::fmt_u64;
    char *p = out.room(name_len + 80 + ms.markers.size() * 48);
    char *const p0 = p;
    p = fmt_lit(p, name, name_len);
    *p++ = ' ';
    p = fmt_u64(p, ms.range_size);
    p = fmt_lit(p, ms.strand == Strand::FWD ? " + " : " - ", 3);
    p = fmt_u64(p, ms.query_start);
    *p++ = ' ';
    p = fmt_u64(p, ms.query_len);
    if (!ms.markers.empty()) {
        for (const MarkerT m : ms.markers) {
            *p++ = ' ';
            p = fmt_u64(p, get_seq(m));
            *p++ = '/';
            p = fmt_u64(p, get_pos(m));
            *p++ = '/';
            p = fmt_u64(p, get_allele(m));
        }
    } else {
        p = fmt_lit(p, " .", 2);
    }
    *p++ = '\n';
    out.len += static_cast<size_t>(p - p0);
}

struct BatchSeeds {
    std::vector<uint64_t> seed_off;  // 2N+1: sequence 2i = read i forward, 2i+1 = its reverse complement
    rbg_marker_seed_t *seeds = nullptr;
    uint64_t *mk = nullptr;
    ~BatchSeeds() { rbg_free_buffer(seeds); rbg_free_buffer(mk); }
};

// the reference's out_fn (rb_markers.cpp:365-382 / :439-464) for one callback record
// (fills `ms` in place: its marker vector keeps its capacity from seed to seed -- nine seeds per read, no allocation)
void make_seed(const RbMarkersArgs &args, const BatchSeeds &r, const rbg_marker_seed_t &s, Strand strand, uint64_t seq_len, MarkerSeed &ms) {
    ms.markers.clear();
    ms.strand = strand;
    ms.range_size = s.hi - s.lo + 1;
    ms.query_start = strand == Strand::REV ? seq_len - s.qstart - 1 : s.qstart;
    ms.query_len = s.qend - s.qstart;  // q.second - q.first + 1
    if (ms.range_size >= args.min_range && s.mk_end > s.mk_begin) {
        ms.markers.assign(r.mk + s.mk_begin, r.mk + s.mk_end);
        std::sort(ms.markers.begin(), ms.markers.end(), marker_cmp);
        ms.markers.erase(std::unique(ms.markers.begin(), ms.markers.end()), ms.markers.end());
    }
}

// text for reads [i0, i1); first_fwd[i] is the heuristic worker's coin for read i
void format_range(const RbMarkersArgs &args, const BatchView &b, const BatchSeeds &r, const std::vector<uint8_t> &first_fwd,
                  size_t i0, size_t i1, rbg_cli::TextBuf &out_s) {
    rbg_cli::FastOut out(out_s);
    std::vector<MarkerSeed> seeds;
    MarkerSeed one;   // the default mode prints a seed as soon as it is made
    for (size_t i = i0; i < i1; ++i) {
        const char *name = b.name(i);
        const size_t name_len = b.name_len(i);
        const uint64_t seq_len = b.seq_len(i);
        seeds.clear();
        if (!args.heuristic) {  // worker, rb_markers.cpp:357-428: forward, then reverse complement
            for (int st = 0; st < 2; ++st)
                for (uint64_t s = r.seed_off[2 * i + st]; s < r.seed_off[2 * i + st + 1]; ++s) {
                    make_seed(args, r, r.seeds[s], st ? Strand::REV : Strand::FWD, seq_len, one);
                    print_seed(out, name, name_len, one);
                }
        } else {  // worker_heuristic, rb_markers.cpp:429-519
            bool stop = false;
            for (int pass = 0; pass < 2 && !stop; ++pass) {
                const bool fwd = (pass == 0) == (first_fwd[i] != 0);
                const int st = fwd ? 0 : 1;
                for (uint64_t s = r.seed_off[2 * i + st]; s < r.seed_off[2 * i + st + 1]; ++s) {
                    MarkerSeed ms;
                    make_seed(args, r, r.seeds[s], fwd ? Strand::FWD : Strand::REV, seq_len, ms);
                    if (ms.query_len < args.min_seed_len) {  // :447 (before any marker work; same result)
                        continue;
                    }
                    if (args.clear_conflicting) ms.clear_if_conflicting(args.read_len);
                    if (args.clear_identical) ms.filter_identical_pos();
                    // :460: not enough useful sequence left over (unsigned arithmetic as in the reference)
                    if (args.best_strand && args.read_len - (ms.query_start + ms.query_len) < args.min_seed_len) stop = true;
                    seeds.push_back(std::move(ms));
                }
                // `stop` is only looked at between the two strands (:499-502)
            }
            if (args.best_strand && !seeds.empty()) {  // keep_seeds_best_strand, :292-297, :308-314
                auto best = std::max_element(seeds.begin(), seeds.end(),
                                             [](const MarkerSeed &l, const MarkerSeed &r2) { return l.query_len < r2.query_len; });
                const Strand keep = best->strand;
                seeds.erase(std::remove_if(seeds.begin(), seeds.end(), [&](const MarkerSeed &ms) { return ms.strand != keep; }), seeds.end());
            }
            if (args.min_seed_len)  // keep_seeds_by_len, :299-303
                seeds.erase(std::remove_if(seeds.begin(), seeds.end(), [&](const MarkerSeed &ms) { return ms.query_len < args.min_seed_len; }),
                            seeds.end());
        }
        for (const MarkerSeed &ms : seeds) print_seed(out, name, name_len, ms);
    }
    out.finish();
}

// forward (through the nt table) and reverse-complement copies of every read, interleaved; the reads are split
// over `threads` workers once their offsets are known
void make_strands(const BatchView &b, std::string &seqs, std::vector<uint64_t> &off, size_t threads) {
    const size_t N = b.size();
    off.resize(2 * N + 1);
    off[0] = 0;
    for (size_t i = 0; i < N; ++i) {
        const uint64_t len = b.seq_len(i);
        off[2 * i + 1] = off[2 * i] + len;
        off[2 * i + 2] = off[2 * i] + 2 * len;
    }
    seqs.resize(off[2 * N]);
    const size_t T = std::max<size_t>(1, std::min<size_t>(threads, (N + 16383) / 16384));
    auto work = [&](size_t t) {
        for (size_t i = N * t / T; i < N * (t + 1) / T; ++i) {
            const char *src = b.seq(i);
            const size_t len = b.seq_len(i);
            char *f = &seqs[off[2 * i]], *rc = f + len;
            for (size_t u = 0; u < len; ++u) f[u] = static_cast<char>(kNt.t[static_cast<uint8_t>(src[u])]);
            for (size_t u = 0; u < len; ++u) rc[u] = comp(f[len - 1 - u]);  // revc_in_place, rb_markers.cpp:189-198
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

// std::mt19937 with its default seed, one bit per read, low bit first (RandomBoolGenerator, :210-225)
struct RandomBoolGenerator {
    bool get_bool() {
        if (bit_count == 0) {
            data = static_cast<uint32_t>(rng());
            bit_count = 32;
        }
        const bool bit = data & 1;
        data >>= 1;
        bit_count--;
        return bit;
    }
    std::mt19937 rng;
    uint32_t data = 0;
    int bit_count = 0;
};

double g_trace[3] = {0, 0, 0};   // RB_ALIGN_TRACE=1: seconds building the strands, in the library call, sorting + formatting

// One batch between the two stages of the loop: strands + library call | seeds -> text
struct SeedSlot {
    std::string seqs;
    std::vector<uint64_t> off;
    BatchSeeds r;
    int rc = RBG_OK;
    double t_strands = 0, t_query = 0;
};

// stage 1: both strands of every read, one library call for the 2N sequences
void query_batch(const rbwt::RowBowt<> &rb, const RbMarkersArgs &args, const BatchView &b, SeedSlot &slot) {
    const size_t N = b.size();
    const uint64_t ft_k = rb.ftab_k();
    const auto t0 = std::chrono::steady_clock::now();
    make_strands(b, slot.seqs, slot.off, static_cast<size_t>(args.threads));
    const auto t1 = std::chrono::steady_clock::now();
    rbg_free_buffer(slot.r.seeds);
    rbg_free_buffer(slot.r.mk);
    slot.r.seeds = nullptr;
    slot.r.mk = nullptr;
    slot.r.seed_off.resize(2 * N + 1);
    slot.rc = rbg_get_markers_greedy_seeding(rb.handle(), reinterpret_cast<const uint8_t *>(slot.seqs.data()), slot.off.data(), 2 * N, args.wsize,
                                             args.max_range, ft_k, slot.r.seed_off.data(), &slot.r.seeds, &slot.r.mk);
    slot.t_strands = std::chrono::duration<double>(t1 - t0).count();
    slot.t_query = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

// stage 2 (`pieces` is a pool that keeps its buffers from window to window: `used` counts the ones of this window)
void format_batch(const RbMarkersArgs &args, const BatchView &b, SeedSlot &slot, RandomBoolGenerator &booler, std::vector<rbg_cli::TextBuf> &pieces,
                  size_t &used) {
    const size_t N = b.size();
    rbwt::detail::check(slot.rc, "rbg_get_markers_greedy_seeding");
    const BatchSeeds &r = slot.r;
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<uint8_t> first_fwd(N, 1);
    if (args.heuristic)
        for (size_t i = 0; i < N; ++i) first_fwd[i] = booler.get_bool() ? 1 : 0;  // :483
    const size_t T = std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(args.threads), (N + 4095) / 4096, size_t(64)}));
    const size_t first_piece = used;
    used += T;
    if (pieces.size() < used) pieces.resize(used);
    for (size_t i = first_piece; i < used; ++i) pieces[i].clear();
    std::vector<std::thread> workers;
    for (size_t t = 1; t < T; ++t)
        workers.emplace_back([&, t] { format_range(args, b, r, first_fwd, N * t / T, N * (t + 1) / T, pieces[first_piece + t]); });
    format_range(args, b, r, first_fwd, 0, N / T, pieces[first_piece]);
    for (auto &w : workers) w.join();
    const auto t3 = std::chrono::steady_clock::now();
    g_trace[0] += slot.t_strands;
    g_trace[1] += slot.t_query;
    g_trace[2] += std::chrono::duration<double>(t3 - t2).count();
}

}  // namespace

int main(int argc, char **argv) {
    const RbMarkersArgs args = parse_args(argc, argv);
    auto start = std::chrono::high_resolution_clock::now();
    std::cerr << "(rle_string_sd) loading rowbowt + markers" << (args.ftab ? " and ftab" : "") << std::endl;  // rb_markers.cpp:553-557
    rbwt::LoadRbwtFlag flag = rbwt::LoadRbwtFlag::MA;  // :541-546
    if (args.ftab) flag = flag | rbwt::LoadRbwtFlag::FT;
    // (the library's automatic layout: the run-indexed replica.  Until round 5 the tool asked for RBG_LAYOUT_PREFER_SLOTS because the seeding
    //  kernels are a quarter faster on slot tables; that is 2 ms of a 190 ms query loop for two million reads, against a replica five times
    //  the size and its load time -- profiles/r05_rb_markers_layout.txt.  RBG_LAYOUT=prefer-slots brings it back.)
    rbwt::RowBowt<> rb = rbwt::load_rowbowt<>(args.inpre, flag, args.device);
    if (rb.ftab_k() && rb.ftab_k() - 1 > args.wsize) {  // rowbowt.hpp:423-426
        std::cerr << "ERROR: wsize cannot be greater than or equal to ftab k size. please rebuild ftab with smaller k\n";
        exit(1);
    }
    auto stop = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> diff = stop - start;
    std::cerr << "loading rowbowt + markers took: " << diff.count() << " seconds\n";

    start = std::chrono::high_resolution_clock::now();
    InputSource input;  // :568-572 (the file is scanned in place, window by window: cli_input.hpp)
    if (!input.open(args.fastq_fname, static_cast<unsigned>(std::max<uint64_t>(1, args.threads)), uint64_t(256) << 20)) {
        fprintf(stderr, "invalid file\n");
        exit(1);
    }
    RandomBoolGenerator booler;
    // three overlapped stages: scan window i+1 | query + format window i (in batches of --batch reads) | write window i-1
    int err = 0;
    Window cur, nxt;
    err = input.next(cur);
    std::future<void> writer;
    std::vector<rbg_cli::TextBuf> pieces, writing;
    size_t used = 0, writing_used = 0;
    SeedSlot slots[2];
    while (true) {
        std::future<int> scanner;
        const bool more = err == 0;
        if (more) scanner = std::async(std::launch::async, [&input, &nxt] { return input.next(nxt); });
        used = 0;
        {   // two stages over the window's batches: batch j + 1 is searched while batch j's seeds are sorted and printed
            const uint64_t ft_k = rb.ftab_k();
            if (ft_k)
                for (size_t i = 0; i < cur.size(); ++i)
                    if (cur.recs.seq_len[i] < ft_k) {  // the reference dies in std::string::substr (rowbowt.hpp:431)
                        fprintf(stderr, "ERROR: read shorter than the ftab k-mer size (%llu)\n", static_cast<unsigned long long>(ft_k));
                        exit(1);
                    }
            const size_t nb = (cur.size() + args.batch - 1) / args.batch;
            auto view = [&](size_t j) { return BatchView{&cur, j * args.batch, std::min<size_t>(cur.size() - j * args.batch, args.batch)}; };
            std::future<void> ahead;
            if (nb) query_batch(rb, args, view(0), slots[0]);
            for (size_t j = 0; j < nb; ++j) {
                if (ahead.valid()) ahead.get();
                if (j + 1 < nb) {
                    SeedSlot *ns = &slots[(j + 1) & 1];
                    const BatchView nv = view(j + 1);
                    ahead = std::async(std::launch::async, [&rb, &args, nv, ns] { query_batch(rb, args, nv, *ns); });
                }
                format_batch(args, view(j), slots[j & 1], booler, pieces, used);
            }
        }
        if (writer.valid()) writer.get();
        writing.swap(pieces);
        writing_used = used;
        writer = std::async(std::launch::async, [&writing, &writing_used] {
            for (size_t i = 0; i < writing_used; ++i) fwrite(writing[i].data(), 1, writing[i].size(), stdout);
        });
        if (!more) break;
        err = scanner.get();
        std::swap(cur, nxt);
    }
    if (writer.valid()) writer.get();
    fflush(stdout);
    switch (err) {  // rb_markers.cpp:581-590
        case -2: fprintf(stderr, "ERROR: truncated quality string\n"); exit(1);
        case -3: fprintf(stderr, "ERROR: error reading stream\n"); exit(1);
        default: break;
    }
    stop = std::chrono::high_resolution_clock::now();
    diff = stop - start;
    if (std::getenv("RB_ALIGN_TRACE"))
        fprintf(stderr, "rb_markers loop: strands %.3f s, library call %.3f s, seeds -> text %.3f s\n", g_trace[0], g_trace[1], g_trace[2]);
    std::cerr << "counting markers took: " << diff.count() << " seconds" << std::endl;
    return 0;
}
