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

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../include/rowbowt_gpu.hpp"
#include "fastx.hpp"
#include "fastx_index.hpp"
#include "cli_input.hpp"
#include "rbg_thread_team.hpp"

namespace {

using rbg_cli::put_u64;
using rbg_cli::RecordSpans;

using rbg_cli::InputSource;
using rbg_cli::Window;

struct RbAlignArgs {  // rb_align.cpp:17-24
    std::string inpre, fastq_fname, outpre;
    int sam = 0, markers = 0;
    int device = 0;
    int gpus = 1;     // replicas: devices device .. device + gpus - 1; every batch is sharded over them
    std::vector<int> devices;  // --devices a,b,...: the replicas' devices, in shard order (overrides --gpu/--gpus)
    uint64_t batch = 1u << 18;   // (a 256 MB window of 100 bp FASTQ holds 1.2 M reads: several batches per window keep both pipeline stages busy)
    bool batch_given = false;    // -s doubles the default: every library call waits 3 ms for its first device operation (0.36 -> 0.28 s per 10 M reads)
    // input scanning and output formatting workers: an eighth of the CPUs, 8..32, within the container's CPU quota
    int threads = static_cast<int>(std::min({32u, std::max(8u, std::thread::hardware_concurrency() / 8), std::max(2u, rbg_hostpath::cpu_budget())}));
    uint64_t window_mb = 256;  // input bytes scanned per pipeline step
};

void print_help() {  // rb_align.cpp:26-35
    fprintf(stderr, "rb_align");
    fprintf(stderr, "Usage: rb_align [options] <index_prefix> <input_fastq_name>\n");
    fprintf(stderr, "    --output_prefix/-o <basename>    output prefix\n");
    fprintf(stderr, "    --markers/-m                     print markers\n");
    fprintf(stderr, "    --sam/-s                         print locations\n");
    fprintf(stderr, "    --gpu <n>                        HIP device ordinal (default 0)\n");
    fprintf(stderr, "    --gpus <G>                       replicate the index on G devices (from --gpu on) and shard every batch over them\n");
    fprintf(stderr, "    --devices <a,b,...>              the same with an explicit device list\n");
    fprintf(stderr, "    --batch <n>                      reads per GPU batch (default 262144; 524288 with -s)\n");
    fprintf(stderr, "    --threads <n>                    input scanning / output formatting threads (default: an eighth of the CPUs, 8..32)\n");
    fprintf(stderr, "    --window-mb <n>                  input bytes scanned per pipeline step (default 256)\n");
    fprintf(stderr, "    <input_prefix>                   index prefix\n");
    fprintf(stderr, "    <input_fastq>                    input fastq\n");
}

RbAlignArgs parse_args(int argc, char **argv) {  // rb_align.cpp:37-84
    RbAlignArgs args;
    static struct option long_options[] = {{"output_prefix", required_argument, 0, 'o'},
                                           {"markers", no_argument, 0, 'm'},
                                           {"sam", no_argument, 0, 's'},
                                           {"gpu", required_argument, 0, 'g'},
                                           {"gpus", required_argument, 0, 'G'},
                                           {"devices", required_argument, 0, 'D'},
                                           {"batch", required_argument, 0, 'b'},
                                           {"threads", required_argument, 0, 't'},
                                           {"window-mb", required_argument, 0, 'W'},
                                           {0, 0, 0, 0}};
    int c, long_index = 0;
    while ((c = getopt_long(argc, argv, "o:smh", long_options, &long_index)) != -1) {
        switch (c) {
            case 'o': args.outpre = optarg; break;
            case 'h': print_help(); exit(0);
            case 's': args.sam = 1; break;
            case 'm': args.markers = 1; break;
            case 'g': args.device = atoi(optarg); break;
            case 'G': args.gpus = atoi(optarg); break;
            case 'D':
                for (const char *p = optarg; *p;) {
                    args.devices.push_back(atoi(p));
                    while (*p && *p != ',') ++p;
                    if (*p == ',') ++p;
                }
                break;
            case 'b': args.batch = strtoull(optarg, nullptr, 10); args.batch_given = true; break;
            case 't': args.threads = atoi(optarg); break;
            case 'W': args.window_mb = strtoull(optarg, nullptr, 10); break;
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
    if (!args.batch_given && args.sam) args.batch = 1u << 19;
    if (args.gpus < 1) args.gpus = 1;
    if (args.threads < 1) args.threads = 1;
    if (args.window_mb < 1) args.window_mb = 1;
    if (args.devices.empty())
        for (int g = 0; g < args.gpus; ++g) args.devices.push_back(args.device + g);
    args.device = args.devices[0];
    return args;
}

// results of one shard of a batch: reads [begin, end) of the batch, answered by one replica
struct BatchResult {
    uint64_t begin = 0, end = 0;
    std::vector<uint64_t> lo, hi, k, loc_off, mk_off;
    uint64_t *locs = nullptr, *mk = nullptr;
    // -s: the shard's text as the library made it on the device (rbg_align_text); handed to the writer, which
    // gives the buffer back (rbg_release_text)
    const char *text = nullptr;
    uint64_t text_len = 0;
    rbg_index *text_owner = nullptr;
    BatchResult() = default;
    BatchResult(const BatchResult &) = delete;
    BatchResult &operator=(const BatchResult &) = delete;
    ~BatchResult() { rbg_free_buffer(locs); rbg_free_buffer(mk); if (text) (void)rbg_release_text(text_owner, text); }
};

// rb_get_range + locs_at + markers_at (rb_align.cpp:95-145) for reads [begin, end) of a batch on one replica.
// Runs on one worker thread per replica: a failing library call is RETURNED (code + the call's name) and reported by
// the main thread after the join -- exiting or throwing from a worker would take the process down mid-write.
struct ShardError {
    int rc = RBG_OK;
    const char *what = "";
};
// -s (with or without -m) prints 975 bytes per read on a pangenome index; the library writes that text with kernels unless
// RB_ALIGN_HOST_TEXT=1 asks for the host formatter below (A/B, tests: the two are byte-identical).  The one line per read
// of the count-only report stays with the host formatter: its 30 bytes per read cost less there (0.05 s per 10 M reads)
// than a library call per batch does (measured: 7.4e7 reads/s against 2.8e7 through rbg_align_text with k = NULL).
bool device_text(const RbAlignArgs &args) {
    static const bool off = [] { const char *e = std::getenv("RB_ALIGN_HOST_TEXT"); return e && e[0] == '1'; }();
    return args.sam && !off;
}

ShardError query_shard(rbg_index *ix, const RbAlignArgs &args, const Window &b, uint64_t begin, uint64_t end, BatchResult &r) {
    try {
        r.begin = begin;
        r.end = end;
        const uint64_t N = end - begin;
        if (N == 0) return {};
        // the sequences are read where the scanner found them (rbg_find_range_spans): no copy into a batch
        const uint8_t *base = reinterpret_cast<const uint8_t *>(b.base);
        const uint64_t *sb = b.recs.seq_begin.data() + begin;
        const uint32_t *sl = b.recs.seq_len.data() + begin;
        int rc;
        r.lo.resize(N); r.hi.resize(N);
        if (args.sam) {  // rb_get_range(sa=true), rb_align.cpp:99-103
            r.k.resize(N);
            if ((rc = rbg_find_range_spans(ix, base, sb, sl, N, r.lo.data(), r.hi.data(), r.k.data()))) return {rc, "rbg_find_range_spans"};
            if (device_text(args)) {   // locations, documents and decimals on the device: the finished text comes back
                r.text_owner = ix;
                if ((rc = rbg_align_text(ix, r.lo.data(), r.hi.data(), r.k.data(), N, static_cast<uint64_t>(-1), args.markers ? RBG_TEXT_MARKERS : 0, b.base, b.recs.name_begin.data() + begin,
                                         b.recs.name_len.data() + begin, &r.text, &r.text_len)))
                    return {rc, "rbg_align_text"};
                return {};
            }
            r.loc_off.resize(N + 1);
            if ((rc = rbg_locs_at(ix, r.lo.data(), r.hi.data(), r.k.data(), N, static_cast<uint64_t>(-1), r.loc_off.data(), &r.locs)))
                return {rc, "rbg_locs_at"};  // rb_align.cpp:125
        } else {
            if ((rc = rbg_find_range_spans(ix, base, sb, sl, N, r.lo.data(), r.hi.data(), nullptr))) return {rc, "rbg_find_range_spans"};
        }
        if (args.markers) {  // rb_align.cpp:138
            r.mk_off.resize(N + 1);
            if ((rc = rbg_markers_at(ix, r.lo.data(), r.hi.data(), N, r.mk_off.data(), &r.mk))) return {rc, "rbg_markers_at"};
        }
        return {};
    } catch (const std::bad_alloc &) {
        return {RBG_ENOMEM, "result arrays of a batch"};
    }
}

using rbg_cli::FastOut;
using rbg_cli::fmt_lit;
using rbg_cli::fmt_u64;

// the document table of the index (rbg_doc_table), fetched once when -s is given
struct DocView {
    uint64_t n = 0, size = 0;
    const uint64_t *starts = nullptr;
    const char *const *names = nullptr;
    std::vector<size_t> name_len;
    size_t max_name = 0;
} g_docs;

// the text of rb_report (rb_align.cpp:118-145) for reads [i0, i1)
void format_range(const rbwt::RowBowt<> &rb, const RbAlignArgs &args, const Window &b, const BatchResult &r, size_t g0,
                  size_t g1, rbg_cli::TextBuf &out_s) {
    static const char kNoMarkers[] = "no markers (consider building the marker array with a larger window size)";
    FastOut out(out_s);
    for (size_t gi = g0; gi < g1; ++gi) {   // gi: index in the window; i: index in the shard's results
        const size_t i = gi - r.begin;
        const size_t nl = b.recs.name_len[gi];
        char *p = out.room(nl + 96);
        char *const p0 = p;
        p = fmt_lit(p, b.base + b.recs.name_begin[gi], nl);
        p = fmt_lit(p, " (", 2);
        p = fmt_u64(p, r.lo[i]);
        *p++ = ',';
        p = fmt_u64(p, r.hi[i]);
        p = fmt_lit(p, "), count=", 9);
        p = fmt_u64(p, r.hi[i] - r.lo[i] + 1);  // unsigned wrap for the empty range, like the reference
        *p++ = '\n';
        out.len += static_cast<size_t>(p - p0);
        if (args.sam) {
            // "\tlocs: " + per location "<pos>/<doc>:<offset> " (rb_align.cpp:126-131), resolved from the document table
            // in place: a call per position through the ABI was a third of the formatting time (41 locations per read
            // on the bench index)
            const uint64_t nloc = r.loc_off[i + 1] - r.loc_off[i];
            p = out.room(16 + nloc * (44 + g_docs.max_name));
            char *const q0 = p;
            p = fmt_lit(p, "\tlocs: ", 7);
            for (uint64_t t = r.loc_off[i]; t < r.loc_off[i + 1]; ++t) {
                const uint64_t pos = r.locs[t];
                const uint64_t q = pos + 1 > g_docs.size ? g_docs.size : pos + 1;    // doclist.hpp:77-79 (pos + 1 wraps like the reference's)
                const uint64_t k = static_cast<uint64_t>(std::lower_bound(g_docs.starts, g_docs.starts + g_docs.n, q) - g_docs.starts);
                if (k == 0) rbwt::detail::die("rbg_resolve_offset", RBG_EARG);   // (the reference indexes doc_names_[-1] here)
                p = fmt_u64(p, pos);
                *p++ = '/';
                p = fmt_lit(p, g_docs.names[k - 1], g_docs.name_len[k - 1]);
                *p++ = ':';
                p = fmt_u64(p, pos - g_docs.starts[k - 1]);                                       // doclist.hpp:48
                *p++ = ' ';
            }
            *p++ = '\n';
            out.len += static_cast<size_t>(p - q0);
        }
        if (args.markers) {
            const uint64_t nm = r.mk_off[i + 1] - r.mk_off[i];
            p = out.room(16 + sizeof(kNoMarkers) + nm * 44);
            char *const q0 = p;
            p = fmt_lit(p, "\tmarkers: ", 10);
            if (nm == 0) p = fmt_lit(p, kNoMarkers, sizeof(kNoMarkers) - 1);
            for (uint64_t t = r.mk_off[i]; t < r.mk_off[i + 1]; ++t) {
                p = fmt_u64(p, get_pos(r.mk[t]));
                *p++ = '/';
                p = fmt_u64(p, get_allele(r.mk[t]));
                *p++ = ' ';
            }
            *p++ = '\n';
            out.len += static_cast<size_t>(p - q0);
        }
    }
    out.finish();
}

// RB_ALIGN_TRACE=1: seconds the main loop spent in the library calls, in formatting, waiting for the scanner and for the writer
double g_trace_query = 0, g_trace_format = 0, g_trace_scan_wait = 0, g_trace_write_wait = 0;
std::shared_future<void> g_text_writer[2];   // device_text(): the last two of the chain of per-batch writes (each waits for the one before it)

// query + format one batch: the batch is sharded over the replicas (contiguous blocks), the shards are
// queried concurrently, formatting is split over worker threads, pieces concatenated in read order
// (`pieces` is a pool that keeps its strings -- and their pages -- from window to window: `used` counts the ones of this
// window; fresh 12 MB strings per batch cost more in page faults than the formatting itself)
// One batch in flight between the two stages: the shards' results (one BatchResult per replica, kept from batch to batch
// like the pieces: the lo / hi / k arrays of a 4 M-read batch are 100 MB of pages)
struct BatchSlot {
    std::vector<std::unique_ptr<BatchResult>> res;
    std::vector<ShardError> err;
    double query_s = 0;
};

// stage 1: the batch is sharded over the replicas (contiguous blocks), the shards are queried concurrently
void query_batch(const std::vector<rbg_index *> &reps, const RbAlignArgs &args, const Window &b, size_t w0, size_t w1, BatchSlot &slot) {
    const size_t N = w1 - w0;
    const int G = static_cast<int>(reps.size());
    while (slot.res.size() < static_cast<size_t>(G)) slot.res.emplace_back(new BatchResult());
    slot.err.assign(G, ShardError());
    const auto t_q0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    auto work = [&](int g) {
        uint64_t s0 = 0, s1 = 0;
        (void)rbg_shard_bounds(N, g, G, &s0, &s1);
        slot.err[g] = query_shard(reps[g], args, b, w0 + s0, w0 + s1, *slot.res[g]);
    };
    for (int g = 1; g < G; ++g) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
    slot.query_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_q0).count();
}

// stage 2: formatting is split over worker threads, pieces concatenated in read order
// (`pieces` is a pool that keeps its strings -- and their pages -- from window to window: `used` counts the ones of this
// window; fresh 12 MB strings per batch cost more in page faults than the formatting itself)
struct ShardFailure {};   // a replica's call failed (already reported): leave through main()
void format_batch(const rbwt::RowBowt<> &rb, const RbAlignArgs &args, const Window &b, size_t w0, size_t w1, BatchSlot &slot,
                  std::vector<rbg_cli::TextBuf> &pieces, size_t &used) {
    const size_t N = w1 - w0;
    const int G = static_cast<int>(slot.err.size());
    for (int g = 0; g < G; ++g)
        if (slot.err[g].rc) {   // what the single-replica path says, from the main thread, once every worker is done
            fprintf(stderr, "%s (replica %d): %s\n", slot.err[g].what, g, rbg_strerror(slot.err[g].rc));
            // not exit(1) from here: the next batch's query and the chain of text writes may still be running on other
            // threads, inside library calls and fwrite -- main() unwinds (the futures join), flushes and leaves
            throw ShardFailure{};
        }
#define res(g) (*slot.res[(g)])
    const auto t_q1 = std::chrono::steady_clock::now();
    g_trace_query += slot.query_s;
    if (device_text(args)) {
        // nothing to format: the shards' texts go to the writer in shard order, behind the previous batch's
        struct Out { rbg_index *ix; const char *p; uint64_t n; };
        std::vector<Out> outs;
        for (int g = 0; g < G; ++g) {
            outs.push_back({res(g).text_owner, res(g).text, res(g).text_len});
            res(g).text = nullptr;
            res(g).text_len = 0;
        }
        if (g_text_writer[0].valid()) {   // at most two batches of text wait for stdout: the pinned buffers are a few hundred MB each
            g_text_writer[0].wait();
            g_trace_write_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_q1).count();
        }
        const std::shared_future<void> prev = g_text_writer[1];
        g_text_writer[0] = prev;
        g_text_writer[1] = std::async(std::launch::async, [outs, prev] {
            if (prev.valid()) prev.wait();
            for (const Out &o : outs) {
                (void)rbg_wait_text(o.ix, o.p);   // (its copy-out ran under the next batch's search)
                if (o.n) fwrite(o.p, 1, o.n, stdout);
                (void)rbg_release_text(o.ix, o.p);
            }
        }).share();
        return;
    }
    const size_t T = std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(args.threads), (N + 4095) / 4096, size_t(64)}));
    // piece (g, t): reads of shard g, t-th slice
    const size_t first_piece = used;
    used += static_cast<size_t>(G) * T;
    if (pieces.size() < used) pieces.resize(used);
    for (size_t i = first_piece; i < used; ++i) pieces[i].clear();
    std::vector<std::thread> workers;
    for (int g = 0; g < G; ++g)
        for (size_t t = 0; t < T; ++t) {
            const size_t n = res(g).end - res(g).begin;
            const size_t a = res(g).begin + n * t / T, z = res(g).begin + n * (t + 1) / T;
            if (a == z) continue;
            rbg_cli::TextBuf *dst = &pieces[first_piece + static_cast<size_t>(g) * T + t];
            const BatchResult *r = &res(g);
            if (g == 0 && t == 0) continue;  // done on this thread below
            workers.emplace_back([&rb, &args, &b, r, a, z, dst] { format_range(rb, args, b, *r, a, z, *dst); });
        }
    {
        const size_t n = res(0).end - res(0).begin;
        format_range(rb, args, b, res(0), res(0).begin, res(0).begin + n / T, pieces[first_piece]);
    }
    for (auto &w : workers) w.join();
    // the text is made: the ragged buffers (3 GB of locations per 10 M reads on a pangenome index) go back now, the
    // fixed-size arrays stay for the next batch
    for (int g = 0; g < G; ++g) {
        rbg_free_buffer(res(g).locs);
        rbg_free_buffer(res(g).mk);
        res(g).locs = res(g).mk = nullptr;
    }
    g_trace_format += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_q1).count();
#undef res
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
    // --gpus G: the device index is built once and copied peer to peer to the other devices
    std::vector<rbg_index *> reps{rb.handle()};
    struct Replicas {
        std::vector<rbg_index *> owned;
        ~Replicas() { for (rbg_index *r : owned) rbg_free(r); }
    } replicas;
    if (args.devices.size() > 1) {   // every further device at once: the peer copies overlap (one xGMI link per target)
        const int G = static_cast<int>(args.devices.size()) - 1;
        std::vector<rbg_index *> more(G, nullptr);
        const int rc = rbg_replicate_many(rb.handle(), args.devices.data() + 1, G, more.data());
        if (rc) {
            fprintf(stderr, "rb_align: replicas on %d further device(s): %s\n", G, rbg_strerror(rc));
            exit(1);
        }
        for (rbg_index *r : more) {
            replicas.owned.push_back(r);
            reps.push_back(r);
        }
    }
    if (args.sam) {   // load_rbwt asked for the DL flag: a missing .docs has already ended the run there (rowbowt_io.hpp:166-169)
        rbwt::detail::check(rbg_doc_table(rb.handle(), &g_docs.n, &g_docs.starts, &g_docs.names, &g_docs.size), "rbg_doc_table");
        for (uint64_t d = 0; d < g_docs.n; ++d) {
            g_docs.name_len.push_back(std::strlen(g_docs.names[d]));
            g_docs.max_name = std::max(g_docs.max_name, g_docs.name_len.back());
        }
    }
    auto stop = std::chrono::high_resolution_clock::now();
    const std::chrono::duration<double> index_load_time = stop - start;

    InputSource input;  // rb_align.cpp:169-173
    if (!input.open(args.fastq_fname, static_cast<unsigned>(args.threads), args.window_mb << 20)) {
        fprintf(stderr, "invalid file\n");
        exit(1);
    }
    if (device_text(args)) {
        // RB_ALIGN_TEXT_STREAMS=<n>: every replica answers its shard as n sub-shards on n threads.  One is the default: the
        // library copies a text out on its own stream under the caller's next calls, and more callers only compete for the
        // CPUs that pack the reads (0.54 s per 10 M reads with one, 0.66 with two, 0.99 with four on a 16-CPU quota).
        const char *e = std::getenv("RB_ALIGN_TEXT_STREAMS");
        const int S = e && std::atoi(e) >= 1 ? std::min(8, std::atoi(e)) : 1;
        const std::vector<rbg_index *> once = reps;
        reps.clear();
        for (rbg_index *r : once)
            for (int t = 0; t < S; ++t) reps.push_back(r);
        // the texts are copied out into pinned buffers: FOUR per half-shard, made before the clock starts.  With three, whether a fourth had
        // to be allocated inside the loop (0.25-0.3 s of hipHostMalloc for 0.7 GB) depended on a race between the writer giving a buffer
        // back and the next batch's text being ready: the process-to-process bimodality of `rb_align -s -m` in round 3 (2.0e7 against
        // 3.4e7 reads/s) -- profiles/r04_numa_probe.txt lists the pinned allocations of 32 processes beside their rates.
        for (rbg_index *r : once) (void)rbg_reserve_text(r, (args.batch / reps.size() + 1) * 1100 + (size_t(1) << 20), 4 * S);
    }
    start = std::chrono::high_resolution_clock::now();
    // three overlapped stages: scan window i+1 | query + format window i (in GPU batches of --batch reads) | write window i-1
    int err = 0;
    Window cur, nxt;
    err = input.next(cur);
    std::future<void> writer;
    std::vector<rbg_cli::TextBuf> pieces, writing;
    size_t used = 0, writing_used = 0;
    BatchSlot slots[2];
    try {
    while (true) {
        std::future<int> scanner;
        const bool more = err == 0;
        if (more) scanner = std::async(std::launch::async, [&input, &nxt] { return input.next(nxt); });
        used = 0;
        // two stages over the window's batches: batch j + 1 is queried (library calls: packing, GPU, results back) while
        // batch j is formatted -- with forty locations per read the text is four fifths of a batch's time
        {
            auto bounds = [&](size_t j, size_t &w0, size_t &w1) { w0 = j * args.batch; w1 = std::min<size_t>(cur.size(), w0 + args.batch); };
            const size_t nb = (cur.size() + args.batch - 1) / args.batch;
            std::future<void> ahead;
            size_t a0 = 0, a1 = 0;
            if (nb) { bounds(0, a0, a1); query_batch(reps, args, cur, a0, a1, slots[0]); }
            for (size_t j = 0; j < nb; ++j) {
                if (ahead.valid()) ahead.get();
                size_t w0, w1;
                bounds(j, w0, w1);
                if (j + 1 < nb) {
                    bounds(j + 1, a0, a1);
                    BatchSlot *nxt_slot = &slots[(j + 1) & 1];
                    ahead = std::async(std::launch::async, [&reps, &args, &cur, a0, a1, nxt_slot] { query_batch(reps, args, cur, a0, a1, *nxt_slot); });
                }
                format_batch(rb, args, cur, w0, w1, slots[j & 1], pieces, used);
            }
        }
        {
            const auto tw0 = std::chrono::steady_clock::now();
            if (writer.valid()) writer.get();
            g_trace_write_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        }
        writing.swap(pieces);
        writing_used = used;
        writer = std::async(std::launch::async, [&writing, &writing_used] {
            for (size_t i = 0; i < writing_used; ++i) fwrite(writing[i].data(), 1, writing[i].size(), stdout);
        });
        if (!more) break;
        {
            const auto ts0 = std::chrono::steady_clock::now();
            err = scanner.get();
            g_trace_scan_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
        }
        std::swap(cur, nxt);
    }
    } catch (const ShardFailure &) {   // (the loop's futures -- scanner, query ahead -- joined while unwinding)
        if (writer.valid()) writer.get();
        if (g_text_writer[1].valid()) g_text_writer[1].wait();
        fflush(stdout);
        return 1;
    }
    if (writer.valid()) writer.get();
    {
        const auto tw0 = std::chrono::steady_clock::now();
        if (g_text_writer[1].valid()) g_text_writer[1].wait();
        g_trace_write_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
    }
    fflush(stdout);
    stop = std::chrono::high_resolution_clock::now();
    const std::chrono::duration<double> total_query_time = stop - start;
    switch (err) {  // rb_align.cpp:182-191
        case -2:
            fprintf(stderr, "ERROR: truncated quality string\n");
            exit(1);
            break;
        case -3:
            fprintf(stderr, "ERROR: error reading stream\n");
            exit(1);
            break;
        default:
            break;
    }
    if (std::getenv("RB_ALIGN_TRACE"))
        fprintf(stderr, "rb_align loop: library calls %.3f s, formatting %.3f s, waiting for the scanner %.3f s, for the writer %.3f s\n", g_trace_query,
                g_trace_format, g_trace_scan_wait, g_trace_write_wait);
    std::cerr << index_load_time.count() << " " << total_query_time.count() << std::endl;  // rb_align.cpp:192
    return 0;
}
