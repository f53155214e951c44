// shim_threads.cpp -- an UNMODIFIED threaded caller of the reference's one-query-at-a-time methods (the shape of
// rb_markers.cpp:318-535: a pool of threads, each calling const query methods on one shared RowBowt): T threads call
// RowBowt::find_range / find_range_w_toehold / get_markers_greedy_seeding with ONE read per call through
// rowbowt_gpu.hpp.  Answers are compared with the batch forms; the time and rbg_combine_stats show what the
// micro-batching queue of the library does for such a caller.
//   usage: shim_threads <index prefix> <fasta of the indexed text> <threads> <queries>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <random>
#include <thread>

#include "rowbowt_gpu.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string prefix = argv[1];
    const unsigned T = static_cast<unsigned>(std::atoi(argv[3]));
    const size_t Q = static_cast<size_t>(std::atol(argv[4]));
    std::string text, line;
    {
        std::ifstream fa(argv[2]);
        while (std::getline(fa, line))
            if (!line.empty() && line[0] != '>') text += line;
    }
    if (text.size() < 1000) { std::fprintf(stderr, "no text\n"); return 2; }
    auto rb = rbwt::load_rowbowt<rbwt::rle_string_t>(prefix, rbwt::LoadRbwtFlag::SA | rbwt::LoadRbwtFlag::MA);
    using RB = rbwt::RowBowt<rbwt::rle_string_t>;
    std::mt19937_64 rng(7);
    std::vector<std::string> queries(Q);
    for (size_t i = 0; i < Q; ++i) {
        const size_t len = 30 + rng() % 60, at = rng() % (text.size() - len);
        queries[i] = text.substr(at, len);
        if (rng() % 8 == 0) queries[i][rng() % len] = "ACGT"[rng() % 4];
    }
    std::vector<RB::range_t> want;
    std::vector<RB::LFData> want_lf;
    rb.find_range_batch(queries, want);
    rb.find_range_w_toehold_batch(queries, want_lf);
    std::atomic<size_t> bad{0}, seeds_seen{0};
    uint64_t st0[2], st1[2];
    rbg_combine_stats(rb.handle(), st0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t)
        pool.emplace_back([&, t] {
            for (size_t i = t; i < Q; i += T) {
                const auto r = rb.find_range(queries[i]);
                if (r != want[i]) bad++;
                const auto lf = rb.find_range_w_toehold(queries[i]);
                if (lf.rn != want_lf[i].rn || lf.ssamp != want_lf[i].ssamp) bad++;
                if (i % 4 == 0) {
                    size_t n = 0;
                    rb.get_markers_greedy_seeding(queries[i], 10, 1000, [&](RB::range_t, std::pair<size_t, size_t>, std::vector<MarkerT>) { ++n; });
                    seeds_seen += n;
                }
            }
        });
    for (auto &th : pool) th.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    rbg_combine_stats(rb.handle(), st1);
    const uint64_t launches = st1[0] - st0[0], requests = st1[1] - st0[1];
    std::printf("%u threads, %zu queries x (find_range + find_range_w_toehold) + %zu marker seedings: %.3f s = %.0f calls/s; "
                "%llu one-read calls in %llu launches (%.1f per launch); seeds %zu; mismatches %zu\n",
                T, Q, (Q + 3) / 4, secs, static_cast<double>(requests ? requests : 2 * Q + (Q + 3) / 4) / secs,
                static_cast<unsigned long long>(requests), static_cast<unsigned long long>(launches),
                launches ? static_cast<double>(requests) / static_cast<double>(launches) : 0.0, seeds_seen.load(), bad.load());
    if (bad.load()) return 1;
    std::printf("shim threads ok\n");
    return 0;
}
