// Exercises rowbowt_gpu.hpp exactly the way the reference's tests/rb_tests.cpp exercises
// rowbowt.hpp: same calls, same expected values (reference rb_tests.cpp:47-58, :115-120, :131-140,
// :147-173, :83-95).  argv[1] = directory holding small.fa.* and *_query.fq, argv[2] = a prefix
// that additionally has a .docs file.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "rowbowt_gpu.hpp"

#define EXPECT_EQ(a, b)                                                                        \
    do {                                                                                       \
        if (!((a) == (b))) { std::fprintf(stderr, "FAIL %s:%d: %s != %s\n", __FILE__, __LINE__, #a, #b); std::exit(2); } \
    } while (0)

static std::vector<std::string> read_fasta(const std::string &fn) {
    std::ifstream ifs(fn);
    std::vector<std::string> out;
    std::string line;
    while (std::getline(ifs, line))
        if (!line.empty() && line[0] != '>') out.push_back(line);
    return out;
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    const std::string dir = argv[1];
    using RB = rbwt::RowBowt<ri::rle_string_sd>;
    using range_t = RB::range_t;
    RB rb = rbwt::load_rowbowt<ri::rle_string_sd>(dir + "/small.fa", rbwt::LoadRbwtFlag::SA | rbwt::LoadRbwtFlag::MA);
    const auto simple = read_fasta(dir + "/simple_query.fq");
    const auto errq = read_fasta(dir + "/error_query.fq");
    EXPECT_EQ(simple.size(), 6u);
    // CountTester, rb_tests.cpp:104-121
    const range_t want[6] = {{24279, 24280}, {24175, 24175}, {27430, 27432}, {27430, 27432}, {17409, 17409}, {17416, 17417}};
    for (int i = 0; i < 6; ++i) {
        EXPECT_EQ(rb.find_range(simple[i]), want[i]);
        EXPECT_EQ(rb.count(simple[i]), want[i].second - want[i].first + 1);
    }
    // LocateTester, rb_tests.cpp:29-59
    std::vector<uint64_t> locs, all_locs;
    for (const auto &q : simple) {
        auto ret = rb.find_range_w_toehold(q);
        locs.clear();
        rb.locs_at(ret.rn, ret.ssamp, static_cast<uint64_t>(-1), locs);
        all_locs.insert(all_locs.end(), locs.begin(), locs.end());
    }
    const uint64_t gold[12] = {20306, 286, 10296, 11897, 21907, 1887, 11897, 21907, 1887, 4644, 14654, 24664};
    EXPECT_EQ(all_locs.size(), 12u);
    for (int i = 0; i < 12; ++i) EXPECT_EQ(all_locs[i], gold[i]);
    // locs_at appends (toehold_sa.hpp:42-45)
    auto ret0 = rb.find_range_w_toehold(simple[0]);
    std::vector<uint64_t> app = {7};
    rb.locs_at(ret0.rn, ret0.ssamp, static_cast<uint64_t>(-1), app);
    EXPECT_EQ(app.size(), 3u);
    EXPECT_EQ(app[0], 7u);
    EXPECT_EQ(rb.locs_at(ret0.rn, ret0.ssamp, 1).size(), 1u);
    // failed search: {1,0}, ssamp 0 (rowbowt.hpp:153-159)
    auto bad = rb.find_range_w_toehold(errq[0]);
    EXPECT_EQ(bad.rn, range_t(1, 0));
    EXPECT_EQ(bad.ssamp, 0u);
    EXPECT_EQ(rb.count(errq[0]), 0u);
    // MarkerTester, rb_tests.cpp:123-141
    const int mpos[6] = {289, 289, -1, -1, 4650, 4650}, mall[6] = {0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 6; ++i) {
        auto lf = rb.find_range_w_markers(simple[i], 10, -1);
        if (mpos[i] < 0) { EXPECT_EQ(lf.markers.size(), 0u); continue; }
        EXPECT_EQ(get_pos(lf.markers[0]), static_cast<uint64_t>(mpos[i]));
        EXPECT_EQ(static_cast<int>(get_allele(lf.markers[0])), mall[i]);
        EXPECT_EQ(lf.rn, want[i]);
    }
    // markers_at appends and works on a single index (rowbowt.hpp:272-290)
    std::vector<MarkerT> mk = {99};
    rb.markers_at(want[0], mk);
    EXPECT_EQ(mk[0], 99u);
    EXPECT_EQ(mk.size() >= 2, true);
    EXPECT_EQ(rb.markers_at(want[0].first).size() >= 1, true);
    // FTab tests' answers (rb_tests.cpp:147-173): ftab is result-neutral
    EXPECT_EQ(rb.find_range("TTCGTCGTAA"), range_t(28942, 28944));
    EXPECT_EQ(rb.find_range("GTATCGTGGAA"), range_t(21142, 21144));
    EXPECT_EQ(rb.find_range("TGGAGATATTG"), range_t(27180, 27182));
    // LF chaining == find_range (rowbowt.hpp:127-129); full_range :115-118; get_f :719
    range_t r = rb.full_range();
    EXPECT_EQ(r, range_t(0, 30030));
    for (auto it = simple[0].rbegin(); it != simple[0].rend(); ++it) r = rb.LF(r, static_cast<uint8_t>(*it));
    EXPECT_EQ(r, want[0]);
    EXPECT_EQ(rb.get_f()['C'], 7650u);
    // GreedyLocateTester, rb_tests.cpp:67-96
    const std::vector<std::vector<uint64_t>> g = {{10296, 20306, 286}, {10296}, {11897, 21907, 1887}, {11897, 21907, 1887}, {}, {14654, 4644}};
    for (int i = 0; i < 6; ++i) {
        auto l = rb.find_locs_greedy_seeding(errq[i], 10, static_cast<uint64_t>(-1));
        if (g[i].empty()) { EXPECT_EQ(l.size(), 0u); continue; }
        for (size_t t = 0; t < g[i].size(); ++t) EXPECT_EQ(l[t], g[i][t]);
    }
    // get_markers_greedy_seeding with the callback rb_markers passes (rowbowt.hpp:406-482): values from the oracle
    {
        std::vector<range_t> ranges;
        std::vector<std::pair<size_t, size_t>> qs;
        size_t nmk = 0;
        auto fn = [&](range_t p, std::pair<size_t, size_t> q, std::vector<MarkerT> mbuf) { ranges.push_back(p); qs.push_back(q); nmk += mbuf.size(); };
        rb.get_markers_greedy_seeding(simple[0], 19, 1000, fn);
        EXPECT_EQ(ranges.size(), 1u);
        EXPECT_EQ(ranges[0], want[0]);
        EXPECT_EQ(qs[0], (std::pair<size_t, size_t>(0, 19)));
        EXPECT_EQ(nmk, 2u);
        ranges.clear(); qs.clear();
        rb.get_markers_greedy_seeding(errq[0], 19, 1000, fn);
        EXPECT_EQ(ranges.size(), 2u);
        EXPECT_EQ(ranges[0], range_t(10661, 10663));
        EXPECT_EQ(qs[0], (std::pair<size_t, size_t>(5, 19)));
        EXPECT_EQ(ranges[1], range_t(24205, 24294));
        EXPECT_EQ(qs[1], (std::pair<size_t, size_t>(0, 3)));
    }
    // LoadRbwtFlag::FT (rowbowt_io.hpp:187): the .ftab written for this index is accepted, its k drives the ftab
    // variant of the seeding (rowbowt.hpp:430-433), disable_ft()/enable_ft() (:760-766) switch it
    {
        const std::string ft = std::string(argv[2]) + ".ftab";
        EXPECT_EQ(rbg_write_ftab(rb.handle(), 6, ft.c_str()), 0);
        rb.load_ftab(ft);
        EXPECT_EQ(rb.ftab_k(), 6u);
        auto collect = [&](const std::string &q, uint64_t w) {
            std::vector<std::pair<range_t, std::pair<size_t, size_t>>> v;
            rb.get_markers_greedy_seeding(q, w, 1000, [&](range_t p, std::pair<size_t, size_t> qq, std::vector<MarkerT>) { v.emplace_back(p, qq); });
            return v;
        };
        auto direct = [&](const std::string &q, uint64_t w, uint64_t k) {
            const uint64_t off[2] = {0, q.size()};
            uint64_t so[2], *mk = nullptr;
            rbg_marker_seed_t *sd = nullptr;
            EXPECT_EQ(rbg_get_markers_greedy_seeding(rb.handle(), reinterpret_cast<const uint8_t *>(q.data()), off, 1, w, 1000, k, so, &sd, &mk), 0);
            std::vector<std::pair<range_t, std::pair<size_t, size_t>>> v;
            for (uint64_t s = 0; s < so[1]; ++s) v.emplace_back(range_t(sd[s].lo, sd[s].hi), std::make_pair(size_t(sd[s].qstart), size_t(sd[s].qend - 1)));
            rbg_free_buffer(sd);
            rbg_free_buffer(mk);
            return v;
        };
        for (int i = 0; i < 6; ++i) {
            EXPECT_EQ(collect(errq[i], 8) == direct(errq[i], 8, 6), true);
            rb.disable_ft();
            EXPECT_EQ(collect(errq[i], 8) == direct(errq[i], 8, 0), true);
            rb.enable_ft();
        }
        RB rbf = rbwt::load_rowbowt<>(argv[2], rbwt::LoadRbwtFlag::SA | rbwt::LoadRbwtFlag::FT);
        EXPECT_EQ(rbf.ftab_k(), 6u);
    }
    // resolve_offset through a .docs file (rowbowt.hpp:623-625)
    RB rb2 = rbwt::load_rowbowt<>(argv[2], rbwt::LoadRbwtFlag::SA | rbwt::LoadRbwtFlag::DL);
    auto x = rb2.resolve_offset(20306);
    EXPECT_EQ(x.first, std::string("hap2"));
    EXPECT_EQ(x.second, 286u);
    std::printf("shim goldens ok\n");
    return 0;
}
