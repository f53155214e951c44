// rowbowt_gpu.hpp -- source-compatible stand-in for the reference's query API on top of the
// C-ABI (include/rbg.h).  A caller written against
//     rbwt::load_rowbowt<StringT>(prefix, flag)            include/rowbowt_io.hpp:176-189
//     rbwt::RowBowt<StringT>::find_range / count / find_range_w_toehold / locs_at /
//         markers_at / find_range_w_markers / resolve_offset / LF / full_range / get_f
//                                                          include/rowbowt.hpp
// compiles against this header unchanged (swap the two #includes) and gets bit-identical
// answers, computed on an MI355X.  Per-read calls cost a kernel launch each, so the class also
// offers *_batch forms; rb_align (rowbowt_amd/csrc/rb_align.cpp) uses those.
//
// Error behaviour follows the reference: a missing file prints to stderr and exit(1)s
// (rowbowt_io.hpp:166-169); a query on a structure that was not loaded returns the default value
// (rowbowt.hpp:171, :273, :294-297); a failed search is in-band {1,0}.  Anything the reference
// would not survive (no GPU, HIP failure) also goes to stderr + exit(1).
//
// Also offered (next-row f4): find_locs_greedy_seeding, get_markers_greedy_seeding (with and without
// a loaded ftab).  Not implemented here: lmem and overlap seeding.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rbg.h"

// pfbwt-f marker_array.hpp helpers used at rb_align.cpp:142 (bit layout: SURVEY.md 8b-format)
using MarkerT = uint64_t;
inline uint64_t get_pos(MarkerT m) { return m & ((uint64_t(1) << 48) - 1); }
inline uint64_t get_seq(MarkerT m) { return (m >> 48) & 0xFFF; }
inline uint8_t get_allele(MarkerT m) { return static_cast<uint8_t>((m >> 60) & 0xF); }

namespace ri {
struct rle_string_sd {};  // tag only: keeps `RowBowt<ri::rle_string_sd>` spelling valid
}

namespace rbwt {

using rle_string_t = ri::rle_string_sd;

// rowbowt_io.hpp:146-158
enum class LoadRbwtFlag { NONE = 0, SA = 1, MA = 2, DL = 4, FT = 8 };
inline constexpr LoadRbwtFlag operator|(LoadRbwtFlag a, LoadRbwtFlag b) {
    return static_cast<LoadRbwtFlag>(static_cast<int>(a) | static_cast<int>(b));
}
inline constexpr LoadRbwtFlag operator&(LoadRbwtFlag a, LoadRbwtFlag b) {
    return static_cast<LoadRbwtFlag>(static_cast<int>(a) & static_cast<int>(b));
}

namespace detail {
[[noreturn]] inline void die(const char *what, int rc) {
    std::cerr << what << ": " << rbg_strerror(rc) << std::endl;
    std::exit(1);
}
inline void check(int rc, const char *what) {
    if (rc != RBG_OK) die(what, rc);
}
struct Batch {  // reads -> the C-ABI's (seqs, off) layout
    std::string seqs;
    std::vector<uint64_t> off{0};
    void add(const std::string &s) { seqs += s; off.push_back(seqs.size()); }
    const uint8_t *data() const { return reinterpret_cast<const uint8_t *>(seqs.data()); }
    uint64_t size() const { return off.size() - 1; }
};
struct LibBuf {  // library-malloc'ed ragged output
    uint64_t *p = nullptr;
    ~LibBuf() { rbg_free_buffer(p); }
};
}  // namespace detail

template <typename RLEString = rle_string_t>
class RowBowt {
   public:
    using range_t = std::pair<uint64_t, uint64_t>;

    // rowbowt.hpp:133-165
    struct LFData {
        LFData() {}
        LFData(range_t r, uint64_t s, uint64_t e, uint64_t ss) : rn(r), qstart(s), qend(e), ssamp(ss) {}
        void clear() { rn = {1, 0}; qstart = 0; qend = 0; ssamp = 0; markers.clear(); }
        range_t rn = {1, 0};
        uint64_t qstart = 0;
        uint64_t qend = 0;
        uint64_t ssamp = 0;
        std::vector<MarkerT> markers;
    };

    RowBowt() {}
    explicit RowBowt(rbg_index *ix) : ix_(ix, rbg_free) {
        rbg_info_t info;
        detail::check(rbg_info(ix, &info), "rbg_info");
        n_ = info.n;
        has_tsa_ = info.has_tsa;
        has_ma_ = info.has_markers;
        f_.resize(256);
        detail::check(rbg_get_f(ix, f_.data()), "rbg_get_f");
    }

    rbg_index *handle() const { return ix_.get(); }

    // rowbowt.hpp:115-118
    range_t full_range() const { return {0, n_ - 1}; }

    // rowbowt.hpp:74-88
    range_t LF(range_t rn, uint8_t c) const {
        range_t out;
        detail::check(rbg_lf(ix_.get(), &rn.first, &rn.second, &c, 1, &out.first, &out.second), "rbg_lf");
        return out;
    }
    range_t LF(uint64_t s, uint64_t e, uint8_t c) const { return LF(range_t(s, e), c); }

    // rowbowt.hpp:121-131
    range_t find_range(const std::string &query) const {
        const uint64_t off[2] = {0, query.size()};
        range_t r;
        detail::check(rbg_find_range(ix_.get(), reinterpret_cast<const uint8_t *>(query.data()), off, 1, &r.first, &r.second),
                      "rbg_find_range");
        return r;
    }

    // rowbowt.hpp:266-269
    uint64_t count(const std::string &query) const {
        auto rn = find_range(query);
        return rn.second >= rn.first ? (rn.second - rn.first) + 1 : 0;
    }

    // rowbowt.hpp:169-184
    LFData find_range_w_toehold(const std::string &query) const {
        LFData lf;
        if (!has_tsa_) return lf;  // :171
        const uint64_t off[2] = {0, query.size()};
        detail::check(rbg_find_range_w_toehold(ix_.get(), reinterpret_cast<const uint8_t *>(query.data()), off, 1,
                                               &lf.rn.first, &lf.rn.second, &lf.ssamp), "rbg_find_range_w_toehold");
        return lf;
    }

    // rowbowt.hpp:613-621 (appends to locs, like ToeholdSA::locate_range toehold_sa.hpp:37-49)
    std::vector<uint64_t> &locs_at(range_t range, uint64_t k, uint64_t max_hits, std::vector<uint64_t> &locs) const {
        uint64_t off[2];
        detail::LibBuf buf;
        detail::check(rbg_locs_at(ix_.get(), &range.first, &range.second, &k, 1, max_hits, off, &buf.p), "rbg_locs_at");
        locs.insert(locs.end(), buf.p, buf.p + off[1]);
        return locs;
    }
    std::vector<uint64_t> locs_at(range_t range, uint64_t k, uint64_t max_hits) const {
        std::vector<uint64_t> locs;
        locs_at(range, k, max_hits, locs);
        return locs;
    }

    // rowbowt.hpp:623-625
    std::pair<std::string, uint64_t> resolve_offset(uint64_t i) const {
        const char *name = nullptr;
        uint64_t off = 0;
        detail::check(rbg_resolve_offset(ix_.get(), i, &name, &off), "rbg_resolve_offset");
        return std::make_pair(std::string(name), off);
    }

    // rowbowt.hpp:272-290 ("WARNING: does not clear markers!")
    std::vector<MarkerT> &markers_at(range_t r, std::vector<MarkerT> &markers) const {
        if (!has_ma_) return markers;  // :283
        uint64_t off[2];
        detail::LibBuf buf;
        detail::check(rbg_markers_at(ix_.get(), &r.first, &r.second, 1, off, &buf.p), "rbg_markers_at");
        markers.insert(markers.end(), buf.p, buf.p + off[1]);
        return markers;
    }
    std::vector<MarkerT> markers_at(range_t r) const {
        std::vector<MarkerT> markers;
        return markers_at(r, markers);
    }
    std::vector<MarkerT> &markers_at(uint64_t i, std::vector<MarkerT> &markers) const { return markers_at(range_t(i, i), markers); }
    std::vector<MarkerT> markers_at(uint64_t i) const { return markers_at(range_t(i, i)); }

    // rowbowt.hpp:292-339
    LFData find_range_w_markers(const std::string query, uint64_t wsize, uint64_t max_range) const {
        LFData lf;
        if (!has_ma_) {
            std::cerr << "warning: no marker array found!\n";  // :295
            return lf;
        }
        if (query.size() < wsize) {
            std::cerr << "warning: query (size=" << query.size() << ") is less than wsize (" << wsize << ")\n";  // :300
            return lf;
        }
        const uint64_t off[2] = {0, query.size()};
        uint64_t mk_off[2];
        detail::LibBuf buf;
        detail::check(rbg_find_range_w_markers(ix_.get(), reinterpret_cast<const uint8_t *>(query.data()), off, 1, wsize,
                                               max_range, &lf.rn.first, &lf.rn.second, mk_off, &buf.p),
                      "rbg_find_range_w_markers");
        lf.markers.assign(buf.p, buf.p + mk_off[1]);
        if (lf.rn.second >= lf.rn.first) { lf.qstart = 0; lf.qend = query.size(); }  // :336-337
        return lf;
    }

    // rowbowt.hpp:406-482: fn(range, (q.first, q.second), mbuf) once per seed, right to left, exactly as the
    // reference calls it (q.second = q.first - 1, wrapped, for an empty seed); goes through the ftab when
    // one was loaded with LoadRbwtFlag::FT and not disabled (:430)
    template <typename F>
    void get_markers_greedy_seeding(const std::string query, uint64_t wsize, uint64_t max_range, F fn) const {
        const uint64_t ft_k = disable_ft_ ? 0 : ft_k_;
        if (ft_k_ && ft_k_ - 1 > wsize) {  // :423-426
            std::cerr << "ERROR: wsize cannot be greater than or equal to ftab k size. please rebuild ftab with smaller k\n";
            std::exit(1);
        }
        if (ft_k && query.size() < ft_k) {  // std::string::substr(pos > size()) throws in the reference (:431)
            std::cerr << "rowbowt_gpu: query shorter than the ftab k-mer size" << std::endl;
            std::exit(1);
        }
        const uint64_t off[2] = {0, query.size()};
        uint64_t seed_off[2];
        rbg_marker_seed_t *seeds = nullptr;
        detail::LibBuf mk;
        detail::check(rbg_get_markers_greedy_seeding(ix_.get(), reinterpret_cast<const uint8_t *>(query.data()), off, 1, wsize,
                                                     max_range, ft_k, seed_off, &seeds, &mk.p), "rbg_get_markers_greedy_seeding");
        std::unique_ptr<rbg_marker_seed_t, void (*)(void *)> hold(seeds, rbg_free_buffer);
        for (uint64_t s = 0; s < seed_off[1]; ++s) {
            const rbg_marker_seed_t &d = seeds[s];
            fn(range_t(d.lo, d.hi), std::make_pair(static_cast<size_t>(d.qstart), static_cast<size_t>(d.qend - 1)),
               std::vector<MarkerT>(mk.p + d.mk_begin, mk.p + d.mk_end));
        }
    }

    // rowbowt.hpp:633-662 (get_seeds_greedy_w_sample :222-256 + locate_from_longest_seed :664-685)
    std::vector<uint64_t> &find_locs_greedy_seeding(std::string s, uint64_t min_length, uint64_t max_hits,
                                                    std::vector<uint64_t> &locs) const {
        locs.clear();
        if (!has_tsa_) return locs;
        const uint64_t off[2] = {0, s.size()};
        uint64_t loc_off[2];
        detail::LibBuf buf;
        detail::check(rbg_find_locs_greedy_seeding(ix_.get(), reinterpret_cast<const uint8_t *>(s.data()), off, 1, min_length,
                                                   max_hits, loc_off, &buf.p), "rbg_find_locs_greedy_seeding");
        locs.assign(buf.p, buf.p + loc_off[1]);
        return locs;
    }
    std::vector<uint64_t> find_locs_greedy_seeding(std::string s, uint64_t min_length, uint64_t max_hits) const {
        std::vector<uint64_t> locs;
        return find_locs_greedy_seeding(s, min_length, max_hits, locs);
    }
    // the seed locate_from_longest_seed would pick among get_seeds_greedy_w_sample(query, min_length)
    LFData longest_greedy_seed(const std::string &query, uint64_t min_length) const {
        LFData lf;
        if (!has_tsa_) return lf;
        const uint64_t off[2] = {0, query.size()};
        detail::check(rbg_greedy_longest_seed(ix_.get(), reinterpret_cast<const uint8_t *>(query.data()), off, 1, min_length,
                                              &lf.rn.first, &lf.rn.second, &lf.qstart, &lf.qend, &lf.ssamp), "rbg_greedy_longest_seed");
        return lf;
    }

    const std::vector<uint64_t> &get_f() const { return f_; }  // rowbowt.hpp:719

    // ftab (LoadRbwtFlag::FT): find_range is result-neutral under it (rowbowt.hpp:124-125); the marker-seed
    // variant is not.  load_ftab verifies the file is build_ftab(k) of this index and keeps k (ftab.hpp:15-27).
    void load_ftab(const std::string &fname) {
        const int rc = rbg_check_ftab(ix_.get(), fname.c_str(), &ft_k_);
        if (rc == RBG_EIO) { std::cerr << "bad file" << std::endl; std::exit(1); }  // rowbowt_io.hpp:167
        detail::check(rc, "load_ftab (the .ftab must be the one rb_build -f made for this index)");
    }
    uint64_t ftab_k() const { return ft_k_; }
    void disable_ft() { disable_ft_ = true; }   // rowbowt.hpp:760-766
    void enable_ft() { disable_ft_ = false; }

    // ---- batched forms (one launch for all reads) ------------------------------------------------
    void find_range_batch(const std::vector<std::string> &queries, std::vector<range_t> &out) const {
        detail::Batch b;
        for (const auto &q : queries) b.add(q);
        std::vector<uint64_t> lo(b.size()), hi(b.size());
        detail::check(rbg_find_range(ix_.get(), b.data(), b.off.data(), b.size(), lo.data(), hi.data()), "rbg_find_range");
        out.resize(b.size());
        for (uint64_t i = 0; i < b.size(); ++i) out[i] = {lo[i], hi[i]};
    }
    void find_range_w_toehold_batch(const std::vector<std::string> &queries, std::vector<LFData> &out) const {
        out.assign(queries.size(), LFData());
        if (!has_tsa_) return;
        detail::Batch b;
        for (const auto &q : queries) b.add(q);
        std::vector<uint64_t> lo(b.size()), hi(b.size()), k(b.size());
        detail::check(rbg_find_range_w_toehold(ix_.get(), b.data(), b.off.data(), b.size(), lo.data(), hi.data(), k.data()),
                      "rbg_find_range_w_toehold");
        for (uint64_t i = 0; i < b.size(); ++i) { out[i].rn = {lo[i], hi[i]}; out[i].ssamp = k[i]; }
    }
    // locs_at for many (range, toehold) triples: loc_off[N+1] + concatenated locations
    void locs_at_batch(const std::vector<LFData> &lfs, uint64_t max_hits, std::vector<uint64_t> &loc_off,
                       std::vector<uint64_t> &locs) const {
        const uint64_t N = lfs.size();
        std::vector<uint64_t> lo(N), hi(N), k(N);
        for (uint64_t i = 0; i < N; ++i) { lo[i] = lfs[i].rn.first; hi[i] = lfs[i].rn.second; k[i] = lfs[i].ssamp; }
        loc_off.assign(N + 1, 0);
        detail::LibBuf buf;
        detail::check(rbg_locs_at(ix_.get(), lo.data(), hi.data(), k.data(), N, max_hits, loc_off.data(), &buf.p), "rbg_locs_at");
        locs.assign(buf.p, buf.p + loc_off[N]);
    }
    void markers_at_batch(const std::vector<range_t> &ranges, std::vector<uint64_t> &mk_off, std::vector<MarkerT> &mk) const {
        const uint64_t N = ranges.size();
        mk_off.assign(N + 1, 0);
        mk.clear();
        if (!has_ma_) return;
        std::vector<uint64_t> lo(N), hi(N);
        for (uint64_t i = 0; i < N; ++i) { lo[i] = ranges[i].first; hi[i] = ranges[i].second; }
        detail::LibBuf buf;
        detail::check(rbg_markers_at(ix_.get(), lo.data(), hi.data(), N, mk_off.data(), &buf.p), "rbg_markers_at");
        mk.assign(buf.p, buf.p + mk_off[N]);
    }

   private:
    std::shared_ptr<rbg_index> ix_;
    uint64_t n_ = 0;
    bool has_tsa_ = false, has_ma_ = false, disable_ft_ = false;
    uint64_t ft_k_ = 0;
    std::vector<uint64_t> f_;
};

// rowbowt_io.hpp:176-189.  `device` is the only addition (defaults to HIP device 0).
template <typename StringT = rle_string_t>
RowBowt<StringT> load_rowbowt(std::string prefix, LoadRbwtFlag flag, int device = 0) {
    std::cerr << "loading type: rbg flat index (MI355X)" << std::endl;  // cf. rowbowt_io.hpp:180
    rbg_index *ix = nullptr;
    const int rc = rbg_load(prefix.c_str(), static_cast<int>(flag), device, &ix);
    if (rc == RBG_EIO) {
        std::cerr << "bad file" << std::endl;  // rowbowt_io.hpp:167
        std::exit(1);
    }
    detail::check(rc, "load_rowbowt");
    RowBowt<StringT> rb(ix);
    if (static_cast<int>(flag & LoadRbwtFlag::FT)) rb.load_ftab(prefix + ".ftab");  // rowbowt_io.hpp:21,187
    return rb;
}

}  // namespace rbwt
