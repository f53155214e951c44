// rbg_host.cpp -- readers for the reference's sdsl-serialised index files + the flattener.
// See rbg_host.hpp for the reference file:line each piece replaces.
#include "rbg_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <queue>
#include <sstream>
#include <thread>
#include <type_traits>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/rbg.h"
#include "rbg_thread_team.hpp"

namespace rbg {

unsigned load_threads() {
    static const unsigned n = [] {
        unsigned t = rbg_hostpath::cpu_budget();
        if (const char *e = std::getenv("RBG_LOAD_THREADS")) { const int v = std::atoi(e); if (v > 0) t = static_cast<unsigned>(v); }
        return std::max(1u, std::min(t, 64u));
    }();
    return n;
}

void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)> &fn, uint64_t min_per_thread) {
    unsigned T = load_threads();
    if (min_per_thread == 0) min_per_thread = 1;
    if (n / min_per_thread < T) T = static_cast<unsigned>(std::max<uint64_t>(1, n / min_per_thread));
    if (T <= 1) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T, t); });
    fn(0, n / T, 0);
    for (auto &w : th) w.join();
}

namespace {

// ---- byte cursor over a whole file --------------------------------------------------------------
class Cursor {
   public:
    bool open(const std::string &fname) {
        std::ifstream ifs(fname, std::ios::binary | std::ios::ate);
        if (!ifs.good()) return false;
        std::streamsize sz = ifs.tellg();
        ifs.seekg(0);
        buf_.resize(static_cast<size_t>(sz));
        if (sz > 0 && !ifs.read(reinterpret_cast<char *>(buf_.data()), sz)) return false;
        return true;
    }
    template <typename T>
    T get() {
        T v{};
        if (!take(sizeof(T))) return v;
        std::memcpy(&v, buf_.data() + pos_ - sizeof(T), sizeof(T));
        return v;
    }
    const uint8_t *span(size_t nbytes) { return take(nbytes) ? buf_.data() + pos_ - nbytes : nullptr; }
    bool ok() const { return ok_; }
    bool at_end() const { return ok_ && pos_ == buf_.size(); }
    size_t remaining() const { return buf_.size() - pos_; }
    void fail() { ok_ = false; }

   private:
    bool take(size_t n) {
        if (!ok_ || n > buf_.size() - pos_) { ok_ = false; return false; }
        pos_ += n;
        return true;
    }
    std::vector<uint8_t> buf_;
    size_t pos_ = 0;
    bool ok_ = true;
};

// sdsl::int_vector<w> as the reference's files hold it: one u64 header whose low 56 bits are the
// length in bits and whose top byte is the element width, then ceil(bits/64) words.
struct PackedVec {
    uint64_t bits = 0;
    unsigned width = 1;
    std::vector<uint64_t> words;
    uint64_t size() const { return bits / width; }
    uint64_t at(uint64_t i) const {
        const uint64_t b = i * width;
        const unsigned sh = b & 63;
        uint64_t x = words[b >> 6] >> sh;
        if (sh + width > 64) x |= words[(b >> 6) + 1] << (64 - sh);
        return width == 64 ? x : x & ((uint64_t(1) << width) - 1);
    }
    bool bit(uint64_t i) const { return (words[i >> 6] >> (i & 63)) & 1; }
};

bool read_packed(Cursor &c, PackedVec &v, bool keep) {
    const uint64_t h = c.get<uint64_t>();
    v.bits = h & ((uint64_t(1) << 56) - 1);
    v.width = static_cast<unsigned>(h >> 56);
    if (!c.ok() || v.width == 0 || v.width > 64) { c.fail(); return false; }
    const uint64_t nwords = (v.bits + 63) / 64;
    if (nwords > c.remaining() / 8) { c.fail(); return false; }
    const uint8_t *p = c.span(nwords * 8);
    if (!p) return false;
    if (keep) {
        v.words.resize(nwords + 1);
        std::memcpy(v.words.data(), p, nwords * 8);
        v.words[nwords] = 0;
    }
    return true;
}

// sdsl::select_support_mcl<b>: arg_cnt, then (if non-zero) superblock vector, mini_or_long bit
// vector and one vector per 4096-argument superblock.  Nothing in it is needed.
void skip_select_support(Cursor &c) {
    const uint64_t arg_cnt = c.get<uint64_t>();
    if (!c.ok() || arg_cnt == 0) return;
    PackedVec tmp;
    read_packed(c, tmp, false);
    read_packed(c, tmp, false);
    const uint64_t sb = (arg_cnt + 4095) >> 12;
    for (uint64_t i = 0; i < sb && c.ok(); ++i) read_packed(c, tmp, false);
}

// sdsl::sd_vector<> (Elias-Fano): size, wl, low, high, select_1, select_0.  Decoded to the
// ascending list of set positions.
bool read_sd_vector(Cursor &c, uint64_t &universe, std::vector<uint64_t> &ones) {
    universe = c.get<uint64_t>();
    const unsigned wl = c.get<uint8_t>();
    PackedVec low, high;
    if (!read_packed(c, low, true) || !read_packed(c, high, true)) return false;
    skip_select_support(c);
    skip_select_support(c);
    if (!c.ok() || wl >= 64) { c.fail(); return false; }
    const uint64_t m = low.size();
    ones.clear();
    ones.reserve(m);
    const uint64_t nwords = (high.bits + 63) / 64;
    for (uint64_t w = 0; w < nwords && ones.size() < m; ++w) {
        uint64_t x = high.words[w];
        if (w == nwords - 1 && (high.bits & 63)) x &= (uint64_t(1) << (high.bits & 63)) - 1;
        while (x && ones.size() < m) {
            const uint64_t p = w * 64 + static_cast<unsigned>(__builtin_ctzll(x));
            const uint64_t k = ones.size();
            ones.push_back(((p - k) << wl) | (wl ? low.at(k) : 0));
            x &= x - 1;
        }
    }
    if (ones.size() != m) { c.fail(); return false; }
    for (uint64_t k = 1; k < m; ++k)
        if (ones[k] <= ones[k - 1]) { c.fail(); return false; }
    if (m && ones.back() >= universe) { c.fail(); return false; }
    return true;
}

// ri::sparse_sd_vector::load (sparse_sd_vector.hpp:194-200): u, then the sd_vector unless u == 0.
bool read_sparse(Cursor &c, uint64_t &u, std::vector<uint64_t> &ones) {
    u = c.get<uint64_t>();
    ones.clear();
    if (!c.ok()) return false;
    if (u == 0) return true;
    uint64_t inner = 0;
    if (!read_sd_vector(c, inner, ones)) return false;
    if (inner != u) { c.fail(); return false; }
    return true;
}

// sdsl::wt_huff<> -> the plain symbol sequence.  Each element is recovered with the wavelet
// tree's own access walk (bit at the node, rank inside the node, descend) over a popcount
// directory built here; the serialised rank/select supports are skipped.
// `expect` = the length the caller already knows from data it decoded (the number of run-length entries):
// the serialised size is never trusted for an allocation or a decode loop.
bool read_wt_huff(Cursor &c, uint64_t expect, std::vector<uint8_t> &seq) {
    const uint64_t size = c.get<uint64_t>();
    if (!c.ok() || size != expect) { c.fail(); return false; }
    (void)c.get<uint64_t>();  // sigma
    PackedVec bv, tmp;
    if (!read_packed(c, bv, true)) return false;
    read_packed(c, tmp, false);  // rank_support_v
    skip_select_support(c);
    skip_select_support(c);
    const uint64_t n_nodes = c.get<uint64_t>();
    if (!c.ok() || n_nodes == 0 || n_nodes > 0xFFFF) { c.fail(); return false; }
    struct Node { uint64_t bv_pos, bv_pos_rank; uint16_t parent, child[2]; };
    std::vector<Node> nodes(n_nodes);
    for (auto &nd : nodes) {  // 22-byte packed records
        nd.bv_pos = c.get<uint64_t>();
        nd.bv_pos_rank = c.get<uint64_t>();
        nd.parent = c.get<uint16_t>();
        nd.child[0] = c.get<uint16_t>();
        nd.child[1] = c.get<uint16_t>();
    }
    uint16_t c_to_leaf[256];
    for (auto &x : c_to_leaf) x = c.get<uint16_t>();
    c.span(256 * 8);  // path[256]
    if (!c.ok()) return false;
    std::vector<int> leaf_symbol(n_nodes, -1);
    for (int s = 0; s < 256; ++s)
        if (c_to_leaf[s] != 0xFFFF) {
            if (c_to_leaf[s] >= n_nodes) { c.fail(); return false; }
            leaf_symbol[c_to_leaf[s]] = s;
        }
    // popcount directory: ones before each 64-bit word
    const uint64_t nwords = (bv.bits + 63) / 64;
    std::vector<uint64_t> before(nwords + 1, 0);
    for (uint64_t w = 0; w < nwords; ++w) before[w + 1] = before[w] + __builtin_popcountll(bv.words[w]);
    auto rank1 = [&](uint64_t p) {  // ones in bv[0,p)
        const uint64_t w = p >> 6;
        const unsigned b = p & 63;
        return before[w] + (b ? __builtin_popcountll(bv.words[w] & ((uint64_t(1) << b) - 1)) : 0);
    };
    seq.resize(size);
    for (uint64_t i = 0; i < size; ++i) {
        uint32_t v = 0;
        uint64_t pos = i;
        unsigned depth = 0;
        while (leaf_symbol[v] < 0) {
            const uint64_t p = nodes[v].bv_pos + pos;
            if (p >= bv.bits || ++depth > 256) { c.fail(); return false; }
            const uint64_t ones_in_node = rank1(p) - nodes[v].bv_pos_rank;
            const int b = bv.bit(p);
            pos = b ? ones_in_node : pos - ones_in_node;
            v = nodes[v].child[b];
            if (v >= n_nodes) { c.fail(); return false; }
        }
        seq[i] = static_cast<uint8_t>(leaf_symbol[v]);
    }
    return true;
}

}  // namespace

// ---- .rbwt ---------------------------------------------------------------------------------------
int parse_rbwt(const std::string &fname, RawRle &out) {
    Cursor c;
    if (!c.open(fname)) return RBG_EIO;
    out = RawRle();
    out.n = c.get<uint64_t>();
    out.R = c.get<uint64_t>();
    out.B = c.get<uint64_t>();
    if (!c.ok() || out.n == 0 || out.R == 0 || out.B == 0 || out.R > out.n) return RBG_EFORMAT;
    uint64_t runs_u = 0;
    std::vector<uint64_t> block_ends;  // `runs`: last position of every B-th run (rle_string.hpp:68,78)
    if (!read_sparse(c, runs_u, block_ends)) return RBG_EFORMAT;
    std::vector<std::vector<uint64_t>> letter_ones(256);
    std::vector<uint64_t> letter_size(256, 0);
    for (int s = 0; s < 256; ++s)
        if (!read_sparse(c, letter_size[s], letter_ones[s])) return RBG_EFORMAT;
    // R as the per-letter vectors (decoded from bytes that are really in the file) give it: a header that
    // claims more cannot make the head decoder allocate or loop beyond what the file holds
    uint64_t letter_runs = 0;
    for (int s = 0; s < 256; ++s) letter_runs += letter_ones[s].size();
    if (letter_runs != out.R) return RBG_EFORMAT;
    if (!read_wt_huff(c, out.R, out.heads)) return RBG_EFORMAT;
    if (!c.at_end() || out.heads.size() != out.R || runs_u != out.n) return RBG_EFORMAT;
    // run lengths: the k-th run of symbol s has length ones[k] - ones[k-1] in s-only space
    // (sparse_sd_vector::gapAt, sparse_sd_vector.hpp:150-154; used by rle_string::run_at :238-242)
    out.lens.resize(out.R);
    std::vector<uint64_t> next(256, 0);
    uint64_t total = 0;
    for (uint64_t i = 0; i < out.R; ++i) {
        const uint8_t s = out.heads[i];
        if (i && s == out.heads[i - 1]) return RBG_EFORMAT;
        const uint64_t k = next[s]++;
        if (k >= letter_ones[s].size()) return RBG_EFORMAT;
        const uint64_t len = k ? letter_ones[s][k] - letter_ones[s][k - 1] : letter_ones[s][0] + 1;
        out.lens[i] = len;
        total += len;
        // cross-check against the block-sampled `runs` vector
        if (i % out.B == out.B - 1 && i != out.R - 1) {
            const uint64_t blk = i / out.B;
            if (blk >= block_ends.size() || block_ends[blk] != total - 1) return RBG_EFORMAT;
        }
    }
    for (int s = 0; s < 256; ++s) {
        if (next[s] != letter_ones[s].size()) return RBG_EFORMAT;
        if (!letter_ones[s].empty() && letter_ones[s].back() + 1 != letter_size[s]) return RBG_EFORMAT;
    }
    if (total != out.n) return RBG_EFORMAT;
    return RBG_OK;
}

// ---- .tsa ----------------------------------------------------------------------------------------
int parse_tsa(const std::string &fname, RawTsa &out) {
    Cursor c;
    if (!c.open(fname)) return RBG_EIO;
    out = RawTsa();
    out.r = c.get<uint64_t>();  // r first, then n (toehold_sa.hpp:86-87)
    out.n = c.get<uint64_t>();
    uint64_t u = 0;
    if (!read_sparse(c, u, out.pred_pos)) return RBG_EFORMAT;
    PackedVec sl, p2r;
    if (!read_packed(c, sl, true) || !read_packed(c, p2r, true)) return RBG_EFORMAT;
    if (!c.at_end() || u != out.n || out.pred_pos.size() != out.r || sl.size() != out.r || p2r.size() != out.r)
        return RBG_EFORMAT;
    out.samples_last.resize(out.r);
    out.pred_to_run.resize(out.r);
    for (uint64_t i = 0; i < out.r; ++i) {
        out.samples_last[i] = sl.at(i);
        out.pred_to_run[i] = p2r.at(i);
        if (out.samples_last[i] >= out.n || out.pred_to_run[i] >= out.r) return RBG_EFORMAT;
    }
    return RBG_OK;
}

// rle_string(std::string fname, uint64_t B), rle_string.hpp:44-97, at byte level: formatted
// extraction (`ifs >> c`) skips the whitespace bytes \t \n \v \f \r and space, every other byte is a
// BWT symbol, byte 0 is stored as 1 (TERMINATOR)
int read_raw_bwt(const std::string &fname, RawRle &out) {
    std::ifstream ifs(fname, std::ios::binary);
    if (!ifs.good()) return RBG_EIO;
    out = RawRle();
    out.B = 2;
    std::vector<char> buf(1 << 20);
    int last = -1;
    uint64_t len = 0;
    while (ifs) {
        ifs.read(buf.data(), static_cast<std::streamsize>(buf.size()));
        const std::streamsize got = ifs.gcount();
        for (std::streamsize i = 0; i < got; ++i) {
            unsigned char ch = static_cast<unsigned char>(buf[i]);
            if (ch == ' ' || (ch >= 9 && ch <= 13)) continue;
            if (ch == 0) ch = 1;
            if (ch == last) { ++len; continue; }
            if (last >= 0) { out.heads.push_back(static_cast<uint8_t>(last)); out.lens.push_back(len); }
            last = ch;
            len = 1;
        }
    }
    if (last < 0) return RBG_EFORMAT;
    out.heads.push_back(static_cast<uint8_t>(last));
    out.lens.push_back(len);
    out.R = out.heads.size();
    out.n = 0;
    for (uint64_t l : out.lens) out.n += l;
    return RBG_OK;
}

// read_run_starts / read_run_ends, toehold_sa.hpp:133-155: (x, y) pairs, x ignored
int read_raw_samples(const std::string &fname, std::vector<uint64_t> &y_out) {
    std::ifstream ifs(fname, std::ios::binary | std::ios::ate);
    if (!ifs.good()) return RBG_EIO;
    const std::streamsize sz = ifs.tellg();
    ifs.seekg(0);
    const uint64_t pairs = static_cast<uint64_t>(sz) / 16;  // a trailing partial pair is dropped, as the reference's loop does
    std::vector<uint64_t> raw(pairs * 2);
    if (pairs && !ifs.read(reinterpret_cast<char *>(raw.data()), static_cast<std::streamsize>(pairs * 16))) return RBG_EIO;
    y_out.resize(pairs);
    for (uint64_t i = 0; i < pairs; ++i) y_out[i] = raw[2 * i + 1];
    return RBG_OK;
}

// toehold_sa.hpp:133-155 (sample = y ? y-1 : n-1) and build_phi :105-131
void tsa_from_samples(uint64_t n, uint64_t r, const uint64_t *ssa_y, const uint64_t *esa_y, RawTsa &out) {
    out = RawTsa();
    out.n = n;
    out.r = r;
    out.samples_last.resize(r);
    std::vector<uint64_t> key(r), order(r);
    parallel_for(r, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) {
            key[i] = ssa_y[i] ? ssa_y[i] - 1 : n - 1;
            out.samples_last[i] = esa_y[i] ? esa_y[i] - 1 : n - 1;
            order[i] = i;
        }
    });
    // order the runs by the text position of their first row: a stable LSD radix sort on (position, run), 11 bits per
    // pass, every pass split over the worker threads (per-thread histograms, then each thread scatters its own chunk
    // to the places the prefix sums give it) -- a comparison sort through the index array took 7 s for the bench index
    // (3.7e7 runs), the serial radix sort 0.4 s there and 9 s at 3.1e8 runs
    {
        int bits = 1;
        while (bits < 64 && (n >> bits)) ++bits;
        constexpr int kDigit = 11;
        constexpr size_t kBuckets = size_t(1) << kDigit;
        std::vector<uint64_t> key2(r), order2(r);
        const unsigned T = std::max(1u, std::min<unsigned>(load_threads(), static_cast<unsigned>(r / 65536 + 1)));
        std::vector<uint64_t> hist(static_cast<size_t>(T) * kBuckets);
        for (int shift = 0; shift < bits; shift += kDigit) {
            auto chunk = [&](unsigned t, uint64_t &b, uint64_t &e) { b = r * t / T; e = r * (t + 1) / T; };
            auto count = [&](unsigned t) {
                uint64_t b, e;
                chunk(t, b, e);
                uint64_t *h = hist.data() + static_cast<size_t>(t) * kBuckets;
                std::fill(h, h + kBuckets, 0);
                for (uint64_t i = b; i < e; ++i) ++h[(key[i] >> shift) & (kBuckets - 1)];
            };
            auto scatter = [&](unsigned t) {
                uint64_t b, e;
                chunk(t, b, e);
                uint64_t *h = hist.data() + static_cast<size_t>(t) * kBuckets;
                for (uint64_t i = b; i < e; ++i) {
                    const uint64_t dst = h[(key[i] >> shift) & (kBuckets - 1)]++;
                    key2[dst] = key[i];
                    order2[dst] = order[i];
                }
            };
            auto on_all = [&](const std::function<void(unsigned)> &f) {
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T; ++t) th.emplace_back(f, t);
                f(0);
                for (auto &w : th) w.join();
            };
            on_all(count);
            uint64_t sum = 0;   // bucket-major, thread-minor: stable
            for (size_t d = 0; d < kBuckets; ++d)
                for (unsigned t = 0; t < T; ++t) {
                    uint64_t &h = hist[static_cast<size_t>(t) * kBuckets + d];
                    const uint64_t c = h;
                    h = sum;
                    sum += c;
                }
            on_all(scatter);
            key.swap(key2);
            order.swap(order2);
        }
    }
    out.pred_pos.swap(key);       // the sorted first-row positions
    out.pred_to_run.swap(order);
}

// ---- .mab ----------------------------------------------------------------------------------------
int parse_mab(const std::string &fname, RawMarkers &out) {
    Cursor c;
    if (!c.open(fname)) return RBG_EIO;
    out = RawMarkers();
    uint64_t u0 = 0, u1 = 0, u2 = 0;
    std::vector<uint64_t> firsts;
    if (!read_sd_vector(c, u0, out.start) || !read_sd_vector(c, u1, out.end) || !read_sd_vector(c, u2, firsts))
        return RBG_EFORMAT;
    const uint64_t count = c.get<uint64_t>();
    if (!c.ok() || count > c.remaining() / 8) return RBG_EFORMAT;
    out.vals.resize(count);
    const uint8_t *p = c.span(count * 8);
    if (count) std::memcpy(out.vals.data(), p, count * 8);
    out.wsize = c.get<int32_t>();
    const uint64_t nruns = out.start.size();
    if (!c.at_end() || out.end.size() != nruns || firsts.size() != nruns) return RBG_EFORMAT;
    out.off.assign(firsts.begin(), firsts.end());
    out.off.push_back(count);
    for (uint64_t j = 0; j < nruns; ++j)
        if (out.end[j] < out.start[j] || out.off[j] > out.off[j + 1]) return RBG_EFORMAT;
    return RBG_OK;
}

// ---- .docs ---------------------------------------------------------------------------------------
int parse_docs(const std::string &fname, RawDocs &out) {
    std::ifstream ifs(fname);
    if (!ifs.good()) return RBG_EIO;
    out = RawDocs();
    std::string name;
    uint64_t pos = 0;
    while (ifs >> name >> pos) {  // doclist.hpp:62
        out.names.push_back(name);
        out.starts.push_back(pos);
    }
    out.sorted = out.starts;
    std::sort(out.sorted.begin(), out.sorted.end());
    return RBG_OK;
}

// ---- native cache file ------------------------------------------------------------------------------
namespace {
// v.resize(count) for the gigabyte arrays of a load (old contents dropped): a vector's zero fill is ONE thread touching
// every page for the first time -- about 1 GB/s, and a load at r = 3e8 makes a dozen 2.5 GB arrays.  Here the worker
// threads touch the reserved storage first (uint64_t and the like: no constructors), so that the fill that follows runs
// over resident pages.
template <typename T>
void resize_parallel(std::vector<T> &v, uint64_t count) {
    static_assert(std::is_trivial<T>::value, "raw storage is written before the elements exist");
    std::vector<T>().swap(v);
    if (count >= (uint64_t(1) << 20)) {
        v.reserve(count);
        T *raw = v.data();
        parallel_for(count, [&](uint64_t b, uint64_t e, unsigned) { std::memset(static_cast<void *>(raw + b), 0, (e - b) * sizeof(T)); }, uint64_t(1) << 18);
    }
    v.resize(count);
}
template <typename T>
void copy_parallel(std::vector<T> &dst, const std::vector<T> &src) {
    resize_parallel(dst, src.size());
    parallel_for(src.size(), [&](uint64_t b, uint64_t e, unsigned) { std::memcpy(dst.data() + b, src.data() + b, (e - b) * sizeof(T)); }, uint64_t(1) << 18);
}
// Version 2 (written since round 3) differs from 1 in the checksum only: one FlatSum per chunk of 2^20 words, and the
// file's last word is the FlatSum over those -- a checksum of checksums that sixteen threads verify in a fraction of a
// second where the single chain of version 1 took 1.5 s per 9 GB.  Version 1 files are still read.
constexpr char kFlatMagic[8] = {'R', 'B', 'G', 'P', 'U', 'I', 'X', '2'};
constexpr char kFlatMagicV1[8] = {'R', 'B', 'G', 'P', 'U', 'I', 'X', '1'};
constexpr uint64_t kSumChunkWords = uint64_t(1) << 20;

struct FlatSum {  // order-sensitive 64-bit checksum over 8-byte words
    uint64_t h = 0x9E3779B97F4A7C15ull;
    void words(const uint64_t *w, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            h ^= w[i];
            h *= 0xFF51AFD7ED558CCDull;
            h = (h << 29) | (h >> 35);
        }
    }
};

class FlatWriter {
   public:
    // (written under a temporary name and renamed when complete: a reader -- another rank of the node, which maps the
    //  file -- never sees a partial one)
    explicit FlatWriter(const std::string &fname)
        : final_(fname), tmp_(fname + ".tmp." + std::to_string(static_cast<long>(getpid()))), fp_(std::fopen(tmp_.c_str(), "wb")) {}
    ~FlatWriter() {
        if (fp_) { std::fclose(fp_); std::remove(tmp_.c_str()); }   // not finished: nothing is left behind
    }
    bool ok() const { return fp_ && ok_; }
    void u64(uint64_t v) { raw(&v, 8); }
    // section of `count` values narrowed to `width` bytes, zero-padded to a multiple of 8
    void section(const uint64_t *v, uint64_t count, unsigned width) {
        if (width == 8) { raw(v, count * 8); return; }
        std::vector<uint32_t> tmp(1 << 16);
        for (uint64_t i = 0; i < count; i += tmp.size()) {
            const uint64_t m = std::min<uint64_t>(tmp.size(), count - i);
            for (uint64_t t = 0; t < m; ++t) tmp[t] = static_cast<uint32_t>(v[i + t]);
            raw(tmp.data(), m * 4);
        }
        pad();
    }
    void bytes(const void *p, uint64_t count) { raw(p, count); pad(); }
    bool finish() {
        pad();
        if (chunk_left_ != kSumChunkWords) outer_.words(&sum_.h, 1);   // the last, partial chunk
        const uint64_t h = outer_.h;
        if (fp_ && std::fwrite(&h, 8, 1, fp_) != 1) ok_ = false;
        if (fp_ && std::fclose(fp_) != 0) ok_ = false;
        fp_ = nullptr;
        if (ok_ && std::rename(tmp_.c_str(), final_.c_str()) != 0) ok_ = false;
        if (!ok_) std::remove(tmp_.c_str());
        return ok_;
    }

   private:
    void raw(const void *p, uint64_t nbytes) {  // checksum runs over whole words; carry_ holds a partial one
        const unsigned char *c = static_cast<const unsigned char *>(p);
        if (fp_ && nbytes && std::fwrite(c, 1, nbytes, fp_) != nbytes) ok_ = false;
        while (nbytes) {
            if (ncarry_ == 0 && nbytes >= 8) {
                const uint64_t nw = nbytes / 8;
                if ((reinterpret_cast<uintptr_t>(c) & 7) == 0) {
                    feed(reinterpret_cast<const uint64_t *>(c), nw);
                } else {
                    for (uint64_t i = 0; i < nw; ++i) { uint64_t w; std::memcpy(&w, c + 8 * i, 8); feed(&w, 1); }
                }
                c += nw * 8;
                nbytes -= nw * 8;
                continue;
            }
            carry_[ncarry_++] = *c++;
            --nbytes;
            if (ncarry_ == 8) { uint64_t w; std::memcpy(&w, carry_, 8); feed(&w, 1); ncarry_ = 0; }
        }
    }
    void feed(const uint64_t *w, uint64_t n) {   // the chunked checksum (kSumChunkWords)
        while (n) {
            const uint64_t m = std::min(n, chunk_left_);
            sum_.words(w, m);
            w += m; n -= m; chunk_left_ -= m;
            if (chunk_left_ == 0) { outer_.words(&sum_.h, 1); sum_ = FlatSum(); chunk_left_ = kSumChunkWords; }
        }
    }
    void pad() {
        static const unsigned char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ncarry_) raw(zeros, 8 - ncarry_);
    }
    std::string final_, tmp_;
    FILE *fp_;
    bool ok_ = true;
    FlatSum sum_, outer_;
    uint64_t chunk_left_ = kSumChunkWords;
    unsigned char carry_[8];
    unsigned ncarry_ = 0;
};

uint64_t padded8(uint64_t nbytes) { return (nbytes + 7) & ~uint64_t(7); }

void widen(const unsigned char *src, uint64_t count, unsigned width, std::vector<uint64_t> &out) {
    resize_parallel(out, count);
    parallel_for(count, [&](uint64_t b, uint64_t e, unsigned) {
        if (width == 8) { std::memcpy(out.data() + b, src + 8 * b, (e - b) * 8); return; }
        for (uint64_t i = b; i < e; ++i) { uint32_t v; std::memcpy(&v, src + 4 * i, 4); out[i] = v; }
    }, uint64_t(1) << 18);
}
}  // namespace

int write_flat(const std::string &fname, const FlatBundle &b) {
    const RawRle &r = b.rle;
    if (r.R == 0 || r.heads.size() != r.R || r.lens.size() != r.R) return RBG_EARG;
    uint64_t max_len = 0;
    for (uint64_t l : r.lens) max_len = std::max(max_len, l);
    const unsigned len_width = max_len < (uint64_t(1) << 32) ? 4 : 8;
    const unsigned pos_width = r.n < (uint64_t(1) << 32) ? 4 : 8;
    std::string docs;
    if (b.has_dl)
        for (size_t i = 0; i < b.dl.names.size(); ++i) docs += b.dl.names[i] + " " + std::to_string(b.dl.starts[i]) + "\n";
    FlatWriter w(fname);
    if (!w.ok()) return RBG_EIO;
    w.bytes(kFlatMagic, 8);
    w.u64((b.has_tsa ? 1u : 0u) | (b.has_ma ? 2u : 0u) | (b.has_dl ? 4u : 0u));
    w.u64(r.n); w.u64(r.R); w.u64(r.B); w.u64(len_width); w.u64(pos_width);
    w.u64(b.has_ma ? b.ma.start.size() : 0);
    w.u64(b.has_ma ? b.ma.vals.size() : 0);
    w.u64(static_cast<uint64_t>(static_cast<int64_t>(b.has_ma ? b.ma.wsize : 0)));
    w.u64(docs.size());
    w.bytes(r.heads.data(), r.R);
    w.section(r.lens.data(), r.R, len_width);
    if (b.has_tsa) {
        if (b.tsa.pred_pos.size() != r.R || b.tsa.samples_last.size() != r.R || b.tsa.pred_to_run.size() != r.R) return RBG_EARG;
        w.section(b.tsa.pred_pos.data(), r.R, pos_width);
        w.section(b.tsa.samples_last.data(), r.R, pos_width);
        w.section(b.tsa.pred_to_run.data(), r.R, pos_width);
    }
    if (b.has_ma) {
        const uint64_t nr = b.ma.start.size();
        if (b.ma.end.size() != nr || b.ma.off.size() != nr + 1 || b.ma.off[nr] != b.ma.vals.size()) return RBG_EARG;
        w.section(b.ma.start.data(), nr, pos_width);
        w.section(b.ma.end.data(), nr, pos_width);
        w.section(b.ma.off.data(), nr + 1, 8);
        w.section(b.ma.vals.data(), b.ma.vals.size(), 8);
    }
    if (b.has_dl) w.bytes(docs.data(), docs.size());
    return w.finish() ? RBG_OK : RBG_EIO;
}

int read_flat(const std::string &fname, FlatBundle &b) {
    // the file is mapped, not copied (9 GB at n = 5e10: a zero-filled buffer and a read() into it were 4 s of the load)
    struct Mapped {
        int fd = -1;
        void *p = MAP_FAILED;
        uint64_t sz = 0;
        ~Mapped() { if (p != MAP_FAILED) munmap(p, sz); if (fd >= 0) close(fd); }
    } mf;
    mf.fd = open(fname.c_str(), O_RDONLY | O_CLOEXEC);
    if (mf.fd < 0) return RBG_EIO;
    struct stat sb;
    if (fstat(mf.fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return RBG_EIO;
    const uint64_t sz = mf.sz = static_cast<uint64_t>(sb.st_size);
    constexpr uint64_t kHeader = 8 * 11;
    if (sz < kHeader + 8 || (sz & 7)) return RBG_EFORMAT;
    mf.p = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (mf.p == MAP_FAILED) return RBG_EIO;
    (void)madvise(mf.p, sz, MADV_WILLNEED);
    struct { const uint64_t *w; uint64_t n; const uint64_t *data() const { return w; } uint64_t size() const { return n; }
             uint64_t operator[](uint64_t i) const { return w[i]; } uint64_t back() const { return w[n - 1]; } } file{static_cast<const uint64_t *>(mf.p), sz / 8};
    const auto t_begin = std::chrono::steady_clock::now();
    const bool v1 = std::memcmp(file.data(), kFlatMagicV1, 8) == 0;
    if (!v1 && std::memcmp(file.data(), kFlatMagic, 8) != 0) return RBG_EFORMAT;
    if (v1) {
        FlatSum sum;
        sum.words(file.data(), file.size() - 1);
        if (sum.h != file.back()) return RBG_EFORMAT;
    } else {
        const uint64_t words = file.size() - 1, nchunks = (words + kSumChunkWords - 1) / kSumChunkWords;
        std::vector<uint64_t> hs(nchunks);
        parallel_for(nchunks, [&](uint64_t c0, uint64_t c1, unsigned) {
            for (uint64_t c = c0; c < c1; ++c) {
                FlatSum s;
                s.words(file.data() + c * kSumChunkWords, std::min(kSumChunkWords, words - c * kSumChunkWords));
                hs[c] = s.h;
            }
        }, 1);
        FlatSum outer;
        outer.words(hs.data(), hs.size());
        if (outer.h != file.back()) return RBG_EFORMAT;
    }
    const auto t_sum = std::chrono::steady_clock::now();
    b = FlatBundle();
    const uint64_t flags = file[1];
    RawRle &r = b.rle;
    r.n = file[2]; r.R = file[3]; r.B = file[4];
    const uint64_t len_width = file[5], pos_width = file[6], ma_nruns = file[7], ma_nvals = file[8], docs_bytes = file[10];
    b.ma.wsize = static_cast<int32_t>(static_cast<int64_t>(file[9]));
    b.has_tsa = flags & 1; b.has_ma = flags & 2; b.has_dl = flags & 4;
    if ((flags & ~uint64_t(7)) || r.n == 0 || r.R == 0 || r.R > r.n || r.B == 0) return RBG_EFORMAT;
    if ((len_width != 4 && len_width != 8) || (pos_width != 4 && pos_width != 8)) return RBG_EFORMAT;
    if (pos_width == 4 && r.n >= (uint64_t(1) << 32)) return RBG_EFORMAT;
    if (!b.has_ma && (ma_nruns || ma_nvals)) return RBG_EFORMAT;
    if (!b.has_dl && docs_bytes) return RBG_EFORMAT;
    const uint64_t body = sz - kHeader - 8;
    // every count is bounded by the file size before anything is multiplied or allocated
    if (r.R > body || ma_nruns > body / 4 || ma_nvals > body / 8 || docs_bytes > body) return RBG_EFORMAT;
    uint64_t need = padded8(r.R) + padded8(r.R * len_width);
    if (b.has_tsa) need += 3 * padded8(r.R * pos_width);
    if (b.has_ma) need += 2 * padded8(ma_nruns * pos_width) + (ma_nruns + 1) * 8 + ma_nvals * 8;
    need += padded8(docs_bytes);
    if (need != body) return RBG_EFORMAT;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(file.data()) + kHeader;
    r.heads.assign(p, p + r.R);
    p += padded8(r.R);
    widen(p, r.R, static_cast<unsigned>(len_width), r.lens);
    p += padded8(r.R * len_width);
    {
        const unsigned T = load_threads();
        std::vector<uint64_t> part(T + 1, 0);
        std::vector<int> bad(T + 1, 0);
        parallel_for(r.R, [&](uint64_t b0, uint64_t e0, unsigned t) {
            uint64_t sum = 0;
            for (uint64_t i = b0; i < e0; ++i) {
                if (r.lens[i] == 0 || r.lens[i] > r.n - sum) { bad[t] = 1; return; }   // (no chunk may exceed n by itself: no overflow)
                if (i && r.heads[i] == r.heads[i - 1]) { bad[t] = 1; return; }           // runs are maximal
                sum += r.lens[i];
            }
            part[t] = sum;
        });
        uint64_t total = 0;
        for (unsigned t = 0; t <= T; ++t) {
            if (bad[t] || part[t] > r.n - total) return RBG_EFORMAT;
            total += part[t];
        }
        if (total != r.n) return RBG_EFORMAT;
    }
    if (b.has_tsa) {
        RawTsa &t = b.tsa;
        t.r = r.R; t.n = r.n;
        widen(p, r.R, static_cast<unsigned>(pos_width), t.pred_pos);
        p += padded8(r.R * pos_width);
        widen(p, r.R, static_cast<unsigned>(pos_width), t.samples_last);
        p += padded8(r.R * pos_width);
        widen(p, r.R, static_cast<unsigned>(pos_width), t.pred_to_run);
        p += padded8(r.R * pos_width);
        std::vector<int> bad(load_threads() + 1, 0);
        parallel_for(r.R, [&](uint64_t b0, uint64_t e0, unsigned th) {
            for (uint64_t i = b0; i < e0; ++i) {
                if (t.pred_pos[i] >= r.n || (i && t.pred_pos[i] <= t.pred_pos[i - 1])) bad[th] = 1;
                if (t.samples_last[i] >= r.n || t.pred_to_run[i] >= r.R) bad[th] = 1;
            }
        });
        for (int v : bad) if (v) return RBG_EFORMAT;
    }
    if (b.has_ma) {
        RawMarkers &m = b.ma;
        widen(p, ma_nruns, static_cast<unsigned>(pos_width), m.start);
        p += padded8(ma_nruns * pos_width);
        widen(p, ma_nruns, static_cast<unsigned>(pos_width), m.end);
        p += padded8(ma_nruns * pos_width);
        widen(p, ma_nruns + 1, 8, m.off);
        p += (ma_nruns + 1) * 8;
        widen(p, ma_nvals, 8, m.vals);
        p += ma_nvals * 8;
        if (m.off[0] != 0 || m.off[ma_nruns] != ma_nvals) return RBG_EFORMAT;
        for (uint64_t i = 0; i < ma_nruns; ++i) {
            if (m.end[i] < m.start[i] || m.end[i] >= r.n || m.off[i + 1] < m.off[i]) return RBG_EFORMAT;
            if (i && m.start[i] <= m.end[i - 1]) return RBG_EFORMAT;
        }
    }
    if (b.has_dl) {
        std::istringstream ss(std::string(reinterpret_cast<const char *>(p), docs_bytes));
        std::string name;
        uint64_t pos = 0;
        while (ss >> name >> pos) {
            b.dl.names.push_back(name);
            b.dl.starts.push_back(pos);
        }
        b.dl.sorted = b.dl.starts;
        std::sort(b.dl.sorted.begin(), b.dl.sorted.end());
    }
    if (std::getenv("RBG_VERBOSE"))
        std::fprintf(stderr, "rbg: cache file of %.2f GB: checksum %.2f s, arrays decoded and checked %.2f s\n", sz / 1e9,
                     std::chrono::duration<double>(t_sum - t_begin).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sum).count());
    return RBG_OK;
}

// ---- flatten -------------------------------------------------------------------------------------
namespace {
// largest shift <= 8 (slot offsets are 8-bit, rbg_dev.h) with at most `per_slot` items per bucket on average
uint32_t auto_shift(uint64_t n, uint64_t items, double per_slot) {
    if (items == 0) items = 1;
    uint32_t s = 0;
    while (s < 8 && static_cast<double>(items) * static_cast<double>(uint64_t(2) << s) <= per_slot * static_cast<double>(n)) ++s;
    return s;
}

// Multi-symbol LF steps.  For the "major" symbols (the <= 4 most frequent non-terminator symbols)
// a depth-d table T_d[x_{d-1} .. x_1 x_0] describes the rows p whose d preceding text characters
// are x_{d-1} .. x_0, i.e. bwt[p] = x_0, bwt[LF(p)] = x_1, ...: runs of such rows (start, cum),
// F = first row of the SA interval of the d-mer, samp = SA - d at the end of each run.  Then
//   LF^d([lo,hi], d-mer) = F + rank_d(., d-mer)
// is exactly d nested RowBowt::LF calls (rowbowt.hpp:74-88), and the toehold after d LF_w_loc
// calls (rowbowt.hpp:555-573) is k-d when row hi carries the d-mer, else samp of the last run
// starting before hi.
//
// T_{d+1} is composed from T_d: the rows with bwt = c are LF-mapped, run by run, onto a row-ordered
// segmentation G_d of [0,n) by depth-d id; every overlap piece is one depth-(d+1) run.  The SA value
// at a piece's last row is known: either its LF image ends a G_d segment (that segment's sample), or
// the row ends the c-run (that run's samples_last_ minus d).
}  // namespace

uint32_t kmer_table_shift(uint64_t n, uint64_t nruns, uint32_t depth, const FlattenOptions &opt) {
    uint32_t s = opt.rank_bucket_shift >= 0 ? static_cast<uint32_t>(opt.rank_bucket_shift) : auto_shift(n, nruns, 1.5);
    if (depth >= 4 && opt.deep_bucket_shift >= 0) s = static_cast<uint32_t>(opt.deep_bucket_shift);  // levels of 4-mers and deeper
    return s;
}

namespace {
struct Segmentation {          // row-ordered cover of [0,n): segment g = [start[g], start[g+1])
    std::vector<uint64_t> start;  // + sentinel n
    std::vector<uint32_t> id;     // depth-d table index, or kNoId
    std::vector<uint64_t> samp;   // SA - d at the segment's last row (only with a toehold SA)
};
constexpr uint32_t kNoId = 0xFFFFFFFFu;

// `next` (optional): the row-ordered segmentation by depth-(d+1) id, for composing the level after this one.
// The pieces a worker finds are already in row order within its symbol, so G_{d+1} is an M-way merge of M
// lists (a heap merge over all n_ids * M tables took as long as the composition itself).
int compose(const HostIndex &ix, const std::vector<uint32_t> &major_slot, const Segmentation &G, uint32_t depth,
            uint32_t n_ids, const std::vector<SymTable> &prev, bool with_samples, const FlattenOptions &opt,
            std::vector<SymTable> &out, Segmentation *next) {
    const uint32_t M = static_cast<uint32_t>(major_slot.size());
    out.assign(static_cast<size_t>(n_ids) * M, SymTable());
    // The runs of symbol m are cut into C chunks at run boundaries; worker (m, c) walks its chunk and keeps
    // what it finds per depth-d id with counts relative to the chunk (a piece never spans two c-runs, so
    // chunks need no stitching); a second step per symbol turns them into the tables (id, m).
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t C = std::max(1u, std::min(8u, hw / std::max(1u, M)));
    struct Piece { uint64_t start, len, samp; uint32_t id; };
    struct Part {
        std::vector<std::vector<uint64_t>> start, cum, samp;  // per id
        std::vector<uint64_t> total;                          // rows per id in this chunk
        std::vector<Piece> rows;                              // the same pieces in row order (only for `next`)
        int rc = RBG_OK;
    };
    std::vector<Part> parts(static_cast<size_t>(M) * C);
    auto walk = [&](uint32_t m, uint32_t c) {
        Part &P = parts[static_cast<size_t>(m) * C + c];
        P.start.assign(n_ids, {}); P.cum.assign(n_ids, {}); P.samp.assign(n_ids, {}); P.total.assign(n_ids, 0);
        const SymTable &tc = ix.sym[major_slot[m]];
        const uint64_t k0 = tc.nruns * c / C, k1 = tc.nruns * (c + 1) / C;
        if (k0 >= k1) return;
        // first segment of G that can overlap the LF image of run k0
        const uint64_t Q0 = tc.F + tc.cum[k0];
        uint64_t g = static_cast<uint64_t>(std::upper_bound(G.start.begin(), G.start.end(), Q0) - G.start.begin());
        g = g ? g - 1 : 0;
        for (uint64_t k = k0; k < k1; ++k) {
            const uint64_t s = tc.start[k], len = tc.cum[k + 1] - tc.cum[k];
            const uint64_t Q = tc.F + tc.cum[k];  // LF of row s (rowbowt.hpp:65-68)
            uint64_t q = Q;
            while (q < Q + len) {
                while (G.start[g + 1] <= q) ++g;
                const uint64_t qend = std::min(Q + len, G.start[g + 1]);
                const uint32_t id = G.id[g];
                if (id != kNoId) {
                    P.start[id].push_back(s + (q - Q));
                    P.cum[id].push_back(P.total[id]);
                    P.total[id] += qend - q;
                    if (with_samples) {
                        uint64_t v;
                        if (qend == G.start[g + 1]) v = G.samp[g];
                        else {
                            if (tc.samp[k] < depth) { P.rc = RBG_EFORMAT; return; }  // would need the terminator inside the k-mer
                            v = tc.samp[k] - depth;
                        }
                        P.samp[id].push_back(v);
                    }
                    if (next) P.rows.push_back(Piece{s + (q - Q), qend - q, with_samples ? P.samp[id].back() : 0, id * M + m});
                }
                q = qend;
            }
        }
    };
    {
        std::vector<std::thread> workers;
        for (uint32_t m = 0; m < M; ++m)
            for (uint32_t c = 0; c < C; ++c)
                if (m || c) workers.emplace_back(walk, m, c);
        walk(0, 0);
        for (auto &w : workers) w.join();
    }
    std::vector<int> rcs(M, RBG_OK);
    auto work = [&](uint32_t m) {
        const SymTable &tc = ix.sym[major_slot[m]];
        for (uint32_t c = 0; c < C; ++c)
            if (parts[static_cast<size_t>(m) * C + c].rc) { rcs[m] = parts[static_cast<size_t>(m) * C + c].rc; return; }
        for (uint32_t id = 0; id < n_ids; ++id) {
            SymTable &t = out[static_cast<size_t>(id) * M + m];
            size_t nr = 0;
            for (uint32_t c = 0; c < C; ++c) nr += parts[static_cast<size_t>(m) * C + c].start[id].size();
            t.start.reserve(nr + 1); t.cum.reserve(nr + 1);
            if (with_samples) t.samp.reserve(nr);
            uint64_t base = 0;
            for (uint32_t c = 0; c < C; ++c) {
                Part &P = parts[static_cast<size_t>(m) * C + c];
                t.start.insert(t.start.end(), P.start[id].begin(), P.start[id].end());
                for (uint64_t v : P.cum[id]) t.cum.push_back(base + v);
                if (with_samples) t.samp.insert(t.samp.end(), P.samp[id].begin(), P.samp[id].end());
                base += P.total[id];
                std::vector<uint64_t>().swap(P.start[id]);
                std::vector<uint64_t>().swap(P.cum[id]);
                std::vector<uint64_t>().swap(P.samp[id]);
            }
            const SymTable &tp = prev[id];
            t.byte = tc.byte;
            t.nruns = t.start.size();
            t.total = base;
            t.start.push_back(ix.n);
            t.cum.push_back(t.total);
            // F_{d+1}[id, c] = F_d[id] + rank_d(F[c], id): rows of the id-interval followed by a smaller symbol
            const uint64_t i = tc.F;
            const uint64_t kk = std::lower_bound(tp.start.begin(), tp.start.begin() + tp.nruns, i) - tp.start.begin();
            uint64_t rk = 0;
            if (kk > 0) rk = tp.cum[kk - 1] + std::min(i - tp.start[kk - 1], tp.cum[kk] - tp.cum[kk - 1]);
            t.F = tp.F + rk;
            t.shift = kmer_table_shift(ix.n, t.nruns, depth + 1, opt);
            if (t.shift > 12 || (t.shift > 8 && (ix.n >> 40))) { rcs[m] = RBG_EARG; return; }  // wide buckets carry 40-bit ranks (rbg_dev.h)
            if (t.nruns >= 0xFFFFFFF0ull) { rcs[m] = RBG_EARG; return; }
        }
    };
    std::vector<std::thread> workers;
    for (uint32_t m = 1; m < M; ++m) workers.emplace_back(work, m);
    work(0);
    for (auto &w : workers) w.join();
    for (int rc : rcs)
        if (rc) return rc;
    if (next) {
        size_t total = 0;
        for (const Part &P : parts) total += P.rows.size();
        next->start.clear(); next->id.clear(); next->samp.clear();
        next->start.reserve(2 * total + 2); next->id.reserve(2 * total + 2);
        if (with_samples) next->samp.reserve(2 * total + 2);
        // cursor per symbol over its chunks' lists (chunk lists of one symbol follow each other in row order)
        std::vector<uint32_t> chunk(M, 0);
        std::vector<size_t> at(M, 0);
        auto head = [&](uint32_t m) -> const Piece * {
            while (chunk[m] < C && at[m] >= parts[static_cast<size_t>(m) * C + chunk[m]].rows.size()) { ++chunk[m]; at[m] = 0; }
            return chunk[m] < C ? &parts[static_cast<size_t>(m) * C + chunk[m]].rows[at[m]] : nullptr;
        };
        uint64_t pos = 0;
        while (true) {
            const Piece *best = nullptr;
            uint32_t bm = 0;
            for (uint32_t m = 0; m < M; ++m) {
                const Piece *h = head(m);
                if (h && (!best || h->start < best->start)) { best = h; bm = m; }
            }
            if (!best) break;
            if (best->start > pos) {  // gap: rows whose context leaves the major alphabet
                next->start.push_back(pos); next->id.push_back(kNoId);
                if (with_samples) next->samp.push_back(0);
            }
            next->start.push_back(best->start); next->id.push_back(best->id);
            if (with_samples) next->samp.push_back(best->samp);
            pos = best->start + best->len;
            ++at[bm];
        }
        if (pos < ix.n) {
            next->start.push_back(pos); next->id.push_back(kNoId);
            if (with_samples) next->samp.push_back(0);
        }
        next->start.push_back(ix.n);
    }
    return RBG_OK;
}

// the k-mer alphabet: the <= 4 most frequent non-terminator symbols; false = no k-mer steps for this index
bool choose_major(HostIndex &out, const FlattenOptions &opt) {
    std::memset(out.major_of, 0xFF, sizeof(out.major_of));
    out.nmajor = 0;
    out.major_slot.clear();
    out.clear_kmer();
    if (opt.kmer_steps < 2 || out.sigma < 2) return false;
    // the terminator: the smallest symbol, occurring once (rle_string.hpp:59,62 maps 0 -> 1).
    // Without one the wrap argument of DESIGN.md 2b does not hold: keep single steps only.
    if (out.sym[0].total != 1) return false;
    std::vector<uint32_t> order;
    for (uint32_t s = 1; s < out.sigma; ++s) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return out.sym[a].total != out.sym[b].total ? out.sym[a].total > out.sym[b].total : a < b;
    });
    if (order.size() > 4) order.resize(4);
    std::sort(order.begin(), order.end());
    const uint32_t M = static_cast<uint32_t>(order.size());
    out.nmajor = M;
    for (uint32_t m = 0; m < M; ++m) {
        out.major_byte[m] = out.sym[order[m]].byte;
        out.major_of[out.sym[order[m]].byte] = static_cast<uint8_t>(m);
    }
    out.major_slot = order;
    return true;
}

int build_kmer_tables(HostIndex &out, const RawTsa *tsa, const FlattenOptions &opt) {
    if (!choose_major(out, opt)) return RBG_OK;
    if (opt.defer_kmer) { out.kmer_deferred = static_cast<uint32_t>(std::min(kMaxKmerDepth, opt.kmer_steps)); return RBG_OK; }
    return compose_kmer_tables_host(out, opt.kmer_steps, opt);
}
}  // namespace

int compose_kmer_tables_host(HostIndex &out, int kmer_steps, const FlattenOptions &opt_in) {
    FlattenOptions opt = opt_in;
    opt.kmer_steps = kmer_steps;
    out.clear_kmer();
    out.kmer_deferred = 0;
    if (out.nmajor == 0 || kmer_steps < 2) return RBG_OK;
    if (kmer_steps > kMaxKmerDepth) opt.kmer_steps = kmer_steps = kMaxKmerDepth;
    RawTsa tsa_view;   // compose() only needs samples_last
    const RawTsa *tsa = nullptr;
    if (out.has_tsa) { tsa_view.samples_last = out.samples_last; tsa = &tsa_view; }
    const std::vector<uint32_t> order = out.major_slot;
    const uint32_t M = out.nmajor;
    // depth 1: the BWT runs themselves, id = major index of the head, sample = samples_last_ (SA - 1)
    Segmentation G;
    G.start = out.run_start;
    G.id.resize(out.r);
    for (uint64_t g = 0; g < out.r; ++g) {
        const uint8_t m = out.major_of[out.run_heads[g]];
        G.id[g] = m == 0xFF ? kNoId : m;
    }
    if (tsa) G.samp = tsa->samples_last;
    std::vector<SymTable> depth1(M);
    for (uint32_t m = 0; m < M; ++m) depth1[m] = out.sym[order[m]];
    // depth d + 1 from depth d's tables and segmentation, one sweep per depth
    uint32_t n_ids = M;
    for (int d = 1; d < kmer_steps; ++d, n_ids *= M) {
        Segmentation Gn;
        const bool more = d + 1 < kmer_steps;
        const int rc = compose(out, order, G, static_cast<uint32_t>(d), n_ids, d == 1 ? depth1 : out.kmer(static_cast<uint32_t>(d)), tsa != nullptr, opt,
                               out.kmer(static_cast<uint32_t>(d + 1)), more ? &Gn : nullptr);
        if (rc) return rc;
        G = std::move(Gn);
    }
    return RBG_OK;
}

namespace {
struct StageTimer {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageTimer(const char *w) : what(w) {}
    ~StageTimer() {
        if (std::getenv("RBG_VERBOSE"))
            std::fprintf(stderr, "rbg: %s %.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};
}  // namespace

int flatten(const RawRle &rle, const RawTsa *tsa, const FlattenOptions &opt, HostIndex &out) {
    StageTimer whole("flatten (incl. k-mer tables)");
    const uint64_t R = rle.R;
    if (R == 0 || rle.heads.size() != R || rle.lens.size() != R) return RBG_EARG;
    if (R >= 0xFFFFFFF0ull) return RBG_EARG;  // run ordinals (DevSym::ord) are 32-bit
    // (and, checked below once n is known: rank values must fit the 48 bits a RankSlot holds)
    if (tsa && (tsa->r != R || tsa->samples_last.size() != R || tsa->pred_pos.size() != R || tsa->pred_to_run.size() != R))
        return RBG_EFORMAT;
    out = HostIndex();
    out.r = R;
    copy_parallel(out.run_heads, rle.heads);
    resize_parallel(out.run_start, R + 1);
    // Two passes over the runs, both split over the worker threads (contiguous chunks): the first sums each chunk's
    // lengths and counts its runs and symbols per head, a prefix over the chunks then tells every chunk where its rows
    // begin and where its runs go in each symbol's table, the second writes run_start and the tables' entries in place.
    const unsigned T = std::max(1u, std::min<unsigned>(load_threads(), static_cast<unsigned>(R / 65536 + 1)));
    struct ChunkSum {
        uint64_t len = 0, cnt[256], nr[256];
        bool zero = false;
    };
    std::vector<ChunkSum> cs(T);
    auto chunk = [&](unsigned t, uint64_t &b, uint64_t &e) { b = R * t / T; e = R * (t + 1) / T; };
    auto on_all = [&](const std::function<void(unsigned)> &f) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(f, t);
        f(0);
        for (auto &w : th) w.join();
    };
    on_all([&](unsigned t) {
        uint64_t b, e;
        chunk(t, b, e);
        ChunkSum &c = cs[t];
        std::memset(c.cnt, 0, sizeof(c.cnt));
        std::memset(c.nr, 0, sizeof(c.nr));
        for (uint64_t i = b; i < e; ++i) {
            const uint64_t l = rle.lens[i];
            if (l == 0) c.zero = true;
            c.len += l;
            c.cnt[rle.heads[i]] += l;
            c.nr[rle.heads[i]]++;
        }
    });
    uint64_t cnt[256] = {0}, nr[256] = {0};
    std::vector<uint64_t> pos0(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) {
        if (cs[t].zero) return RBG_EARG;
        pos0[t + 1] = pos0[t] + cs[t].len;
        for (int c = 0; c < 256; ++c) { cnt[c] += cs[t].cnt[c]; nr[c] += cs[t].nr[c]; }
    }
    out.run_start[R] = pos0[T];
    out.n = pos0[T];
    if (out.n >> 48) return RBG_EARG;  // RankSlot carries 48-bit ranks (rbg_dev.h)
    if (tsa && tsa->n != out.n) return RBG_EFORMAT;
    // F column (RowBowt::build_f, rowbowt.hpp:770-778) and slots
    std::memset(out.lut, 0xFF, sizeof(out.lut));
    uint64_t acc = 0;
    for (int s = 0; s < 256; ++s) {
        out.f[s] = acc;
        acc += cnt[s];
        if (cnt[s]) {
            out.lut[s] = static_cast<uint8_t>(out.sym.size());
            SymTable t;
            t.byte = static_cast<uint8_t>(s);
            t.nruns = nr[s];
            t.total = cnt[s];
            t.F = out.f[s];
            out.sym.push_back(std::move(t));
        }
    }
    out.f[256] = acc;
    out.sigma = static_cast<uint32_t>(out.sym.size());
    if (out.sigma > 255) return RBG_EARG;  // slot 0xFF is the "absent" marker
    // 32-bit positions need n below the two reserved marker values (rbg_dev.h kSent / kOvf)
    out.pos_bytes = opt.force_pos_bytes ? opt.force_pos_bytes : (out.n < 0xFFFFFFF0ull ? 4 : 8);
    if (out.pos_bytes == 4 && out.n >= 0xFFFFFFF0ull) return RBG_EARG;
    for (SymTable &t : out.sym) {
        resize_parallel(t.start, t.nruns + 1);
        resize_parallel(t.cum, t.nruns + 1);
        if (tsa) resize_parallel(t.samp, t.nruns);
    }
    // where each chunk's runs go in each symbol's table, and how many of the symbol precede them
    std::vector<uint64_t> ord0(static_cast<size_t>(T) * 256, 0), cum0(static_cast<size_t>(T) * 256, 0);
    for (int c = 0; c < 256; ++c) {
        uint64_t o = 0, q = 0;
        for (unsigned t = 0; t < T; ++t) {
            ord0[static_cast<size_t>(t) * 256 + c] = o;
            cum0[static_cast<size_t>(t) * 256 + c] = q;
            o += cs[t].nr[c];
            q += cs[t].cnt[c];
        }
    }
    on_all([&](unsigned t) {
        uint64_t b, e;
        chunk(t, b, e);
        uint64_t pos = pos0[t];
        uint64_t *o = ord0.data() + static_cast<size_t>(t) * 256, *q = cum0.data() + static_cast<size_t>(t) * 256;
        for (uint64_t i = b; i < e; ++i) {
            const uint8_t h = rle.heads[i];
            SymTable &tb = out.sym[out.lut[h]];
            const uint64_t k = o[h]++;
            out.run_start[i] = pos;
            tb.start[k] = pos;
            tb.cum[k] = q[h];
            if (tsa) tb.samp[k] = tsa->samples_last[i];
            q[h] += rle.lens[i];
            pos += rle.lens[i];
        }
    });
    for (SymTable &t : out.sym) {
        t.start[t.nruns] = out.n;
        t.cum[t.nruns] = t.total;
        t.shift = opt.rank_bucket_shift >= 0 ? static_cast<uint32_t>(opt.rank_bucket_shift) : auto_shift(out.n, t.nruns, 1.5);
        if (t.shift > 12 || (t.shift > 8 && (out.n >> 40))) return RBG_EARG;  // wide buckets carry 40-bit ranks (rbg_dev.h)
    }
    if (tsa) {
        out.has_tsa = true;
        copy_parallel(out.samples_last, tsa->samples_last);
        copy_parallel(out.pred_pos, tsa->pred_pos);
        resize_parallel(out.phi_base, R);
        std::vector<int> bad(T, 0);
        on_all([&](unsigned t) {
            uint64_t b, e;
            chunk(t, b, e);
            for (uint64_t j = b; j < e; ++j) {
                const uint64_t run = tsa->pred_to_run[j];
                out.phi_base[j] = run ? tsa->samples_last[run - 1] : 0;  // toehold_sa.hpp:67-70
                if (j && out.pred_pos[j] <= out.pred_pos[j - 1]) bad[t] = 1;
            }
        });
        for (int v : bad)
            if (v) return RBG_EFORMAT;
        out.last_run_sample = (tsa->samples_last[R - 1] + 1) % out.n;  // toehold_sa.hpp:97-99
        out.phi_shift = opt.phi_bucket_shift >= 0 ? static_cast<uint32_t>(opt.phi_bucket_shift) : auto_shift(out.n, R, 0.75);
        if (out.phi_shift > 8) return RBG_EARG;
    }
    StageTimer km("k-mer tables");
    return build_kmer_tables(out, tsa, opt);
}

}  // namespace rbg
