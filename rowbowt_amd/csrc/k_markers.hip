// k_markers.hip -- K4 and the seeding kernels: markers_at, find_range_w_markers, greedy seeds, marker seeds, single LF steps (rowbowt.hpp:74-88, :222-339, :406-482)
#include "rbg_device.hpp"

namespace rbg {
namespace {

// ---- K4: markers (rbg_device.hpp marker_query: at_range from the bucket records, or the directory + run arrays) ------------
__global__ __launch_bounds__(256) void k_markers_count(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                       const uint64_t *__restrict__ hi, const uint64_t N,
                                                       uint64_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        uint64_t cnt = 0;
        if (hi[i] >= lo[i]) {
            uint64_t src, c;
            if (marker_query(ix, lo[i], hi[i], &src, &c)) cnt = c;
        }
        out[i + 1] = cnt;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

__global__ __launch_bounds__(256) void k_markers_fill(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                      const uint64_t *__restrict__ hi, const uint64_t N,
                                                      const uint64_t *__restrict__ mk_off, uint64_t *__restrict__ mk) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        if (hi[i] < lo[i]) continue;
        uint64_t src, cnt;
        if (!marker_query(ix, lo[i], hi[i], &src, &cnt)) continue;
        uint64_t *dst = mk + mk_off[i];
        for (uint64_t t = 0; t < cnt; ++t) dst[t] = ix.mk_vals[src + t];
    }
}


// ---- find_range_w_markers (rowbowt.hpp:292-339) --------------------------------------------------
// Backward search that queries the marker array at every window end (:315-323) and once more at
// the end when (m-1) % wsize != 0 (:328-335).  Window results are PREPENDED in the reference
// (:320,:333), so query q's markers land at  total - (c_0 + ... + c_q).
// FILL=false: count pass (writes lo/hi and per-read totals to cnt_out[i+1]);
// FILL=true : re-walks the read and writes the markers at mk_off[i].
template <typename P, bool FILL>
__global__ __launch_bounds__(256) void k_find_range_markers(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                            const uint64_t *__restrict__ off, const uint64_t N,
                                                            const uint64_t wsize, const uint64_t max_range,
                                                            uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                            uint64_t *__restrict__ cnt_out,
                                                            const uint64_t *__restrict__ mk_off, uint64_t *__restrict__ mk) {
    __shared__ uint8_t s_lut[256];
    __shared__ DevSym s_sym[kLdsSyms];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) s_lut[t] = ix.lut[t];
    const int nlds = ix.sigma < static_cast<uint32_t>(kLdsSyms) ? static_cast<int>(ix.sigma) : kLdsSyms;
    for (int t = threadIdx.x; t < nlds; t += blockDim.x) s_sym[t] = ix.syms[t];
    __syncthreads();
    const uint64_t *__restrict__ words = reinterpret_cast<const uint64_t *>(seqs);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) cnt_out[0] = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint64_t beg = off[i], end = off[i + 1], m = end - beg;
        uint64_t total = 0;
        uint64_t lo = 1, hi = 0;
        bool alive = m >= wsize;  // rowbowt.hpp:299-302: shorter queries return the default LFData
        // fill pass: a read with nothing to emit (including every read that dies: lf.clear() drops
        // what earlier windows collected) must not write at all
        if (FILL && mk_off[i + 1] == mk_off[i]) continue;
        if (alive) {
            lo = 0; hi = ix.n - 1;
            uint64_t window_ei = m, acc = 0;
            const uint64_t want = FILL ? mk_off[i + 1] - mk_off[i] : 0;
            uint64_t *dst = FILL ? mk + mk_off[i] : nullptr;
            uint64_t cur_wi = ~uint64_t(0), w = 0;
            for (uint64_t s = 0; s <= m; ++s) {
                bool query;
                if (s < m) {
                    const uint64_t p = end - 1 - s;
                    const uint64_t wi = p >> 3;
                    if (wi != cur_wi) { w = words[wi]; cur_wi = wi; }
                    const uint32_t c = static_cast<uint32_t>(w >> ((p & 7) * 8)) & 0xFFu;
                    const uint32_t slot = s_lut[c];
                    if (slot == 0xFFu) { alive = false; break; }
                    const DevSym S = slot < static_cast<uint32_t>(kLdsSyms) ? s_sym[slot] : ix.syms[slot];
                    RankAux q;
                    uint64_t c_before, c_upto, bh;
                    rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);
                    const uint64_t c_inside = c_upto - c_before;
                    if (c_inside == 0) { alive = false; break; }
                    lo = S.F + c_before;
                    hi = lo + c_inside - 1;
                    query = window_ei - (m - s) >= wsize;  // :315
                    if (query) window_ei = m - s;          // :322
                } else {
                    query = (m - 1) % wsize != 0;          // :328
                }
                if (query && hi - lo + 1 <= max_range) {   // :318,:331
                    uint64_t src, cnt;
                    if (marker_query(ix, lo, hi, &src, &cnt)) {
                        acc += cnt;
                        if (FILL) {
                            uint64_t *d = dst + (want - acc);
                            for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
                        }
                    }
                }
            }
            total = alive ? acc : 0;
            if (!alive) { lo = 1; hi = 0; }  // lf.clear(), :311-313
        }
        if (!FILL) {
            lo_out[i] = lo;
            hi_out[i] = hi;
            cnt_out[i + 1] = total;
        }
    }
}

// ---- greedy seeding (next-row f4): RowBowt::get_seeds_greedy_w_sample (rowbowt.hpp:222-256)
// reduced on the fly by locate_from_longest_seed's choice (rowbowt.hpp:669-677): per read, the
// first seed of strictly greatest length.  Seeds are maximal exact matches found right to left;
// the base that ends a seed is skipped.  A k-mer gather is attempted first; when it comes back
// empty the symbol that actually ends the seed is found with single reference steps.
template <typename P>
__device__ __forceinline__ bool lf_w_loc(const DevSym &S, const uint8_t *__restrict__ dense, uint32_t adv, uint64_t &lo, uint64_t &hi,
                                         uint64_t &k) {
    RankAux q;
    uint64_t c_before, c_upto, bh;
    rank_pair<P>(S, dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);
    const uint64_t c_inside = c_upto - c_before;
    if (c_inside == 0) return false;
    if (q.inside) k = k - adv;
    else k = pred_sample<P>(S, bh, q);
    lo = S.F + c_before;
    hi = lo + c_inside - 1;
    return true;
}

template <typename P>
__global__ __launch_bounds__(1024) void k_greedy_seed(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                     const uint64_t *__restrict__ off, const uint64_t N,
                                                     const uint64_t min_length, uint64_t *__restrict__ lo_out,
                                                     uint64_t *__restrict__ hi_out, uint64_t *__restrict__ qs_out,
                                                     uint64_t *__restrict__ qe_out, uint64_t *__restrict__ ss_out,
                                                     const uint32_t max_k) {
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    DevSym *s_tab = reinterpret_cast<DevSym *>(s_dyn);
    const uint32_t ksteps = ix.kmer_steps < max_k ? ix.kmer_steps : max_k;  // deepest level staged (the launcher sized the LDS for it)
    stage_tables(ix, s_tab, s_lut, s_lut2, ksteps >= 5);
    const uint32_t M = ix.nmajor;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint64_t beg = off[i], m = off[i + 1] - beg;
        const uint64_t first_k = ix.last_run_sample;  // rowbowt.hpp:230
        const uint64_t fhi = ix.n - 1;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;
        uint64_t k = first_k, pk = ~uint64_t(0), ei = m;
        uint64_t b_lo = 1, b_hi = 0, b_qs = 0, b_qe = 0, b_k = 0, b_len = 0;
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        uint64_t j = m;  // next symbol to consume is q[j-1]
        auto lf1 = [&](uint32_t c) -> bool {  // one reference step (rowbowt.hpp:235); an absent symbol is an empty range (:76)
            const uint32_t slot = s_lut[c];
            if (slot == 0xFFu) return false;
            if (slot < static_cast<uint32_t>(kLdsSyms)) { const DevSym S = s_tab[slot]; return lf_w_loc<P>(S, ix.dense, 1u, lo, hi, k); }
            return lf_w_loc<P>(ix.syms[slot], ix.dense, 1u, lo, hi, k);
        };
        // the longest k-mer (2..min(cap, kmer_steps) symbols, all with k-mer tables) ending at byte p; success
        // is identical to *len nested LF_w_loc calls
        auto lfk = [&](uint64_t p, uint32_t c, uint64_t cap, uint32_t *len) -> bool {
            *len = 0;
            const uint32_t m0 = s_lut2[c];
            if (m0 == 0xFFu || cap < 2 || ksteps < 2) return false;
            const uint32_t m1 = s_lut2[rd.at(p - 1)];
            if (m1 == 0xFFu) return false;
            uint32_t adv = 2, idx = kOff2 + m1 * M + m0;
            if (ksteps >= 3 && cap >= 3) {
                const uint32_t m2 = s_lut2[rd.at(p - 2)];
                if (m2 != 0xFFu) {
                    adv = 3;
                    idx = kOff3 + (m2 * M + m1) * M + m0;
                    if (ksteps >= 4 && cap >= 4) {
                        const uint32_t m3 = s_lut2[rd.at(p - 3)];
                        if (m3 != 0xFFu) {
                            adv = 4;
                            idx = kOff4 + ((m3 * M + m2) * M + m1) * M + m0;
                            if (ksteps >= 5 && cap >= 5) {
                                const uint32_t m4 = s_lut2[rd.at(p - 4)];
                                if (m4 != 0xFFu) { adv = 5; idx = kOff5 + (((m4 * M + m3) * M + m2) * M + m1) * M + m0; }
                            }
                        }
                    }
                }
            }
            *len = adv;
            const DevSym S = s_tab[idx];
            return lf_w_loc<P>(S, ix.dense, adv, lo, hi, k);
        };
        auto on_ok = [&](uint32_t adv) {
            j -= adv;
            plo = lo; phi = hi; pk = k;  // rowbowt.hpp:248-249
        };
        auto on_fail = [&]() {  // q[j-1] ends the seed q[j, ei)  (rowbowt.hpp:236-246; m-i == j here)
            if (ei - j >= min_length && ei - j > b_len) { b_len = ei - j; b_lo = plo; b_hi = phi; b_qs = j; b_qe = ei; b_k = pk; }
            k = first_k;
            lo = 0; hi = fhi; plo = 0; phi = fhi;
            j -= 1;      // skip the base that failed
            ei = j;
        };
        while (j > 0) {
            const uint64_t p = beg + j - 1;
            const uint32_t c = rd.at(p);
            // a fresh seed: the state after its first ftab_k symbols (range and toehold, searched from
            // first_k like here) is one gather in the device table; an empty entry means the word does not
            // occur and the steps below find where it stops
            if (j == ei && ix.ftab_k && j >= ix.ftab_k) {
                uint64_t idx = 0, pw = 1;
                bool all_major = true;
                for (uint32_t t = 0; t < ix.ftab_k; ++t) {
                    const uint32_t mm = s_lut2[rd.at(p - t)];
                    all_major = all_major && mm != 0xFFu;
                    idx += (mm & 3u) * pw;
                    pw *= M;
                }
                uint64_t flo, fhi2, fk;
                if (all_major && ftab_lookup<P>(ix, idx, flo, fhi2, fk) && flo <= fhi2) {
                    lo = flo; hi = fhi2; k = fk;
                    on_ok(ix.ftab_k);
                    continue;
                }
            }
            uint32_t len;
            if (lfk(p, c, j, &len)) { on_ok(len); continue; }
            if (len == 0) {
                if (lf1(c)) on_ok(1u); else on_fail();
                continue;
            }
            // the range died inside q[j-len, j) (lo/hi/k are untouched by a failed step): halve the window
            // until one symbol is left -- at most three more gathers -- that symbol is the failing base
            while (len > 1) {
                const uint32_t half = len / 2;
                const uint64_t p2 = beg + j - 1;
                const uint32_t c2 = rd.at(p2);
                uint32_t l2;
                const bool ok2 = half >= 2 ? lfk(p2, c2, half, &l2) : lf1(c2);
                if (ok2) { on_ok(half); len -= half; } else len = half;
            }
            on_fail();
        }
        if (ei >= min_length && ei > b_len) { b_len = ei; b_lo = plo; b_hi = phi; b_qs = 0; b_qe = ei; b_k = pk; }  // :252-254
        lo_out[i] = b_lo;
        hi_out[i] = b_hi;
        qs_out[i] = b_qs;
        qe_out[i] = b_qe;
        ss_out[i] = b_k;
    }
}

// ---- marker seeds (next-row f4): RowBowt::get_markers_greedy_seeding without an ftab
// (rowbowt.hpp:406-482; rb_markers' default path, rb_markers.cpp:411-413).  One record per call of
// the reference's callback: {range lo, range hi, q.first, seed_ei (= q.second + 1), first marker,
// one past last marker}; the markers of a seed are what every window query along it appended to
// mbuf (:437-441, :469-472), plus one more query when the seed ends or the read does (:445-447,
// :478-480).  FILL=false counts records and markers per read; FILL=true re-walks and writes.
// k-mer steps are used where no window query can fall inside them and a fresh seed takes its first
// ftab_k symbols from the device state table; a k-mer step that comes back empty is narrowed down with
// two more gathers so the failing base is the reference's.
// LOG (count pass only): also leave the marker-seed log behind (rbg_dev.h SeedLog) so that the fill pass copies instead of
// walking again; FILL with lg.base set: walk only the sequences the log lists as over quota.
template <typename P, bool FILL, bool LOG = false>
__global__ __launch_bounds__(1024) void k_marker_seeds(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                      const uint64_t *__restrict__ off, const uint64_t N,
                                                      const uint64_t wsize, const uint64_t max_range, const uint64_t ftab_k,
                                                      uint64_t *__restrict__ seed_cnt, uint64_t *__restrict__ mk_cnt,
                                                      const uint64_t *__restrict__ seed_off,
                                                      const uint64_t *__restrict__ mk_off,
                                                      uint64_t *__restrict__ seeds, uint64_t *__restrict__ mk, const uint32_t max_k,
                                                      const SeedLog lg) {
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    DevSym *s_tab = reinterpret_cast<DevSym *>(s_dyn);
    const uint32_t ksteps = ix.kmer_steps < max_k ? ix.kmer_steps : max_k;  // deepest level staged (the launcher sized the LDS for it)
    stage_tables(ix, s_tab, s_lut, s_lut2, ksteps >= 5);
    const uint32_t M = ix.nmajor;
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) { seed_cnt[0] = 0; mk_cnt[0] = 0; }
    const bool have_ma = ix.mk_nruns != 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const bool listed = FILL && lg.base != nullptr;       // fill pass after a logged count pass: only the sequences over quota
    const uint64_t Neff = listed ? static_cast<uint64_t>(lg.nsel[0]) : N;
    for (uint64_t j_ = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j_ < Neff; j_ += stride) {
        const uint64_t i = listed ? static_cast<uint64_t>(lg.nsel[4 + j_]) : j_;
        const uint64_t beg = off[i], m = off[i + 1] - beg;
        const uint64_t fhi = ix.n - 1;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;    // range, prev_range (:427-428)
        uint64_t window_ei = m, seed_ei = m;              // :434
        uint64_t ns = 0, tot = 0, mb_begin = 0;           // mbuf == markers [mb_begin, tot) of this read
        uint64_t *srec = FILL ? seeds + 6 * seed_off[i] : nullptr;
        const uint64_t mbase = FILL ? mk_off[i] : 0;
        // the log of this sequence (LOG): {ns, nw} | records | window entries
        unsigned char *lbase = LOG ? lg.base + i * lg.stride : nullptr;
        SeedLogRec<P> *lrec = reinterpret_cast<SeedLogRec<P> *>(lbase + 8);
        SeedLogWin *lwin = reinterpret_cast<SeedLogWin *>(lbase + 8 + static_cast<size_t>(lg.qs) * sizeof(SeedLogRec<P>));
        uint32_t nw = 0;
        bool lover = LOG && (m >> 32) != 0;               // (positions in the read are logged as 32 bits)
        auto update_mbuf = [&](uint64_t l, uint64_t h) {  // :437-441
            if (!have_ma || h - l + 1 > max_range) return;
            uint64_t src, cnt;
            if (!marker_query(ix, l, h, &src, &cnt)) return;
            if (FILL) {
                uint64_t *d = mk + mbase + tot;
                for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
            }
            if (LOG) {
                if (nw < lg.qw && ((src + cnt) >> 32) == 0) lwin[nw] = SeedLogWin{static_cast<uint32_t>(src), static_cast<uint32_t>(cnt)};
                else lover = true;
                ++nw;
            }
            tot += cnt;
        };
        auto emit = [&](uint64_t l, uint64_t h, uint64_t qs, uint64_t qe) {  // fn(range, (qs, qe-1), mbuf)
            if (FILL) {
                uint64_t *d = srec + 6 * ns;
                d[0] = l; d[1] = h; d[2] = qs; d[3] = qe; d[4] = mbase + mb_begin; d[5] = mbase + tot;
            }
            if (LOG) {
                if (ns < lg.qs && (tot >> 32) == 0)
                    lrec[ns] = SeedLogRec<P>{static_cast<P>(l), static_cast<P>(h), static_cast<uint32_t>(qs), static_cast<uint32_t>(qe),
                                             static_cast<uint32_t>(mb_begin), static_cast<uint32_t>(tot)};
                else lover = true;
            }
            ++ns;
        };
        auto finish = [&]() {
            if (!FILL) {
                seed_cnt[i + 1] = ns;
                mk_cnt[i + 1] = tot;
            }
            if (LOG) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(lbase);
                hdr[0] = lover ? kSeedLogOverflow : static_cast<uint32_t>(ns);
                hdr[1] = nw;
            }
        };
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        // one reference LF step on (lo,hi) with symbol c; false = empty range, (lo,hi) untouched
        auto lf1 = [&](uint32_t c) -> bool {
            const uint32_t slot = s_lut[c];
            if (slot == 0xFFu) return false;
            const DevSym S = slot < static_cast<uint32_t>(kLdsSyms) ? s_tab[slot] : ix.syms[slot];
            RankAux q;
            uint64_t c_before, c_upto, bh;
            rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);
            if (c_upto <= c_before) return false;
            lo = S.F + c_before;
            hi = lo + (c_upto - c_before) - 1;
            return true;
        };
        // the longest k-mer (2..min(cap, kmer_steps) symbols, all with k-mer tables) ending at byte p:
        // *len = its length (0: none applies); returns true when the range survived it
        auto lfk = [&](uint64_t p, uint32_t c, uint64_t cap, uint32_t *len) -> bool {
            *len = 0;
            const uint32_t m0 = s_lut2[c];
            if (m0 == 0xFFu || cap < 2 || ksteps < 2) return false;
            const uint32_t m1 = s_lut2[rd.at(p - 1)];
            if (m1 == 0xFFu) return false;
            uint32_t adv = 2, idx = kOff2 + m1 * M + m0;
            if (ksteps >= 3 && cap >= 3) {
                const uint32_t m2 = s_lut2[rd.at(p - 2)];
                if (m2 != 0xFFu) {
                    adv = 3;
                    idx = kOff3 + (m2 * M + m1) * M + m0;
                    if (ksteps >= 4 && cap >= 4) {
                        const uint32_t m3 = s_lut2[rd.at(p - 3)];
                        if (m3 != 0xFFu) {
                            adv = 4;
                            idx = kOff4 + ((m3 * M + m2) * M + m1) * M + m0;
                            if (ksteps >= 5 && cap >= 5) {
                                const uint32_t m4 = s_lut2[rd.at(p - 4)];
                                if (m4 != 0xFFu) { adv = 5; idx = kOff5 + (((m4 * M + m3) * M + m2) * M + m1) * M + m0; }
                            }
                        }
                    }
                }
            }
            *len = adv;
            const DevSym S = s_tab[idx];
            RankAux q;
            uint64_t c_before, c_upto, bh;
            rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);
            if (c_upto <= c_before) return false;
            lo = S.F + c_before;
            hi = lo + (c_upto - c_before) - 1;
            return true;
        };
        if (ftab_k) {
            // ---- with the ftab of k-mer size K (rb_markers --ftab): the reference's loop in its own
            // index i; search_ftab (:746-758) on the table build_ftab(K) makes for this index (:726-744)
            // is find_range of an ACGT-only k-mer, done here as K steps from the full range
            const uint64_t K = ftab_k;
            auto ftab_hit = [&](uint64_t e) -> bool {   // k-mer q[e-K, e); on a hit (lo,hi) is its range
                for (uint64_t t = e - K; t < e; ++t) {
                    const uint32_t c = rd.at(beg + t);
                    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
                }
                lo = 0; hi = fhi;
                uint64_t e2 = e;
                while (e2 > e - K) {
                    const uint64_t p = beg + e2 - 1;
                    const uint32_t c = rd.at(p);
                    uint32_t adv;
                    if (!lfk(p, c, e2 - (e - K), &adv)) {
                        if (adv) return false;       // a k-mer of the word is absent: so is the word
                        if (!lf1(c)) return false;
                        adv = 1;
                    }
                    e2 -= adv;
                }
                return true;
            };
            uint64_t i2 = 0;
            if (m >= K) {                                  // :430-433 (a shorter read makes the reference throw)
                if (ftab_hit(m)) i2 = K; else { lo = 0; hi = fhi; }
                plo = lo; phi = hi;
            }
            for (; i2 < m; ++i2) {
                if (lf1(rd.at(beg + m - i2 - 1))) {        // :443
                    if (window_ei - (m - i2 - 1) >= wsize) {   // :469-472
                        update_mbuf(lo, hi);
                        window_ei = m - i2 - 1;
                    }
                    plo = lo; phi = hi;                    // :473
                } else {                                   // :444-467
                    if (seed_ei - (m - i2) >= wsize) update_mbuf(plo, phi);
                    emit(plo, phi, m - i2, seed_ei);
                    mb_begin = tot;
                    plo = 0; phi = fhi;
                    seed_ei = m - i2 - 1;
                    window_ei = m - i2 - 1;
                    lo = 0; hi = fhi;
                    if (m - i2 - 1 >= K) {
                        // :454-464.  search_ftab answers an absent k-mer with {full_range(), 0} (:757), so the
                        // reference's test `range.first <= range.second` (:459) holds for a miss too and its loop
                        // always leaves on the first iteration: a hit continues from the k-mer's range, a miss
                        // from the FULL range, the K bases skipped either way (i += K, then the outer ++i)
                        if (!ftab_hit(m - i2 - 1)) { lo = 0; hi = fhi; }
                        i2 += K;                           // :460
                        plo = lo; phi = hi;                // :461
                    }
                }
            }
            if (hi >= lo && seed_ei - (m - i2) >= wsize) update_mbuf(lo, hi);   // :478-480
            emit(lo, hi, m - i2, seed_ei);                                      // :481
            finish();
            continue;
        }
        uint64_t j = m;  // m - i of the reference; the next symbol consumed is q[j-1]
        auto on_ok = [&](uint32_t adv) {              // adv symbols consumed, range still non-empty
            j -= adv;
            if (window_ei - j >= wsize) {             // :469-472 (m-i-1 == j after the step)
                update_mbuf(lo, hi);
                window_ei = j;
            }
            plo = lo; phi = hi;                       // :473
        };
        auto on_fail = [&]() {                        // q[j-1] empties the range: the seed q[j, seed_ei) ends (:444-466)
            if (seed_ei - j >= wsize) update_mbuf(plo, phi);
            emit(plo, phi, j, seed_ei);
            mb_begin = tot;
            plo = 0; phi = fhi; lo = 0; hi = fhi;
            j -= 1;                                   // the failing base is skipped
            seed_ei = j;
            window_ei = j;
        };
        while (j > 0) {
            const uint64_t p = beg + j - 1;
            const uint32_t c = rd.at(p);
            // symbols that may be consumed before the next window query fires (after the step that
            // leaves j' with j' + wsize <= window_ei, :469)
            uint64_t dist = j + wsize > window_ei ? j + wsize - window_ei : 1;
            if (dist == 0) dist = 1;
            const uint64_t cap = dist < j ? dist : j;
            // a fresh seed starts from the full range: the state after its first ftab_k symbols is one
            // gather in the device table (what k_find_range computes for that word; empty = the word does
            // not occur, then the steps below find where it stops)
            if (j == seed_ei && ix.ftab_k && cap >= ix.ftab_k) {
                uint64_t idx = 0, pw = 1;
                bool all_major = true;
                for (uint32_t t = 0; t < ix.ftab_k; ++t) {
                    const uint32_t mm = s_lut2[rd.at(p - t)];
                    all_major = all_major && mm != 0xFFu;
                    idx += (mm & 3u) * pw;
                    pw *= M;
                }
                uint64_t flo, fhi2, fk;
                if (all_major && ftab_lookup<P>(ix, idx, flo, fhi2, fk) && flo <= fhi2) {
                    lo = flo; hi = fhi2;
                    on_ok(ix.ftab_k);
                    continue;
                }
            }
            uint32_t len;
            if (lfk(p, c, cap, &len)) { on_ok(len); continue; }
            if (len == 0) {                           // no k-mer applies: one reference step (:443)
                if (lf1(c)) on_ok(1u); else on_fail();
                continue;
            }
            // the range died inside q[j-len, j) (lo/hi are untouched by a failed step): halve the window until
            // one symbol is left -- at most three more gathers -- that symbol is the failing base
            while (len > 1) {
                const uint32_t half = len / 2;
                const uint64_t p2 = beg + j - 1;
                const uint32_t c2 = rd.at(p2);
                uint32_t l2;
                const bool ok2 = half >= 2 ? lfk(p2, c2, half, &l2) : lf1(c2);
                if (ok2) { on_ok(half); len -= half; } else len = half;
            }
            on_fail();
        }
        if (hi >= lo && seed_ei >= wsize) update_mbuf(lo, hi);   // :478-480 (m-i == 0)
        emit(lo, hi, 0, seed_ei);                                // :481
        finish();
    }
}

// The fill pass from the log: eight lanes per sequence copy its seed records (48 bytes each) and the markers of its
// window queries into place; a sequence over quota is appended to the list behind lg.nsel for the walking kernel.
template <typename P>
__global__ __launch_bounds__(256) void k_marker_seeds_from_log(const DevIndex ix, const uint64_t N, const uint64_t *__restrict__ seed_off,
                                                               const uint64_t *__restrict__ mk_off, uint64_t *__restrict__ seeds,
                                                               uint64_t *__restrict__ mk, const SeedLog lg) {
    constexpr uint32_t G = 8;
    const uint64_t stride = (static_cast<uint64_t>(gridDim.x) * blockDim.x) / G;
    const uint32_t sub = threadIdx.x & (G - 1);
    for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) / G; i < N; i += stride) {
        const unsigned char *lbase = lg.base + i * lg.stride;
        const uint2 hdr = *reinterpret_cast<const uint2 *>(lbase);
        if (hdr.x == kSeedLogOverflow) {
            if (sub == 0) lg.nsel[4 + atomicAdd(lg.nsel, 1u)] = static_cast<uint32_t>(i);
            continue;
        }
        const SeedLogRec<P> *lrec = reinterpret_cast<const SeedLogRec<P> *>(lbase + 8);
        const SeedLogWin *lwin = reinterpret_cast<const SeedLogWin *>(lbase + 8 + static_cast<size_t>(lg.qs) * sizeof(SeedLogRec<P>));
        const uint64_t mbase = mk_off[i];
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(seeds + 6 * seed_off[i]);
        for (uint32_t t = sub; t < hdr.x; t += G) {
            const SeedLogRec<P> r = lrec[t];
            dst[3 * t + 0] = make_ulonglong2(static_cast<uint64_t>(r.lo), static_cast<uint64_t>(r.hi));
            dst[3 * t + 1] = make_ulonglong2(r.qs, r.qe);
            dst[3 * t + 2] = make_ulonglong2(mbase + r.mb, mbase + r.me);
        }
        uint64_t at = mbase;
        for (uint32_t w = 0; w < hdr.y; ++w) {
            const SeedLogWin q = lwin[w];
            for (uint32_t t = sub; t < q.cnt; t += G) mk[at + t] = ix.mk_vals[q.src + t];
            at += q.cnt;
        }
    }
}

// ---- single LF step for N (range, symbol) triples: RowBowt::LF(range_t, uint8_t), rowbowt.hpp:74-88
template <typename P>
__global__ __launch_bounds__(256) void k_lf(const DevIndex ix, const uint64_t *__restrict__ lo_in,
                                            const uint64_t *__restrict__ hi_in, const uint8_t *__restrict__ sym,
                                            const uint64_t N, uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint64_t lo = lo_in[i], hi = hi_in[i];
        const uint32_t slot = ix.lut[sym[i]];
        uint64_t nlo = 1, nhi = 0;
        // hi >= n is outside rle_string::rank's domain (assert(i<=n), rle_string.hpp:132): answer {1,0}
        if (slot != 0xFFu && hi < ix.n && lo <= hi + 1) {
            const DevSym S = ix.syms[slot];
            RankAux q;
            uint64_t c_before, c_upto, bh;
            rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);
            if (c_upto > c_before) { nlo = S.F + c_before; nhi = nlo + (c_upto - c_before) - 1; }
        }
        lo_out[i] = nlo;
        hi_out[i] = nhi;
    }
}

__global__ __launch_bounds__(256) void k_count(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                               const uint64_t N, uint64_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride)
        out[i] = hi[i] >= lo[i] ? hi[i] - lo[i] + 1 : 0;  // RowBowt::count, rowbowt.hpp:266-269
}

}  // namespace

int launch_markers_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                        uint64_t *mk_off, void *tmp, size_t tmp_bytes, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_markers_count, dim3(grid_for(cfg, N)), dim3(cfg.block_threads), 0, st, ix, lo, hi, N, mk_off);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    return scan_in_place(mk_off + 1, N, tmp, tmp_bytes, st);
}

int launch_markers_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                        const uint64_t *mk_off, uint64_t *mk, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_markers_fill, dim3(grid_for(cfg, N)), dim3(cfg.block_threads), 0, st, ix, lo, hi, N, mk_off, mk);
    return static_cast<int>(hipGetLastError());
}


int launch_find_range_markers_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off,
                                   uint64_t N, uint64_t wsize, uint64_t max_range, uint64_t *lo, uint64_t *hi,
                                   uint64_t *mk_off, void *tmp, size_t tmp_bytes, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ix.layout == 2) {   // run-indexed layout: the cooperative kernel (k_runs_seeds.hip)
        const int rc2 = launch_find_range_markers_runs(ix, cfg, seqs, off, N, wsize, max_range, lo, hi, mk_off, nullptr, nullptr, false, stream);
        if (rc2) return rc2;
        return scan_in_place(mk_off + 1, N, tmp, tmp_bytes, st);
    }
    const dim3 grid(grid_for(cfg, N)), block(cfg.block_threads);
    if (ix.pos_bytes == 4)
        hipLaunchKernelGGL((k_find_range_markers<uint32_t, false>), grid, block, 0, st, ix, seqs, off, N, wsize, max_range, lo, hi, mk_off, nullptr, nullptr);
    else
        hipLaunchKernelGGL((k_find_range_markers<uint64_t, false>), grid, block, 0, st, ix, seqs, off, N, wsize, max_range, lo, hi, mk_off, nullptr, nullptr);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    return scan_in_place(mk_off + 1, N, tmp, tmp_bytes, st);
}

int launch_find_range_markers_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off,
                                   uint64_t N, uint64_t wsize, uint64_t max_range, const uint64_t *mk_off, uint64_t *mk,
                                   void *stream) {
    if (N == 0) return 0;
    if (ix.layout == 2) return launch_find_range_markers_runs(ix, cfg, seqs, off, N, wsize, max_range, nullptr, nullptr, nullptr, mk_off, mk, true, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(cfg, N)), block(cfg.block_threads);
    if (ix.pos_bytes == 4)
        hipLaunchKernelGGL((k_find_range_markers<uint32_t, true>), grid, block, 0, st, ix, seqs, off, N, wsize, max_range, nullptr, nullptr, nullptr, mk_off, mk);
    else
        hipLaunchKernelGGL((k_find_range_markers<uint64_t, true>), grid, block, 0, st, ix, seqs, off, N, wsize, max_range, nullptr, nullptr, nullptr, mk_off, mk);
    return static_cast<int>(hipGetLastError());
}

int launch_marker_seeds_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range, uint64_t ftab_k, uint64_t *seed_off, uint64_t *mk_off, void *tmp,
                             size_t tmp_bytes, void *stream, void *log, size_t log_bytes, unsigned long long *stats) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    SeedLog lg = make_seed_log(log, log_bytes, N, ix.pos_bytes);   // base == nullptr: no log (two walks)
    if (stats) {
        if (ix.layout != 2 || lg.base || ftab_k) return static_cast<int>(hipErrorNotSupported);   // the instrumented walks: run-indexed layout, no log, no --ftab
        lg.stats = stats;
    }
    int rc;
    if (ix.layout == 2) {   // run-indexed layout: k_runs_seeds.hip
        rc = launch_marker_seeds_runs(ix, cfg, seqs, off, N, wsize, max_range, seed_off, mk_off, nullptr, nullptr, nullptr, nullptr, false, stream, lg, ftab_k);
    } else {
#define RBG_MSP(PT, LG)                                                                                                                  \
    do {                                                                                                                                  \
        auto kern = k_marker_seeds<PT, false, LG>;                                                                                        \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);                                                            \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, seed_off, mk_off, nullptr, nullptr, nullptr, \
                           nullptr, kSeedKmerLevel, lg);                                                                                  \
    } while (0)
        if (ix.pos_bytes == 4) { if (lg.base) RBG_MSP(uint32_t, true); else RBG_MSP(uint32_t, false); }
        else { if (lg.base) RBG_MSP(uint64_t, true); else RBG_MSP(uint64_t, false); }
#undef RBG_MSP
        rc = static_cast<int>(hipGetLastError());
    }
    if (rc) return rc;
    rc = scan_in_place(seed_off + 1, N, tmp, tmp_bytes, st);
    if (rc) return rc;
    return scan_in_place(mk_off + 1, N, tmp, tmp_bytes, st);
}

int launch_marker_seeds_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range, uint64_t ftab_k, const uint64_t *seed_off, const uint64_t *mk_off,
                             uint64_t *seeds, uint64_t *mk, void *stream, void *log, size_t log_bytes, unsigned long long *stats) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SeedLog lg = make_seed_log(log, log_bytes, N, ix.pos_bytes);
    if (stats) {
        if (ix.layout != 2 || lg.base || ftab_k) return static_cast<int>(hipErrorNotSupported);
        lg.stats = stats;
    }
    if (reinterpret_cast<uintptr_t>(seeds) & 15) lg.base = nullptr;   // (the copy stores records as 16-byte pieces)
    if (lg.base) {
        // copy what the count pass logged; the sequences over quota end up listed behind lg.nsel and are walked below
        int rc = static_cast<int>(hipMemsetAsync(lg.nsel, 0, 16, st));
        if (rc) return rc;
        const uint64_t threads = N * 8;
        const int grid = static_cast<int>(std::min<uint64_t>((threads + 255) / 256, 256ull * 64));
        if (ix.pos_bytes == 4) hipLaunchKernelGGL((k_marker_seeds_from_log<uint32_t>), dim3(grid), dim3(256), 0, st, ix, N, seed_off, mk_off, seeds, mk, lg);
        else hipLaunchKernelGGL((k_marker_seeds_from_log<uint64_t>), dim3(grid), dim3(256), 0, st, ix, N, seed_off, mk_off, seeds, mk, lg);
        rc = static_cast<int>(hipGetLastError());
        if (rc) return rc;
    }
    if (ix.layout == 2)
        return launch_marker_seeds_runs(ix, cfg, seqs, off, N, wsize, max_range, nullptr, nullptr, seed_off, mk_off, seeds, mk, true, stream, lg, ftab_k);
#define RBG_MSF(PT)                                                                                                                       \
    do {                                                                                                                                  \
        auto kern = k_marker_seeds<PT, true, false>;                                                                                      \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, lg.base ? 256 : 0, kSeedKmerLevel);                                            \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, nullptr, nullptr, seed_off, mk_off, seeds, mk, \
                           kSeedKmerLevel, lg);                                                                                           \
    } while (0)
    if (ix.pos_bytes == 4) RBG_MSF(uint32_t); else RBG_MSF(uint64_t);
#undef RBG_MSF
    return static_cast<int>(hipGetLastError());
}

int launch_greedy_seed(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                       uint64_t min_length, uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, void *stream, unsigned long long *stats) {
    if (N == 0) return 0;
    if (ix.layout == 2) return launch_greedy_seed_runs(ix, cfg, seqs, off, N, min_length, lo, hi, qs, qe, ss, stream, stats);
    if (stats) return static_cast<int>(hipErrorNotSupported);   // (instrumented on the run-indexed layout only)
    hipStream_t st = static_cast<hipStream_t>(stream);
#define RBG_GS(PT)                                                                                                        \
    do {                                                                                                                  \
        auto kern = k_greedy_seed<PT>;                                                                                    \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);                                            \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, min_length, lo, hi, qs, qe, ss, kSeedKmerLevel); \
    } while (0)
    if (ix.pos_bytes == 4) RBG_GS(uint32_t); else RBG_GS(uint64_t);
#undef RBG_GS
    return static_cast<int>(hipGetLastError());
}

int launch_lf(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym,
              uint64_t N, uint64_t *lo_out, uint64_t *hi_out, void *stream) {
    if (N == 0) return 0;
    if (ix.layout == 2) return launch_lf_runs(ix, cfg, lo, hi, sym, N, lo_out, hi_out, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(cfg, N)), block(cfg.block_threads);
    if (ix.pos_bytes == 4) hipLaunchKernelGGL((k_lf<uint32_t>), grid, block, 0, st, ix, lo, hi, sym, N, lo_out, hi_out);
    else hipLaunchKernelGGL((k_lf<uint64_t>), grid, block, 0, st, ix, lo, hi, sym, N, lo_out, hi_out);
    return static_cast<int>(hipGetLastError());
}

int launch_count_from_ranges(const uint64_t *lo, const uint64_t *hi, uint64_t N, uint64_t *count, void *stream) {
    if (N == 0) return 0;
    LaunchCfg cfg;
    hipLaunchKernelGGL(k_count, dim3(grid_for(cfg, N)), dim3(256), 0, static_cast<hipStream_t>(stream), lo, hi, N, count);
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
