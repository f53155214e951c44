// k_search.hip -- gfx950 (MI355X, wave64) kernels for the rb_align hot path: the backward search.
//
//  K1/K2  k_find_range<P,TOEHOLD>  batched backward search, one lane walks one read.
//         Replaces RowBowt::find_range (rowbowt.hpp:121-131) / find_range_w_toehold (:169-184),
//         i.e. m x { RowBowt::LF :74-88 -> 2 x rle_string::rank rle_string.hpp:131-161 } and the
//         toehold update of LF_w_loc (:555-573).
//  K3     k_locate_fill<P>         phi chains, ToeholdSA::locate_range toehold_sa.hpp:37-49 / phi :56-72.
//  K4     k_markers_*              MarkerArray::at_range behind RowBowt::markers_at rowbowt.hpp:282-285.
//  also   k_find_range_markers, k_greedy_seed, k_marker_seeds (windowed / greedy seeding, rowbowt.hpp:222-339, :406-482),
//         k_pack_reads + k_find_range_packed (opt-in 2-bit reads), k_build_rank_slots / k_build_phi_slots (tables at load).
//
// Integer gather kernels: no MFMA (nothing here is a contraction).  The bound is HBM / fabric
// transactions per LF step, so the layout (rbg_dev.h) makes one rank = ONE aligned 4-word slot
// (direct-addressed by position >> shift) that already answers the rank, and both ranks of a step
// share that slot whenever lo and hi+1 fall in the same bucket.
#include "rbg_device.hpp"

namespace rbg {
namespace {

// The record table lives in dynamic LDS: kTabMax records (17 KB) up to 4-mer steps, kTab5 (66 KB) with the
// 5-mer level, then launched as 1024-thread workgroups so that two of them still give 8 waves per SIMD.
// STATS = the instrumented instantiation (rbg_find_range_stats_dev): the same walk, plus counts of what it
// touched (rbg_dev.h SearchStat) summed into stats[]; bench.py derives the bytes of the algorithm AS RUN from
// them.  The timed launches are the STATS = false ones (no counter exists in them).
template <typename P, bool TOEHOLD, bool USE_FTAB, bool STATS = false>
__global__ __launch_bounds__(1024, 8) void k_find_range(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                    const uint64_t *__restrict__ off, const uint64_t N,
                                                    uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                    uint64_t *__restrict__ ss_out, const uint32_t *__restrict__ sel,
                                                    const uint32_t *__restrict__ nsel,
                                                    unsigned long long *__restrict__ stats = nullptr) {
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    DevSym *s_tab = reinterpret_cast<DevSym *>(s_dyn);
    // sel != nullptr: only the reads sel[0 .. *nsel) (the ones the packed path hands back)
    const uint64_t Neff = sel ? static_cast<uint64_t>(*nsel) : N;
    if (Neff == 0) return;
    stage_tables(ix, s_tab, s_lut, s_lut2, true);
    const uint32_t M = ix.nmajor;
    const uint32_t ksteps = ix.kmer_steps;

    // reads handled by this lane follow from the loop bounds; only the matches and their widths are accumulated
    unsigned long long c_occ = 0;
    uint32_t c_matched = 0;
    unsigned long long st[kStatSearchN] = {0, 0, 0, 0, 0, 0, 0, 0};  // STATS only (dead code otherwise)
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t first_ = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint64_t j_ = first_; j_ < Neff; j_ += stride) {
        uint64_t beg, p;
        {
            const uint64_t i0 = sel ? static_cast<uint64_t>(sel[j_]) : j_;
            beg = off[i0];
            p = off[i0 + 1];
        }
        const uint64_t p_end = p;
        uint64_t p_min = p;  // STATS: lowest read byte fetched (the k-mer look-ahead reads left of the consumed symbols)
        uint64_t lo = 0, hi = ix.n - 1;  // full_range(), rowbowt.hpp:115-118
        uint64_t k = TOEHOLD ? ix.last_run_sample : 0;
        // The toehold only flows forward through "k - adv" (row hi carries the symbol); a step that
        // re-samples overwrites it.  While the range is still wide almost every step re-samples
        // (bwt[hi] is a random symbol), so the two gathers of a re-sample (run ordinal, sample) are
        // deferred to the end of the read: only the LAST re-sample of a read is ever used, and what the
        // later steps subtract from it is accumulated in k itself (k = 0 at the re-sample, k -= adv after
        // it, toehold = sample + k in wrapping 64-bit arithmetic, exactly the reference's chain of k - 1).
        bool pend = false;
        uint32_t pend_tab = 0;   // which record to re-sample from: s_tab index, or kHbmRec | symbol slot
        uint64_t pend_b = 0;
        uint32_t pend_v = 0;     // runs before the position inside the bucket, or (overflow bucket) the position's offset in it
        bool pend_abs = false;
        // the two gathers of a deferred re-sample: run ordinal from `ord`, then the run's sample
        auto resample = [&]() -> uint64_t {
            const DevSym *rec = (pend_tab & kHbmRec) ? ix.syms + (pend_tab & ~kHbmRec) : s_tab + pend_tab;
            const void *samp = rec->samp;
            const uint64_t j = pred_run<P>(*rec, pend_b, pend_abs, pend_v);
            return static_cast<uint64_t>(as_global<P>(samp)[j]);
        };
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        bool alive = true;
        // ftab (rowbowt.hpp:124-125, :745-758): while the range is still wide a step costs two slot
        // gathers (lo and hi+1 fall in different buckets) plus toehold re-sampling; the state after the
        // last ftab_k symbols is looked up with one gather instead.  Result-neutral: the table holds
        // what this very kernel computes for that word.
        if (USE_FTAB && ix.ftab_k && p - beg >= ix.ftab_k) {
            uint64_t idx = 0, pw = 1;
            bool all_major = true;
            for (uint32_t t = 1; t <= ix.ftab_k; ++t) {  // right to left; leftmost symbol = most significant digit
                const uint32_t mm = s_lut2[rd.at(p - t)];
                all_major = all_major && mm != 0xFFu;
                idx += (mm & 3u) * pw;
                pw *= M;
            }
            uint64_t flo, fhi2, fk;
            if (STATS) { p_min = p - ix.ftab_k; if (all_major) st[kStFtab] += 1; }
            if (all_major && ftab_lookup<P>(ix, idx, flo, fhi2, fk)) {
                lo = flo; hi = fhi2;
                if (TOEHOLD) k = fk;
                p -= ix.ftab_k;
                if (STATS) st[kStSymbols] += ix.ftab_k;
                if (hi < lo) { alive = false; p = beg; }
            }
        }
        // one LF step (or several nested ones) through the record S; false = range emptied
        auto step = [&](const DevSym &S, uint32_t adv, uint32_t tab) -> bool {
            RankAux q;
            uint64_t c_before, c_upto, bh;
            if (STATS) {
                RankAux pa;
                const bool two = (lo >> S.shift) != ((hi + 1) >> S.shift);
                rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q, &pa);
                st[kStSteps] += 1;
                st[kStSlots] += two ? 2 : 1;
                st[kStDense] += (pa.dense ? 1 : 0) + ((two || !pa.dense) && q.dense ? 1 : 0);
                st[kStSearch] += (pa.ovf ? 1 : 0) + (q.ovf ? 1 : 0);
                st[kStSymbols] += adv;
            } else {
                rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);  // rowbowt.hpp:79,83
            }
            const uint64_t c_inside = c_upto - c_before;
            if (c_inside == 0) return false;                               // rowbowt.hpp:85
            if (TOEHOLD) {
                // LF_w_loc, rowbowt.hpp:559-566.  Either hi holds the symbol (bwt_[hi]==c -> k-1 per
                // nested step), or the last run starting before hi ends before hi and its last row
                // is select(rank(hi,c)-1,c), whose run-end sample is samples_last_[run] (resp. SA-adv).
                if (q.inside) {
                    k = k - adv;   // with a re-sample pending, k is minus the distance walked since it
                } else {
                    pend = true;
                    pend_tab = tab;
                    pend_b = bh;
                    pend_abs = q.ovf;
                    pend_v = q.nbefore;
                    k = 0;
                }
            }
            lo = S.F + c_before;           // rowbowt.hpp:86
            hi = lo + c_inside - 1;        // rowbowt.hpp:87
            return true;
        };
        while (p > beg) {  // right-to-left over the read (rowbowt.hpp:127-129, :175-181)
            --p;
            const uint32_t c = rd.at(p);
            if (STATS && p < p_min) p_min = p;
            // Up to five reference iterations in one gather: when this symbol and its left
            // neighbours all have k-mer tables, LF(LF(LF(range,x0),x1),x2) == F3[x2x1x0] + rank3(.),
            // and the toehold after the nested LF_w_loc calls is k-adv if row hi carries the k-mer,
            // else the k-mer run sample.  An empty result is {1,0} whichever of the
            // nested steps emptied it.  Otherwise: one reference step (rowbowt.hpp:74-88, :555-573).
            uint32_t adv = 1, idx = 0;
            const uint32_t m0 = s_lut2[c];
            if (m0 != 0xFFu) {
                // extend to the left while the symbols have k-mer tables: the table index is the k-mer read as
                // a base-M number whose least significant digit is the symbol next to the suffix
                uint32_t acc = m0, pw = M;
#pragma unroll 1  // unrolled, the five table-index computations stay live together and spill
                for (uint32_t t = 1; t < 5; ++t) {
                    if (t >= ksteps || p < beg + t) break;
                    const uint32_t mm = s_lut2[rd.at(p - t)];
                    if (STATS && p - t < p_min) p_min = p - t;
                    if (mm == 0xFFu) break;
                    acc += mm * pw;
                    pw *= M;
                    adv = t + 1;
                }
                idx = (adv == 5 ? kOff5 : adv == 4 ? kOff4 : adv == 3 ? kOff3 : kOff2) + acc;
            }
            bool ok;
            if (adv == 1) {
                const uint32_t slot = s_lut[c];
                if (slot == 0xFFu) { alive = false; break; }  // symbol absent: f_[c] >= f_[c+1], rowbowt.hpp:76
                if (slot < static_cast<uint32_t>(kLdsSyms)) { const DevSym Sc = s_tab[slot]; ok = step(Sc, 1u, slot); }
                else ok = step(ix.syms[slot], 1u, kHbmRec | slot);  // rare symbols: record read from HBM field by field
            } else {
                // copy the 48-byte record with three wide LDS reads: reading it field by field makes
                // every lane hit the same two banks (records are 16 dwords apart)
                const DevSym Sc = s_tab[idx];
                ok = step(Sc, adv, idx);
            }
            if (!ok) { alive = false; break; }
            p -= adv - 1;                  // the left neighbours are consumed too
        }
        if (TOEHOLD && alive && pend) k += resample();
        if (STATS) {
            if (TOEHOLD && alive && pend) st[kStResample] += 1;
            if (p_end > p_min) st[kStChunks] += ((p_end - 1) >> 4) - (p_min >> 4) + 1;  // aligned 16-byte read chunks fetched
        }
        if (!alive) { lo = 1; hi = 0; k = 0; }  // {1,0}; LFData::clear rowbowt.hpp:153-159
        const uint64_t i = sel ? static_cast<uint64_t>(sel[j_]) : j_;  // (re-read rather than kept live through the search)
        lo_out[i] = lo;
        hi_out[i] = hi;
        if (TOEHOLD) ss_out[i] = k;
        if (alive) { c_matched += 1; c_occ += hi - lo + 1; }
    }
    unsigned long long c_reads = first_ < Neff ? (Neff - first_ + stride - 1) / stride : 0;
    c_reads = wave_sum(c_reads);
    const unsigned long long w_matched = wave_sum(static_cast<unsigned long long>(c_matched));
    c_occ = wave_sum(c_occ);
    if ((threadIdx.x & (kWave - 1)) == 0 && c_reads) {
        atomicAdd(&ix.counters[0], c_reads);
        if (w_matched) atomicAdd(&ix.counters[1], w_matched);
        if (c_occ) atomicAdd(&ix.counters[2], c_occ);
    }
    if (STATS) {
#pragma unroll
        for (int t = 0; t < kStatSearchN; ++t) {
            const unsigned long long v = wave_sum(st[t]);
            if ((threadIdx.x & (kWave - 1)) == 0 && v) atomicAdd(&stats[t], v);
        }
    }
}

// ---- packed reads (2 bits per symbol) ------------------------------------------------------------
// One lane per read fetching its own bytes costs 7 uncoalesced 16-byte requests per 100 bp read,
// a fifth of all L2 requests of K1/K2.  k_pack_reads reads the byte stream once with
// coalesced loads and rewrites each read as 2-bit codes of the major alphabet in the order the
// search consumes them (symbol q[m-1-t] at bits [2t, 2t+2)), 64 symbols per 16-byte chunk.  The
// packed search kernel then needs ceil(m/64) requests per read, and a k-mer step's table index is the
// next 2k bits of the stream (no per-symbol LUT lookups).  A read with a symbol outside the major
// alphabet is flagged and listed in sel[] for the byte kernel.
constexpr int kPackLdsBytes = 40 * 1024;

__global__ __launch_bounds__(256) void k_pack_count(const uint64_t *__restrict__ off, const uint64_t N,
                                                    uint64_t *__restrict__ chunk_off, uint32_t *__restrict__ nsel) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride)
        chunk_off[i + 1] = (off[i + 1] - off[i] + 63) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) { chunk_off[0] = 0; *nsel = 0; }
}

__global__ __launch_bounds__(256) void k_pack_reads(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                    const uint64_t *__restrict__ off, const uint64_t N,
                                                    const uint64_t *__restrict__ chunk_off, uint2 *__restrict__ meta,
                                                    uint4 *__restrict__ chunks, uint32_t *__restrict__ sel,
                                                    uint32_t *__restrict__ nsel) {
    __shared__ uint4 s_raw[kPackLdsBytes / 16];
    __shared__ uint8_t s_lut2[256];
    const bool usable = ix.nmajor == 4;  // 2-bit codes are indices into a 4-symbol major alphabet
    for (int t = threadIdx.x; t < 256; t += blockDim.x) s_lut2[t] = usable ? ix.lut2[t] : 0xFFu;
    const uint4 *__restrict__ gsrc = reinterpret_cast<const uint4 *>(seqs);
    const uint64_t ngroups = (N + blockDim.x - 1) / blockDim.x;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint64_t i0 = g * blockDim.x;
        const uint64_t i1 = i0 + blockDim.x < N ? i0 + blockDim.x : N;
        const uint64_t b0 = off[i0], b1 = off[i1];
        const uint64_t a0 = b0 & ~uint64_t(15);
        const bool fits = b1 - a0 <= static_cast<uint64_t>(kPackLdsBytes);
        __syncthreads();  // previous group's readers are done with s_raw (and s_lut2 is staged)
        if (fits) {
            const uint64_t nch = (b1 - a0 + 15) >> 4;
            for (uint64_t c = threadIdx.x; c < nch; c += blockDim.x) s_raw[c] = gsrc[(a0 >> 4) + c];
        }
        __syncthreads();
        const uint64_t i = i0 + threadIdx.x;
        if (i < i1) {
            const uint64_t beg = off[i], m = off[i + 1] - beg;
            const uint8_t *lsrc = reinterpret_cast<const uint8_t *>(s_raw) + (beg - a0);
            uint4 *dst = chunks + chunk_off[i];
            uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            uint32_t bad = m >= 0x80000000ull;  // the packed length field is 31 bits: such a read goes to the byte kernel
            for (uint64_t t0 = 0; t0 < m && !bad; t0 += 16) {  // 16 symbols = one 32-bit word
                uint32_t acc = 0;
                const uint32_t lim = m - t0 < 16 ? static_cast<uint32_t>(m - t0) : 16u;
                for (uint32_t u = 0; u < lim; ++u) {
                    const uint64_t pos = m - 1 - (t0 + u);
                    const uint32_t code = s_lut2[fits ? lsrc[pos] : seqs[beg + pos]];
                    bad |= code == 0xFFu;
                    acc |= (code & 3u) << (2 * u);
                }
                const uint32_t wi = static_cast<uint32_t>(t0 >> 4) & 3u;
                if (wi == 0) w0 = acc; else if (wi == 1) w1 = acc; else if (wi == 2) w2 = acc; else w3 = acc;
                if (wi == 3 || t0 + 16 >= m) {
                    dst[t0 >> 6] = make_uint4(w0, w1, w2, w3);
                    w0 = w1 = w2 = w3 = 0;
                }
            }
            meta[i] = make_uint2(static_cast<uint32_t>(chunk_off[i]), bad ? 0x80000000u : static_cast<uint32_t>(m));
            if (bad) sel[atomicAdd(nsel, 1u)] = static_cast<uint32_t>(i);
        }
    }
}

// per-lane reader of a packed read: take(nb) returns the next nb (<= 32) bits
struct BitStream {
    const uint4 *__restrict__ cp;
    uint4 w;
    uint32_t widx;    // next word of w to hand out; 4 = fetch the next chunk first
    uint32_t navail;
    uint64_t sr;
    __device__ __forceinline__ uint32_t next_word() {
        if (widx == 4) { w = *cp++; widx = 0; }
        const uint32_t v = widx == 0 ? w.x : widx == 1 ? w.y : widx == 2 ? w.z : w.w;
        ++widx;
        return v;
    }
    __device__ __forceinline__ uint32_t take(uint32_t nb) {
        if (navail < nb) {
            sr |= static_cast<uint64_t>(next_word()) << navail;
            navail += 32;
        }
        const uint32_t v = static_cast<uint32_t>(sr & ((uint64_t(1) << nb) - 1));
        sr >>= nb;
        navail -= nb;
        return v;
    }
};

// k_find_range over packed reads: the same steps in the same order as the byte kernel takes for a
// read made of major symbols only (ftab word, then min(kmer_steps, remaining) symbols per gather),
// so ranges and toeholds are identical; flagged reads are left to the byte kernel (sel list).
template <typename P, bool TOEHOLD>
__global__ __launch_bounds__(1024, 8) void k_find_range_packed(const DevIndex ix, const uint2 *__restrict__ meta,
                                                           const uint4 *__restrict__ chunks, const uint64_t N,
                                                           uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                           uint64_t *__restrict__ ss_out) {
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    __shared__ uint8_t s_mslot[4];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    DevSym *s_tab = reinterpret_cast<DevSym *>(s_dyn);
    stage_tables(ix, s_tab, s_lut, s_lut2, true);
    for (int t = threadIdx.x; t < 256; t += blockDim.x)
        if (s_lut2[t] != 0xFFu) s_mslot[s_lut2[t] & 3u] = s_lut[t];  // major index -> symbol slot
    __syncthreads();
    const uint32_t ksteps = ix.kmer_steps;

    unsigned long long c_reads = 0, c_matched = 0, c_occ = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint2 mt = meta[i];
        if (mt.y & 0x80000000u) continue;  // has a non-major symbol: byte kernel
        uint32_t r = mt.y;                 // symbols still to consume
        uint64_t lo = 0, hi = ix.n - 1;    // full_range(), rowbowt.hpp:115-118
        uint64_t k = TOEHOLD ? ix.last_run_sample : 0;
        bool pend = false;                 // deferred toehold re-sample, as in k_find_range
        uint32_t pend_tab = 0;
        uint64_t pend_b = 0;
        uint32_t pend_v = 0;
        bool pend_abs = false;
        auto resample = [&]() -> uint64_t {
            const DevSym *rec = (pend_tab & kHbmRec) ? ix.syms + (pend_tab & ~kHbmRec) : s_tab + pend_tab;
            const void *samp = rec->samp;
            const uint64_t j = pred_run<P>(*rec, pend_b, pend_abs, pend_v);
            return static_cast<uint64_t>(as_global<P>(samp)[j]);
        };
        BitStream bs{chunks + mt.x, make_uint4(0, 0, 0, 0), 4u, 0u, 0ull};
        bool alive = true;
        if (ix.ftab_k && r >= ix.ftab_k) {  // the first ftab_k symbols are the low 2*ftab_k bits
            BitStream probe = bs;           // consumed only if the entry is usable
            const uint64_t idx = probe.take(2 * ix.ftab_k);
            uint64_t flo, fhi2, fk;
            if (ftab_lookup<P>(ix, idx, flo, fhi2, fk)) {
                bs = probe;
                lo = flo; hi = fhi2;
                if (TOEHOLD) k = fk;
                r -= ix.ftab_k;
                if (hi < lo) { alive = false; r = 0; }
            }
        }
        auto step = [&](const DevSym &S, uint32_t adv, uint32_t tab) -> bool {
            RankAux q;
            uint64_t c_before, c_upto, bh;
            rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);      // rowbowt.hpp:79,83
            const uint64_t c_inside = c_upto - c_before;
            if (c_inside == 0) return false;                               // rowbowt.hpp:85
            if (TOEHOLD) {                                                 // LF_w_loc, rowbowt.hpp:559-566
                if (q.inside) {
                    k = k - adv;   // with a re-sample pending, k is minus the distance walked since it
                } else {
                    pend = true;
                    pend_tab = tab;
                    pend_b = bh;
                    pend_abs = q.ovf;
                    pend_v = q.nbefore;
                    k = 0;
                }
            }
            lo = S.F + c_before;           // rowbowt.hpp:86
            hi = lo + c_inside - 1;        // rowbowt.hpp:87
            return true;
        };
        while (r > 0) {
            const uint32_t a = r < ksteps ? r : ksteps;
            const uint32_t v = bs.take(2 * a);
            bool ok;
            if (a == 1) {
                const uint32_t slot = s_mslot[v];
                if (slot < static_cast<uint32_t>(kLdsSyms)) { const DevSym Sc = s_tab[slot]; ok = step(Sc, 1u, slot); }
                else ok = step(ix.syms[slot], 1u, kHbmRec | slot);
            } else {
                const uint32_t idx = (a == 5 ? kOff5 : a == 4 ? kOff4 : a == 3 ? kOff3 : kOff2) + v;
                const DevSym Sc = s_tab[idx];
                ok = step(Sc, a, idx);
            }
            if (!ok) { alive = false; break; }
            r -= a;
        }
        if (TOEHOLD && alive && pend) k += resample();
        if (!alive) { lo = 1; hi = 0; k = 0; }
        lo_out[i] = lo;
        hi_out[i] = hi;
        if (TOEHOLD) ss_out[i] = k;
        c_reads += 1;
        if (alive) { c_matched += 1; c_occ += hi - lo + 1; }
    }
    c_reads = wave_sum(c_reads);
    c_matched = wave_sum(c_matched);
    c_occ = wave_sum(c_occ);
    if ((threadIdx.x & (kWave - 1)) == 0 && c_reads) {
        atomicAdd(&ix.counters[0], c_reads);
        if (c_matched) atomicAdd(&ix.counters[1], c_matched);
        if (c_occ) atomicAdd(&ix.counters[2], c_occ);
    }
}

}  // namespace

template <bool USE_FTAB, bool STATS = false>
int launch_find_range_impl(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                           uint64_t *lo, uint64_t *hi, uint64_t *ssamp, const uint32_t *sel, const uint32_t *nsel, void *stream,
                           unsigned long long *stats = nullptr) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool toe = ssamp != nullptr;
    // sel mode: the number of reads is only known on the device; a fixed modest grid loops over it
    const int cap = sel ? 512 : 0;
#define RBG_LAUNCH_FR(PT, TOE)                                                                                      \
    do {                                                                                                            \
        auto kern = k_find_range<PT, TOE, USE_FTAB, STATS>;                                                         \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, cap);                                                    \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, lo, hi, ssamp, sel, nsel, stats);    \
    } while (0)
    if (ix.pos_bytes == 4) {
        if (toe) RBG_LAUNCH_FR(uint32_t, true); else RBG_LAUNCH_FR(uint32_t, false);
    } else {
        if (toe) RBG_LAUNCH_FR(uint64_t, true); else RBG_LAUNCH_FR(uint64_t, false);
    }
#undef RBG_LAUNCH_FR
    return static_cast<int>(hipGetLastError());
}

int launch_find_range_stats(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                            uint64_t *lo, uint64_t *hi, uint64_t *ssamp, unsigned long long *stats, void *stream) {
    if (ix.layout == 2) return launch_find_range_runs(ix, cfg, seqs, off, N, lo, hi, ssamp, stream, stats);  // (its sums mean other things: rbg.h)
    return ix.ftab_k ? launch_find_range_impl<true, true>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream, stats)
                     : launch_find_range_impl<false, true>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream, stats);
}

int launch_find_range(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    if (ix.layout == 2) return launch_find_range_runs(ix, cfg, seqs, off, N, lo, hi, ssamp, stream);  // run-indexed layout (k_runs.hip)
    // without a table the table-free instantiation runs (it is also the one that BUILDS the table,
    // so profiles show that one-off launch under its own kernel name)
    return ix.ftab_k ? launch_find_range_impl<true>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream)
                     : launch_find_range_impl<false>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream);
}


// ---- packed reads: workspace layout, pack, search -------------------------------------------------
// [ meta uint2[N] | nsel u32 (+pad to 16) | sel u32[N] | chunk_off u64[N+1] | scan temp | chunks uint4[total/64 + N] ]
namespace {
struct PackLayout {
    size_t meta, nsel, sel, chunk_off, scan_tmp, scan_tmp_bytes, chunks, total;
};
PackLayout pack_layout(uint64_t N, uint64_t total_bytes) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    PackLayout L;
    size_t o = 0;
    L.meta = o; o = up(o + N * sizeof(uint2));
    L.nsel = o; o = up(o + 16);
    L.sel = o; o = up(o + N * 4);
    L.chunk_off = o; o = up(o + (N + 1) * 8);
    size_t tb = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb, static_cast<uint64_t *>(nullptr), static_cast<uint64_t *>(nullptr), static_cast<int64_t>(N ? N : 1));
    L.scan_tmp_bytes = tb;
    L.scan_tmp = o; o = up(o + tb);
    L.chunks = o; o = up(o + (total_bytes / 64 + N + 1) * 16);
    L.total = o;
    return L;
}
}  // namespace

size_t pack_ws_bytes(uint64_t N, uint64_t total_bytes) { return pack_layout(N, total_bytes).total; }

int launch_pack_reads(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t total_bytes, void *ws, size_t ws_bytes, void *stream) {
    if (N == 0) return 0;
    const PackLayout L = pack_layout(N, total_bytes);
    if (ws_bytes < L.total) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *b = static_cast<char *>(ws);
    uint64_t *chunk_off = reinterpret_cast<uint64_t *>(b + L.chunk_off);
    uint32_t *nsel = reinterpret_cast<uint32_t *>(b + L.nsel);
    hipLaunchKernelGGL(k_pack_count, dim3(grid_for(cfg, N)), dim3(256), 0, st, off, N, chunk_off, nsel);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    rc = scan_in_place(chunk_off + 1, N, b + L.scan_tmp, L.scan_tmp_bytes, st);
    if (rc) return rc;
    const uint64_t groups = (N + 255) / 256;
    const int grid = static_cast<int>(std::min<uint64_t>(groups, 256ull * 8));
    hipLaunchKernelGGL(k_pack_reads, dim3(grid), dim3(256), 0, st, ix, seqs, off, N, chunk_off, reinterpret_cast<uint2 *>(b + L.meta),
                       reinterpret_cast<uint4 *>(b + L.chunks), reinterpret_cast<uint32_t *>(b + L.sel), nsel);
    return static_cast<int>(hipGetLastError());
}

// the packed search alone, on a batch packed elsewhere (the host path of rbg_capi.hip packs on the CPU): flagged
// reads (meta.y bit 31) are skipped, the caller searches them from their bytes
int launch_find_range_packed_only(const DevIndex &ix, const LaunchCfg &cfg, const uint2 *meta, const uint4 *chunks, uint64_t N,
                                  uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    if (N == 0) return 0;
    if (ix.layout == 2) return launch_find_range_runs_packed(ix, cfg, meta, chunks, N, lo, hi, ssamp, stream);   // run-indexed layout (k_runs.hip)
    hipStream_t st = static_cast<hipStream_t>(stream);
#define RBG_LAUNCH_FRP(PT, TOE)                                                                       \
    do {                                                                                              \
        auto kern = k_find_range_packed<PT, TOE>;                                                     \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern);                                           \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, meta, chunks, N, lo, hi, ssamp);     \
    } while (0)
    if (ix.pos_bytes == 4) {
        if (ssamp) RBG_LAUNCH_FRP(uint32_t, true); else RBG_LAUNCH_FRP(uint32_t, false);
    } else {
        if (ssamp) RBG_LAUNCH_FRP(uint64_t, true); else RBG_LAUNCH_FRP(uint64_t, false);
    }
#undef RBG_LAUNCH_FRP
    return static_cast<int>(hipGetLastError());
}

int launch_find_range_packed(const DevIndex &ix, const LaunchCfg &cfg, const void *ws, const uint8_t *seqs, const uint64_t *off,
                             uint64_t N, uint64_t total_bytes, uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    if (N == 0) return 0;
    const PackLayout L = pack_layout(N, total_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *b = static_cast<const char *>(ws);
    const uint2 *meta = reinterpret_cast<const uint2 *>(b + L.meta);
    const uint4 *chunks = reinterpret_cast<const uint4 *>(b + L.chunks);
    const bool toe = ssamp != nullptr;
    int rc;
    if (ix.layout == 2) {   // run-indexed layout (k_runs.hip); the flagged reads go to its byte kernel below
        rc = launch_find_range_runs_packed(ix, cfg, meta, chunks, N, lo, hi, ssamp, stream);
        if (rc) return rc;
        const uint32_t *sel2 = reinterpret_cast<const uint32_t *>(b + L.sel), *nsel2 = reinterpret_cast<const uint32_t *>(b + L.nsel);
        return launch_find_range_runs_sel(ix, cfg, seqs, off, N, lo, hi, ssamp, sel2, nsel2, stream);
    }
#define RBG_LAUNCH_FRP(PT, TOE)                                                                       \
    do {                                                                                              \
        auto kern = k_find_range_packed<PT, TOE>;                                                     \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern);                                           \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, meta, chunks, N, lo, hi, ssamp);     \
    } while (0)
    if (ix.pos_bytes == 4) {
        if (toe) RBG_LAUNCH_FRP(uint32_t, true); else RBG_LAUNCH_FRP(uint32_t, false);
    } else {
        if (toe) RBG_LAUNCH_FRP(uint64_t, true); else RBG_LAUNCH_FRP(uint64_t, false);
    }
#undef RBG_LAUNCH_FRP
    rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    // the reads the packed form cannot express (a symbol outside the major alphabet)
    const uint32_t *sel = reinterpret_cast<const uint32_t *>(b + L.sel), *nsel = reinterpret_cast<const uint32_t *>(b + L.nsel);
    return ix.ftab_k ? launch_find_range_impl<true>(ix, cfg, seqs, off, N, lo, hi, ssamp, sel, nsel, stream)
                     : launch_find_range_impl<false>(ix, cfg, seqs, off, N, lo, hi, ssamp, sel, nsel, stream);
}

}  // namespace rbg
