// k_runs.hip -- the run-indexed layout (RBG_LAYOUT_RUNS): K1/K2 and K3 with space proportional to r.  rank and phi are
// predecessor searches over sorted run boundaries, as in the reference (rle_string::rank rle_string.hpp:131-161,
// ToeholdSA::phi toehold_sa.hpp:56-72): a directory (or a bucket record) has already cut the search down to the handful
// of entries of one bucket, and the lane that owns the query fetches and scans exactly those (rbg_runs2_device.hpp).  One
// lane owns one read (K1/K2) or one phi chain (K3); bucket records are fetched by quads of lanes.
#include <type_traits>

#include "rbg_runs2_device.hpp"

namespace rbg {
namespace {

// ---- K1 / K2 over the run-indexed layout ------------------------------------------------------------------------
// RowBowt::find_range (rowbowt.hpp:121-131) / find_range_w_toehold (:169-184).  A step consumes up to run_ksteps read
// symbols through the k-mer depth's table, exactly as k_find_range does with its slot tables (k_search.hip; DESIGN.md
// 2b: identical to that many nested RowBowt::LF / LF_w_loc calls, rowbowt.hpp:74-88, :555-573); both ranks of all
// the wave's reads are answered cooperatively.  The device ftab (a constant-size state table, result-neutral) still
// replaces the first ftab_k steps.
// (4 waves per SIMD at 4-byte positions: the toehold variant sits at 130 VGPRs without the bound, one wave per SIMD less)
// PACKED: the reads arrive as 2-bit codes (k_pack_reads / the host path's CPU packer: meta[i] = {first chunk, length or
// bit 31 = "has a symbol outside the k-mer alphabet: byte kernel"}, 64 symbols per 16-byte chunk in consumption order);
// same steps in the same order as the byte form takes for a read of k-mer symbols only.
// STATS: the instrumented instantiation (rbg_find_range_stats_dev), sums as listed in rbg_runs_device.hpp.
struct PackedBits {   // per-lane reader of a packed read: peek / drop of up to 32 bits
    const uint4 *__restrict__ cp;
    uint4 w;
    uint32_t widx, navail;
    uint64_t sr;
    __device__ __forceinline__ uint32_t next_word() {
        if (widx == 4) { w = *cp++; widx = 0; }
        // (the word in front, the others moved up: indexing the vector by widx becomes a dynamic index, and one into scratch once the
        //  reader is captured by the step's lambda)
        const uint32_t v = w.x;
        w.x = w.y; w.y = w.z; w.z = w.w;
        ++widx;
        return v;
    }
    __device__ __forceinline__ uint32_t take(uint32_t nb) {
        if (navail < nb) { sr |= static_cast<uint64_t>(next_word()) << navail; navail += 32; }
        const uint32_t v = static_cast<uint32_t>(sr & ((uint64_t(1) << nb) - 1));
        sr >>= nb;
        navail -= nb;
        return v;
    }
};

// (the staging of a wave's reads as 2-bit codes in LDS -- stage_read, StageTab, kStageCap -- lives in rbg_runs_device.hpp: the seeding kernels use it too)
// GLDS: the bucket records arrive by LDS-direct loads (rbg_runs2_device.hpp lane_lf2_quad) instead of quad permutes
// STAGE (byte form only): the wave stages its reads as 2-bit codes in LDS first (above)
template <typename P, bool TOEHOLD, bool PACKED = false, bool STATS = false, bool GLDS = false, bool STAGE = false>
__global__ __launch_bounds__(512, STATS ? 2 : 4) void k_find_range_runs(const DevIndex ix, const void *__restrict__ src_a,
                                                        const void *__restrict__ src_b, const uint64_t N,
                                                        uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                        uint64_t *__restrict__ ss_out, unsigned long long *__restrict__ stats,
                                                        const uint32_t *__restrict__ sel, const uint32_t *__restrict__ nsel) {
    // byte form: src_a = seqs, src_b = off;  packed form: src_a = chunks (uint4), src_b = meta (uint2)
    // sel != nullptr (byte form): only the reads sel[0 .. *nsel) -- the ones the packed form could not express
    const uint64_t Neff = sel ? static_cast<uint64_t>(*nsel) : N;
    const uint8_t *__restrict__ seqs = static_cast<const uint8_t *>(src_a);
    const uint64_t *__restrict__ off = static_cast<const uint64_t *>(src_b);
    const uint4 *__restrict__ chunks = static_cast<const uint4 *>(src_a);
    const uint2 *__restrict__ meta = static_cast<const uint2 *>(src_b);
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    __shared__ uint8_t s_mslot[4];
    __shared__ __align__(16) unsigned char s_tile[GLDS ? 8 * kTileBytes : 16];   // (eight waves per workgroup)
    __shared__ uint32_t s_codes[STAGE ? 8 * kStageWords * 64 : 1];               // STAGE: the waves' reads as 2-bit codes
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {
        s_lut[t] = ix.lut[t];
        s_lut2[t] = ix.nmajor ? ix.lut2[t] : 0xFFu;
    }
    RBG_RUN_SEARCH2_SHARED;
    const RunSearch2<P> S2 = stage_run_search2<P>(ix, s_tab_first, s_ent2, s_dir2, s_rec2, s_dyn);
    const uint32_t *tab_first = s_tab_first;
    lds_byte *tile = (lds_byte *)s_tile + (GLDS ? (threadIdx.x >> 6) * kTileBytes : 0u);   // (C-style: a cast into the LDS address space)
    if (PACKED || STAGE) {
        for (int t = threadIdx.x; t < 256; t += blockDim.x)
            if (s_lut2[t] != 0xFFu) s_mslot[s_lut2[t] & 3u] = s_lut[t];   // major index -> symbol slot
        __syncthreads();
    }
    const uint32_t D = ix.run_ksteps, DMASK = ix.run_depth_mask | 1u;
    const uint32_t M = ix.nmajor;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    lds_u32 *codes = (lds_u32 *)s_codes + (STAGE ? (threadIdx.x >> 6) * (kStageWords * 64u) + lane : 0u);   // this lane's column: word w at codes[64 w]
    const StageTab stage_tab{ix.stage_code[0], ix.stage_code[1], ix.stage_byte[0], ix.stage_byte[1], ix.stage_shift};
    const bool stage_on = STAGE && ix.stage_ok != 0 && M == 4;

    unsigned long long c_occ = 0, c_reads = 0;
    uint32_t c_matched = 0;
    unsigned long long st[kStatSearchN] = {0, 0, 0, 0, 0, 0, 0, 0};   // STATS only (dead code otherwise)
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t wave_first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + (threadIdx.x & ~(kWave - 1));
    for (uint64_t base = wave_first; base < Neff; base += stride) {  // the lanes of a wave iterate together
        bool valid = base + lane < Neff;
        const uint64_t i = (sel && valid) ? static_cast<uint64_t>(sel[base + lane]) : base + lane;
        uint64_t beg = 0, p = 0;
        PackedBits bs{chunks, make_uint4(0, 0, 0, 0), 4u, 0u, 0ull};
        if (PACKED) {
            if (valid) {
                const uint2 mt = meta[i];
                if (mt.y & 0x80000000u) valid = false;           // has a non-major symbol: the byte kernel answers it
                else { p = mt.y; bs.cp = chunks + mt.x; }        // p = symbols still to consume (beg = 0)
            }
        } else if (valid) { beg = off[i]; p = off[i + 1]; }
        // STAGE: the wave's reads become 2-bit codes in LDS (stage_read above) unless one of them is too long or holds a symbol outside the k-mer
        // alphabet; the walk below is instantiated for both forms, the wave takes one
        bool staged = false;
        if constexpr (STAGE && !PACKED) {
            if (stage_on && __ballot(valid && p - beg > kStageCap) == 0) {
                uint32_t nch = 0;
                const bool bad = valid && stage_read(reinterpret_cast<const uint4 *>(seqs), beg, p, stage_tab, codes, nch);
                if (STATS) st[kStChunks] += nch;
                staged = __ballot(bad) == 0;
                if (staged) { p -= beg; beg = 0; }
                else if (STATS) st[kStChunks] -= nch;          // (the byte walk counts its own)
            }
        }
        auto walk = [&](auto staged_tag) __attribute__((always_inline)) {
            constexpr bool STAGED = decltype(staged_tag)::value;
            const uint64_t p_end = p;
            uint64_t p_min = p;                                    // STATS: lowest read byte fetched
            const uint32_t m32 = static_cast<uint32_t>(p);         // STAGED: the read's length (p counts the symbols still to consume, beg = 0)
            uint64_t lo = 0, hi = ix.n - 1;                       // full_range(), rowbowt.hpp:115-118
            uint64_t k = TOEHOLD ? ix.last_run_sample : 0;
            bool alive = valid;
            bool pend = false;                                     // deferred toehold re-sample (k_search.hip): the last one is the only one used
            uint32_t pend_d = 0, pend_rec = 0;
            uint64_t pend_e = 0;
            ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
            if (valid && ix.ftab_k && p - beg >= ix.ftab_k) {      // rowbowt.hpp:124-125, :745-758 (k_search.hip)
                uint64_t idx = 0;
                bool all_major = true;
                PackedBits probe = bs;                             // consumed only if the entry is usable
                if constexpr (STAGED) {
                    const uint32_t w0 = codes[0];                  // the first ftab_k (<= 16) symbols are the low 2 * ftab_k bits of the first word
                    idx = ix.ftab_k >= 16u ? w0 : (w0 & ((1u << (2u * ix.ftab_k)) - 1u));
                } else if (PACKED) {
                    idx = probe.take(2 * ix.ftab_k);               // the first ftab_k symbols are the low 2 * ftab_k bits
                } else {
                    uint64_t pw = 1;
                    for (uint32_t t = 1; t <= ix.ftab_k; ++t) {
                        const uint32_t mm = s_lut2[rd.at(p - t)];
                        all_major = all_major && mm != 0xFFu;
                        idx += (mm & 3u) * pw;
                        pw *= M;
                    }
                }
                uint64_t flo, fhi2, fk;
                if (STATS) { if (!STAGED) p_min = p - ix.ftab_k; if (all_major) st[kStFtab] += 1; }
                if (all_major && ftab_lookup<P>(ix, idx, flo, fhi2, fk)) {
                    if (PACKED) bs = probe;
                    lo = flo; hi = fhi2;
                    if (TOEHOLD) k = fk;
                    p -= ix.ftab_k;
                    if (STATS) st[kStSymbols] += ix.ftab_k;
                    if (hi < lo) { alive = false; p = beg; }
                }
            }
            // The steps of a read do not depend on the search: which table a step goes through is a matter of the read's symbols alone.  So the
            // NEXT step is chosen, and its table's record fetched (from LDS, or for depths 6-8 from the global array: an L2 round trip), while
            // the current step's bucket record is on its way -- the dependent chain of a step is its record (and, rarely, a scan), nothing else.
            struct Pick { uint32_t d, adv, rec; bool ok; };
            auto pick_at = [&](const uint64_t pp) __attribute__((always_inline)) -> Pick {        // the step that consumes the symbols ending at pp (exclusive: pp > beg)
                Pick s{0u, 1u, 0u, true};
                if constexpr (STAGED) {
                    s.adv = pp < D ? static_cast<uint32_t>(pp) : D;
                    s.adv = 32u - static_cast<uint32_t>(__clz(DMASK & ((2u << (s.adv - 1)) - 1u)));   // the deepest depth kept that fits
                    const uint32_t t0 = m32 - static_cast<uint32_t>(pp);                               // symbols consumed so far
                    const lds_u32 *cw = codes + (t0 >> 4) * 64u;
                    const uint32_t v = __builtin_amdgcn_alignbit(cw[64], cw[0], (t0 & 15u) * 2u) & ((1u << (2u * s.adv)) - 1u);
                    s.rec = run_record(tab_first, s.adv, s.adv == 1 ? static_cast<uint32_t>(s_mslot[v]) : v);
                    s.d = s.adv - 1;
                } else if (PACKED) {
                    s.adv = pp < D ? static_cast<uint32_t>(pp) : D;
                    s.adv = 32u - static_cast<uint32_t>(__clz(DMASK & ((2u << (s.adv - 1)) - 1u)));   // the deepest depth kept that fits
                    const uint32_t v = bs.take(2 * s.adv);
                    s.rec = run_record(tab_first, s.adv, s.adv == 1 ? static_cast<uint32_t>(s_mslot[v]) : v);
                    s.d = s.adv - 1;
                } else {
                    const uint64_t q = pp - 1;
                    const uint32_t c = rd.at(q);
                    if (STATS && q < p_min) p_min = q;
                    const uint32_t m0 = s_lut2[c];
                    uint32_t acc = m0;
                    if (m0 != 0xFFu) {                              // the longest run of major symbols among the next D (k_search.hip)
                        uint32_t pw = M, run_acc = m0;              // (of its prefixes, the longest whose depth has run lists)
    #pragma unroll 1
                        for (uint32_t t = 1; t < static_cast<uint32_t>(kMaxRunDepth); ++t) {
                            if (t >= D || q < beg + t) break;
                            const uint32_t mm = s_lut2[rd.at(q - t)];
                            if (STATS && q - t < p_min) p_min = q - t;
                            if (mm == 0xFFu) break;
                            run_acc += mm * pw;
                            pw *= M;
                            if ((DMASK >> t) & 1u) { s.adv = t + 1; acc = run_acc; }
                        }
                    }
                    if (s.adv == 1) {
                        const uint32_t slot = s_lut[c];
                        // symbol absent (f_[c] >= f_[c+1], rowbowt.hpp:76).  (An index with more than kLdsSyms symbols is
                        // never given this layout: upload() keeps the slot tables for it.)
                        if (slot == 0xFFu || slot >= static_cast<uint32_t>(kLdsSyms)) s.ok = false;
                        else s.rec = run_record(tab_first, 1u, slot);
                    } else {
                        s.d = s.adv - 1;
                        s.rec = run_record(tab_first, s.adv, acc);
                    }
                }
                return s;
            };
            // (the packed form keeps the step's choice at the top of the loop: the extra state of looking ahead sent it to scratch, and no kernel
            //  of this layout may spill -- tests/test_capi_host.py test_no_run_indexed_kernel_spills)
            constexpr bool AHEAD = !PACKED || STAGED;
            Pick cur{0u, 1u, 0u, false};
            RunHot Rcur{0, 0};
            if (AHEAD && alive && p > beg) {
                cur = pick_at(p);
                if (cur.ok) Rcur = load_run_tab<P>(S2, cur.d, cur.rec);
            }
            while (__ballot(alive && p > beg)) {                   // right-to-left over the reads (rowbowt.hpp:127-129, :175-181)
                bool stepping = alive && p > beg;
                if (!AHEAD && stepping) {
                    cur = pick_at(p);
                    if (cur.ok) Rcur = load_run_tab<P>(S2, cur.d, cur.rec);
                }
                if (stepping && !cur.ok) { alive = false; stepping = false; }
                const uint32_t d = cur.d, adv = cur.adv, rec = cur.rec;
                const RunHot R = Rcur;
                // the step after this one (taken only if this one leaves the range non-empty)
                const uint64_t p_next = p - adv;
                if (AHEAD && stepping && p_next > beg) {
                    cur = pick_at(p_next);
                    if (cur.ok) Rcur = load_run_tab<P>(S2, cur.d, cur.rec);
                }
                RunStep r;
                // rank(lo, c), rank(hi + 1, c): rowbowt.hpp:79,83
                lane_lf2_quad<P, STATS, false, GLDS>(S2, stepping, d, rec, R, lo, hi + 1, r, st, tile);   // (every lane of the wave: the records are fetched by quads)
                if (stepping) {
                    if (STATS) st[kStSymbols] += adv;
                    const uint64_t c_inside = r.c_upto - r.c_before;
                    if (c_inside == 0) {                            // rowbowt.hpp:85 (whichever of the nested steps emptied the range)
                        alive = false;
                    } else {
                        if (TOEHOLD) {                              // LF_w_loc, rowbowt.hpp:559-566, `adv` times nested
                            if (r.inside) k = k - adv;
                            else { pend = true; pend_d = d; pend_rec = rec; pend_e = r.samp_e; k = 0; }
                        }
                        lo = r.F + r.c_before;                      // rowbowt.hpp:86
                        hi = lo + c_inside - 1;                     // rowbowt.hpp:87
                        p = p_next;                                 // the left neighbours are consumed too
                    }
                }
            }
            if (TOEHOLD && alive && pend) {
                k += run_step_sample2<P>(ix, S2, pend_d, pend_rec, pend_e);
                if (STATS) st[kStResample] += 1;
            }
            if (STATS && !PACKED && !STAGED && p_end > p_min) st[kStChunks] += ((p_end - 1) >> 4) - (p_min >> 4) + 1;
            if (STATS && PACKED && valid) st[kStChunks] += (p_end + 63) >> 6;
            if (!alive) { lo = 1; hi = 0; k = 0; }                 // {1,0}; LFData::clear rowbowt.hpp:153-159
            if (valid) {
                lo_out[i] = lo;
                hi_out[i] = hi;
                if (TOEHOLD) ss_out[i] = k;
                c_reads += 1;
                if (alive) { c_matched += 1; c_occ += hi - lo + 1; }
            }
        };
        if (STAGE && staged) walk(std::integral_constant<bool, STAGE>{}); else walk(std::false_type{});
    }
    c_reads = wave_sum(c_reads);
    const unsigned long long w_matched = wave_sum(static_cast<unsigned long long>(c_matched));
    c_occ = wave_sum(c_occ);
    if (lane == 0 && c_reads) {
        atomicAdd(&ix.counters[0], c_reads);
        if (w_matched) atomicAdd(&ix.counters[1], w_matched);
        if (c_occ) atomicAdd(&ix.counters[2], c_occ);
    }
    if (STATS) {
#pragma unroll
        for (int t = 0; t < kStatSearchN; ++t) {
            const unsigned long long v = wave_sum(st[t]);
            if (lane == 0 && v) atomicAdd(&stats[t], v);
        }
    }
}

template <typename P> struct ChunkR { static constexpr int v = 8; };   // as in k_locate.hip (flush windows of eight locations: 64 bytes)
// the toehold of a locus key (k_locate.hip locus_key / locus_toehold: the chains' order by locus)
__device__ __forceinline__ uint64_t locus_toehold_r(const DevIndex &ix, const uint64_t key, const uint64_t *__restrict__ k, const uint64_t i) {
    if (key == ~uint64_t(0)) return k[i];
    const uint32_t low = ix.order_lowbits, db = ix.order_dbits;
    const uint64_t doc = (key >> low) & ((uint64_t(1) << db) - 1u);
    return ix.order_docs[doc] + (((key >> (db + low)) << low) | (key & ((uint64_t(1) << low) - 1u)));
}

// ---- K3 over the run-indexed layout: ToeholdSA::locate_range (toehold_sa.hpp:37-49) -- every lane walks its own chain and
// answers its own phi (toehold_sa.hpp:56-72; rbg_runs2_device.hpp lane_phi).  The chains, the staging of the values in LDS and
// the coalesced flush are k_locate_fill's (k_locate.hip; chains in toehold order); a phi step is one directory
// gather (two neighbouring counts, plus the super count at 8-byte positions) and one scan of the bucket's few sampled
// positions, no cross-lane traffic.  Works for ordered and unordered walks alike (the order only decides how well
// neighbouring lanes share sectors).  STATS: [kLsPhiSteps] phi evaluations, [kLsPhiOvf] sampled positions the scans
// and the narrowing rounds needed (8 or 12 bytes each; 7 pivot keys per round), [kLsChains], [kLsLocs].
template <typename P, typename OUT = uint64_t, bool STATS = false, bool SUB = false, bool HI8 = false>
__global__ __launch_bounds__(256, 4) void k_locate_fill_runs2(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                          const uint64_t *__restrict__ hi, const uint64_t *__restrict__ k,
                                                          const uint64_t N, const uint64_t max_hits,
                                                          const uint64_t *__restrict__ loc_off, OUT *__restrict__ locs,
                                                          const uint64_t *__restrict__ sub, const uint32_t *__restrict__ order,
                                                          const uint64_t *__restrict__ skeys, unsigned long long *__restrict__ stats) {
    constexpr int kChunkR = ChunkR<P>::v;
    constexpr bool RING = true;
    using Stage = ChainStage<P, kChunkR, SUB, RING, HI8>;
    __shared__ Stage S;   // (rbg_device.hpp: the staging and the flush are k_locate_fill's)
    const uint64_t out_elem0 = reinterpret_cast<uintptr_t>(locs) / sizeof(OUT);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    unsigned long long c_locs = 0;
    unsigned long long st_phi = 0, st_ent = 0, st_chains = 0;   // STATS only
    const uint64_t n = ix.n;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * blockDim.x + wv * kWave; base < N; base += stride) {
        const uint64_t j = base + lane;
        uint64_t i = j;
        if (order && j < N) i = order[j];
        uint64_t occ = 0, k1 = 0, dst = 0;
        if (i < N && j < N) {
            dst = loc_off[i];
            if (skeys) {
                if (ix.order_docs) k1 = locus_toehold_r(ix, skeys[j], k, i);
                else if (sizeof(P) == 4) { const uint32_t k32 = reinterpret_cast<const uint32_t *>(skeys)[j]; k1 = k32 == 0xFFFFFFFFu ? k[i] : k32; }   // (k_locate.hip k_keys32)
                else k1 = skeys[j];
                occ = loc_off[i + 1] - dst;
            } else {
                const uint64_t l = lo[i], h = hi[i];
                occ = h >= l ? h - l + 1 : 0;                  // toehold_sa.hpp:38-39
                if (occ > max_hits) occ = max_hits;
                k1 = k[i];
            }
        }
        const uint64_t minus = (SUB && i < N && j < N) ? sub[i] : 0;
        // the window grid of this read: its first location sits a elements past a CH-element boundary of the output array (RING; else a = 0)
        const uint32_t a = (RING && occ) ? static_cast<uint32_t>((out_elem0 + dst) & static_cast<uint64_t>(kChunkR - 1)) : 0u;
        S.dst[wv][lane] = dst - a;   // (wraps for a read at the very start of a misaligned array; + v >= a brings it back)
        if (SUB) S.minus[wv][lane] = minus;
        const bool off_text = Stage::kSentinel && k1 >= n;     // a toehold below zero: its owner stores that location (ChainStage)
        if (off_text && occ) locs[dst] = static_cast<OUT>(k1 - minus);
        c_locs += occ;
        if (STATS && occ) st_chains += 1;
        uint64_t wmax = occ + a;      // the chain's extent in virtual columns
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(wmax, o, kWave);
            wmax = other > wmax ? other : wmax;
        }
        for (uint64_t t0 = 0; t0 < wmax; t0 += kChunkR) {
            const uint32_t cnt = chain_round_count<kChunkR>(occ, t0);
            S.bounds[wv][lane] = chain_window_bounds<kChunkR>(occ, a, t0);
#pragma unroll 1
            for (int e = 0; e < kChunkR; ++e) {
                const bool mine = static_cast<uint32_t>(e) < cnt;
                if (mine && (e || t0)) {                       // toehold_sa.hpp:44: k = phi(k)
                    uint64_t s;
                    if (k1 >= n) {
                        // a toehold below zero (k_locate.hip phi_step): outside phi's domain -- the last sample is its predecessor
                        s = (ix.phi_last_base + (k1 - ix.phi_last_pos)) % n;
                    } else {
                        bool found;
                        uint64_t val;
                        uint32_t ents, rounds;
                        lane_phi<P>(ix, k1, found, val, ents, rounds);
                        if (STATS) { st_phi += 1; st_ent += ents + 7u * rounds; }
                        // no sampled position before k1: circular predecessor = the last one, delta = i + 1
                        // (sparse_sd_vector.hpp:141-143, toehold_sa.hpp:59,65); else prev_sample + delta (toehold_sa.hpp:65-71)
                        s = found ? val : ix.phi_last_base + k1 + 1;
                        if (s >= n) s -= n;
                    }
                    k1 = s;
                }
                if (mine) chain_put(S, wv, lane, a + static_cast<uint32_t>(t0) + e, k1, e == 0 && t0 == 0 && off_text);
            }
            wave_lds_sync();
            chain_flush(S, wv, lane, t0, locs);
            wave_lds_sync();
        }
        wave_lds_sync();
    }
    c_locs = wave_sum(c_locs);
    if (lane == 0 && c_locs) atomicAdd(&ix.counters[3], c_locs);
    if (STATS) {
        st_phi = wave_sum(st_phi);
        st_ent = wave_sum(st_ent);
        st_chains = wave_sum(st_chains);
        if (lane == 0) {
            if (st_phi) atomicAdd(&stats[kLsPhiSteps], st_phi);
            if (st_ent) atomicAdd(&stats[kLsPhiOvf], st_ent);
            if (st_chains) atomicAdd(&stats[kLsChains], st_chains);
            if (c_locs) atomicAdd(&stats[kLsLocs], c_locs);
        }
    }
}

}  // namespace

// `packed`: src_a = chunks, src_b = meta (see the kernel); stats != nullptr: the instrumented instantiation
int launch_find_range_runs_impl(const DevIndex &ix, const LaunchCfg &cfg, const void *src_a, const void *src_b, uint64_t N,
                                uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream, bool packed, unsigned long long *stats,
                                const uint32_t *sel = nullptr, const uint32_t *nsel = nullptr) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = run_search2_lds(ix);
    // how a quad's bucket records travel (rbg_runs2_device.hpp): LDS-direct loads (the default), or RBG_REC_FETCH=quad: registers + quad permutes
    static const bool glds = [] { const char *e = std::getenv("RBG_REC_FETCH"); return !(e && std::strcmp(e, "quad") == 0); }();
    // the byte form stages its reads as 2-bit codes in LDS (STAGE above; needs an alphabet the register tables express: DevIndex::stage_ok);
    // RBG_STAGE_READS=0: every lane walks its read through a cursor of 16-byte chunks as before round 6 (A/B, profiles/r06_k2_sectors.md)
    static const bool stage = [] { const char *e = std::getenv("RBG_STAGE_READS"); return !(e && e[0] == '0'); }();
    LaunchCfg c = cfg;   // 512-thread workgroups: the staged tables and top level are shared by eight waves
    c.block_threads = 512;
    c.max_blocks = cfg.max_blocks > 0 ? std::max(1, cfg.max_blocks / 2) : 256 * 16;
    // sel mode: the number of reads is only known on the device; a fixed modest grid loops over it
    const dim3 grid(sel ? std::min(grid_for(c, N), 256) : grid_for(c, N)), block(512);
#define RBG_LAUNCH_FRR1(PT, TOE, PK, STS, GL, SG)                                                      \
    do {                                                                                               \
        auto kern = k_find_range_runs<PT, TOE, PK, STS, GL, SG>;                                       \
        raise_lds(kern, lds, (GL ? 8 * kTileBytes : 0) + (SG ? 8 * kStageWaveBytes : 0));             \
        hipLaunchKernelGGL(kern, grid, block, lds, st, ix, src_a, src_b, N, lo, hi, ssamp, stats, sel, nsel);     \
    } while (0)
#define RBG_LAUNCH_FRR(PT, TOE, PK, STS)                                                               \
    do {                                                                                               \
        if (glds && stage && !PK && !sel) RBG_LAUNCH_FRR1(PT, TOE, false, STS, true, true);            \
        else if (glds) RBG_LAUNCH_FRR1(PT, TOE, PK, STS, true, false);                                 \
        else RBG_LAUNCH_FRR1(PT, TOE, PK, STS, false, false);                                          \
    } while (0)
#define RBG_LAUNCH_FRR2(PT, TOE)                                                                       \
    do {                                                                                               \
        if (stats) { if (packed) return static_cast<int>(hipErrorNotSupported); RBG_LAUNCH_FRR(PT, TOE, false, true); } \
        else if (packed) RBG_LAUNCH_FRR(PT, TOE, true, false);                                         \
        else RBG_LAUNCH_FRR(PT, TOE, false, false);                                                    \
    } while (0)
    if (ix.pos_bytes == 4) {
        if (ssamp) RBG_LAUNCH_FRR2(uint32_t, true); else RBG_LAUNCH_FRR2(uint32_t, false);
    } else {
        if (ssamp) RBG_LAUNCH_FRR2(uint64_t, true); else RBG_LAUNCH_FRR2(uint64_t, false);
    }
#undef RBG_LAUNCH_FRR2
#undef RBG_LAUNCH_FRR
#undef RBG_LAUNCH_FRR1
    return static_cast<int>(hipGetLastError());
}

int launch_find_range_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                           uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream, unsigned long long *stats) {
    return launch_find_range_runs_impl(ix, cfg, seqs, off, N, lo, hi, ssamp, stream, false, stats);
}

int launch_find_range_runs_sel(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                               uint64_t *hi, uint64_t *ssamp, const uint32_t *sel, const uint32_t *nsel, void *stream) {
    return launch_find_range_runs_impl(ix, cfg, seqs, off, N, lo, hi, ssamp, stream, false, nullptr, sel, nsel);
}

int launch_find_range_runs_packed(const DevIndex &ix, const LaunchCfg &cfg, const uint2 *meta, const uint4 *chunks, uint64_t N,
                                  uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    return launch_find_range_runs_impl(ix, cfg, chunks, meta, N, lo, hi, ssamp, stream, true, nullptr);
}

int launch_locate_fill_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint64_t *k,
                            uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs, const uint64_t *sub,
                            const void *order, const uint64_t *skeys, void *stream, unsigned long long *stats, uint32_t *locs32) {
    if (N == 0) return 0;
    if ((stats || locs32) && sub) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(cfg, N)), block(256);
    const uint32_t *perm = static_cast<const uint32_t *>(order);
    const bool hi8 = ix.pos_bytes == 8 && ix.n < kChainHi8Limit && chain_hi8_enabled();
#define RBG_LAUNCH_LFR2(PT, OUT, STS, SB, DST) \
    do { \
        if (sizeof(PT) == 8 && hi8) hipLaunchKernelGGL((k_locate_fill_runs2<PT, OUT, STS, SB, sizeof(PT) == 8>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, DST, sub, perm, skeys, stats); \
        else hipLaunchKernelGGL((k_locate_fill_runs2<PT, OUT, STS, SB, false>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, DST, sub, perm, skeys, stats); \
    } while (0)
    if (locs32) {
        if (ix.pos_bytes != 4) return static_cast<int>(hipErrorInvalidValue);
        RBG_LAUNCH_LFR2(uint32_t, uint32_t, false, false, locs32);
    } else if (stats) {
        if (ix.pos_bytes == 4) RBG_LAUNCH_LFR2(uint32_t, uint64_t, true, false, locs); else RBG_LAUNCH_LFR2(uint64_t, uint64_t, true, false, locs);
    } else if (sub) {
        if (ix.pos_bytes == 4) RBG_LAUNCH_LFR2(uint32_t, uint64_t, false, true, locs); else RBG_LAUNCH_LFR2(uint64_t, uint64_t, false, true, locs);
    } else if (ix.pos_bytes == 4) {
        RBG_LAUNCH_LFR2(uint32_t, uint64_t, false, false, locs);
    } else {
        RBG_LAUNCH_LFR2(uint64_t, uint64_t, false, false, locs);
    }
#undef RBG_LAUNCH_LFR2
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
