// k_runs_seeds.hip -- the kernels beside the rb_align path on the run-indexed layout (RBG_LAYOUT_RUNS): single LF steps
// (RowBowt::LF, rowbowt.hpp:74-88), find_range_w_markers (:292-339), greedy seeds (:222-256 reduced by :669-677) and
// marker seeds (get_markers_greedy_seeding, :406-482).  Same outputs as their slot-table versions in k_markers.hip; what
// differs is how a rank is answered: there one lane reads one slot, here the lane reads its bucket's record (fetched by its
// quad) or directory entry and scans the few run-list entries it names (rbg_runs2_device.hpp), and a step may consume up to
// run_ksteps symbols through the k-mer depth's run list.  Each kernel is written as a state machine: per round every lane
// that still has work names its next step (or does lane-local work that needs no rank: ftab entries, absent symbols,
// marker queries), all lanes of the wave take their steps together (the quads fetch records for one another), every lane
// applies its result.
#include <type_traits>

#include "rbg_runs2_device.hpp"

namespace rbg {
namespace {

// (the marker query -- MarkerArray::at_range as {src, cnt} -- is rbg_device.hpp marker_query: bucket records, or the directory + run arrays)

// the step that consumes the longest k-mer of at most `cap` symbols (all of them k-mer symbols) ending at byte p:
// adv = its length, (d, rec) = depth index and record of its table.  adv == 1: the single symbol's table, or
// ok = false when the byte does not occur in the index (f_[c] >= f_[c+1], rowbowt.hpp:76: an empty range).
struct StepPick {
    uint32_t adv, d, rec;
    bool ok;
};
__device__ __forceinline__ StepPick pick_step(ByteCursor &rd, const uint8_t *s_lut, const uint8_t *s_lut2, const uint32_t *s_tab_first,
                                              const uint64_t p, const uint64_t cap, const uint32_t D, const uint32_t DMASK, const uint32_t M) {
    StepPick s{1u, 0u, 0u, true};
    const uint32_t c = rd.at(p);
    const uint32_t m0 = s_lut2[c];
    uint32_t acc = m0;
    if (m0 != 0xFFu) {
        uint32_t pw = M, run_acc = m0;   // (the longest prefix of the run whose depth has run lists: DevIndex::run_depth_mask)
#pragma unroll 1
        for (uint32_t t = 1; t < static_cast<uint32_t>(kMaxRunDepth); ++t) {
            if (t >= D || t >= cap) break;
            const uint32_t mm = s_lut2[rd.at(p - t)];
            if (mm == 0xFFu) break;
            run_acc += mm * pw;
            pw *= M;
            if ((DMASK >> t) & 1u) { s.adv = t + 1; acc = run_acc; }
        }
    }
    if (s.adv == 1) {
        const uint32_t slot = s_lut[c];
        if (slot == 0xFFu || slot >= static_cast<uint32_t>(kLdsSyms)) s.ok = false;   // (more than kLdsSyms symbols never get this layout)
        else s.rec = run_record(s_tab_first, 1u, slot);
    } else {
        s.d = s.adv - 1;
        s.rec = run_record(s_tab_first, s.adv, acc);
    }
    return s;
}

// the state after the word of ftab_k k-mer symbols that ends at byte p, from the device table; false: not usable
// (a symbol outside the k-mer alphabet, the "search it step by step" marker, or a word that does not occur)
template <typename P>
__device__ __forceinline__ bool ftab_state(const DevIndex &ix, ByteCursor &rd, const uint8_t *s_lut2, const uint64_t p, const uint32_t M, uint64_t &lo,
                                           uint64_t &hi, uint64_t &k) {
    uint64_t idx = 0, pw = 1;
    bool all_major = true;
    for (uint32_t t = 0; t < ix.ftab_k; ++t) {
        const uint32_t mm = s_lut2[rd.at(p - t)];
        all_major = all_major && mm != 0xFFu;
        idx += (mm & 3u) * pw;
        pw *= M;
    }
    uint64_t flo, fhi, fk;
    if (!all_major || !ftab_lookup<P>(ix, idx, flo, fhi, fk) || flo > fhi) return false;
    lo = flo; hi = fhi; k = fk;
    return true;
}

// ---- the same two on a read STAGED as 2-bit codes in LDS (rbg_runs_device.hpp stage_read: every symbol of the read is a k-mer symbol) -----------
// t = consumption index of the symbol the step starts at (symbols between it and the read's end); cap >= 1 symbols may be consumed
__device__ __forceinline__ StepPick pick_step_staged(const lds_u32 *codes, const uint8_t *s_mslot, const uint32_t *s_tab_first, const uint32_t t, const uint64_t cap,
                                                     const uint32_t D, const uint32_t DMASK) {
    StepPick s{1u, 0u, 0u, true};
    const uint32_t lim = cap < D ? static_cast<uint32_t>(cap) : D;
    s.adv = 32u - static_cast<uint32_t>(__clz(DMASK & ((2u << (lim - 1)) - 1u)));   // the deepest depth kept that fits
    const uint32_t v = staged_bits(codes, t, s.adv);
    s.rec = run_record(s_tab_first, s.adv, s.adv == 1 ? static_cast<uint32_t>(s_mslot[v]) : v);
    s.d = s.adv - 1;
    return s;
}
template <typename P>
__device__ __forceinline__ bool ftab_state_staged(const DevIndex &ix, const lds_u32 *codes, const uint32_t t, uint64_t &lo, uint64_t &hi, uint64_t &k) {
    uint64_t flo, fhi, fk;
    if (!ftab_lookup<P>(ix, staged_bits(codes, t, ix.ftab_k), flo, fhi, fk) || flo > fhi) return false;
    lo = flo; hi = fhi; k = fk;
    return true;
}

// STAGES: the kernel stages its waves' reads (34.8 KB of LDS per workgroup: the two seed walks; the others read bytes)
#define RBG_SEED_STAGE_SHARED                                                                     \
    __shared__ uint32_t s_codes[8 * kStageWords * 64];                                            \
    __shared__ uint8_t s_mslot[4];                                                                \
    for (int t = threadIdx.x; t < 256; t += blockDim.x)                                           \
        if (s_lut2[t] != 0xFFu) s_mslot[s_lut2[t] & 3u] = s_lut[t];                               \
    __syncthreads();                                                                              \
    lds_u32 *codes = (lds_u32 *)s_codes + (threadIdx.x >> 6) * (kStageWords * 64u) + lane;        \
    const StageTab stage_tab{ix.stage_code[0], ix.stage_code[1], ix.stage_byte[0], ix.stage_byte[1], ix.stage_shift}; \
    static_assert(kStageCap >= 16, "");                                                           \
    const bool stage_on = ix.stage_ok != 0 && M == 4

#define RBG_SEED_KERNEL_PROLOGUE(P)                                                               \
    __shared__ uint8_t s_lut[256];                                                                \
    __shared__ uint8_t s_lut2[256];                                                               \
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {                                         \
        s_lut[t] = ix.lut[t];                                                                     \
        s_lut2[t] = ix.nmajor ? ix.lut2[t] : 0xFFu;                                               \
    }                                                                                             \
    RBG_RUN_SEARCH2_SHARED;                                                                       \
    const RunSearch2<P> S2 = stage_run_search2<P>(ix, s_tab_first, s_ent2, s_dir2, s_rec2, s_dyn); \
    const uint32_t *tab_first = s_tab_first;                                                      \
    const uint32_t D = ix.run_ksteps, DMASK = ix.run_depth_mask | 1u, M = ix.nmajor;              \
    const uint32_t lane = threadIdx.x & (kWave - 1);                                              \
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;                        \
    const uint64_t wave_first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + (threadIdx.x & ~(kWave - 1))

// both ranks of the step of every lane of the wave
// LEAN: one record in registers at a time (rbg_runs2_device.hpp lane_lf2) -- for the instantiation whose state would otherwise cost
// a workgroup per CU (512 threads: two or four waves per SIMD, nothing between: the greedy seeds at 8-byte positions); the others
// fetch both records of a step, and both scans of crowded buckets, together
// QUAD: the first record of a step fetched by the lane's quad (rbg_runs2_device.hpp lane_lf2_quad); not where its registers would cost the workgroup
template <typename P, bool LEAN = false, bool QUAD = true, bool STATS = false>
__device__ __forceinline__ void seeds_lf2(const RunSearch2<P> &S2, const bool stepping, const uint32_t d, const uint32_t rec, const uint64_t q0, const uint64_t q1,
                                          RunStep &r, unsigned long long *st = nullptr) {
    if constexpr (QUAD) lane_lf2_quad<P, STATS, LEAN>(S2, stepping, d, rec, q0, q1, r, st);   // (every lane calls)
    else { if (stepping) lane_lf2<P, STATS, LEAN>(S2, d, rec, q0, q1, r, st); }
}
// the instrumented instantiations' sums, added to the launch's array at the end of the kernel
template <int NS>
__device__ __forceinline__ void seed_stats_flush(unsigned long long (&st)[NS], unsigned long long *__restrict__ stats, const uint32_t lane) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const unsigned long long v = wave_sum(st[t]);
        if (lane == 0 && v) atomicAdd(&stats[t], v);
    }
}

// ---- single LF step for N (range, symbol) triples: RowBowt::LF(range_t, uint8_t), rowbowt.hpp:74-88 -------------------
template <typename P>
__global__ __launch_bounds__(512, 4) void k_lf_runs(const DevIndex ix, const uint64_t *__restrict__ lo_in,
                                                                        const uint64_t *__restrict__ hi_in, const uint8_t *__restrict__ sym,
                                                                        const uint64_t N, uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out) {
    RBG_SEED_KERNEL_PROLOGUE(P);
    (void)D; (void)DMASK; (void)M; (void)s_lut2;
    for (uint64_t base = wave_first; base < N; base += stride) {
        const uint64_t i = base + lane;
        const bool valid = i < N;
        uint64_t lo = 0, hi = 0;
        uint32_t rec = 0;
        bool stepping = false;
        if (valid) {
            lo = lo_in[i]; hi = hi_in[i];
            const uint32_t slot = s_lut[sym[i]];
            // hi >= n is outside rle_string::rank's domain (assert(i<=n), rle_string.hpp:132): answer {1,0}
            stepping = slot != 0xFFu && slot < static_cast<uint32_t>(kLdsSyms) && hi < ix.n && lo <= hi + 1;
            if (stepping) rec = run_record(tab_first, 1u, slot);
        }
        RunStep r;
        seeds_lf2<P>(S2, stepping, 0u, rec, lo, hi + 1, r);
        if (valid) {
            uint64_t nlo = 1, nhi = 0;
            if (stepping && r.c_upto > r.c_before) { nlo = r.F + r.c_before; nhi = nlo + (r.c_upto - r.c_before) - 1; }
            lo_out[i] = nlo;
            hi_out[i] = nhi;
        }
    }
}

// ---- find_range_w_markers (rowbowt.hpp:292-339): single steps, a marker query at every window end ---------------------
// (k_markers.hip k_find_range_markers; window results are PREPENDED in the reference, :320,:333)
template <typename P, bool FILL>
__global__ __launch_bounds__(512, 4) void k_find_range_markers_runs(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                            const uint64_t *__restrict__ off, const uint64_t N,
                                                            const uint64_t wsize, const uint64_t max_range,
                                                            uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                            uint64_t *__restrict__ cnt_out,
                                                            const uint64_t *__restrict__ mk_off, uint64_t *__restrict__ mk) {
    RBG_SEED_KERNEL_PROLOGUE(P);
    (void)D; (void)DMASK; (void)M;
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) cnt_out[0] = 0;
    for (uint64_t base = wave_first; base < N; base += stride) {
        const uint64_t i = base + lane;
        const bool valid = i < N;
        uint64_t beg = 0, end = 0, m = 0;
        if (valid) { beg = off[i]; end = off[i + 1]; m = end - beg; }
        uint64_t lo = 1, hi = 0;
        bool alive = valid && m >= wsize;              // rowbowt.hpp:299-302: shorter queries return the default LFData
        // fill pass: a read with nothing to emit (every read that dies: lf.clear() drops what earlier windows collected) writes nothing
        if (FILL && valid && mk_off[i + 1] == mk_off[i]) alive = false;
        const bool started = alive;
        uint64_t window_ei = m, acc = 0, s = 0;
        const uint64_t want = (FILL && valid) ? mk_off[i + 1] - mk_off[i] : 0;
        uint64_t *dst = (FILL && valid) ? mk + mk_off[i] : nullptr;
        if (alive) { lo = 0; hi = ix.n - 1; }
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        auto query = [&]() {                           // :318,:331
            if (hi - lo + 1 > max_range) return;
            uint64_t src, cnt;
            if (!marker_query(ix, lo, hi, &src, &cnt)) return;
            acc += cnt;
            if (FILL) {
                uint64_t *d = dst + (want - acc);
                for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
            }
        };
        while (__ballot(alive && s < m)) {
            bool stepping = alive && s < m;
            uint32_t rec = 0;
            if (stepping) {
                const uint32_t slot = s_lut[rd.at(end - 1 - s)];
                if (slot == 0xFFu || slot >= static_cast<uint32_t>(kLdsSyms)) { alive = false; stepping = false; }
                else rec = run_record(tab_first, 1u, slot);
            }
            RunStep r;
            seeds_lf2<P>(S2, stepping, 0u, rec, lo, hi + 1, r);
            if (stepping) {
                const uint64_t c_inside = r.c_upto - r.c_before;
                if (c_inside == 0) alive = false;
                else {
                    lo = r.F + r.c_before;
                    hi = lo + c_inside - 1;
                    if (window_ei - (m - s) >= wsize) {     // :315
                        window_ei = m - s;                  // :322
                        query();
                    }
                    ++s;
                }
            }
        }
        if (alive && (m - 1) % wsize != 0) query();         // :328 (s == m)
        if (!FILL && valid) {
            if (started && !alive) { lo = 1; hi = 0; }      // lf.clear(), :311-313
            lo_out[i] = lo;
            hi_out[i] = hi;
            cnt_out[i + 1] = (started && alive) ? acc : 0;
        }
    }
}

// ---- greedy seeding: RowBowt::get_seeds_greedy_w_sample (rowbowt.hpp:222-256) reduced on the fly by
// locate_from_longest_seed's choice (:669-677): per read the first seed of strictly greatest length (k_markers.hip
// k_greedy_seed).  A k-mer step that comes back empty is narrowed by halving until the failing base is the reference's.
// (three waves per SIMD -- at four the register limit of 128 sends a few of this kernel's values to scratch, and a kernel of this layout
//  that spilled faulted on the device in round 4, profiles/r04_fault_note.txt: none of them is allowed to)
template <typename P, bool QUAD, bool STATS = false>
__global__ __launch_bounds__(512, STATS ? 2 : 3) void k_greedy_seed_runs(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                     const uint64_t *__restrict__ off, const uint64_t N,
                                                     const uint64_t min_length, uint64_t *__restrict__ lo_out,
                                                     uint64_t *__restrict__ hi_out, uint64_t *__restrict__ qs_out,
                                                     uint64_t *__restrict__ qe_out, uint64_t *__restrict__ ss_out,
                                                     unsigned long long *__restrict__ stats = nullptr) {
    RBG_SEED_KERNEL_PROLOGUE(P);
    RBG_SEED_STAGE_SHARED;
    unsigned long long st[kSeedStatN] = {};            // STATS only (dead code otherwise)
    const uint64_t first_k = ix.last_run_sample;       // rowbowt.hpp:230
    const uint64_t fhi = ix.n - 1;
    for (uint64_t base = wave_first; base < N; base += stride) {
        const uint64_t i = base + lane;
        const bool valid = i < N;
        uint64_t beg = 0, m = 0;
        if (valid) { beg = off[i]; m = off[i + 1] - beg; }
        // the wave's reads as 2-bit codes in LDS (k_runs.hip STAGE: every chunk of a read fetched once, back to back) unless one is too long or
        // holds a symbol outside the k-mer alphabet
        bool staged = false;
        if (stage_on && __ballot(valid && m > kStageCap) == 0) {
            uint32_t nch = 0;
            const bool bad = valid && stage_read(reinterpret_cast<const uint4 *>(seqs), beg, beg + m, stage_tab, codes, nch);
            staged = __ballot(bad) == 0;
            if (STATS && staged) st[kStChunks] += nch;
        }
        auto walk = [&](auto staged_tag) __attribute__((always_inline)) {
        constexpr bool STAGED = decltype(staged_tag)::value;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;
        uint64_t k = first_k, pk = ~uint64_t(0), ei = m;
        // DEFERRED re-samples (as in k_find_range_runs): a step whose row hi does not carry the symbol re-samples the toehold from the run list -- two
        // dependent gathers behind the step's record -- but only the LAST re-sample before a seed's end is ever used: the step leaves {table, entry}
        // behind and k = 0, later steps subtract from k, and the sample is fetched when a seed becomes the read's best (toehold = sample + k in
        // wrapping arithmetic: exactly the reference's chain of k - 1, rowbowt.hpp:555-573).  pend_* / ppend_*: of k and of pk (the state before the step).
        bool pend = false, ppend = false;
        uint32_t pend_d = 0, pend_rec = 0, ppend_d = 0, ppend_rec = 0;
        uint64_t pend_e = 0, ppend_e = 0;
        uint64_t b_lo = 1, b_hi = 0, b_qs = 0, b_qe = 0, b_k = 0, b_len = 0;
        uint64_t j = m;                                 // next symbol to consume is q[j-1]
        uint32_t nlen = 0;                              // > 0: a k-mer step over q[j-nlen, j) came back empty and is being narrowed
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        if (STATS) { rd.fetched = &st[kStChunks]; if (valid) st[kSdSequences] += 1; }
        auto on_ok = [&](uint32_t adv) {
            j -= adv;
            if (STATS) st[kStSymbols] += adv;
            plo = lo; phi = hi; pk = k;                 // rowbowt.hpp:248-249
            ppend = pend; ppend_d = pend_d; ppend_rec = pend_rec; ppend_e = pend_e;
        };
        bool b_pend = false;                            // ... and of the best seed's toehold: fetched once, when the read is done
        uint32_t b_pd = 0, b_prec = 0;
        uint64_t b_pe = 0;
        auto take_best = [&]() { b_k = pk; b_pend = ppend; b_pd = ppend_d; b_prec = ppend_rec; b_pe = ppend_e; };
        auto on_fail = [&]() {                          // q[j-1] ends the seed q[j, ei)  (rowbowt.hpp:236-246; m-i == j here)
            if (ei - j >= min_length && ei - j > b_len) { b_len = ei - j; b_lo = plo; b_hi = phi; b_qs = j; b_qe = ei; take_best(); }
            k = first_k;
            pend = false; ppend = false;
            lo = 0; hi = fhi; plo = 0; phi = fhi;
            j -= 1;                                     // skip the base that failed
            ei = j;
        };
        while (__ballot(valid && j > 0)) {
            bool stepping = false;
            StepPick pick{1u, 0u, 0u, true};
            while (valid && j > 0 && !stepping) {       // lane-local work until a rank is needed
                const uint64_t p = beg + j - 1;
                if (nlen == 0) {
                    // a fresh seed: the state after its first ftab_k symbols is one gather in the device table
                    const uint32_t tc = static_cast<uint32_t>(m - j);   // STAGED: symbols between q[j-1] and the read's end
                    if (j == ei && ix.ftab_k && j >= ix.ftab_k) {
                        if (STATS) st[kStFtab] += 1;
                        if (STAGED ? ftab_state_staged<P>(ix, codes, tc, lo, hi, k) : ftab_state<P>(ix, rd, s_lut2, p, M, lo, hi, k)) { pend = false; on_ok(ix.ftab_k); continue; }
                    }
                    pick = STAGED ? pick_step_staged(codes, s_mslot, tab_first, tc, j, D, DMASK) : pick_step(rd, s_lut, s_lut2, tab_first, p, j, D, DMASK, M);
                    if (!pick.ok) { on_fail(); continue; }
                } else {
                    // == nlen / 2 symbols: the window holds k-mer symbols only
                    pick = STAGED ? pick_step_staged(codes, s_mslot, tab_first, static_cast<uint32_t>(m - j), nlen / 2, D, DMASK)
                                  : pick_step(rd, s_lut, s_lut2, tab_first, p, nlen / 2, D, DMASK, M);
                }
                stepping = true;
            }
            RunStep r;
            seeds_lf2<P, (QUAD || sizeof(P) == 8), QUAD, STATS>(S2, stepping, pick.d, pick.rec, lo, hi + 1, r, st);
            if (stepping) {
                const uint64_t c_inside = r.c_upto - r.c_before;
                const bool ok = c_inside != 0;
                if (ok) {                               // LF_w_loc, rowbowt.hpp:555-573, pick.adv times nested
                    if (r.inside) k = k - pick.adv;
                    else { pend = true; pend_d = pick.d; pend_rec = pick.rec; pend_e = r.samp_e; k = 0; }
                    lo = r.F + r.c_before;
                    hi = lo + c_inside - 1;
                }
                if (nlen == 0) {
                    if (ok) on_ok(pick.adv);
                    else if (pick.adv == 1) on_fail();
                    else nlen = pick.adv;               // the range died inside q[j-adv, j): halve until one symbol is left
                } else {
                    if (ok) { on_ok(pick.adv); nlen -= pick.adv; } else nlen = pick.adv;
                }
                if (nlen == 1) { on_fail(); nlen = 0; }  // that symbol is the failing base
            }
        }
        if (valid) {
            if (ei >= min_length && ei > b_len) { b_len = ei; b_lo = plo; b_hi = phi; b_qs = 0; b_qe = ei; take_best(); }  // :252-254
            if (b_pend) { b_k += run_step_sample2<P>(ix, S2, b_pd, b_prec, b_pe); if (STATS) st[kStResample] += 1; }
            lo_out[i] = b_lo;
            hi_out[i] = b_hi;
            qs_out[i] = b_qs;
            qe_out[i] = b_qe;
            ss_out[i] = b_k;
        }
        };
        if (staged) walk(std::true_type{}); else walk(std::false_type{});
    }
    if (STATS) seed_stats_flush(st, stats, lane);
}

// ---- marker seeds: RowBowt::get_markers_greedy_seeding without an ftab file (rowbowt.hpp:406-482; rb_markers' default
// path, rb_markers.cpp:411-413) -- k_markers.hip k_marker_seeds<P, FILL>'s default mode; the --ftab mode of the tool stays
// with that kernel (lane by lane).  One record per call of the reference's callback: {range lo, range hi, q.first,
// seed_ei, first marker, one past last marker}.
// LOG / lg: the marker-seed log (rbg_dev.h SeedLog), as in k_markers.hip k_marker_seeds
template <typename P, bool FILL, bool LOG, bool STATS = false>
__global__ __launch_bounds__(512, STATS ? 2 : 4) void k_marker_seeds_runs(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                      const uint64_t *__restrict__ off, const uint64_t N,
                                                      const uint64_t wsize, const uint64_t max_range,
                                                      uint64_t *__restrict__ seed_cnt, uint64_t *__restrict__ mk_cnt,
                                                      const uint64_t *__restrict__ seed_off, const uint64_t *__restrict__ mk_off,
                                                      uint64_t *__restrict__ seeds, uint64_t *__restrict__ mk, const SeedLog lg) {
    RBG_SEED_KERNEL_PROLOGUE(P);
    RBG_SEED_STAGE_SHARED;
    unsigned long long st[kSeedStatN] = {};               // STATS only (dead code otherwise)
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) { seed_cnt[0] = 0; mk_cnt[0] = 0; }
    const bool have_ma = ix.mk_nruns != 0;
    const uint64_t fhi = ix.n - 1;
    const bool listed = FILL && lg.base != nullptr;       // fill pass after a logged count pass: only the sequences over quota
    const uint64_t Neff = listed ? static_cast<uint64_t>(lg.nsel[0]) : N;
    for (uint64_t base = wave_first; base < Neff; base += stride) {
        const bool valid = base + lane < Neff;
        const uint64_t i = (listed && valid) ? static_cast<uint64_t>(lg.nsel[4 + base + lane]) : base + lane;
        uint64_t beg = 0, m = 0;
        if (valid) { beg = off[i]; m = off[i + 1] - beg; }
        bool staged = false;                              // the wave's reads as 2-bit codes in LDS (as in k_greedy_seed_runs)
        if (stage_on && __ballot(valid && m > kStageCap) == 0) {
            uint32_t nch = 0;
            const bool bad = valid && stage_read(reinterpret_cast<const uint4 *>(seqs), beg, beg + m, stage_tab, codes, nch);
            staged = __ballot(bad) == 0;
            if (STATS && staged) st[kStChunks] += nch;
        }
        auto walk = [&](auto staged_tag) __attribute__((always_inline)) {
        constexpr bool STAGED = decltype(staged_tag)::value;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;    // range, prev_range (:427-428)
        uint64_t window_ei = m, seed_ei = m;              // :434
        uint64_t ns = 0, tot = 0, mb_begin = 0;           // mbuf == markers [mb_begin, tot) of this read
        uint64_t *srec = (FILL && valid) ? seeds + 6 * seed_off[i] : nullptr;
        const uint64_t mbase = (FILL && valid) ? mk_off[i] : 0;
        uint64_t j = m;                                   // m - i of the reference; the next symbol consumed is q[j-1]
        uint32_t nlen = 0;
        uint64_t unused_k = 0;
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        if (STATS) { rd.fetched = &st[kStChunks]; if (valid) st[kSdSequences] += 1; }
        unsigned char *lbase = (LOG && valid) ? lg.base + i * lg.stride : nullptr;
        // (the log's two arrays are addressed from lbase where they are written: two more pointers held over the whole walk cost the
        //  logging instantiation its fourth wave per SIMD)
        auto lrec = [&]() { return reinterpret_cast<SeedLogRec<P> *>(lbase + 8); };
        auto lwin = [&]() { return reinterpret_cast<SeedLogWin *>(lbase + 8 + static_cast<size_t>(lg.qs) * sizeof(SeedLogRec<P>)); };
        uint32_t nw = 0;
        bool lover = LOG && (m >> 32) != 0;
        auto update_mbuf = [&](uint64_t l, uint64_t h) {  // :437-441
            if (!have_ma || h - l + 1 > max_range) return;
            uint64_t src, cnt;
            if (STATS) st[kSdMarkerQueries] += 1;
            if (!marker_query(ix, l, h, &src, &cnt, STATS ? st : nullptr)) return;
            if (STATS && FILL) st[kSdMarkerVals] += cnt;
            if (FILL) {
                uint64_t *d = mk + mbase + tot;
                for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
            }
            if (LOG) {
                if (nw < lg.qw && ((src + cnt) >> 32) == 0) lwin()[nw] = SeedLogWin{static_cast<uint32_t>(src), static_cast<uint32_t>(cnt)};
                else lover = true;
                ++nw;
            }
            tot += cnt;
        };
        auto emit = [&](uint64_t l, uint64_t h, uint64_t qs, uint64_t qe) {  // fn(range, (qs, qe-1), mbuf)
            if (STATS && FILL) st[kSdSeedRecs] += 1;
            if (FILL) {
                uint64_t *d = srec + 6 * ns;
                d[0] = l; d[1] = h; d[2] = qs; d[3] = qe; d[4] = mbase + mb_begin; d[5] = mbase + tot;
            }
            if (LOG) {
                if (ns < lg.qs && (tot >> 32) == 0)
                    lrec()[ns] = SeedLogRec<P>{static_cast<P>(l), static_cast<P>(h), static_cast<uint32_t>(qs), static_cast<uint32_t>(qe),
                                             static_cast<uint32_t>(mb_begin), static_cast<uint32_t>(tot)};
                else lover = true;
            }
            ++ns;
        };
        auto on_ok = [&](uint32_t adv) {              // adv symbols consumed, range still non-empty
            j -= adv;
            if (STATS) st[kStSymbols] += adv;
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
        while (__ballot(valid && j > 0)) {
            bool stepping = false;
            StepPick pick{1u, 0u, 0u, true};
            while (valid && j > 0 && !stepping) {
                const uint64_t p = beg + j - 1;
                if (nlen == 0) {
                    // symbols that may be consumed before the next window query fires (:469)
                    uint64_t dist = j + wsize > window_ei ? j + wsize - window_ei : 1;
                    if (dist == 0) dist = 1;
                    const uint64_t cap = dist < j ? dist : j;
                    const uint32_t tc = static_cast<uint32_t>(m - j);   // STAGED: symbols between q[j-1] and the read's end
                    if (j == seed_ei && ix.ftab_k && cap >= ix.ftab_k) {
                        if (STATS) st[kStFtab] += 1;
                        if (STAGED ? ftab_state_staged<P>(ix, codes, tc, lo, hi, unused_k) : ftab_state<P>(ix, rd, s_lut2, p, M, lo, hi, unused_k)) { on_ok(ix.ftab_k); continue; }
                    }
                    pick = STAGED ? pick_step_staged(codes, s_mslot, tab_first, tc, cap, D, DMASK) : pick_step(rd, s_lut, s_lut2, tab_first, p, cap, D, DMASK, M);
                    if (!pick.ok) { on_fail(); continue; }
                } else {
                    pick = STAGED ? pick_step_staged(codes, s_mslot, tab_first, static_cast<uint32_t>(m - j), nlen / 2, D, DMASK)
                                  : pick_step(rd, s_lut, s_lut2, tab_first, p, nlen / 2, D, DMASK, M);
                }
                stepping = true;
            }
            RunStep r;
            seeds_lf2<P, true, true, STATS>(S2, stepping, pick.d, pick.rec, lo, hi + 1, r, st);
            if (stepping) {
                const uint64_t c_inside = r.c_upto - r.c_before;
                const bool ok = c_inside != 0;
                if (ok) {
                    lo = r.F + r.c_before;
                    hi = lo + c_inside - 1;
                }
                if (nlen == 0) {
                    if (ok) on_ok(pick.adv);
                    else if (pick.adv == 1) on_fail();
                    else nlen = pick.adv;
                } else {
                    if (ok) { on_ok(pick.adv); nlen -= pick.adv; } else nlen = pick.adv;
                }
                if (nlen == 1) { on_fail(); nlen = 0; }
            }
        }
        if (valid) {
            if (hi >= lo && seed_ei >= wsize) update_mbuf(lo, hi);   // :478-480 (m-i == 0)
            emit(lo, hi, 0, seed_ei);                                // :481
            if (!FILL) {
                seed_cnt[i + 1] = ns;
                mk_cnt[i + 1] = tot;
            }
            if (LOG) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(lbase);
                hdr[0] = lover ? kSeedLogOverflow : static_cast<uint32_t>(ns);
                hdr[1] = nw;
            }
        }
        };
        if (staged) walk(std::true_type{}); else walk(std::false_type{});
    }
    if (STATS) seed_stats_flush(st, lg.stats, lane);
}

// ---- marker seeds with an ftab of k-mer size K (rb_markers --ftab; rowbowt.hpp:406-482 with :430-433, :454-464) on format 2:
// the reference's loop in its own index, one lane per sequence (k_markers.hip k_marker_seeds' ftab branch, whose ranks on this
// layout were a per-lane binary search of the symbol's run list until round 4).  search_ftab (:746-758) on the table
// build_ftab(K) makes for this index is find_range of an ACGT-only k-mer: K symbols from the full range, taken as k-mer
// steps through the depths' run lists; every other step of the loop is a single symbol, as in the reference.
template <typename P, bool FILL, bool LOG>
__global__ __launch_bounds__(512, sizeof(P) == 8 ? 3 : 4) void k_marker_seeds_ftab_runs2(const DevIndex ix, const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                                                    const uint64_t N, const uint64_t wsize, const uint64_t max_range, const uint64_t K,
                                                                    uint64_t *__restrict__ seed_cnt, uint64_t *__restrict__ mk_cnt,
                                                                    const uint64_t *__restrict__ seed_off, const uint64_t *__restrict__ mk_off,
                                                                    uint64_t *__restrict__ seeds, uint64_t *__restrict__ mk, const SeedLog lg) {
    RBG_SEED_KERNEL_PROLOGUE(P);
    (void)lane; (void)wave_first;
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) { seed_cnt[0] = 0; mk_cnt[0] = 0; }
    const bool have_ma = ix.mk_nruns != 0;
    const uint64_t fhi = ix.n - 1;
    const bool listed = FILL && lg.base != nullptr;
    const uint64_t Neff = listed ? static_cast<uint64_t>(lg.nsel[0]) : N;
    for (uint64_t j_ = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j_ < Neff; j_ += stride) {
        const uint64_t i = listed ? static_cast<uint64_t>(lg.nsel[4 + j_]) : j_;
        const uint64_t beg = off[i], m = off[i + 1] - beg;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;    // range, prev_range (:427-428)
        uint64_t window_ei = m, seed_ei = m;              // :434
        uint64_t ns = 0, tot = 0, mb_begin = 0;
        uint64_t *srec = FILL ? seeds + 6 * seed_off[i] : nullptr;
        const uint64_t mbase = FILL ? mk_off[i] : 0;
        unsigned char *lbase = LOG ? lg.base + i * lg.stride : nullptr;
        // (the log's two arrays are addressed from lbase where they are written: two more pointers held over the whole walk cost the
        //  logging instantiation its fourth wave per SIMD)
        auto lrec = [&]() { return reinterpret_cast<SeedLogRec<P> *>(lbase + 8); };
        auto lwin = [&]() { return reinterpret_cast<SeedLogWin *>(lbase + 8 + static_cast<size_t>(lg.qs) * sizeof(SeedLogRec<P>)); };
        uint32_t nw = 0;
        bool lover = LOG && (m >> 32) != 0;
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        auto update_mbuf = [&](uint64_t l, uint64_t h) {  // :437-441
            if (!have_ma || h - l + 1 > max_range) return;
            uint64_t src, cnt;
            if (!marker_query(ix, l, h, &src, &cnt)) return;
            if (FILL) {
                uint64_t *d = mk + mbase + tot;
                for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
            }
            if (LOG) {
                if (nw < lg.qw && ((src + cnt) >> 32) == 0) lwin()[nw] = SeedLogWin{static_cast<uint32_t>(src), static_cast<uint32_t>(cnt)};
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
                    lrec()[ns] = SeedLogRec<P>{static_cast<P>(l), static_cast<P>(h), static_cast<uint32_t>(qs), static_cast<uint32_t>(qe),
                                             static_cast<uint32_t>(mb_begin), static_cast<uint32_t>(tot)};
                else lover = true;
            }
            ++ns;
        };
        // one step of at most `cap` symbols ending at byte p on (lo, hi): *adv = symbols consumed; false = empty range, (lo, hi) untouched
        auto step = [&](uint64_t p, uint64_t cap, uint32_t *adv) -> bool {
            const StepPick pick = pick_step(rd, s_lut, s_lut2, tab_first, p, cap, D, DMASK, M);
            *adv = pick.adv;
            if (!pick.ok) return false;
            RunStep r;
            lane_lf2<P>(S2, pick.d, pick.rec, lo, hi + 1, r);
            if (r.c_upto <= r.c_before) return false;
            lo = r.F + r.c_before;
            hi = lo + (r.c_upto - r.c_before) - 1;
            return true;
        };
        auto ftab_hit = [&](uint64_t e) -> bool {   // k-mer q[e-K, e); on a hit (lo,hi) is its range
            for (uint64_t t = e - K; t < e; ++t) {
                const uint32_t c = rd.at(beg + t);
                if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
            }
            lo = 0; hi = fhi;
            uint64_t e2 = e;
            while (e2 > e - K) {
                uint32_t adv;
                if (!step(beg + e2 - 1, e2 - (e - K), &adv)) return false;   // a k-mer of the word is absent: so is the word
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
            uint32_t adv1;
            if (step(beg + m - i2 - 1, 1, &adv1)) {        // :443
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
                    // :454-464 (k_markers.hip: search_ftab answers an absent k-mer with the full range, so the reference's
                    // loop always leaves on its first iteration: a hit continues from the k-mer's range, a miss from the
                    // FULL range, the K bases skipped either way)
                    if (!ftab_hit(m - i2 - 1)) { lo = 0; hi = fhi; }
                    i2 += K;                           // :460
                    plo = lo; phi = hi;                // :461
                }
            }
        }
        if (hi >= lo && seed_ei - (m - i2) >= wsize) update_mbuf(lo, hi);   // :478-480
        emit(lo, hi, m - i2, seed_ei);                                      // :481
        if (!FILL) {
            seed_cnt[i + 1] = ns;
            mk_cnt[i + 1] = tot;
        }
        if (LOG) {
            uint32_t *hdr = reinterpret_cast<uint32_t *>(lbase);
            hdr[0] = lover ? kSeedLogOverflow : static_cast<uint32_t>(ns);
            hdr[1] = nw;
        }
    }
}

struct SeedLaunch {
    dim3 grid, block;
    size_t lds;
};
SeedLaunch seed_launch(const DevIndex &ix, const LaunchCfg &cfg, uint64_t N) {
    LaunchCfg c = cfg;   // 512-thread workgroups: the staged tables are shared by eight waves
    c.block_threads = 512;
    c.max_blocks = cfg.max_blocks > 0 ? std::max(1, cfg.max_blocks / 2) : 256 * 16;
    return SeedLaunch{dim3(grid_for(c, N)), dim3(512), run_search2_lds(ix)};
}

}  // namespace

#define RBG_LAUNCH_SEEDK(KERN, ...)                                                     \
    do {                                                                                \
        auto kern = KERN;                                                               \
        raise_lds(kern, L.lds, 8 * kStageWaveBytes);   /* (the two seed walks hold the staged reads: static LDS) */ \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, __VA_ARGS__);          \
    } while (0)

int launch_lf_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym, uint64_t N,
                   uint64_t *lo_out, uint64_t *hi_out, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const SeedLaunch L = seed_launch(ix, cfg, N);
    if (ix.pos_bytes == 4) RBG_LAUNCH_SEEDK((k_lf_runs<uint32_t>), lo, hi, sym, N, lo_out, hi_out);
    else RBG_LAUNCH_SEEDK((k_lf_runs<uint64_t>), lo, hi, sym, N, lo_out, hi_out);
    return static_cast<int>(hipGetLastError());
}

int launch_find_range_markers_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                                   uint64_t wsize, uint64_t max_range, uint64_t *lo, uint64_t *hi, uint64_t *cnt, const uint64_t *mk_off,
                                   uint64_t *mk, bool fill, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const SeedLaunch L = seed_launch(ix, cfg, N);
#define RBG_FRM(PT)                                                                                                                           \
    do {                                                                                                                                     \
        if (fill) RBG_LAUNCH_SEEDK((k_find_range_markers_runs<PT, true>), seqs, off, N, wsize, max_range, lo, hi, cnt, mk_off, mk);           \
        else RBG_LAUNCH_SEEDK((k_find_range_markers_runs<PT, false>), seqs, off, N, wsize, max_range, lo, hi, cnt, mk_off, mk);               \
    } while (0)
    if (ix.pos_bytes == 4) RBG_FRM(uint32_t); else RBG_FRM(uint64_t);
#undef RBG_FRM
    return static_cast<int>(hipGetLastError());
}

int launch_greedy_seed_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                            uint64_t min_length, uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, void *stream, unsigned long long *stats) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const SeedLaunch L = seed_launch(ix, cfg, N);
    if (stats) {   // the instrumented instantiation (rbg_greedy_longest_seed_stats_dev): same walk, same outputs
        if (ix.pos_bytes == 4) RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint32_t, false, true>), seqs, off, N, min_length, lo, hi, qs, qe, ss, stats);
        else RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint64_t, false, true>), seqs, off, N, min_length, lo, hi, qs, qe, ss, stats);
        return static_cast<int>(hipGetLastError());
    }
    // every lane fetches its own record: the quad fetch of k_find_range_runs (rbg_runs2_device.hpp lane_lf2_quad) measured the same here (6.80 ms
    // against 6.77 per 10 M reads on the bench index, profiles/r05_experiments.txt) and costs the 8-byte instantiation its fourth wave per SIMD
    // (131 VGPRs).  RBG_GREEDY_FETCH=quad keeps the A/B runnable.
    static const bool quad = [] { const char *e = std::getenv("RBG_GREEDY_FETCH"); return e && e[0] == 'q'; }();
    if (quad) {
        if (ix.pos_bytes == 4) RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint32_t, true>), seqs, off, N, min_length, lo, hi, qs, qe, ss, nullptr);
        else RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint64_t, true>), seqs, off, N, min_length, lo, hi, qs, qe, ss, nullptr);
    } else {
        if (ix.pos_bytes == 4) RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint32_t, false>), seqs, off, N, min_length, lo, hi, qs, qe, ss, nullptr);
        else RBG_LAUNCH_SEEDK((k_greedy_seed_runs<uint64_t, false>), seqs, off, N, min_length, lo, hi, qs, qe, ss, nullptr);
    }
    return static_cast<int>(hipGetLastError());
}

int launch_marker_seeds_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize,
                             uint64_t max_range, uint64_t *seed_cnt, uint64_t *mk_cnt, const uint64_t *seed_off, const uint64_t *mk_off,
                             uint64_t *seeds, uint64_t *mk, bool fill, void *stream, const SeedLog &lg, uint64_t ftab_k) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    SeedLaunch L = seed_launch(ix, cfg, N);
    if (fill && lg.base) L.grid = dim3(std::min<unsigned>(L.grid.x, 128u));   // the listed sequences only: their number is on the device
    if (ftab_k) {   // rb_markers --ftab
#define RBG_MSF(PT)                                                                                                                          \
    do {                                                                                                                                     \
        if (fill) RBG_LAUNCH_SEEDK((k_marker_seeds_ftab_runs2<PT, true, false>), seqs, off, N, wsize, max_range, ftab_k, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
        else if (lg.base) RBG_LAUNCH_SEEDK((k_marker_seeds_ftab_runs2<PT, false, true>), seqs, off, N, wsize, max_range, ftab_k, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
        else RBG_LAUNCH_SEEDK((k_marker_seeds_ftab_runs2<PT, false, false>), seqs, off, N, wsize, max_range, ftab_k, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
    } while (0)
        if (ix.pos_bytes == 4) RBG_MSF(uint32_t); else RBG_MSF(uint64_t);
#undef RBG_MSF
        return static_cast<int>(hipGetLastError());
    }
    if (lg.stats && !lg.base) {   // the instrumented instantiations (rbg_marker_seeds_stats_dev): same walks, same outputs
#define RBG_MSS(PT)                                                                                                                          \
    do {                                                                                                                                     \
        if (fill) RBG_LAUNCH_SEEDK((k_marker_seeds_runs<PT, true, false, true>), seqs, off, N, wsize, max_range, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
        else RBG_LAUNCH_SEEDK((k_marker_seeds_runs<PT, false, false, true>), seqs, off, N, wsize, max_range, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
    } while (0)
        if (ix.pos_bytes == 4) RBG_MSS(uint32_t); else RBG_MSS(uint64_t);
#undef RBG_MSS
        return static_cast<int>(hipGetLastError());
    }
#define RBG_MSR(PT)                                                                                                                          \
    do {                                                                                                                                     \
        if (fill) RBG_LAUNCH_SEEDK((k_marker_seeds_runs<PT, true, false>), seqs, off, N, wsize, max_range, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
        else if (lg.base) RBG_LAUNCH_SEEDK((k_marker_seeds_runs<PT, false, true>), seqs, off, N, wsize, max_range, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
        else RBG_LAUNCH_SEEDK((k_marker_seeds_runs<PT, false, false>), seqs, off, N, wsize, max_range, seed_cnt, mk_cnt, seed_off, mk_off, seeds, mk, lg); \
    } while (0)
    if (ix.pos_bytes == 4) RBG_MSR(uint32_t); else RBG_MSR(uint64_t);
#undef RBG_MSR
    return static_cast<int>(hipGetLastError());
}
#undef RBG_LAUNCH_SEEDK

}  // namespace rbg
