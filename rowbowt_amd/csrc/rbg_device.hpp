// rbg_device.hpp -- device-side building blocks shared by the kernel translation units (k_search.hip,
// k_locate.hip, k_markers.hip, k_build.hip): the rank over one slot table, the staged record table, the read
// cursor, and the host-side launch geometry.  Everything lives in an anonymous namespace: each unit gets its
// own inlined copy.
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "rbg_dev.h"

namespace rbg {
namespace {

constexpr int kWave = 64;

// A slot is 4 words and must arrive as ONE request: left to itself the compiler fetches the word
// that decides a branch first and the rest later (two gather requests instead of one; seen in the
// ISA of k_locate_fill), so slots are loaded through a vector type.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// A table pointer that a kernel read out of a record staged in LDS (DevSym, DevTree) has no known address space, and the
// compiler loads through it with FLAT instructions: they count against the LDS counter as well as the memory counter, so
// every wait for an LDS read also waits for the gathers in flight.  The index lives in device memory: say so.
#define RBG_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const RBG_GLOBAL T *as_global(const T *p) { return (const RBG_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ const RBG_GLOBAL T *as_global(const void *p) { return (const RBG_GLOBAL T *)p; }

// The bucket of a position under a PER-LANE shift (the run-indexed layout: every table has its own bucket width, sh < 32): q >> sh in
// 32-bit pieces.  Not the 64-bit shift -- on gfx950 a v_lshrrev_b64 / v_lshlrev_b64 whose AMOUNT sits in the last VGPR the kernel
// allocates (and the allocation ends on a granule of eight) reads its amount from elsewhere: observed in k_lf_runs (72 VGPRs, amount in
// v71: the result was shifted by the contents of v0), as a rank answered from another bucket's record or a memory fault, depending on what the
// neighbouring wave had left in the register file (profiles/r05_shift64_last_vgpr.md).  A shift by a kernel argument is scalar and safe;
// tests/test_capi_host.py scans the ISA of every kernel for the pattern.
template <typename P>
__device__ __forceinline__ uint64_t pos_bucket(const uint64_t q, const uint32_t sh) {
    const uint32_t lo = static_cast<uint32_t>(q), hi = static_cast<uint32_t>(q >> 32);
    if constexpr (sizeof(P) == 4) return lo >> sh;   // (4-byte positions: q <= n < 2^32)
    else return (static_cast<uint64_t>(hi >> sh) << 32) | __builtin_amdgcn_alignbit(hi, lo, sh);
}
// low word of the bucket's first position, and the position's offset into its bucket
__device__ __forceinline__ uint32_t pos_bucket_base32(const uint64_t q, const uint32_t sh) { return static_cast<uint32_t>(q) & (~0u << sh); }
__device__ __forceinline__ uint32_t pos_bucket_offset(const uint64_t q, const uint32_t sh) { return static_cast<uint32_t>(q) & ~(~0u << sh); }

template <typename SlotT>
__device__ __forceinline__ SlotT load_slot(const SlotT *p) {
    SlotT s;
    if constexpr (sizeof(SlotT) == 16) {
        const u32x4 t = *as_global<u32x4>(static_cast<const void *>(p));
        __builtin_memcpy(&s, &t, 16);
    } else {
        static_assert(sizeof(SlotT) == 32, "slot is 4 words of 4 or 8 bytes");
        const u64x2 a = *as_global<u64x2>(static_cast<const void *>(p));
        const u64x2 b = *(as_global<u64x2>(static_cast<const void *>(p)) + 1);
        __builtin_memcpy(&s, &a, 16);
        __builtin_memcpy(reinterpret_cast<char *>(&s) + 16, &b, 16);
    }
    return s;
}

// ---- rank over one symbol table -----------------------------------------------------------------
// Returns # of the symbol in BWT[0,i)  ==  rle_string::rank(i,c) (rle_string.hpp:131-161).
// `sl` is the RankSlot of bucket b = i >> shift, already in registers (rbg_dev.h): nothing else is
// read unless the bucket holds more than 4 run starts.
// *nbefore = # runs of the symbol that start in [B0, i)  (-> ordinal of the predecessor run),
// *inside  = position i-1 holds the symbol; both feed the toehold update.
struct RankAux {
    uint32_t nbefore;  // overflow bucket: i - B0 instead (the run ordinal is searched only if it is needed)
    bool inside;
    bool ovf;
    bool dense;        // answered from a dense overflow table (read by the instrumented kernels only)
};

// # runs of the symbol that start before i, searched in the run list of bucket b (overflow buckets without
// a dense table only)
template <typename P>
__device__ __forceinline__ uint64_t search_runs(const RunEnt<P> *__restrict__ ent_, uint64_t a, uint64_t z, uint64_t i) {
    const RBG_GLOBAL RunEnt<P> *ent = as_global(ent_);
    while (z - a > 4) {
        const uint64_t mid = a + ((z - a) >> 1);
        if (static_cast<uint64_t>(ent[mid].start) < i) a = mid + 1; else z = mid;
    }
    while (a < z && static_cast<uint64_t>(ent[a].start) < i) ++a;
    return a;
}
template <typename P>
__device__ __forceinline__ uint64_t runs_before(const DevSym &S, uint64_t b, uint64_t i) {
    const RBG_GLOBAL uint32_t *ord = as_global(S.ord);
    return search_runs<P>(static_cast<const RunEnt<P> *>(S.ent), ord[b], ord[b + 1], i);
}

template <typename P>
__device__ __forceinline__ uint64_t rank_in_slot(const DevSym &S, const RankSlot &sl, uint64_t b, uint64_t i,
                                                 const uint8_t *__restrict__ dense, RankAux *aux) {
    const uint32_t w1 = sl.w1, w2 = sl.w2, w3 = sl.w3;
    const bool wide = S.shift > kMaxNarrowShift;
    aux->dense = false;
    const uint32_t cnt = wide ? (w1 >> 21) & 7u : (w1 >> 9) & 7u;
    if (wide && cnt != kSlotOvf) {  // wide-bucket encoding (rbg_dev.h); overflow buckets share the path below
        const uint32_t o = static_cast<uint32_t>(i - (b << S.shift));
        const uint32_t ext = (w1 >> 8) & 0x1FFFu;
        uint32_t add = o < ext ? o : ext;
        bool in = o ? (o <= ext) : ((w1 >> 24) & 1u);
        uint32_t nb = 0;
#define RBG_RUNW(field)                                           \
    {                                                             \
        const uint32_t run_ = (field) & 0xFFFFFFu;                \
        const uint32_t off_ = run_ & 0xFFFu;                      \
        const uint32_t len_ = (run_ >> 12) + 1u;                  \
        if (run_ != 0xFFFFFFu && o > off_) {                      \
            const uint32_t d_ = o - off_;                         \
            add += d_ < len_ ? d_ : len_;                         \
            in = in || d_ <= len_;                                \
            ++nb;                                                 \
        }                                                         \
    }
        RBG_RUNW(w2)
        RBG_RUNW(w3)
#undef RBG_RUNW
        aux->ovf = false;
        aux->nbefore = nb;
        aux->inside = in;
        return (static_cast<uint64_t>(sl.r0) | (static_cast<uint64_t>(w1 & 0xFFu) << 32)) + add;
    }
    if (cnt == kSlotOvf) {
        const uint32_t o = static_cast<uint32_t>(i - (b << S.shift));
        if (dense && !wide) {
            // dense bucket (rbg_dev.h): two bytes per row -- rank(B0 + o) - rank(B0), and 255 if position
            // i-1 holds the symbol, else the number of runs that start in [B0, i)
            const uint32_t e = as_global<uint16_t>(dense + (static_cast<uint64_t>(w2) << 4))[o];
            aux->ovf = false;
            aux->dense = true;
            aux->nbefore = e >> 8;
            aux->inside = (e >> 8) == 255u;
            return (static_cast<uint64_t>(sl.r0) | (static_cast<uint64_t>(w3 >> 16) << 32)) + (e & 0xFFu);
        }
        aux->ovf = true;
        aux->nbefore = o;
        const RBG_GLOBAL RunEnt<P> *ent = as_global<RunEnt<P>>(S.ent);
        const uint64_t a = runs_before<P>(S, b, i);
        if (a == 0) { aux->inside = false; return 0; }
        const uint64_t e_start = ent[a - 1].start, e_cum = ent[a - 1].cum;
        const uint64_t len = static_cast<uint64_t>(ent[a].cum) - e_cum;
        const uint64_t d = i - e_start;
        aux->inside = d <= len;
        return e_cum + (d < len ? d : len);
    }
    const uint32_t o = static_cast<uint32_t>(i - (b << S.shift));
    const uint32_t ext = w1 & 0x1FFu;
    uint32_t add = o < ext ? o : ext;
    bool in = o ? (o <= ext) : ((w1 >> 12) & 1u);
    uint32_t nb = 0;
#define RBG_RUN(field)                                            \
    {                                                             \
        const uint32_t run_ = (field);                            \
        const uint32_t off_ = run_ & 0xFFu;                       \
        const uint32_t len_ = ((run_ >> 8) & 0xFFu) + 1u;         \
        if (o > off_) {                                           \
            const uint32_t d_ = o - off_;                         \
            add += d_ < len_ ? d_ : len_;                         \
            in = in || d_ <= len_;                                \
            ++nb;                                                 \
        }                                                         \
    }
    RBG_RUN(w1 >> 16)
    RBG_RUN(w2 & 0xFFFFu)
    RBG_RUN(w2 >> 16)
    RBG_RUN(w3 & 0xFFFFu)
#undef RBG_RUN
    aux->ovf = false;
    aux->nbefore = nb;
    aux->inside = in;
    return (static_cast<uint64_t>(sl.r0) | (static_cast<uint64_t>(w3 >> 16) << 32)) + add;
}

// a sample of the run-indexed layout's run lists (rbg_dev.h RunsFmt: 4 bytes, or 6 bytes at 8-byte positions)
template <typename P> struct RunList;
template <> struct RunList<uint32_t> {
    static __device__ __forceinline__ uint64_t samp(const void *__restrict__ b, uint64_t i) { return as_global<uint32_t>(b)[i]; }
};
template <> struct RunList<uint64_t> {
    static __device__ __forceinline__ uint64_t samp(const void *__restrict__ b, uint64_t i) {
        const RBG_GLOBAL uint16_t *h = as_global<uint16_t>(b) + 3 * i;
        return static_cast<uint64_t>(h[0]) | (static_cast<uint64_t>(h[1]) << 16) | (static_cast<uint64_t>(h[2]) << 32);
    }
};

// both ranks of one LF step (rowbowt.hpp:79,83).  lo and hi+1 usually share a bucket late in the
// search (the range has narrowed to a few dozen rows), so the step is ONE 4-word load.
template <typename P>
__device__ __forceinline__ void rank_pair(const DevSym &S, const uint8_t *__restrict__ dense, uint64_t lo, uint64_t hi1,
                                          uint64_t *c_before, uint64_t *c_upto, uint64_t *bh_out, RankAux *qaux,
                                          RankAux *paux_out = nullptr) {
    const RankSlot *__restrict__ slots = static_cast<const RankSlot *>(S.slots);
    const uint64_t bl = lo >> S.shift, bh = hi1 >> S.shift;
    const RankSlot sl = load_slot(slots + bl);
    RankSlot sh = sl;
    if (bh != bl) sh = load_slot(slots + bh);
    RankAux paux;
    *c_before = rank_in_slot<P>(S, sl, bl, lo, dense, &paux);
    *c_upto = rank_in_slot<P>(S, sh, bh, hi1, dense, qaux);
    *bh_out = bh;
    if (paux_out) *paux_out = paux;
}

// ordinal of the last run of the symbol that starts before the position a RankAux describes
template <typename P>
__device__ __forceinline__ uint64_t pred_run(const DevSym &S, uint64_t b, bool ovf, uint32_t v) {
    return (ovf ? runs_before<P>(S, b, (b << S.shift) + v) : static_cast<uint64_t>(as_global(S.ord)[b]) + v) - 1;
}

// samples_last_ of the last run of the symbol that starts before i (LF_w_loc, rowbowt.hpp:563-566);
// only taken when position i-1 does not hold the symbol, which is the rare case.
template <typename P>
__device__ __forceinline__ uint64_t pred_sample(const DevSym &S, uint64_t b, const RankAux &aux) {
    const uint64_t j = pred_run<P>(S, b, aux.ovf, aux.nbefore);
    return static_cast<uint64_t>(as_global<P>(S.samp)[j]);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving
    // the cross-lane reads above the writes (and vice versa)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- K3's staging: the values of a wave's 64 phi chains, CH steps at a time, and their coalesced flush (k_locate.hip k_locate_fill, k_runs.hip
// k_locate_fill_runs2).  One lane walks one read's chain (toehold_sa.hpp:37-49); the locations of a read are contiguous in `locs`, but a lane
// storing its own values would make every store instruction touch 64 different lines, and stores are gather-class requests like the slot loads
// (tools/gather_roof.hip).  So the values are staged per wave in LDS and flushed with CH lanes writing one read's (8 CH)-byte segment.
//
// What bounds the walk is the number of chains in flight (a phi step is a dependent gather), and what bounded THAT was this staging's LDS: 26.6 KB
// per 256-thread workgroup until round 6 (values + per lane an 8-byte destination, count, subtrahend and first location, a 4-byte flush alignment)
// = six waves per SIMD at 4-byte positions, five at 8-byte ones.  A workgroup now keeps per lane only what the FLUSHING lanes need:
//   val    the values of the round as 32-bit words, a row of CH * (sizeof(P) / 4) + 1 per lane (the odd row length spreads the owners' column
//          writes and the flush's row reads over the banks; an 8-byte position travels as two words so that the pad stays four bytes);
//   dst    where the read's locations start (8 bytes: 3.5e9 locations per batch at pangenome scale);
//   bounds which columns of THIS round's window are the read's (two bytes, rewritten per round) instead of the 8-byte count;
//   minus  locate_from_longest_seed's subtrahend (rowbowt.hpp:681-683): in the SUB instantiation only.
// = 19.3 KB (20.0 with the ring below): eight workgroups per CU, eight waves per SIMD (the kernels stay within 64 VGPRs).  profiles/r06_k3_lean_ab.txt:
// K3 -10 % at r = 1.2e8 (9.9-10.3 -> 8.9-9.2 ms per 10 M x 150 bp), -2.4 % on the bench index.
// The first location of a chain is the toehold itself, a text position except when it wrapped below zero (LF_w_loc, rowbowt.hpp:561: k - 1 at
// text position 0): at 4-byte positions such a value is staged as the all-ones word -- no position of such an index --, the flush skips it and
// the owner stores that one location itself (ChainStage::kSentinel; the high-byte form of 8-byte positions does the same).
// FLUSH WINDOWS ON BOUNDARIES OF THE OUTPUT ARRAY (RING; round 6, second half).  A scattered store's cost depends on its alignment by a factor of
// three (tools/scatter_width.hip, profiles/r06_scatter_width.txt: a 128-byte segment wherever a read's locations start 1.19 TB/s, on a 64-byte boundary
// 2.9; a 64-byte segment 0.85 against 2.89).  Aligning by making a lane WAIT before its first step lost twice (it breaks the phase of the walk).  Here
// nobody waits: step t of a chain is staged at VIRTUAL column v = a + t, a = (index of the read's first location in the output array) mod CH, in a ring of
// 2 CH columns; after the round of steps [t0, t0 + CH) every lane has produced all v < t0 + CH, so the WINDOW v in [t0, t0 + CH) is flushed -- it lies on
// a CH-element boundary of the output array for every read -- and the a values beyond it wait in the ring for the next flush.  Per lane and round the
// flushing lanes need the window's bounds, two bytes, instead of a count.
// 8-byte positions below 2^40 (HI8: every index built so far -- n = 3.0e11 at r = 1.07e9 is 2^38.1) stage a value as its low word + ONE high byte: 24 KB per
// workgroup with the ring, six waves per SIMD, where whole 8-byte values take 36 KB (four waves).
template <typename P, int CH, bool SUB, bool RING = true, bool HI8 = false>
struct ChainStage {
    static_assert(CH == 8 || CH == 16, "a flush pass serves 64 / CH reads");
    static_assert(!HI8 || sizeof(P) == 8, "the high byte belongs to 8-byte positions");
    static constexpr int W = (sizeof(P) == 8 && !HI8) ? 2 : 1, COLS = RING ? 2 * CH : CH, ROW = COLS * W + 1;
    static constexpr bool kSentinel = sizeof(P) == 4 || HI8;   // a first location outside the text travels as all ones (else it is staged whole)
    uint32_t val[4][kWave][ROW];
    uint64_t dst[4][kWave];                       // the read's first location, minus a: where virtual column 0 would land
    uint64_t minus[SUB ? 4 : 1][SUB ? kWave : 1];
    uint16_t bounds[4][kWave];                    // lo | hi << 8: the window's columns [lo, hi) are the read's (rewritten per round)
    uint8_t hi[HI8 ? 4 : 1][HI8 ? kWave : 1][HI8 ? COLS + 1 : 1];
};
// waves per SIMD the staging's LDS leaves room for (256-thread workgroups, 160 KB per CU): the kernels' launch bound
template <typename ST>
constexpr int chain_stage_waves() {
    constexpr int fit = static_cast<int>(160 * 1024 / sizeof(ST));
    return fit > 8 ? 8 : fit;
}
// the read's share of the round of steps [t0, t0 + CH) (the walk counts in 32 bits)
template <int CH>
__device__ __forceinline__ uint32_t chain_round_count(const uint64_t occ, const uint64_t t0) {
    return occ > t0 ? static_cast<uint32_t>(occ - t0 < static_cast<uint64_t>(CH) ? occ - t0 : CH) : 0u;
}
// the read's share of the WINDOW [t0, t0 + CH) of virtual columns: [max(a, t0), min(a + occ, t0 + CH)) relative to t0, as lo | hi << 8
template <int CH>
__device__ __forceinline__ uint16_t chain_window_bounds(const uint64_t occ, const uint32_t a, const uint64_t t0) {
    const uint64_t end = occ + a;                 // (occ == 0: a == 0, an empty window)
    const uint32_t lo = a > t0 ? (a - t0 < static_cast<uint64_t>(CH) ? static_cast<uint32_t>(a - t0) : CH) : 0u;
    const uint32_t hi = end > t0 ? (end - t0 < static_cast<uint64_t>(CH) ? static_cast<uint32_t>(end - t0) : CH) : 0u;
    return static_cast<uint16_t>(lo | (hi << 8));
}
// stage the value of virtual column v (= a + t); off_text: the value is no text position (only a chain's first can be: ChainStage::kSentinel forms)
template <typename P, int CH, bool SUB, bool RING, bool HI8>
__device__ __forceinline__ void chain_put(ChainStage<P, CH, SUB, RING, HI8> &S, const int wv, const int lane, const uint32_t v, const uint64_t x, const bool off_text) {
    const uint32_t c = v & (ChainStage<P, CH, SUB, RING, HI8>::COLS - 1);
    if (sizeof(P) == 4) {
        S.val[wv][lane][c] = off_text ? 0xFFFFFFFFu : static_cast<uint32_t>(x);
    } else if (HI8) {
        S.val[wv][lane][c] = off_text ? 0xFFFFFFFFu : static_cast<uint32_t>(x);
        S.hi[wv][lane][c] = off_text ? uint8_t(0xFF) : static_cast<uint8_t>(x >> 32);
    } else {
        S.val[wv][lane][2 * c] = static_cast<uint32_t>(x);
        S.val[wv][lane][2 * c + 1] = static_cast<uint32_t>(x >> 32);
    }
}
// after wave_lds_sync(): CH lanes per read, 64 / CH reads per pass; `locs[dst + t0 + e] = value - minus` for the columns [lo, hi) of the window at t0
template <typename P, int CH, bool SUB, bool RING, bool HI8, typename OUT>
__device__ __forceinline__ void chain_flush(const ChainStage<P, CH, SUB, RING, HI8> &S, const int wv, const int lane, const uint64_t t0, OUT *__restrict__ locs) {
    constexpr int G = kWave / CH;
    const uint32_t c0 = static_cast<uint32_t>(t0) & (ChainStage<P, CH, SUB, RING, HI8>::COLS - 1);   // (t0 is a multiple of CH: 0 or CH in the ring)
#pragma unroll
    for (int pass = 0; pass < CH; ++pass) {
        const int s = pass * G + lane / CH;
        const uint32_t e = lane & (CH - 1);
        const uint32_t bd = S.bounds[wv][s];
        if (e >= (bd & 0xFFu) && e < (bd >> 8)) {
            uint64_t x;
            if (sizeof(P) == 4) {
                const uint32_t x32 = S.val[wv][s][c0 + e];
                if (x32 == 0xFFFFFFFFu) continue;   // a first location outside the text: its owner stored it
                x = x32;
            } else if (HI8) {
                const uint32_t x32 = S.val[wv][s][c0 + e], h = S.hi[wv][s][c0 + e];
                if (x32 == 0xFFFFFFFFu && h == 0xFFu) continue;
                x = static_cast<uint64_t>(x32) | (static_cast<uint64_t>(h) << 32);
            } else {
                x = static_cast<uint64_t>(S.val[wv][s][2 * (c0 + e)]) | (static_cast<uint64_t>(S.val[wv][s][2 * (c0 + e) + 1]) << 32);
            }
            locs[S.dst[wv][s] + t0 + e] = static_cast<OUT>(x - (SUB ? S.minus[wv][s] : uint64_t(0)));
        }
    }
}
// positions below this fit the HI8 staging with the all-ones value to spare
constexpr uint64_t kChainHi8Limit = (uint64_t(1) << 40) - 16;
inline bool chain_hi8_enabled() {
    static const bool on = [] { const char *e = std::getenv("RBG_K3_HI8"); return !(e && e[0] == '0'); }();
    return on;
}

// ---- the marker query: MarkerArray::at_range(lo, hi) (rowbowt.hpp:272-290, :318, :437-441) as {src, cnt}: the values mk_vals[src, src + cnt) of all runs with
// start <= hi && end >= lo, in run order.  Runs are disjoint, ascending inclusive SA-index intervals.  From the bucket records (rbg_dev.h MkRec: one or two
// sectors), else from the directory + the run arrays.  st (instrumented seed walks): [kStatSearchN + 1] += records / directory entries read, [+ 2] += run starts /
// ends read, [+ 3] += value offsets read.
__device__ __forceinline__ void marker_span_arrays(const DevIndex &ix, uint64_t lo, uint64_t hi, uint64_t a, uint64_t z, uint64_t *first, uint64_t *last, unsigned long long *st) {
    while (a < ix.mk_nruns && ix.mk_end[a] < lo) { ++a; if (st) st[kStatSearchN + 2] += 1; }
    *first = a;
    if (z < a) z = a;
    while (z < ix.mk_nruns && ix.mk_start[z] <= hi) { ++z; if (st) st[kStatSearchN + 2] += 1; }
    *last = z;
    if (st) st[kStatSearchN + 2] += 2;
}
__device__ __forceinline__ bool marker_query(const DevIndex &ix, uint64_t lo, uint64_t hi, uint64_t *src, uint64_t *cnt, unsigned long long *st = nullptr) {
    if (lo >= ix.n) return false;          // caller-supplied rows beyond the BWT: nothing
    if (hi >= ix.n) hi = ix.n - 1;
    uint64_t f, l;
    if (ix.mk_rec) {
        const uint32_t sh = ix.mk_shift;
        const uint64_t b0 = lo >> sh, b1 = hi >> sh;
        const u32x4 *r0p = reinterpret_cast<const u32x4 *>(ix.mk_rec + b0), *r1p = reinterpret_cast<const u32x4 *>(ix.mk_rec + b1);
        MkRec R0, R1;
        const u32x4 x0 = as_global<u32x4>(static_cast<const void *>(r0p))[0], x1 = as_global<u32x4>(static_cast<const void *>(r0p))[1];
        __builtin_memcpy(&R0, &x0, 16); __builtin_memcpy(reinterpret_cast<char *>(&R0) + 16, &x1, 16);
        if (b1 != b0) {
            const u32x4 y0 = as_global<u32x4>(static_cast<const void *>(r1p))[0], y1 = as_global<u32x4>(static_cast<const void *>(r1p))[1];
            __builtin_memcpy(&R1, &y0, 16); __builtin_memcpy(reinterpret_cast<char *>(&R1) + 16, &y1, 16);
        } else {
            R1 = R0;
        }
        if (st) st[kStatSearchN + 1] += b1 != b0 ? 2 : 1;
        if (R0.nin != kMkRecOverflow && R1.nin != kMkRecOverflow) {
            const uint32_t lo_rel = static_cast<uint32_t>(lo - (b0 << sh)), hi_rel = static_cast<uint32_t>(hi - (b1 << sh));
            uint64_t off_f = static_cast<uint64_t>(R0.off_lo) | (static_cast<uint64_t>(R0.off_hi) << 32);
            uint64_t off_l = static_cast<uint64_t>(R1.off_lo) | (static_cast<uint64_t>(R1.off_hi) << 32);
            uint32_t nf = 0, nl = 0;
#pragma unroll
            for (uint32_t j = 0; j < kMkRecRuns; ++j) {
                const bool bf = j < R0.nin && R0.e_off[j] < lo_rel;      // (ends ascend: a prefix of the listed runs)
                const bool bl = j < R1.nin && R1.s_off[j] <= hi_rel;     // (starts ascend)
                nf += bf ? 1u : 0u; off_f += bf ? R0.cnt[j] : 0u;
                nl += bl ? 1u : 0u; off_l += bl ? R1.cnt[j] : 0u;
            }
            f = static_cast<uint64_t>(R0.a) + nf;
            l = static_cast<uint64_t>(R1.a) + nl;
            if (l <= f) return false;
            *src = off_f;
            *cnt = off_l - off_f;
            return true;
        }
        marker_span_arrays(ix, lo, hi, R0.a, R1.a, &f, &l, st);   // an overflowing bucket: the arrays, from the records' first runs
    } else if (ix.mk_bucket) {
        // first run with end >= lo: the directory gives the first run ending at or after the start of lo's bucket, the answer is at most a bucket's worth of runs
        // further on; one past the last run with start <= hi: every run before the entry of hi's bucket ends, hence starts, before hi
        if (st) st[kStatSearchN + 1] += 2;
        marker_span_arrays(ix, lo, hi, ix.mk_bucket[lo >> ix.mk_shift], ix.mk_bucket[hi >> ix.mk_shift], &f, &l, st);
    } else {
        uint64_t a = 0, z = ix.mk_nruns;
        while (a < z) { const uint64_t m = a + ((z - a) >> 1); if (ix.mk_end[m] < lo) a = m + 1; else z = m; }
        f = a;   // first run with end >= lo
        a = 0; z = ix.mk_nruns;
        while (a < z) { const uint64_t m = a + ((z - a) >> 1); if (ix.mk_start[m] <= hi) a = m + 1; else z = m; }
        l = a;   // one past the last run with start <= hi
    }
    if (l <= f) return false;
    if (st) st[kStatSearchN + 3] += 2;
    *src = ix.mk_off[f];
    *cnt = ix.mk_off[l] - *src;
    return true;
}

// ---- K1 / K2 ------------------------------------------------------------------------------------
// per-lane cursor over the read bytes, fetched as aligned 16-byte chunks while walking right to left
struct ByteCursor {
    const uint4 *__restrict__ chunks;
    uint64_t cur_ci;
    uint4 w;
    unsigned long long *fetched = nullptr;   // instrumented instantiations: counts the chunks fetched (nullptr, never assigned, elsewhere: folded away)
    __device__ __forceinline__ uint32_t at(uint64_t p) {
        const uint64_t ci = p >> 4;
        if (ci != cur_ci) { w = chunks[ci]; cur_ci = ci; if (fetched) ++*fetched; }
        // select + shift (indexing the vector by a run-time lane value makes the compiler spill it)
        const uint32_t sel = static_cast<uint32_t>(p) & 15u;
        const uint64_t lo64 = (static_cast<uint64_t>(w.y) << 32) | w.x;
        const uint64_t hi64 = (static_cast<uint64_t>(w.w) << 32) | w.z;
        const uint64_t half = (sel & 8u) ? hi64 : lo64;
        return static_cast<uint32_t>(half >> ((sel & 7u) * 8)) & 0xFFu;
    }
};

// LDS table of symbol / k-mer records: [singles | 2-mers | 3-mers | 4-mers | 5-mers]
constexpr int kOff2 = kLdsSyms;
constexpr int kOff3 = kOff2 + kMaxMajor * kMaxMajor;
constexpr int kOff4 = kOff3 + kMaxMajor * kMaxMajor * kMaxMajor;
constexpr int kOff5 = kOff4 + kMaxMajor * kMaxMajor * kMaxMajor * kMaxMajor;
constexpr int kTabMax = kOff5;                 // levels up to the 4-mers
constexpr int kTab5 = kOff5 + kMaxMajor * kMaxMajor * kMaxMajor * kMaxMajor * kMaxMajor;  // with the 5-mers (k_find_range's dynamic LDS)
constexpr uint32_t kHbmRec = 0x80000000u;  // record reference: symbol slot whose DevSym lives in HBM (ix.syms), not in s_tab

__device__ __forceinline__ void stage_tables(const DevIndex &ix, DevSym *s_tab, uint8_t *s_lut, uint8_t *s_lut2, bool with5 = false) {
    const uint32_t M = ix.nmajor;
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {
        s_lut[t] = ix.lut[t];
        s_lut2[t] = M ? ix.lut2[t] : 0xFFu;
    }
    const int nlds = ix.sigma < static_cast<uint32_t>(kLdsSyms) ? static_cast<int>(ix.sigma) : kLdsSyms;
    for (int t = threadIdx.x; t < nlds; t += blockDim.x) s_tab[t] = ix.syms[t];
    if (ix.kmer_steps >= 2)
        for (int t = threadIdx.x; t < static_cast<int>(M * M); t += blockDim.x) s_tab[kOff2 + t] = ix.pairs[t];
    if (ix.kmer_steps >= 3)
        for (int t = threadIdx.x; t < static_cast<int>(M * M * M); t += blockDim.x) s_tab[kOff3 + t] = ix.triples[t];
    if (ix.kmer_steps >= 4)
        for (int t = threadIdx.x; t < static_cast<int>(M * M * M * M); t += blockDim.x) s_tab[kOff4 + t] = ix.quads[t];
    if (with5 && ix.kmer_steps >= 5)
        for (int t = threadIdx.x; t < static_cast<int>(M * M * M * M * M); t += blockDim.x) s_tab[kOff5 + t] = ix.quints[t];
    __syncthreads();
}

// ftab entry: the state {lo, hi, toehold} after the word.  At 4-byte positions it is 16 bytes (one request)
// {lo, hi, toehold, 0}: a toehold of 2^64 - 1 (the word's last row is text position 0) is stored as
// 0xFFFFFFFF, and a word whose toehold fits neither is stored as {2, 0}: "search it step by step".  At
// 8-byte positions it is 4 x u64.  Returns false for the step-by-step marker.
template <typename P>
__device__ __forceinline__ bool ftab_lookup(const DevIndex &ix, uint64_t idx, uint64_t &lo, uint64_t &hi, uint64_t &k) {
    if constexpr (sizeof(P) == 4) {
        const uint4 e = static_cast<const uint4 *>(ix.ftab)[idx];
        if (e.x > e.y && e.x != 1u) return false;
        lo = e.x; hi = e.y;
        k = e.z == 0xFFFFFFFFu ? ~uint64_t(0) : static_cast<uint64_t>(e.z);
    } else {
        const ulonglong4 e = static_cast<const ulonglong4 *>(ix.ftab)[idx];
        lo = e.x; hi = e.y; k = e.z;
    }
    return true;
}

// ---- host side: launch geometry, in-place scan ------------------------------------------------------
inline int grid_for(const LaunchCfg &cfg, uint64_t N) {
    const int bt = cfg.block_threads;
    uint64_t blocks = (N + bt - 1) / bt;
    const uint64_t cap = cfg.max_blocks > 0 ? static_cast<uint64_t>(cfg.max_blocks) : 256ull * 32;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return static_cast<int>(blocks);
}


// Launch geometry of the kernels that stage the k-mer records in dynamic LDS: 17 KB and the configured
// workgroup size up to the 4-mer level; 66 KB with the 5-mer level, then 1024-thread workgroups so that
// two of them still give 8 waves per SIMD.  The first launch of a kernel with more than 64 KB of dynamic
// LDS has to raise its limit.
// The seeding kernels (k_greedy_seed, k_marker_seeds) need more registers per lane; with the 66 KB table and
// 1024-thread workgroups they drop to 4 waves per SIMD and lose more than the fifth symbol gains
// (k_marker_seeds 47-50 -> 52 ms per 10M reads): they stage the levels up to 4.
constexpr uint32_t kSeedKmerLevel = 4;

struct KmerLaunch {
    dim3 grid, block;
    size_t lds;
};
template <typename Kernel>
KmerLaunch kmer_launch(const DevIndex &ix, const LaunchCfg &cfg, uint64_t N, Kernel kernel, int grid_cap = 0, uint32_t max_k = 5) {
    const bool five = ix.kmer_steps >= 5 && max_k >= 5;
    LaunchCfg c = cfg;
    if (five) { c.block_threads = 1024; c.max_blocks = cfg.max_blocks > 0 ? std::max(1, cfg.max_blocks / 4) : 256 * 8; }
    KmerLaunch L;
    int g = grid_for(c, N);
    if (grid_cap > 0) g = std::min(g, five ? std::max(1, grid_cap / 4) : grid_cap);
    L.grid = dim3(g);
    L.block = dim3(c.block_threads);
    L.lds = static_cast<size_t>(five ? kTab5 : kTabMax) * sizeof(DevSym);
    if (five) {  // once per kernel (and device): later launches only look the pointer up
        static std::mutex mu;
        static std::set<std::pair<int, const void *>> raised;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const auto key = std::make_pair(dev, reinterpret_cast<const void *>(kernel));
        std::lock_guard<std::mutex> g(mu);
        if (raised.insert(key).second)
            (void)hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(L.lds));
    }
    return L;
}

inline int scan_in_place(uint64_t *vals, uint64_t N, void *tmp, size_t tmp_bytes, hipStream_t st) {
    if (N == 0) return 0;
    size_t need = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, need, vals, vals, static_cast<int64_t>(N));
    if (need > tmp_bytes) return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipcub::DeviceScan::InclusiveSum(tmp, need, vals, vals, static_cast<int64_t>(N), st));
}

}  // namespace
}  // namespace rbg
