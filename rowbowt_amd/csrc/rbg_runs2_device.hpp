// rbg_runs2_device.hpp -- the search of the run-indexed layout (rbg_dev.h DevRunTab2; "format 2" in profiles/ and
// DESIGN_HISTORY.md): one lane answers its own ranks and phi steps.  rle_string::rank (rle_string.hpp:131-161) and
// ToeholdSA::phi (toehold_sa.hpp:56-72) stay predecessor searches over run boundaries / sampled positions in O(r) space, but
// the directory has already cut the search down to the handful of entries of one bucket (and the one before them), so the
// lane that owns the query fetches exactly those -- independent 16-byte requests, all in flight together -- and scans
// them in registers.  Rounds 2-3 spread the same entries over 16, 8 and finally 4 lanes; each halving of the group was
// faster because the work per step is cross-lane choreography, not memory: with one lane per query there is none left
// (DESIGN.md 2c).  At 8-byte positions the entries carry the low 32 bits of {start, cum} only (rbg_dev.h: the bucket number
// is the high part, as in Elias-Fano): half the bytes per entry, 32-bit compares.
#pragma once

#include "rbg_runs_device.hpp"

namespace rbg {
namespace {

constexpr int kRunUniAt = 12;          // s_tab_first[12..15]: the uniform depth's constants (kMaxRunDepth + 1 = 9 words of first records before them)
constexpr uint32_t kLaneMaxZ = 15;     // candidates one scan covers (16 entries = two chunks of four 16-byte requests); more: narrowed first
constexpr uint32_t kLanePhiMaxZ = 8;   // sampled positions one phi scan covers (two chunks of four entries)

template <typename P>
struct RunSearch2 {
    const uint64_t *hot;          // LDS: the hot words (rbg_dev.h: dir_off | dir_shift << 56) of the tables of depths <= kLdsRunDepth
    const uint64_t *ghot;         // device memory: all run_ntabs hot words (the deeper depths' are read from here: rbg_dev.h kMaxRunDepth)
    const DevRunTab2 *gtab;       // device memory: the cold records (`first`: scans of the run list, re-samples)
    const uint32_t *tab_first;    // LDS [kMaxRunDepth + 1]: first record of each depth
    const void *const *ent;       // LDS [kMaxRunDepth]: entry arrays per depth
    const void *const *dir;       // LDS [kMaxRunDepth]: directory arrays per depth
    const void *const *rec;       // LDS [kMaxRunDepth]: bucket records per depth (nullptr: directories)
    const void *rec_any;          // the records of some depth that has them (what a lane without a record of its own fetches for its quad), or nullptr
    uint32_t fill;                // DevIndex::run_fill_shift
};

#define RBG_RUN_SEARCH2_SHARED                                            \
    __shared__ __align__(16) uint32_t s_tab_first[kMaxRunDepth + 1 + 3 + 4];  \
    __shared__ const void *s_ent2[8];                                     \
    __shared__ const void *s_dir2[8];                                     \
    __shared__ const void *s_rec2[8];                                     \
    extern __shared__ __align__(16) unsigned char s_dyn[]

inline size_t run_search2_lds(const DevIndex &ix) { return static_cast<size_t>(ix.run_tab_first[kLdsRunDepth]) * sizeof(uint64_t) + 16; }

// fills the arrays of RBG_RUN_SEARCH2_SHARED and returns the view of them; ends with __syncthreads()
template <typename P>
__device__ __forceinline__ RunSearch2<P> stage_run_search2(const DevIndex &ix, uint32_t *s_tab_first, const void **s_ent2, const void **s_dir2,
                                                           const void **s_rec2, unsigned char *s_dyn) {
    uint64_t *s_hot = reinterpret_cast<uint64_t *>(s_dyn);
    if (threadIdx.x < 8) {
        s_ent2[threadIdx.x] = threadIdx.x < static_cast<uint32_t>(kMaxRunDepth) ? ix.run_ent2[threadIdx.x] : nullptr;
        s_dir2[threadIdx.x] = threadIdx.x < static_cast<uint32_t>(kMaxRunDepth) ? ix.run_dir2[threadIdx.x] : nullptr;
        s_rec2[threadIdx.x] = threadIdx.x < static_cast<uint32_t>(kMaxRunDepth) ? static_cast<const void *>(ix.run_rec2[threadIdx.x]) : nullptr;
    }
    for (uint32_t t = threadIdx.x; t <= static_cast<uint32_t>(kMaxRunDepth); t += blockDim.x) s_tab_first[t] = ix.run_tab_first[t];
    if (threadIdx.x == 0) {   // the uniform depth's constants (DevIndex::run_uni_*), one 8-byte LDS read per step: {first record | depth << 24, stride | shift << 27}
        const bool on = ix.run_uni_depth < static_cast<uint32_t>(kMaxRunDepth);
        s_tab_first[kRunUniAt] = (on ? ix.run_tab_first[ix.run_uni_depth] : 0u) | (on ? ix.run_uni_depth : 0xFFu) << 24;
        s_tab_first[kRunUniAt + 1] = ix.run_uni_stride | ix.run_uni_shift << 27;
    }
    for (uint32_t t = threadIdx.x; t < ix.run_tab_first[kLdsRunDepth]; t += blockDim.x) s_hot[t] = ix.run_hot[t];
    __syncthreads();
    RunSearch2<P> S;
    S.hot = s_hot; S.ghot = ix.run_hot;
 S.gtab = ix.run_tabs2; S.tab_first = s_tab_first; S.ent = s_ent2; S.dir = s_dir2; S.rec = s_rec2;
    S.fill = ix.run_fill_shift;
    S.rec_any = nullptr;
    for (int t = kMaxRunDepth - 1; t >= 0; --t)
        if (ix.run_rec2[t]) S.rec_any = ix.run_rec2[t];
    return S;
}

// what a step reads of its table (rbg_dev.h kRunHotShiftBit): from LDS up to kLdsRunDepth, else one 8-byte load (L2 resident)
struct RunHot {
    uint64_t dir_off;     // the table's first directory entry / bucket record in its depth's array
    uint32_t dir_shift;
};
template <typename P>
__device__ __forceinline__ RunHot load_run_tab(const RunSearch2<P> &S, const uint32_t d, const uint32_t rec) {
    // uniform geometry (DevIndex::run_uni_*): the table's ordinal in its depth x the depth's stride, one 32 x 32 -> 64 bit multiply, no memory request
    // uniform geometry (DevIndex::run_uni_*): the table's ordinal in its depth x the depth's stride, one 32 x 32 -> 64 bit multiply.  Branch-free: the lanes
    // of the uniform depth all name the depth's FIRST hot word (one request for the wave instead of one per lane) and replace what it returns.
    const uint2 u = *reinterpret_cast<const uint2 *>(S.tab_first + kRunUniAt);   // {first record | depth << 24, stride | shift << 27}
    const bool uni = d == (u.x >> 24);
    const uint32_t first = u.x & 0xFFFFFFu;
    const uint64_t w0 = d < static_cast<uint32_t>(kLdsRunDepth) ? S.hot[rec] : as_global<uint64_t>(S.ghot)[uni ? first : rec];
    const uint64_t w = uni ? (static_cast<uint64_t>(rec - first) * (u.y & 0x7FFFFFFu)) | (static_cast<uint64_t>(u.y >> 27) << kRunHotShiftBit) : w0;
    return RunHot{w & ((uint64_t(1) << kRunHotShiftBit) - 1), static_cast<uint32_t>(w >> kRunHotShiftBit)};
}
// index of the table's first entry in its depth's arrays (cold: scans of the run list, re-samples)
template <typename P>
__device__ __forceinline__ uint64_t run_first(const RunSearch2<P> &S, const uint32_t rec) { return as_global<uint64_t>(static_cast<const void *>(S.gtab + rec))[1]; }

typedef unsigned int u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned int u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));

// directory entries b and b + 1 of a table: a = # entries starting below bucket b, e = # below bucket b + 1, h = the
// high part of the rank in bucket b (rbg_dev.h RunDir64; 0 at 4-byte positions)
template <typename P>
__device__ __forceinline__ void load_dir2(const void *__restrict__ dir, const uint64_t at, uint32_t &a, uint32_t &h, uint32_t &e) {
    if constexpr (sizeof(P) == 8) {
        const u32x4a8 w = *as_global<u32x4a8>(static_cast<const void *>(static_cast<const char *>(dir) + at * 8u));
        a = w.x; h = w.y; e = w.z;
    } else {
        const u32x2a4 w = *reinterpret_cast<const RBG_GLOBAL u32x2a4 *>(as_global<uint32_t>(dir) + at);   // (4-byte aligned pair)
        a = w.x; e = w.y; h = 0;
    }
}

// one query against a window of entries: the running state of the scan
struct LaneQ {
    uint32_t qa;       // the position as a distance from the anchor
    uint32_t zlim;     // window entries [0, zlim) are its candidates
    uint32_t c = 0;    // candidates below the position so far
    uint32_t ks = 0, kc = 0, kn = 0;   // the last entry below it {start, cum} and the cum of the entry after that one
    uint32_t ps = 0, pc = 0;
    uint32_t fc = 0;   // cum of the window's first entry: the rank of a position with no entry below it (the table's first entry: cum = F)
    bool pb = false;
    __device__ __forceinline__ void feed(const uint32_t gi, const uint32_t sa, const uint32_t cu) {
        if (gi == 0u) fc = cu;
        const bool nb = gi < zlim && sa < qa;
        const bool sel = pb && !nb;      // entry gi - 1 is the last one below the position
        ks = sel ? ps : ks;
        kc = sel ? pc : kc;
        kn = sel ? cu : kn;
        c += nb ? 1u : 0u;
        pb = nb; ps = sa; pc = cu;
    }
};

// 8-ary narrowing of a crowded bucket by the lane itself: candidates [p, p + z) of the table (entry indices relative to
// the table's first entry), z > kLaneMaxZ; seven pivots a stride apart are fetched together, the range shrinks to the
// stride between the last pivot below the position and the next.  Entry p stays "below or unknown".
__device__ __forceinline__ void lane_narrow(const char *__restrict__ tent, const uint32_t a_lo, const uint32_t qa, uint32_t &p, uint32_t &z, const uint32_t max_z,
                                            const uint32_t ent_bytes, uint32_t &rounds) {
    while (z > max_z) {
        const uint32_t stride = (z + 7u) >> 3;
        uint32_t key[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t at = static_cast<uint32_t>(j + 1) * stride;
            key[j] = at < z ? *as_global<uint32_t>(static_cast<const void *>(tent + static_cast<uint64_t>(p + at) * ent_bytes)) : 0u;
        }
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t at = static_cast<uint32_t>(j + 1) * stride;
            m += (at < z && key[j] - a_lo < qa) ? 1u : 0u;
        }
        const uint32_t adv = m * stride;
        p += adv;
        z = (z - adv) < stride ? (z - adv) : stride;
        ++rounds;
    }
}

// scan of the window [p, p + zw] (zw + 1 <= 16 entries: the candidates and the entry after the last of them) for one or two
// queries.  All requests -- four, or eight when the window is longer than eight entries -- are issued before the first is
// waited for: one memory round trip per scan.
// LEAN (the seeding kernels, whose walks hold enough state that the eight wide loads would cost them a wave per SIMD): four entries
// at a time, a round trip per four -- crowded buckets are the rare case once the bucket records hold six entries.
template <bool LEAN = false>
__device__ __forceinline__ void lane_scan(const char *__restrict__ tent, const uint32_t p, const uint32_t zw, const uint32_t a_lo, LaneQ &A, LaneQ *B) {
    const RBG_GLOBAL char *base = as_global<char>(static_cast<const void *>(tent)) + static_cast<uint64_t>(p) * 8u;
    if constexpr (LEAN) {
#pragma unroll 1
        for (uint32_t g = 0; g <= zw; g += 4u) {
            const uint32_t g1 = g + 2u < zw ? g + 2u : zw;
            const u32x4a8 w0 = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(g < zw ? g : zw) * 8u);
            const u32x4a8 w1 = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(g1) * 8u);
            // (a pair clamped to the window's last one is fed under its own indices: past every candidate, never below, never selected)
            A.feed(g, w0.x - a_lo, w0.y);
            A.feed(g + 1u, w0.z - a_lo, w0.w);
            A.feed(g + 2u, w1.x - a_lo, w1.y);
            A.feed(g + 3u, w1.z - a_lo, w1.w);
            if (B) {
                B->feed(g, w0.x - a_lo, w0.y);
                B->feed(g + 1u, w0.z - a_lo, w0.w);
                B->feed(g + 2u, w1.x - a_lo, w1.y);
                B->feed(g + 3u, w1.z - a_lo, w1.w);
            }
        }
        return;
    }
    u32x4a8 w[4], v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (unconditional, clamped to the window's last pair: a pair beyond it is fed as entries past every candidate --
        //  never below, never selected)
        const uint32_t gj = 2u * j < zw ? 2u * j : zw;
        w[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(gj) * 8u);
    }
    const bool more = zw >= 8u;
    if (more) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t gj = 8u + 2u * j < zw ? 8u + 2u * j : zw;
            v[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(gj) * 8u);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A.feed(2u * j, w[j].x - a_lo, w[j].y);
        A.feed(2u * j + 1u, w[j].z - a_lo, w[j].w);
        if (B) {
            B->feed(2u * j, w[j].x - a_lo, w[j].y);
            B->feed(2u * j + 1u, w[j].z - a_lo, w[j].w);
        }
    }
    if (more) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A.feed(8u + 2u * j, v[j].x - a_lo, v[j].y);
            A.feed(8u + 2u * j + 1u, v[j].z - a_lo, v[j].w);
            if (B) {
                B->feed(8u + 2u * j, v[j].x - a_lo, v[j].y);
                B->feed(8u + 2u * j + 1u, v[j].z - a_lo, v[j].w);
            }
        }
    }
}

// One bucket record in registers (rbg_dev.h RunRec2) and the ranks of one or two positions of its bucket: c = # entries of the table below
// the position counted from the record's first (0: none below at all), lo32 = the rank's low word, inside = the position's left
// neighbour lies in that run, p = the first entry's index relative to the table's first.
// COMPACT records answer by themselves, branch-free: the runs held are disjoint and ascending, so the rank is the first entry's cum plus
// the sum over ALL places of min(max(position - start, 0), length) -- a saturating subtract, a min and an add per place; the places'
// fields are unpacked once for both positions of a step (rank2).
// An OVERFLOWING bucket carries twelve pivots instead of entries: the lane narrows its candidates thirteen-fold from the registers it
// already has, and what is left (eight entries or fewer up to 91 candidates; beyond that lane_narrow's rounds first) is one scan of the
// run list -- pending(): the scan the caller still owes, so that the scans of both positions of a step leave together.
constexpr uint32_t kRecScanZ = 7u;   // candidates the FIRST pass of the scan after an overflowing record takes (+ the entry after them: four 16-byte loads)
constexpr uint32_t kRecScanZ2 = 15u;  // ... and with a second pass of four loads, taken only when the eighth entry still lies below the position
struct LaneRank {
    uint32_t c = 0, lo32 = 0, p = 0, z = 0;   // (p, z): candidates [p, p + z) of the pending scan
    bool inside = false, pending = false;
    uint32_t qa = 0, a_lo = 0;
};
__device__ __forceinline__ uint32_t sub_sat(const uint32_t a, const uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }
struct LaneRec {
    uint32_t w[16];
    __device__ __forceinline__ void load(const void *__restrict__ recs, const uint64_t at) {
        const RBG_GLOBAL u32x4 *p = as_global<u32x4>(static_cast<const void *>(static_cast<const char *>(recs) + at * 64u));
        const u32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w; w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
    }
    __device__ __forceinline__ bool compact() const { return (w[2] & kRec2Compact) != 0; }
    // the overflow form's part of a rank: thirteen-fold narrowing by the pivots; the scan stays pending
    __device__ __forceinline__ void pivots(const uint32_t a_lo, const uint32_t qa, LaneRank &out) const {
        out.qa = qa; out.a_lo = a_lo;
        const uint32_t z = w[3], stride = (z + 12u) / 13u;
        uint32_t m = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRec2Pivots; ++j) m += ((j + 1u) * stride < z && w[4 + j] - a_lo < qa) ? 1u : 0u;
        const uint32_t adv = m * stride;
        out.p = w[0] + adv;
        out.z = (z - adv) < stride ? (z - adv) : stride;
        out.pending = true;
    }
    // compact form, ONE position: qa = the position as a distance from the anchor, o = its offset into the bucket, sh = the table's bucket shift
    template <bool NEED_C>
    __device__ __forceinline__ void rank1(const uint32_t sh, const uint32_t a_lo, const uint32_t qa, const uint32_t o, LaneRank &out) const {
        out.qa = qa; out.a_lo = a_lo; out.p = w[0]; out.pending = false;
        const uint32_t mask = ~(~0u << (sh & 31u));
        const uint32_t s0 = w[4] - a_lo, d0 = qa - s0, om1 = o - 1u;
        const bool b0 = (w[2] & 15u) != 0u && s0 < qa;
        uint32_t add = b0 ? (d0 < w[5] ? d0 : w[5]) : 0u, c = b0 ? 1u : 0u;
        bool ins = b0 && d0 <= w[5];
#pragma unroll
        for (uint32_t k = 0; k < kRec2CompactIn; ++k) {
            const uint32_t v = w[6 + k], off = v & mask, len = v >> (sh & 31u);
            const uint32_t d = sub_sat(o, off);
            add += d < len ? d : len;
            if (NEED_C) { c += off < o ? 1u : 0u; ins = ins || (om1 - off) < len; }
        }
        out.c = NEED_C ? c : 1u; out.lo32 = w[3] + add; out.inside = ins;   // (without NEED_C only "a rank of zero" could hide behind c == 0, and the sum says that too)
    }
    // compact form, BOTH positions of a step (q0 <= q1 in the same bucket): the places are unpacked once
    __device__ __forceinline__ void rank2(const uint32_t sh, const uint32_t a_lo, const uint32_t qa0, const uint32_t o0, const uint32_t qa1, const uint32_t o1,
                                          LaneRank &A, LaneRank &B) const {
        A.qa = qa0; A.a_lo = a_lo; A.p = w[0]; A.pending = false;
        B.qa = qa1; B.a_lo = a_lo; B.p = w[0]; B.pending = false;
        const uint32_t mask = ~(~0u << (sh & 31u));
        const uint32_t s0 = w[4] - a_lo, l0 = w[5], da = qa0 - s0, db = qa1 - s0, om1 = o1 - 1u;
        const bool have = (w[2] & 15u) != 0u, ba = have && s0 < qa0, bb = have && s0 < qa1;
        uint32_t add_a = ba ? (da < l0 ? da : l0) : 0u, add_b = bb ? (db < l0 ? db : l0) : 0u, c = bb ? 1u : 0u;
        bool ins = bb && db <= l0;
#pragma unroll
        for (uint32_t k = 0; k < kRec2CompactIn; ++k) {
            const uint32_t v = w[6 + k], off = v & mask, len = v >> (sh & 31u);
            const uint32_t ea = sub_sat(o0, off), eb = sub_sat(o1, off);
            add_a += ea < len ? ea : len;
            add_b += eb < len ? eb : len;
            c += off < o1 ? 1u : 0u;
            ins = ins || (om1 - off) < len;
        }
        A.c = 1u;   // (c == 0 means "no run below: rank 0" to the caller; the sum is 0 then anyway -- the first entry held is the table's first, cum 0)
        A.lo32 = w[3] + add_a; A.inside = false;
        B.c = c; B.lo32 = w[3] + add_b; B.inside = ins;
    }
};

// the pending scans of one or two ranks (LaneRank::pending): every request of both is issued before the first is waited for.
// The run list is the table's slice of its depth's entry array: ent_d + first * 8, `first` read from the table's cold record here, by
// the lanes that have a scan to make (a dependent load, but only overflowing buckets pay it).
// Up to fifteen candidates are scanned without narrowing: eight entries at once, and the next eight only for the lane whose eighth entry still
// lies below its position (a second round trip for it alone -- the crowded buckets of a 520-haplotype pangenome hold about a hundred entries,
// thirteen-fold pivots leave eight to ten: with a seven-candidate limit nearly every such rank paid a round of seven pivot gathers first,
// 0.93 rounds per read at r = 1.07e9; profiles/r05_pangenome_stream_r1e9.json).
// WIDE = false (the seeding kernels, LEAN: the second pass's registers would cost their logging instantiations a wave per SIMD): seven candidates, narrowed first.
template <typename P, bool WIDE = true>
__device__ __forceinline__ void lane_finish(const RunSearch2<P> &S, const uint32_t d, const uint32_t rec, LaneRank &A, LaneRank *B, uint32_t &rounds, uint32_t &ents) {
    const bool pb = B && B->pending;
    if (!(A.pending || pb)) return;
    const char *__restrict__ tent = static_cast<const char *>(S.ent[d]) + run_first<P>(S, rec) * 8u;
    constexpr uint32_t kMaxZ = WIDE ? kRecScanZ2 : kRecScanZ;
    if (A.pending && A.z > kMaxZ) lane_narrow(tent, A.a_lo, A.qa, A.p, A.z, kMaxZ, 8u, rounds);
    if (pb && B->z > kMaxZ) lane_narrow(tent, B->a_lo, B->qa, B->p, B->z, kMaxZ, 8u, rounds);
    u32x4a8 wa[4], wb[4];
    if (A.pending) {
        const RBG_GLOBAL char *base = as_global<char>(static_cast<const void *>(tent)) + static_cast<uint64_t>(A.p) * 8u;
#pragma unroll
        for (int j = 0; j < 4; ++j) wa[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(2u * j < A.z ? 2u * j : A.z) * 8u);
    }
    if (pb) {
        const RBG_GLOBAL char *base = as_global<char>(static_cast<const void *>(tent)) + static_cast<uint64_t>(B->p) * 8u;
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(2u * j < B->z ? 2u * j : B->z) * 8u);
    }
    auto take = [&](LaneRank &R, const u32x4a8 (&w)[4]) {
        LaneQ Q;
        Q.qa = R.qa; Q.zlim = R.z;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (a pair clamped to the window's last one is fed under its own indices: past every candidate, never below, never selected)
            Q.feed(2u * j, w[j].x - R.a_lo, w[j].y);
            Q.feed(2u * j + 1u, w[j].z - R.a_lo, w[j].w);
        }
        if (WIDE && R.z > kRecScanZ && Q.pb) {   // the eighth entry is still below the position: entries 8 .. 15 (the scan's state carries over)
            const RBG_GLOBAL char *base = as_global<char>(static_cast<const void *>(tent)) + static_cast<uint64_t>(R.p) * 8u;
            u32x4a8 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(8u + 2u * j < R.z ? 8u + 2u * j : R.z) * 8u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Q.feed(8u + 2u * j, v[j].x - R.a_lo, v[j].y);
                Q.feed(8u + 2u * j + 1u, v[j].z - R.a_lo, v[j].w);
            }
            ents += 8u;
        }
        R.c = Q.c;
        const uint32_t dd = Q.qa - Q.ks, len = Q.kn - Q.kc;
        R.lo32 = Q.c ? Q.kc + (dd < len ? dd : len) : Q.fc;   // (none below: the window starts at the table's first entry, whose cum is the table's F)
        R.inside = Q.c != 0u && dd <= len;
        ents += (R.z < 8u ? R.z + 1u : 8u);
    };
    if (A.pending) take(A, wa);
    if (pb) take(*B, wb);
}

// What the instrumented instantiations count on this format (the eight sums of SearchStat): [kStSteps] search steps,
// [kStSlots] bucket records fetched (64 bytes) or directory gathers (two neighbouring entries: 8 or 16 bytes), [kStDense] run-list
// entries the scans needed (8 bytes each), [kStSearch] narrowing rounds (seven 4-byte pivots each), the rest as rbg_runs_device.hpp
// lists them.

// Both ranks of one LF step of ONE lane: rle_string::rank (rle_string.hpp:131-161) in the k-mer table `rec` of depth index d (R: its
// hot word) at positions q0 = lo and q1 = hi + 1 (q0 <= q1).  The cums of this layout carry the table's F (rbg_dev.h): out.c_before /
// out.c_upto are the ROWS F + rank (out.F stays 0), so that LF is lo' = c_before, hi' = c_upto - 1 (rowbowt.hpp:86-87).
// out.samp_e = the entry whose sample a toehold re-sample needs, relative to the table's first (run_step_sample2).
template <typename P, bool STATS = false, bool LEAN = false>
__device__ __forceinline__ void lane_lf2_tab(const RunSearch2<P> &S, const uint32_t d, const uint32_t rec, const RunHot &R, const uint64_t q0, const uint64_t q1, RunStep &out,
                                             unsigned long long *st = nullptr) {
    constexpr bool W = sizeof(P) == 8;
    out.F = 0;
    const uint32_t sh = R.dir_shift;
    const uint64_t b0 = pos_bucket<P>(q0, sh), b1 = pos_bucket<P>(q1, sh);   // (per-lane shifts: rbg_device.hpp pos_bucket)
    if (const void *__restrict__ recs = S.rec[d]) {   // ---- bucket records: one aligned 64-byte record per position (rbg_dev.h RunRec2) ----
        const uint32_t al0 = W ? pos_bucket_base32(q0, sh) - (1u << S.fill) : 0u;
        const uint32_t al1 = W ? pos_bucket_base32(q1, sh) - (1u << S.fill) : 0u;
        const uint32_t qa0 = static_cast<uint32_t>(q0) - al0, qa1 = static_cast<uint32_t>(q1) - al1;
        const uint32_t o0 = pos_bucket_offset(q0, sh), o1 = pos_bucket_offset(q1, sh);
        LaneRank A, B;
        uint32_t h0, h1;
        uint32_t rounds = 0, ents = 0;
        LaneRec r0;
        r0.load(recs, R.dir_off + b0);
        h0 = h1 = r0.w[1];
        if (b1 == b0) {
            if (r0.compact()) r0.rank2(sh, al0, qa0, o0, qa1, o1, A, B);
            else { r0.pivots(al0, qa0, A); r0.pivots(al1, qa1, B); }
            lane_finish<P, !LEAN>(S, d, rec, A, &B, rounds, ents);
        } else if constexpr (LEAN) {
            // (the seeding kernels: one record in registers at a time -- the second is fetched after the first position is answered;
            //  holding both costs sixteen registers on every step and them a workgroup per CU)
            if (r0.compact()) r0.template rank1<false>(sh, al0, qa0, o0, A); else r0.pivots(al0, qa0, A);
            lane_finish<P, !LEAN>(S, d, rec, A, nullptr, rounds, ents);
            r0.load(recs, R.dir_off + b1);
            h1 = r0.w[1];
            if (r0.compact()) r0.template rank1<true>(sh, al1, qa1, o1, B); else r0.pivots(al1, qa1, B);
            lane_finish<P, !LEAN>(S, d, rec, B, nullptr, rounds, ents);
        } else {
            // both records leave together, then both scans of the run list (overflowing buckets): the dependent round trips of a step are
            // record -> scan [-> sample] whatever its two positions meet -- a wave waits for the longest chain among its 64 lanes
            LaneRec r1;
            r1.load(recs, R.dir_off + b1);
            h1 = r1.w[1];
            if (r0.compact()) r0.template rank1<false>(sh, al0, qa0, o0, A); else r0.pivots(al0, qa0, A);
            if (r1.compact()) r1.template rank1<true>(sh, al1, qa1, o1, B); else r1.pivots(al1, qa1, B);
            lane_finish<P, !LEAN>(S, d, rec, A, &B, rounds, ents);
        }
        if (STATS) { st[kStSlots] += b1 != b0 ? 2 : 1; st[kStSearch] += rounds; st[kStDense] += ents; st[kStSteps] += 1; }
        const uint64_t y0 = static_cast<uint64_t>(h0) << 31, y1 = static_cast<uint64_t>(h1) << 31;
        out.c_before = W ? y0 + static_cast<uint32_t>(A.lo32 - static_cast<uint32_t>(y0)) : A.lo32;
        out.c_upto = W ? y1 + static_cast<uint32_t>(B.lo32 - static_cast<uint32_t>(y1)) : B.lo32;
        out.inside = B.c != 0 && B.inside;
        out.samp_e = static_cast<uint64_t>(B.p) + B.c - 1u;
        return;
    }
    const char *__restrict__ tent = static_cast<const char *>(S.ent[d]) + run_first<P>(S, rec) * 8u;
    const void *__restrict__ dir = S.dir[d];
    uint32_t a0, h0, e0, a1, h1, e1;
    load_dir2<P>(dir, R.dir_off + b0, a0, h0, e0);
    a1 = a0; h1 = h0; e1 = e0;
    if (b1 != b0) load_dir2<P>(dir, R.dir_off + b1, a1, h1, e1);
    if (STATS) st[kStSlots] += b1 != b0 ? 2 : 1;
    uint32_t p0 = a0 ? a0 - 1u : 0u, z0 = e0 - p0;     // candidates: the entries of the bucket and the one before them
    uint32_t p1 = a1 ? a1 - 1u : 0u, z1 = e1 - p1;
    // anchors (8-byte positions): every candidate of bucket b lies above (b << sh) - 2^fill (fillers, rbg_dev.h)
    const uint32_t al0 = W ? pos_bucket_base32(q0, sh) - (1u << S.fill) : 0u;
    const uint32_t al1 = W ? pos_bucket_base32(q1, sh) - (1u << S.fill) : 0u;
    const uint32_t qa0 = static_cast<uint32_t>(q0) - al0, qa1 = static_cast<uint32_t>(q1) - al1;
    uint32_t rounds = 0;
    if (z0 > kLaneMaxZ) lane_narrow(tent, al0, qa0, p0, z0, kLaneMaxZ, 8u, rounds);
    if (z1 > kLaneMaxZ) lane_narrow(tent, al1, qa1, p1, z1, kLaneMaxZ, 8u, rounds);
    if (STATS) st[kStSearch] += rounds;
    // one window for both positions when the second's candidates end within reach of the first's start (the usual case: the
    // range has narrowed to one locus); at 8-byte positions only for the same or the next bucket, so that the first
    // bucket's anchor orders the second position too
    const bool shared = p1 + z1 <= p0 + kLaneMaxZ && (!W || b1 - b0 <= 1u);
    LaneQ A, B;
    A.qa = qa0; A.zlim = z0;
    if (shared) {
        B.qa = static_cast<uint32_t>(q1) - al0; B.zlim = p1 + z1 - p0;
        const uint32_t zw = B.zlim > z0 ? B.zlim : z0;
        lane_scan<LEAN>(tent, p0, zw, al0, A, &B);
        if (STATS) st[kStDense] += zw + 1u;
        p1 = p0;
    } else {
        B.qa = qa1; B.zlim = z1;
        lane_scan<LEAN>(tent, p0, z0, al0, A, nullptr);
        lane_scan<LEAN>(tent, p1, z1, al1, B, nullptr);
        if (STATS) st[kStDense] += z0 + z1 + 2u;
    }
    // the row = cum + min(position - start, length of that run), or -- no entry below the position -- the cum of the table's first
    // entry (its F); the high part from the directory (rbg_dev.h RunDir64)
    bool inside = false;
    uint32_t lo32a = A.fc, lo32b = B.fc;
    if (A.c) {
        const uint32_t dd = A.qa - A.ks, len = A.kn - A.kc;
        lo32a = A.kc + (dd < len ? dd : len);
    }
    if (B.c) {
        const uint32_t dd = B.qa - B.ks, len = B.kn - B.kc;
        lo32b = B.kc + (dd < len ? dd : len);
        inside = dd <= len;
    }
    const uint64_t y0 = static_cast<uint64_t>(h0) << 31, y1 = static_cast<uint64_t>(h1) << 31;
    out.c_before = W ? y0 + static_cast<uint32_t>(lo32a - static_cast<uint32_t>(y0)) : lo32a;
    out.c_upto = W ? y1 + static_cast<uint32_t>(lo32b - static_cast<uint32_t>(y1)) : lo32b;
    out.inside = inside;
    out.samp_e = static_cast<uint64_t>(p1) + B.c - 1u;   // (read only when B.c > 0 and the row is not inside the run)
    if (STATS) st[kStSteps] += 1;
}

template <typename P, bool STATS = false, bool LEAN = false>
__device__ __forceinline__ void lane_lf2(const RunSearch2<P> &S, const uint32_t d, const uint32_t rec, const uint64_t q0, const uint64_t q1, RunStep &out,
                                         unsigned long long *st = nullptr) {
    const RunHot R = load_run_tab<P>(S, d, rec);
    lane_lf2_tab<P, STATS, LEAN>(S, d, rec, R, q0, q1, out, st);
}

// ---- the same step with the FIRST record of every lane fetched by its QUAD (k_find_range_runs) ----------------------------------------
// A lane that reads its own 64-byte record issues four 16-byte requests to one line in four instructions: four address translations and four
// L1 lookups where the data is one sector.  Here the four lanes of a quad read the record of each of them in turn -- lane p the p-th sixteen
// bytes, one instruction per record, the quad's four requests coalesced into one line -- and a 4 x 4 transpose through two quad permutes puts
// every record in its owner's registers: a quarter of the L1 accesses for the same sectors (profiles/r04_scale_probe.txt: what the search pays
// for beyond 10 GB of index is not sectors).  Called by ALL lanes of the wave; the rare second record (hi + 1 in another bucket) and the scans
// of overflowing buckets are fetched by the lane itself as in lane_lf2.
__device__ __forceinline__ uint32_t quad_xor1(const uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t quad_xor2(const uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
template <int C>
__device__ __forceinline__ uint32_t quad_bcast(const uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), C * 0x55, 0xF, 0xF, true)); }
// w[4 r + k] = word k of the chunk this lane read in round r (chunk p of the record of quad lane r)  ->  word k of chunk r of its OWN record
__device__ __forceinline__ void quad_transpose16(uint32_t (&w)[16], const uint32_t p) {
    const bool p0 = (p & 1u) != 0u, p1 = (p & 2u) != 0u;
#pragma unroll
    for (int b = 0; b < 4; b += 2)          // distance 1: registers (b, b + 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = w[4 * b + k], hi = w[4 * (b + 1) + k];
            const uint32_t t = quad_xor1(p0 ? lo : hi);
            w[4 * b + k] = p0 ? t : lo;
            w[4 * (b + 1) + k] = p0 ? hi : t;
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)             // distance 2: registers (b, b + 2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = w[4 * b + k], hi = w[4 * (b + 2) + k];
            const uint32_t t = quad_xor2(p1 ? lo : hi);
            w[4 * b + k] = p1 ? t : lo;
            w[4 * (b + 2) + k] = p1 ? hi : t;
        }
}
// GLDS: the same four rounds as LDS-DIRECT loads (global_load_lds_dwordx4: memory -> LDS, no VGPR in between, gfx950): in round r lane l
// sends chunk (l & 3) of the record of quad lane r to  tile + r * kTileRound + 16 l  (the destination of such a load is the wave's base + 16
// bytes per lane: exactly this image), so the record of owner lane L lies contiguous at  tile + (L & 3) * kTileRound + (L & ~3) * 16  and
// the owner reads it back with four ds_read_b128 -- no transpose on the VALU (64 quad permutes and as many selects per record), sixteen
// registers fewer while the loads fly.  kTileRound = 1 KiB + 16: the pad puts the four lanes of a quad on different banks.
constexpr uint32_t kTileRound = 1024u + 16u;
constexpr uint32_t kTileBytes = 4u * kTileRound;     // per wave
typedef __attribute__((address_space(3))) unsigned char lds_byte;
template <int r>
__device__ __forceinline__ void glds_round(const uint32_t a_lo, const uint32_t a_hi, const uint32_t p, lds_byte *tile) {
    const uint64_t an = (static_cast<uint64_t>(quad_bcast<r>(a_hi)) << 32) | quad_bcast<r>(a_lo);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const RBG_GLOBAL void *>(an + 16u * p), reinterpret_cast<__attribute__((address_space(3))) void *>(tile + r * kTileRound), 16, 0, 0);
}
// LEAN: the second record (hi + 1 in another bucket) is fetched into the first one's registers once that is answered (the seeding kernels)
// R: the table's hot word (load_run_tab; the caller may have fetched it a step ahead)
template <typename P, bool STATS = false, bool LEAN = false, bool GLDS = false>
__device__ __forceinline__ void lane_lf2_quad(const RunSearch2<P> &S, const bool stepping, const uint32_t d, const uint32_t rec, const RunHot &R, const uint64_t q0,
                                              const uint64_t q1, RunStep &out, unsigned long long *st = nullptr, lds_byte *tile = nullptr) {
    constexpr bool W = sizeof(P) == 8;
    const uint32_t sh = R.dir_shift;
    const bool by_rec = stepping && S.rec[d] != nullptr;
    if (stepping && !by_rec) lane_lf2_tab<P, STATS, LEAN>(S, d, rec, R, q0, q1, out, st);   // (a depth with directories over its run lists: the lane by itself)
    if (__ballot(by_rec) == 0) return;                                            // (nobody's quad has a record to fetch in this step)
    const uint32_t p = threadIdx.x & 3u;
    const uint64_t b0 = by_rec ? pos_bucket<P>(q0, sh) : 0u, b1 = by_rec ? pos_bucket<P>(q1, sh) : 0u;
    const char *recs = static_cast<const char *>(by_rec ? S.rec[d] : S.rec_any);
    const uint64_t a0 = reinterpret_cast<uint64_t>(recs) + (by_rec ? (R.dir_off + b0) * 64u : 0u);   // (a lane without a record to fetch names the first record there is)
    const uint32_t a_lo = static_cast<uint32_t>(a0), a_hi = static_cast<uint32_t>(a0 >> 32);
    LaneRec r0, r1;
    uint32_t (&w0)[16] = r0.w;
    const bool two = by_rec && b1 != b0;
    if constexpr (GLDS) {
        glds_round<0>(a_lo, a_hi, p, tile);
        glds_round<1>(a_lo, a_hi, p, tile);
        glds_round<2>(a_lo, a_hi, p, tile);
        glds_round<3>(a_lo, a_hi, p, tile);
        if constexpr (!LEAN) { if (two) r1.load(recs, R.dir_off + b1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-direct loads are this wave's own: no barrier, the counter is the wave's)
        const lds_byte *own = tile + (threadIdx.x & 3u) * kTileRound + (threadIdx.x & 60u) * 16u;
        typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
        const u32x4 t0 = reinterpret_cast<lds_u32x4 *>(own)[0], t1 = reinterpret_cast<lds_u32x4 *>(own)[1], t2 = reinterpret_cast<lds_u32x4 *>(own)[2],
                    t3 = reinterpret_cast<lds_u32x4 *>(own)[3];
        w0[0] = t0.x; w0[1] = t0.y; w0[2] = t0.z; w0[3] = t0.w; w0[4] = t1.x; w0[5] = t1.y; w0[6] = t1.z; w0[7] = t1.w;
        w0[8] = t2.x; w0[9] = t2.y; w0[10] = t2.z; w0[11] = t2.w; w0[12] = t3.x; w0[13] = t3.y; w0[14] = t3.z; w0[15] = t3.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the record is in registers before the next step's loads may land on the tile)
    } else {
#define RBG_QUAD_ROUND(r)                                                                                                             \
    {                                                                                                                                  \
        const uint64_t an = (static_cast<uint64_t>(quad_bcast<r>(a_hi)) << 32) | quad_bcast<r>(a_lo);                                  \
        const u32x4 t = *as_global<u32x4>(reinterpret_cast<const void *>(an + 16u * p));                                                \
        w0[4 * r] = t.x; w0[4 * r + 1] = t.y; w0[4 * r + 2] = t.z; w0[4 * r + 3] = t.w;                                                  \
    }
        RBG_QUAD_ROUND(0) RBG_QUAD_ROUND(1) RBG_QUAD_ROUND(2) RBG_QUAD_ROUND(3)
#undef RBG_QUAD_ROUND
        if constexpr (!LEAN) { if (two) r1.load(recs, R.dir_off + b1); }
        quad_transpose16(w0, p);
    }
    if (!by_rec) return;
    out.F = 0;
    const uint32_t al0 = W ? pos_bucket_base32(q0, sh) - (1u << S.fill) : 0u;
    const uint32_t al1 = W ? pos_bucket_base32(q1, sh) - (1u << S.fill) : 0u;
    const uint32_t qa0 = static_cast<uint32_t>(q0) - al0, qa1 = static_cast<uint32_t>(q1) - al1;
    const uint32_t o0 = pos_bucket_offset(q0, sh), o1 = pos_bucket_offset(q1, sh);
    LaneRank A, B;
    uint32_t h0 = w0[1], h1 = w0[1], rounds = 0, ents = 0;
    if (!two) {
        if (r0.compact()) r0.rank2(sh, al0, qa0, o0, qa1, o1, A, B);
        else { r0.pivots(al0, qa0, A); r0.pivots(al1, qa1, B); }
        lane_finish<P, !LEAN>(S, d, rec, A, &B, rounds, ents);
    } else if constexpr (LEAN) {
        if (r0.compact()) r0.template rank1<false>(sh, al0, qa0, o0, A); else r0.pivots(al0, qa0, A);
        lane_finish<P, !LEAN>(S, d, rec, A, nullptr, rounds, ents);
        r0.load(recs, R.dir_off + b1);
        h1 = w0[1];
        if (r0.compact()) r0.template rank1<true>(sh, al1, qa1, o1, B); else r0.pivots(al1, qa1, B);
        lane_finish<P, !LEAN>(S, d, rec, B, nullptr, rounds, ents);
    } else {
        h1 = r1.w[1];
        if (r0.compact()) r0.template rank1<false>(sh, al0, qa0, o0, A); else r0.pivots(al0, qa0, A);
        if (r1.compact()) r1.template rank1<true>(sh, al1, qa1, o1, B); else r1.pivots(al1, qa1, B);
        lane_finish<P, !LEAN>(S, d, rec, A, &B, rounds, ents);
    }
    if (STATS) { st[kStSlots] += two ? 2 : 1; st[kStSearch] += rounds; st[kStDense] += ents; st[kStSteps] += 1; }
    const uint64_t y0 = static_cast<uint64_t>(h0) << 31, y1 = static_cast<uint64_t>(h1) << 31;
    out.c_before = W ? y0 + static_cast<uint32_t>(A.lo32 - static_cast<uint32_t>(y0)) : A.lo32;
    out.c_upto = W ? y1 + static_cast<uint32_t>(B.lo32 - static_cast<uint32_t>(y1)) : B.lo32;
    out.inside = B.c != 0 && B.inside;
    out.samp_e = static_cast<uint64_t>(B.p) + B.c - 1u;
}
// the same with the table's record looked up here (the seeding kernels)
template <typename P, bool STATS = false, bool LEAN = false>
__device__ __forceinline__ void lane_lf2_quad(const RunSearch2<P> &S, const bool stepping, const uint32_t d, const uint32_t rec, const uint64_t q0, const uint64_t q1,
                                              RunStep &out, unsigned long long *st = nullptr) {
    const RunHot R = load_run_tab<P>(S, stepping ? d : 0u, stepping ? rec : 0u);
    lane_lf2_quad<P, STATS, LEAN, false>(S, stepping, d, rec, R, q0, q1, out, st, nullptr);
}

// the sample of the step's predecessor run: entry `rel` of table `rec` (the table's first entry from its cold record, then one gather)
template <typename P>
__device__ __forceinline__ uint64_t run_step_sample2(const DevIndex &ix, const RunSearch2<P> &S, const uint32_t d, const uint32_t rec, const uint64_t rel) {
    return RunList<P>::samp(ix.run_samp[d], run_first<P>(S, rec) + rel);
}

// ---- phi by one lane ------------------------------------------------------------------------------------------------------
// ToeholdSA::phi(i) (toehold_sa.hpp:56-72) for a position q < n: found = a sampled position lies before q, val = its base +
// (q - it) -- not yet reduced mod n; else the caller takes the circular predecessor.  ents = entries the scan needed (STATS).
template <typename P>
__device__ __forceinline__ void lane_phi(const DevIndex &ix, const uint64_t q, bool &found, uint64_t &val, uint32_t &ents, uint32_t &rounds) {
    constexpr bool W = sizeof(P) == 8;
    constexpr uint32_t EB = W ? 12u : 8u;
    const uint32_t sh = ix.phi_dir_shift;
    const uint64_t b = q >> sh;
    const u32x2a4 dw = *reinterpret_cast<const RBG_GLOBAL u32x2a4 *>(as_global<uint32_t>(ix.phi_dir) + b);
    uint64_t g0;
    uint32_t inb = dw.y - dw.x;            // sampled positions inside the bucket (the counts' low words differ by it)
    if constexpr (W) {
        const uint64_t s0 = as_global<uint64_t>(ix.phi_super)[b >> ix.phi_super_shift];
        g0 = s0 + static_cast<uint32_t>(dw.x - static_cast<uint32_t>(s0));   // (fewer than 2^32 entries per super block)
    } else {
        g0 = dw.x;
    }
    const uint64_t pred = g0 ? g0 - 1u : 0u;
    uint32_t z = inb + (g0 ? 1u : 0u);
    const uint32_t a_lo = W ? static_cast<uint32_t>(b << sh) - (1u << ix.run_fill_shift) : 0u;
    const uint32_t qa = static_cast<uint32_t>(q) - a_lo;
    const char *__restrict__ tent = static_cast<const char *>(ix.phi_ent) + pred * EB;
    uint32_t p = 0;
    rounds = 0;
    if (z > kLanePhiMaxZ) lane_narrow(tent, a_lo, qa, p, z, kLanePhiMaxZ, EB, rounds);
    ents = z;
    const RBG_GLOBAL char *base = as_global<char>(static_cast<const void *>(tent)) + static_cast<uint64_t>(p) * EB;
    uint32_t c = 0, ks = 0, kb_lo = 0, kb_hi = 0;
    // (z <= 8 entries: four requests, eight when there are more than four -- all issued before the first is waited for;
    //  requests beyond the last entry re-read it and are fed as entries past every candidate)
    const uint32_t zl = z ? z - 1u : 0u;
    const bool more = z > 4u;
    if constexpr (W) {
        u32x3a4 w[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const RBG_GLOBAL u32x3a4 *>(base + static_cast<uint64_t>(static_cast<uint32_t>(j) < zl ? static_cast<uint32_t>(j) : zl) * 12u);
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const RBG_GLOBAL u32x3a4 *>(base + static_cast<uint64_t>(4u + j < zl ? 4u + j : zl) * 12u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t sa = w[j].x - a_lo;
            const bool nb = static_cast<uint32_t>(j) < z && sa < qa;
            ks = nb ? sa : ks; kb_lo = nb ? w[j].y : kb_lo; kb_hi = nb ? w[j].z : kb_hi;
            c += nb ? 1u : 0u;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t sa = v[j].x - a_lo;
                const bool nb = 4u + j < z && sa < qa;
                ks = nb ? sa : ks; kb_lo = nb ? v[j].y : kb_lo; kb_hi = nb ? v[j].z : kb_hi;
                c += nb ? 1u : 0u;
            }
        }
    } else {
        u32x4a8 w[2], v[2];   // two entries per request (one spare entry follows the sentinel)
#pragma unroll
        for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(2u * j < zl ? 2u * j : zl) * 8u);
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const RBG_GLOBAL u32x4a8 *>(base + static_cast<uint64_t>(4u + 2u * j < zl ? 4u + 2u * j : zl) * 8u);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool nb0 = 2u * j < z && w[j].x < qa;
            ks = nb0 ? w[j].x : ks; kb_lo = nb0 ? w[j].y : kb_lo;
            const bool nb1 = 2u * j + 1u < z && w[j].z < qa;
            ks = nb1 ? w[j].z : ks; kb_lo = nb1 ? w[j].w : kb_lo;
            c += (nb0 ? 1u : 0u) + (nb1 ? 1u : 0u);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool nb0 = 4u + 2u * j < z && v[j].x < qa;
                ks = nb0 ? v[j].x : ks; kb_lo = nb0 ? v[j].y : kb_lo;
                const bool nb1 = 4u + 2u * j + 1u < z && v[j].z < qa;
                ks = nb1 ? v[j].z : ks; kb_lo = nb1 ? v[j].w : kb_lo;
                c += (nb0 ? 1u : 0u) + (nb1 ? 1u : 0u);
            }
        }
    }
    found = c != 0;
    val = ((static_cast<uint64_t>(kb_hi) << 32) | kb_lo) + static_cast<uint32_t>(qa - ks);
}

}  // namespace
}  // namespace rbg
