// rbg_runs_device.hpp -- device-side building blocks of the run-indexed layout (RBG_LAYOUT_RUNS, rbg_dev.h DevTree /
// DevRunTab), shared by k_runs.hip (K1/K2/K3) and k_runs_seeds.hip (seeding, windowed markers, single LF steps):
// the wave-cooperative probes (16-lane rows, quads), the staging of a workgroup's search tables, and coop_lf2 -- both
// ranks of one LF step of every lane of the wave (rle_string::rank, rle_string.hpp:131-161, through the directories).
// Everything lives in an anonymous namespace: each unit gets its own inlined copy.
#pragma once

#include "rbg_device.hpp"

namespace rbg {
namespace {
constexpr int kFan = kTreeFan;                 // entries per block = lanes that probe one block together
constexpr int kRows = kWave / kFan;            // blocks probed by one wave-wide load instruction
static_assert(kFan == 16 && kRows == 4, "the probes below are written for 16-lane rows of a 64-lane wave");

// pair of P as the kernels load it (one request per lane)
template <typename P> struct PairOf;
template <> struct PairOf<uint32_t> { typedef unsigned int vec __attribute__((ext_vector_type(2))); };
template <> struct PairOf<uint64_t> { typedef unsigned long long vec __attribute__((ext_vector_type(2))); };

// value of lane `j` of this lane's 16-lane row
__device__ __forceinline__ uint32_t row_get(uint32_t v, uint32_t rowbase, int j) { return __shfl(v, static_cast<int>(rowbase) | j, kWave); }
__device__ __forceinline__ uint64_t row_get(uint64_t v, uint32_t rowbase, int j) { return __shfl(v, static_cast<int>(rowbase) | j, kWave); }
// the same from a lane chosen at run time (same in every lane of the row)
__device__ __forceinline__ uint32_t row_pick(uint32_t v, uint32_t rowbase, uint32_t j) { return __shfl(v, static_cast<int>(rowbase | j), kWave); }
__device__ __forceinline__ uint64_t row_pick(uint64_t v, uint32_t rowbase, uint32_t j) { return __shfl(v, static_cast<int>(rowbase | j), kWave); }

__device__ __forceinline__ void wave_lds_sync() {  // as in k_locate.hip: orders the wave's own LDS writes and cross-lane reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the next lane's value within the 16-lane row (a DPP move: no LDS traffic; lane 15 of a row gets 0)
__device__ __forceinline__ uint32_t row_next(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x101, 0xF, 0xF, false)); }
__device__ __forceinline__ uint64_t row_next(uint64_t v) {
    return (static_cast<uint64_t>(row_next(static_cast<uint32_t>(v >> 32))) << 32) | row_next(static_cast<uint32_t>(v));
}

// REQUEST SLOTS.  The kernels below are bound by the LDS pipe, not by memory, when every value an owner shares with its
// row travels by ds_bpermute (372 of them per wave and LF step in round 2's first kernels: SQ_ACTIVE_INST_LDS at the
// CU's limit, profiles/).  So an owner WRITES what its row needs to know -- where to probe, how many candidates, the two
// positions -- into its own slots of an LDS area once per pass (kReqSlots x 16 bytes per lane), and in round j the row's
// lanes read owner j's slot with one broadcast ds_read_b128; the rank itself is computed by the lane that holds the run
// (its neighbour's count arrives by DPP), so what travels back is one value per position.
template <typename P> struct ReqSlots { static constexpr int v = sizeof(P) == 8 ? 3 : 2; };

// does round j serve anyone?  (m = ballot of the lanes with a query; round j serves lane j of each row: the test is
// uniform over the wave, so a round without an owner costs a scalar branch instead of its cross-lane traffic)
__device__ __forceinline__ bool round_has_owner(uint64_t m, int j) { return (m & (0x0001000100010001ull << j)) != 0; }

// # lanes of this lane's row for which `pred` holds
__device__ __forceinline__ uint32_t row_count(bool pred, uint32_t rowbase) {
    return static_cast<uint32_t>(__popc(static_cast<uint32_t>(__ballot(pred) >> rowbase) & 0xFFFFu));
}

// # entries of s_top[off, off + n) that are < q (per lane; LDS)
template <typename P>
__device__ __forceinline__ uint32_t top_count(const P *s_top, uint32_t off, uint32_t n, uint64_t q) {
    uint32_t a = 0, z = n;
    while (a < z) {
        const uint32_t mid = (a + z) >> 1;
        if (static_cast<uint64_t>(s_top[off + mid]) < q) a = mid + 1; else z = mid;
    }
    return a;
}

// A query is a predecessor search CLAMPED to a slice [lo_t, hi_t) of the tree's entry array (one table of a k-mer
// depth, rbg_dev.h DevRunTab; the whole array for phi): entries before the slice count as below the query, entries
// from hi_t on as not below it.  For the 16 entries of the block that starts at entry index `gbase` of a level whose
// entries stand `1 << sh` apart in the leaf array, that is: the first `a` lanes are below whatever their key, lanes
// from `z` on are not; both fit five bits.
__device__ __forceinline__ uint32_t clamp_lanes(uint64_t bound, uint64_t gbase, int sh) {
    if (bound <= gbase) return 0;
    const uint64_t d = (bound - gbase + ((uint64_t(1) << sh) - 1)) >> sh;
    return d > static_cast<uint64_t>(kFan) ? static_cast<uint32_t>(kFan) : static_cast<uint32_t>(d);
}
// info word an owner publishes to its row for one level: bit 0 any query, bit 1 second block, bits 2-4 tree,
// bits 5-9 / 10-14 (a, z) of the first block, bits 15-19 / 20-24 (a, z) of the second
__device__ __forceinline__ uint32_t level_info(bool any, bool two, uint32_t tid, uint32_t first, uint32_t b1, uint32_t lo_t, uint32_t hi_t, int sh) {
    const uint64_t g0 = (static_cast<uint64_t>(first) * kFan) << sh, g1 = (static_cast<uint64_t>(b1) * kFan) << sh;
    return (any ? 1u : 0u) | (two ? 2u : 0u) | (tid << 2) | (clamp_lanes(lo_t, g0, sh) << 5) | (clamp_lanes(hi_t, g0, sh) << 10) |
           (clamp_lanes(lo_t, g1, sh) << 15) | (clamp_lanes(hi_t, g1, sh) << 20);
}

// One sampled level (keys only), for up to two queries per lane.  On entry t0 / t1 = # entries of the level ABOVE
// that are below q (>= 1 for a live query): the answer at this level lies in block t - 1.  On return t = # entries of
// THIS level that are below q.  Every lane of the wave must call.
// The wave works in kFan rounds: in round j each 16-lane row serves the queries of ITS lane j -- the row's lanes load
// the 16 keys of that owner's block with one coalesced request (four owners per wave-wide load instruction), the
// owner's query is broadcast along the row, and popcount(ballot) over the row is the answer.  All 16 rounds' loads are
// issued before the first compare, so a level costs one memory round trip per wave, not one per owner.
template <typename P>
__device__ __forceinline__ void coop_level(const DevTree *s_tree, const int l, const uint32_t tid, const uint32_t lo_t, const uint32_t hi_t,
                                           const bool live0, const bool live1, uint32_t &t0, uint32_t &t1, const P q0, const P q1) {
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & (kFan - 1), rowbase = lane & ~static_cast<uint32_t>(kFan - 1);
    const uint64_t m_live = __ballot(live0 || live1);
    if (!m_live) return;
    const uint32_t b0 = t0 - 1, b1 = t1 - 1;
    const bool two = live0 && live1 && b0 != b1;                     // the second query needs a block of its own
    const uint32_t first = live0 ? b0 : b1;
    const uint32_t info = level_info(live0 || live1, two, tid, first, b1, lo_t, hi_t, 4 * (l + 1));
    P va[kFan];
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t ob = row_get(first, rowbase, j);
        va[j] = static_cast<P>(~P(0));
        if (oi & 1u) {
            const DevTree &T = s_tree[(oi >> 2) & 7u];
            const uint64_t i = static_cast<uint64_t>(ob) * kFan + sub;
            if (i < T.lvl_n[l]) va[j] = as_global<P>(T.lvl[l])[i];
        }
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        // padding lanes hold the all-ones key, which no query exceeds (positions stay below it: flatten())
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t a = (oi >> 5) & 31u, z = (oi >> 10) & 31u;
        const bool in = sub < z;
        // (the broadcasts are cross-lane operations: every lane must execute them, so they stay outside the || / &&)
        const P oq0 = row_get(q0, rowbase, j), oq1 = row_get(q1, rowbase, j);
        const uint32_t c0 = row_count(sub < a || (in && va[j] < oq0), rowbase);
        const uint32_t c1 = row_count(sub < a || (in && va[j] < oq1), rowbase);
        if (static_cast<int>(sub) == j) {
            if (live0) t0 = b0 * kFan + c0;
            if (live1 && !two) t1 = b1 * kFan + c1;
        }
    }
    const uint64_t m_two = __ballot(two);
    if (!m_two) return;
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_two, j)) continue;
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t ob = row_get(b1, rowbase, j);
        va[j] = static_cast<P>(~P(0));
        if (oi & 2u) {
            const DevTree &T = s_tree[(oi >> 2) & 7u];
            const uint64_t i = static_cast<uint64_t>(ob) * kFan + sub;
            if (i < T.lvl_n[l]) va[j] = as_global<P>(T.lvl[l])[i];
        }
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_two, j)) continue;
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t a = (oi >> 15) & 31u, z = (oi >> 20) & 31u;
        const P oq1 = row_get(q1, rowbase, j);
        const uint32_t c1 = row_count(sub < a || (sub < z && va[j] < oq1), rowbase);
        if (static_cast<int>(sub) == j && two) t1 = b1 * kFan + c1;
    }
}

// The leaf level: {key, value} pairs, probed the same way.  For each live query returns g = # entries below q (in t),
// the pair before it (key pk, value pv: entry g-1) and the value of entry g (nv; every slice ends with a sentinel).
// Entries g-1 and g are read out of the registers of the row's lanes that loaded them: a rank costs no further gather.
template <typename P, typename L = RunList<P>>
__device__ __forceinline__ void coop_leaf(const DevTree *s_tree, const uint32_t tid, const uint32_t lo_t, const uint32_t hi_t, const bool live0,
                                          const bool live1, uint32_t &t0, uint32_t &t1, const P q0, const P q1, P &pk0, P &pv0, P &nv0, P &pk1,
                                          P &pv1, P &nv1) {
    typedef typename PairOf<P>::vec vec;
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & (kFan - 1), rowbase = lane & ~static_cast<uint32_t>(kFan - 1);
    const uint64_t m_live = __ballot(live0 || live1);
    if (!m_live) return;
    const uint32_t b0 = t0 - 1, b1 = t1 - 1;
    const bool two = live0 && live1 && b0 != b1;
    const uint32_t first = live0 ? b0 : b1;
    const uint32_t info = level_info(live0 || live1, two, tid, first, b1, lo_t, hi_t, 0);
    bool fix0 = false, fix1 = false;  // entry g is the first of the next block: fetched by the owner afterwards
    vec va[kFan];
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t ob = row_get(first, rowbase, j);
        va[j] = vec{static_cast<P>(~P(0)), 0};
        if (oi & 1u) {
            const DevTree &T = s_tree[(oi >> 2) & 7u];
            const uint64_t i = static_cast<uint64_t>(ob) * kFan + sub;
            if (i <= T.m) va[j] = L::load(T.ent, i);   // entry m is the last sentinel
        }
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint32_t oi = row_get(info, rowbase, j);
        const uint32_t a = (oi >> 5) & 31u, z = (oi >> 10) & 31u;
        const bool in = sub < z;
        const P oq0 = row_get(q0, rowbase, j), oq1 = row_get(q1, rowbase, j);   // (cross-lane: outside the || / &&)
        const uint32_t c0 = row_count(sub < a || (in && static_cast<P>(va[j].x) < oq0), rowbase);
        const uint32_t c1 = row_count(sub < a || (in && static_cast<P>(va[j].x) < oq1), rowbase);
        // a live query has c >= 1 (entry 16*block is the sample that was below q one level up); entries c - 1 and c are lanes
        const uint32_t p0 = c0 ? c0 - 1 : 0, n0 = c0 < kFan ? c0 : kFan - 1;
        const uint32_t p1 = c1 ? c1 - 1 : 0, n1 = c1 < kFan ? c1 : kFan - 1;
        const P a_pk0 = row_pick(static_cast<P>(va[j].x), rowbase, p0), a_pv0 = row_pick(static_cast<P>(va[j].y), rowbase, p0), a_nv0 = row_pick(static_cast<P>(va[j].y), rowbase, n0);
        const P a_pk1 = row_pick(static_cast<P>(va[j].x), rowbase, p1), a_pv1 = row_pick(static_cast<P>(va[j].y), rowbase, p1), a_nv1 = row_pick(static_cast<P>(va[j].y), rowbase, n1);
        if (static_cast<int>(sub) == j) {
            if (live0) { t0 = b0 * kFan + c0; pk0 = a_pk0; pv0 = a_pv0; nv0 = a_nv0; fix0 = c0 == kFan; }
            if (live1 && !two) { t1 = b1 * kFan + c1; pk1 = a_pk1; pv1 = a_pv1; nv1 = a_nv1; fix1 = c1 == kFan; }
        }
    }
    const uint64_t m_two = __ballot(two);
    if (m_two) {
#pragma unroll
        for (int j = 0; j < kFan; ++j) {
            if (!round_has_owner(m_two, j)) continue;
            const uint32_t oi = row_get(info, rowbase, j);
            const uint32_t ob = row_get(b1, rowbase, j);
            va[j] = vec{static_cast<P>(~P(0)), 0};
            if (oi & 2u) {
                const DevTree &T = s_tree[(oi >> 2) & 7u];
                const uint64_t i = static_cast<uint64_t>(ob) * kFan + sub;
                if (i <= T.m) va[j] = L::load(T.ent, i);
            }
        }
#pragma unroll
        for (int j = 0; j < kFan; ++j) {
            if (!round_has_owner(m_two, j)) continue;
            const uint32_t oi = row_get(info, rowbase, j);
            const uint32_t a = (oi >> 15) & 31u, z = (oi >> 20) & 31u;
            const P oq1 = row_get(q1, rowbase, j);
            const uint32_t c1 = row_count(sub < a || (sub < z && static_cast<P>(va[j].x) < oq1), rowbase);
            const uint32_t p1 = c1 ? c1 - 1 : 0, n1 = c1 < kFan ? c1 : kFan - 1;
            const P a_pk1 = row_pick(static_cast<P>(va[j].x), rowbase, p1), a_pv1 = row_pick(static_cast<P>(va[j].y), rowbase, p1), a_nv1 = row_pick(static_cast<P>(va[j].y), rowbase, n1);
            if (static_cast<int>(sub) == j && two) { t1 = b1 * kFan + c1; pk1 = a_pk1; pv1 = a_pv1; nv1 = a_nv1; fix1 = c1 == kFan; }
        }
    }
    if (fix0 || fix1) {
        const void *__restrict__ ent = s_tree[tid].ent;
        if (fix0) nv0 = L::val(ent, t0);
        if (fix1) nv1 = L::val(ent, t1);
    }
}

// The phi directory's probe with request slots: the row's lanes load the z candidates start .. start + z - 1 of the
// owner's position (lanes beyond them re-read the last one: the probe touches only the sectors that hold them), the lane
// that holds the last sampled position below q computes phi's value base + (q - pos) itself, and t = start + # of them
// below q travels back with it (val undefined when t == start).  One query per lane; every lane of the wave must call.
template <typename P>
__device__ __forceinline__ void coop_probe_phi(const DevTree &T, uint4 *req, const bool live, const uint32_t start, const uint32_t z, const P q, uint32_t &t,
                                               P &val) {
    typedef typename PairOf<P>::vec vec;
    constexpr int NS = ReqSlots<P>::v;
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & (kFan - 1), rowbase = lane & ~static_cast<uint32_t>(kFan - 1);
    const uint64_t m_live = __ballot(live);
    if (!m_live) return;
    const uint32_t zz = z > static_cast<uint32_t>(kFan) ? static_cast<uint32_t>(kFan) : z;
    wave_lds_sync();
    req[lane * NS + 0] = make_uint4(start, (live ? 1u : 0u) | (zz << 1), static_cast<uint32_t>(q), static_cast<uint32_t>(static_cast<uint64_t>(q) >> 32));
    wave_lds_sync();
    vec va[kFan];
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint4 a = req[(rowbase + j) * NS + 0];
        va[j] = vec{static_cast<P>(~P(0)), 0};
        if (a.y & 1u) {
            const uint32_t oz = a.y >> 1, last = oz ? oz - 1 : 0;
            uint64_t i = static_cast<uint64_t>(a.x) + (sub < last ? sub : last);
            if (i > T.m) i = T.m;   // entry m is the sentinel (never below a query)
            va[j] = PhiList<P>::load(T.ent, i);
        }
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint4 a = req[(rowbase + j) * NS + 0];
        const P oq = sizeof(P) == 8 ? static_cast<P>((static_cast<uint64_t>(a.w) << 32) | a.z) : static_cast<P>(a.z);
        const P key = static_cast<P>(va[j].x);
        const uint32_t c = row_count(sub < (a.y >> 1) && key < oq, rowbase);
        const P v = static_cast<P>(va[j].y) + (oq - key);
        const P a_v = row_pick(v, rowbase, c ? c - 1 : 0);
        if (static_cast<int>(sub) == j && live) { t = start + c; val = a_v; }
    }
}

// The two ranks of an LF step through the directories (K1/K2): position 0 is answered from entries s0 .. s0 + 15 of
// which the first z0 are candidates (the rest lie beyond the bucket, possibly in the next table's slice), position 1
// likewise from s1 / z1 -- sharing the load when s1 == s0.  Returns for each position t = s + # candidates below it, the
// RANK rk = cum[t-1] + min(q - start[t-1], cum[t] - cum[t-1]) (valid when t > s or an entry precedes s) and, for the
// second position, whether it lies inside that run (q - start <= length: the toehold test of the caller).  The lanes
// beyond the candidates re-read the entry after the last one, so a probe touches only the sectors that hold its z + 1
// entries.
// QUADS: four lanes serve an owner, each holding FOUR consecutive entries of its stretch (two 16-byte requests at
// 4-byte positions, four at 8-byte ones), so one wave-wide load instruction serves sixteen owners and a pass takes
// four rounds -- and everything the owner and its lanes tell each other travels by DPP quad permutes (a VALU move: no
// LDS traffic, no request area, no synchronisation): in round J the owner is lane J of each quad, its values are
// broadcast with quad_perm:[J,J,J,J], the number of candidates below the position is a quad sum, and the rank
// computed by the lane that holds the run (its fourth entry's length needs the next lane's first count: DPP row_shl;
// entry 15's is fetched by the owner when all sixteen lie below) comes back as a quad OR, the other three lanes
// contributing 0.  (Half-rows -- eight lanes per owner, two entries per lane, the owner's values through LDS request
// slots -- were the step before: 8.3 / 13.3 ms per 10 M reads at 4- / 8-byte positions against 8.2 / 11.4.)
template <int J> __device__ __forceinline__ uint32_t quad_get(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), J * 0x55, 0xF, 0xF, false));
}
template <int J> __device__ __forceinline__ uint64_t quad_get(uint64_t v) {
    return (static_cast<uint64_t>(quad_get<J>(static_cast<uint32_t>(v >> 32))) << 32) | quad_get<J>(static_cast<uint32_t>(v));
}
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    v += quad_perm<0xB1>(v);   // [1,0,3,2]
    v += quad_perm<0x4E>(v);   // [2,3,0,1]
    return v;
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    v |= quad_perm<0xB1>(v);
    v |= quad_perm<0x4E>(v);
    return v;
}
__device__ __forceinline__ uint64_t quad_or(uint64_t v) {
    return (static_cast<uint64_t>(quad_or(static_cast<uint32_t>(v >> 32))) << 32) | quad_or(static_cast<uint32_t>(v));
}
__device__ __forceinline__ bool round_has_owner4(uint64_t m, int j) { return (m & (0x1111111111111111ull << j)) != 0; }

// A lane's four entries AS LOADED: the words stay untouched until the round that uses them, so that the loads of all four
// rounds are in flight together (anything done to them inside the load step makes every round wait for its own data).
template <typename P> struct QuadRaw;
template <> struct QuadRaw<uint32_t> {
    typedef unsigned int vec4 __attribute__((ext_vector_type(4)));
    typedef PairOf<uint32_t>::vec vec;
    vec4 a, b;
    __device__ __forceinline__ void unpack(vec (&e)[4]) const { e[0] = vec{a.x, a.y}; e[1] = vec{a.z, a.w}; e[2] = vec{b.x, b.y}; e[3] = vec{b.z, b.w}; }
};
template <> struct QuadRaw<uint64_t> {
    typedef unsigned int vec4 __attribute__((ext_vector_type(4)));
    typedef PairOf<uint64_t>::vec vec;
    vec4 w[4];   // one 16-byte entry each
    __device__ __forceinline__ void unpack(vec (&e)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            e[i] = vec{static_cast<uint64_t>(w[i].x) | (static_cast<uint64_t>(w[i].y) << 32), static_cast<uint64_t>(w[i].z) | (static_cast<uint64_t>(w[i].w) << 32)};
    }
};

// the phi list's entries as loaded (rbg_dev.h PhiFmt)
template <typename P> struct PhiRaw;
template <> struct PhiRaw<uint32_t> : QuadRaw<uint32_t> {};
template <> struct PhiRaw<uint64_t> {
    typedef unsigned int vec4 __attribute__((ext_vector_type(4)));
    typedef PairOf<uint64_t>::vec vec;
    vec4 a, b, c;   // four 12-byte entries
    __device__ __forceinline__ void unpack(vec (&e)[4]) const {
        e[0] = PhiList<uint64_t>::unpack(a.x, a.y, a.z);
        e[1] = PhiList<uint64_t>::unpack(a.w, b.x, b.y);
        e[2] = PhiList<uint64_t>::unpack(b.z, b.w, c.x);
        e[3] = PhiList<uint64_t>::unpack(c.y, c.z, c.w);
    }
};

// the four entries start + 4 * sub .. + 3 of the owner's stretch (clamped to entry zc of the stretch and to the array's sentinel)
template <typename P, int J>
__device__ __forceinline__ void quad_load(const DevTree *s_tree, const uint32_t sub, const uint32_t start, const uint32_t info, const bool second,
                                          QuadRaw<P> &raw) {
    const uint32_t os = quad_get<J>(start), oi = quad_get<J>(info);
    // The loads are UNCONDITIONAL: a quad whose owner has no query in this pass reads the first entries of tree 0 (always
    // there, the same for everyone) and quad_round drops what it computes from them.  Loads under `if (owner has a query)`
    // with "never below" defaults on the other path made the compiler wait for each round's data inside the round (the
    // defaults and the loaded words met in different registers): four memory round trips per probe instead of one.
    // Entry indices are 32-bit (upload_tables_runs leaves out a depth with more entries): one multiply-add per address.
    const bool on = (oi & (second ? 2u : 1u)) != 0;
    const DevTree &T = s_tree[on ? (oi >> 2) & 7u : 0u];
    const uint32_t last = on ? static_cast<uint32_t>(T.m) : 0u;   // entry m is the last sentinel; no query: entry 0
    const uint32_t za = second ? 0u : (oi >> 5) & 31u, zb = (!second && (oi & 2u)) ? 0u : (oi >> 10) & 31u;
    uint32_t zc = za > zb ? za : zb;                      // entries 0 .. zc of the stretch are needed (zc: the one after the last candidate)
    if (zc > static_cast<uint32_t>(kFan - 1)) zc = kFan - 1;
    if constexpr (sizeof(P) == 4) {
        // two entries per request (16 bytes at any 8-byte boundary; the arrays end with one spare entry after the sentinel)
        typedef unsigned int vec4 __attribute__((ext_vector_type(4), aligned(8)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t idx = 4u * sub + 2u * h;
            const uint32_t g = min(os + min(idx, zc), last);
            const vec4 w = *as_global<vec4>(static_cast<const void *>(static_cast<const char *>(T.ent) + static_cast<uint64_t>(g) * 8u));
            if (h == 0) raw.a = w; else raw.b = w;
        }
    } else {
        // one 16-byte request per entry, each clamped to entry zc like the pairs above: the probe touches only the sectors
        // that hold entries 0 .. zc
        typedef unsigned int vec4 __attribute__((ext_vector_type(4)));
        const RBG_GLOBAL vec4 *base = as_global<vec4>(T.ent);
#pragma unroll
        for (int i = 0; i < 4; ++i) raw.w[i] = base[min(os + min(4u * sub + i, zc), last)];
    }
}

// one position against the quad's sixteen entries: c = # candidates (the first z entries) below q, rk = the rank, ins = q inside that run
template <typename P>
__device__ __forceinline__ void quad_rank(const uint32_t sub, const typename PairOf<P>::vec (&e)[4], const P next_cum, const uint32_t z, const P q,
                                          uint32_t &c, P &rk, bool &ins) {
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) n += (4u * sub + i < z && static_cast<P>(e[i].x) < q) ? 1u : 0u;
    c = quad_sum(n);
    const bool mine = n > 0 && ((c - 1) >> 2) == sub;          // the last entry below q is this lane's n-th
    // this lane's n-th entry and the count of the one after it (the next lane's first for the fourth)
    P k = static_cast<P>(e[0].x), v = static_cast<P>(e[0].y), vn = static_cast<P>(e[1].y);
    if (n == 2u) { k = static_cast<P>(e[1].x); v = static_cast<P>(e[1].y); vn = static_cast<P>(e[2].y); }
    if (n == 3u) { k = static_cast<P>(e[2].x); v = static_cast<P>(e[2].y); vn = static_cast<P>(e[3].y); }
    if (n == 4u) { k = static_cast<P>(e[3].x); v = static_cast<P>(e[3].y); vn = next_cum; }
    const P l = vn - v;
    const P d = q - k;
    const P r = v + (d < l ? d : l);
    rk = quad_or(mine ? r : P(0));
    ins = quad_or((mine && d <= l) ? 1u : 0u) != 0;
}

template <typename P, int J>
__device__ __forceinline__ void quad_round(const uint32_t sub, const bool second, const bool live, const bool two, const uint32_t s0, const uint32_t s1,
                                           const uint32_t info, const P q0, const P q1, const QuadRaw<P> &raw, uint32_t &t0, uint32_t &t1,
                                           P &rk0, P &rk1, bool &ins1, bool &fix0, bool &fix1) {
    typename PairOf<P>::vec e[4];
    raw.unpack(e);
    const uint32_t oi = quad_get<J>(info);
    const P oq0 = quad_get<J>(q0), oq1 = quad_get<J>(q1);
    const P next_cum = row_next(static_cast<P>(e[0].y));   // (the quad's last lane: entry 15's run, fixed up by the owner)
    uint32_t c0 = 0, c1 = 0;
    P a_r0 = 0, a_r1 = 0;
    bool a_i0 = false, a_i1 = false;
    if (!second) quad_rank<P>(sub, e, next_cum, (oi >> 5) & 31u, oq0, c0, a_r0, a_i0);
    quad_rank<P>(sub, e, next_cum, (oi >> 10) & 31u, oq1, c1, a_r1, a_i1);
    if (static_cast<int>(sub) == J) {
        if (!second && live) {
            t0 = s0 + c0; rk0 = a_r0; fix0 = c0 == kFan;
            if (!two) { t1 = s1 + c1; rk1 = a_r1; ins1 = a_i1; fix1 = c1 == kFan; }
        }
        if (second && two) { t1 = s1 + c1; rk1 = a_r1; ins1 = a_i1; fix1 = c1 == kFan; }
    }
}

template <typename P>
__device__ __forceinline__ void coop_probe2_rank4(const DevTree *s_tree, const uint32_t tid, const bool live, const uint32_t s0, const uint32_t z0,
                                                  const uint32_t s1, const uint32_t z1, const P q0, const P q1, uint32_t &t0, uint32_t &t1, P &rk0,
                                                  P &rk1, bool &ins1) {
    typedef typename PairOf<P>::vec vec;
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & 3u;
    const uint64_t m_live = __ballot(live);
    if (!m_live) return;
    const bool two = live && s1 != s0;
    // bit 0 live, bit 1 second block, bits 2-4 tree, bits 5-9 z0, 10-14 z1
    const uint32_t info = (live ? 1u : 0u) | (two ? 2u : 0u) | (tid << 2) | (z0 << 5) | (z1 << 10);
    bool fix0 = false, fix1 = false;
    QuadRaw<P> e0, e1, e2, e3;
    // (all four rounds' loads in one basic block, no branch between them: every request is in flight before the first wait)
    quad_load<P, 0>(s_tree, sub, s0, info, false, e0);
    quad_load<P, 1>(s_tree, sub, s0, info, false, e1);
    quad_load<P, 2>(s_tree, sub, s0, info, false, e2);
    quad_load<P, 3>(s_tree, sub, s0, info, false, e3);
    if (round_has_owner4(m_live, 0)) quad_round<P, 0>(sub, false, live, two, s0, s1, info, q0, q1, e0, t0, t1, rk0, rk1, ins1, fix0, fix1);
    if (round_has_owner4(m_live, 1)) quad_round<P, 1>(sub, false, live, two, s0, s1, info, q0, q1, e1, t0, t1, rk0, rk1, ins1, fix0, fix1);
    if (round_has_owner4(m_live, 2)) quad_round<P, 2>(sub, false, live, two, s0, s1, info, q0, q1, e2, t0, t1, rk0, rk1, ins1, fix0, fix1);
    if (round_has_owner4(m_live, 3)) quad_round<P, 3>(sub, false, live, two, s0, s1, info, q0, q1, e3, t0, t1, rk0, rk1, ins1, fix0, fix1);
    const uint64_t m_two = __ballot(two);
    if (m_two) {
        // (second blocks are sparse -- most rounds have no owner: skipping them saves a third of the kernel's load instructions)
        if (round_has_owner4(m_two, 0)) quad_load<P, 0>(s_tree, sub, s1, info, true, e0);
        if (round_has_owner4(m_two, 1)) quad_load<P, 1>(s_tree, sub, s1, info, true, e1);
        if (round_has_owner4(m_two, 2)) quad_load<P, 2>(s_tree, sub, s1, info, true, e2);
        if (round_has_owner4(m_two, 3)) quad_load<P, 3>(s_tree, sub, s1, info, true, e3);
        if (round_has_owner4(m_two, 0)) quad_round<P, 0>(sub, true, live, two, s0, s1, info, q0, q1, e0, t0, t1, rk0, rk1, ins1, fix0, fix1);
        if (round_has_owner4(m_two, 1)) quad_round<P, 1>(sub, true, live, two, s0, s1, info, q0, q1, e1, t0, t1, rk0, rk1, ins1, fix0, fix1);
        if (round_has_owner4(m_two, 2)) quad_round<P, 2>(sub, true, live, two, s0, s1, info, q0, q1, e2, t0, t1, rk0, rk1, ins1, fix0, fix1);
        if (round_has_owner4(m_two, 3)) quad_round<P, 3>(sub, true, live, two, s0, s1, info, q0, q1, e3, t0, t1, rk0, rk1, ins1, fix0, fix1);
    }
    if (fix0 || fix1) {   // all 16 loaded entries lie below the position: the run it lands in ends in the next block
        const void *__restrict__ ent = s_tree[tid].ent;
        if (fix0) {
            const vec e = RunList<P>::load(ent, t0 - 1);
            const P len = static_cast<P>(RunList<P>::val(ent, t0)) - static_cast<P>(e.y), d = q0 - static_cast<P>(e.x);
            rk0 = static_cast<P>(e.y) + (d < len ? d : len);
        }
        if (fix1) {
            const vec e = RunList<P>::load(ent, t1 - 1);
            const P len = static_cast<P>(RunList<P>::val(ent, t1)) - static_cast<P>(e.y), d = q1 - static_cast<P>(e.x);
            rk1 = static_cast<P>(e.y) + (d < len ? d : len);
            ins1 = d <= len;
        }
    }
}

// The phi directory's probe by QUADS (K3's ordered walk): four lanes per owner, four consecutive sampled positions per
// lane (two 16-byte requests at 4-byte positions), the owner's start and position by DPP quad permutes, the number
// of samples below the position as a quad sum, phi's value base + (q - pos) from the lane that holds the last of them
// as a quad OR.  t = start + # entries below q (val undefined when t == start).
template <typename P, int J>
__device__ __forceinline__ void phi_quad_load(const DevTree &T, const uint32_t sub, const uint32_t info, PhiRaw<P> &raw) {
    const uint32_t oi = quad_get<J>(info);
    // (unconditional, as in quad_load: a quad without a query reads the list's first entries and its round's result is dropped)
    const uint32_t first = (oi & 0x80000000u) ? (oi & 0x7FFFFFFFu) + 4u * sub : 0u;   // (fewer than 2^31 sampled positions: upload() checks)
    const uint32_t last = static_cast<uint32_t>(T.m);   // entry m is the sentinel (never below a query)
    if constexpr (sizeof(P) == 4) {   // (the array ends with one spare entry after the sentinel)
        typedef unsigned int vec4 __attribute__((ext_vector_type(4), aligned(8)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t g = min(first + 2u * h, last);
            const vec4 w = *as_global<vec4>(static_cast<const void *>(static_cast<const char *>(T.ent) + static_cast<uint64_t>(g) * 8u));
            if (h == 0) raw.a = w; else raw.b = w;
        }
    } else {   // four 12-byte entries = three 16-byte requests at a 4-byte boundary (three spare entries follow the sentinel)
        typedef unsigned int vec4 __attribute__((ext_vector_type(4), aligned(4)));
        const RBG_GLOBAL vec4 *w = as_global<vec4>(static_cast<const void *>(static_cast<const char *>(T.ent) + static_cast<uint64_t>(min(first, last)) * 12u));
        raw.a = w[0]; raw.b = w[1]; raw.c = w[2];
    }
}

template <typename P, int J>
__device__ __forceinline__ void phi_quad_round(const uint32_t sub, const bool live, const uint32_t start, const P q, const PhiRaw<P> &raw,
                                               uint32_t &t, P &val) {
    typename PairOf<P>::vec e[4];
    raw.unpack(e);
    const P oq = quad_get<J>(q);
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) n += static_cast<P>(e[i].x) < oq ? 1u : 0u;
    const uint32_t c = quad_sum(n);
    const bool mine = n > 0 && ((c - 1) >> 2) == sub;
    P k = static_cast<P>(e[0].x), v = static_cast<P>(e[0].y);
    if (n == 2u) { k = static_cast<P>(e[1].x); v = static_cast<P>(e[1].y); }
    if (n == 3u) { k = static_cast<P>(e[2].x); v = static_cast<P>(e[2].y); }
    if (n == 4u) { k = static_cast<P>(e[3].x); v = static_cast<P>(e[3].y); }
    const P a_v = quad_or(mine ? v + (oq - k) : P(0));
    if (static_cast<int>(sub) == J && live) { t = start + c; val = a_v; }
}

template <typename P>
__device__ __forceinline__ void coop_probe_phi4(const DevTree &T, const bool live, const uint32_t start, const P q, uint32_t &t, P &val) {
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & 3u;
    const uint64_t m_live = __ballot(live);
    if (!m_live) return;
    const uint32_t info = (live ? 0x80000000u : 0u) | (start & 0x7FFFFFFFu);   // (fewer than 2^31 sampled positions: upload() checks)
    PhiRaw<P> e0, e1, e2, e3;
    phi_quad_load<P, 0>(T, sub, info, e0);
    phi_quad_load<P, 1>(T, sub, info, e1);
    phi_quad_load<P, 2>(T, sub, info, e2);
    phi_quad_load<P, 3>(T, sub, info, e3);
    if (round_has_owner4(m_live, 0)) phi_quad_round<P, 0>(sub, live, start, q, e0, t, val);
    if (round_has_owner4(m_live, 1)) phi_quad_round<P, 1>(sub, live, start, q, e1, t, val);
    if (round_has_owner4(m_live, 2)) phi_quad_round<P, 2>(sub, live, start, q, e2, t, val);
    if (round_has_owner4(m_live, 3)) phi_quad_round<P, 3>(sub, live, start, q, e3, t, val);
}

// One narrowing round for a crowded bucket: the candidates [s, s + z) of a query (z > 16) are sampled at 16 pivots a
// stride apart; the answer lies between the last pivot below q and the next one, so the range shrinks to at most
// ceil(z / 16) candidates (s moves to that pivot, which is known to be below q -- or stays with z = 1 when not even
// the first candidate is).  Lanes with live == false pass through.  Every lane must call.  By QUADS: the sixteen
// pivots are four per lane of the owner's quad, its (s, z, tree, q) reach the quad by DPP and the number of pivots
// below q is a quad sum -- no LDS traffic.
template <typename P, int J, typename L>
__device__ __forceinline__ void narrow_quad_load(const DevTree *s_tree, const uint32_t sub, const uint32_t flags, const uint32_t s, const uint32_t z, P (&key)[4]) {
    const uint32_t of = quad_get<J>(flags), os = quad_get<J>(s), oz = quad_get<J>(z);
    const uint32_t ost = (oz + kFan - 1) / kFan;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        key[i] = static_cast<P>(~P(0));   // pivots beyond the candidates are never below
        const uint32_t at = (4u * sub + i) * ost;
        if ((of & 1u) && at < oz) key[i] = L::key(s_tree[(of >> 1) & 7u].ent, static_cast<uint64_t>(os) + at);
    }
}
template <typename P, int J>
__device__ __forceinline__ void narrow_quad_round(const uint32_t sub, const bool live, const uint32_t stride, const P q, const P (&key)[4], uint32_t &s, uint32_t &z) {
    const P oq = quad_get<J>(q);
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) n += key[i] < oq ? 1u : 0u;
    const uint32_t c = quad_sum(n);
    if (static_cast<int>(sub) == J && live) {
        if (c == 0) { z = 1; }
        else {
            const uint32_t adv = (c - 1) * stride;
            s += adv;
            z = (z - adv) < stride ? (z - adv) : stride;
        }
    }
}
template <typename P, typename L = RunList<P>>
__device__ __forceinline__ void coop_narrow4(const DevTree *s_tree, const uint32_t tid, const bool live, uint32_t &s, uint32_t &z, const P q) {
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & 3u;
    const uint64_t m_live = __ballot(live);
    if (!m_live) return;
    const uint32_t stride = (z + kFan - 1) / kFan;
    const uint32_t flags = (live ? 1u : 0u) | (tid << 1);
    const uint32_t s_in = s, z_in = z;   // (the owners update s and z while later rounds still broadcast the others')
    P k0[4], k1[4], k2[4], k3[4];
    if (round_has_owner4(m_live, 0)) narrow_quad_load<P, 0, L>(s_tree, sub, flags, s_in, z_in, k0);
    if (round_has_owner4(m_live, 1)) narrow_quad_load<P, 1, L>(s_tree, sub, flags, s_in, z_in, k1);
    if (round_has_owner4(m_live, 2)) narrow_quad_load<P, 2, L>(s_tree, sub, flags, s_in, z_in, k2);
    if (round_has_owner4(m_live, 3)) narrow_quad_load<P, 3, L>(s_tree, sub, flags, s_in, z_in, k3);
    if (round_has_owner4(m_live, 0)) narrow_quad_round<P, 0>(sub, live, stride, q, k0, s, z);
    if (round_has_owner4(m_live, 1)) narrow_quad_round<P, 1>(sub, live, stride, q, k1, s, z);
    if (round_has_owner4(m_live, 2)) narrow_quad_round<P, 2>(sub, live, stride, q, k2, s, z);
    if (round_has_owner4(m_live, 3)) narrow_quad_round<P, 3>(sub, live, stride, q, k3, s, z);
}

// The two ranks of an LF step through the BUCKET RECORDS (rbg_dev.h RunRec): the row's lanes load the 128-byte record of
// the owner's bucket with one coalesced request (lane 0 the header, lane 1 the rank at the bucket's start, lanes 2-15
// the pairs); each pair's lane computes the rank the position would have if it fell into ITS run, popcount(ballot)
// finds the run it does fall into, and one value per position travels back -- no directory gather, no second round
// trip.  r0 / r1: the records (index into the depth's array) of the two positions, o0 / o1 their offsets in the bucket.
// Returns per position c = # runs of the record that start below it (0: no run of the table starts below it at all),
// the rank rk (valid when c > 0) and for the second whether it lies inside its run; ov = one of the two buckets
// holds more runs than a record (the caller reads (e0, count) from the headers and goes through the run list).
template <typename P>
__device__ __forceinline__ void coop_rec2(const RunRec *const *s_rec, uint4 *req, const uint32_t tid, const bool live, const uint32_t r0, const uint32_t r1,
                                          const uint32_t o0, const uint32_t o1, uint32_t &c0, uint32_t &c1, P &rk0, P &rk1, bool &ins1, bool &ov) {
    typedef unsigned int vec2 __attribute__((ext_vector_type(2)));
    constexpr int NS = ReqSlots<P>::v;
    const uint32_t lane = threadIdx.x & (kWave - 1), sub = lane & (kFan - 1), rowbase = lane & ~static_cast<uint32_t>(kFan - 1);
    const uint64_t m_live = __ballot(live);
    if (!m_live) return;
    const bool two = live && r1 != r0;
    const uint32_t info = (live ? 1u : 0u) | (two ? 2u : 0u) | (tid << 2);
    wave_lds_sync();
    req[lane * NS + 0] = make_uint4(r0, info, o0, o1);
    req[lane * NS + 1] = make_uint4(r1, info, o1, 0u);
    wave_lds_sync();
    vec2 va[kFan];
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint4 a = req[(rowbase + j) * NS + 0];
        va[j] = vec2{kRecNoPair, 0u};
        if (a.y & 1u) va[j] = as_global<vec2>(static_cast<const void *>(s_rec[(a.y >> 2) & 7u] + a.x))[sub];
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_live, j)) continue;
        const uint4 a = req[(rowbase + j) * NS + 0];
        const uint32_t off = va[j].x, cl = va[j].y;
        const uint32_t len = row_next(cl) - cl;                   // (lane 15 holds the closing pair: never chosen)
        const bool pair_lane = sub >= 2u;
        const uint32_t n0 = row_count(pair_lane && static_cast<int32_t>(off) < static_cast<int32_t>(a.z), rowbase);
        const uint32_t n1 = row_count(pair_lane && static_cast<int32_t>(off) < static_cast<int32_t>(a.w), rowbase);
        const uint32_t d0 = a.z - off, d1 = a.w - off;
        const uint32_t v0 = cl + (d0 < len ? d0 : len), v1 = cl + (d1 < len ? d1 : len);
        const uint64_t in1 = __ballot(d1 <= len);
        const uint64_t ovm = __ballot(sub == 0u && (cl & kRecOverflow));   // (the header's flags sit in lane 0's second word)
        // pair n - 1 sits in lane n + 1
        const uint32_t a_v0 = row_pick(v0, rowbase, n0 + 1u), a_v1 = row_pick(v1, rowbase, n1 + 1u);
        uint32_t blo = 0, bhi = 0;
        if (sizeof(P) == 8) { blo = row_pick(off, rowbase, 1u); bhi = row_pick(cl, rowbase, 1u); }
        if (static_cast<int>(sub) == j && live) {
            const P base = static_cast<P>((static_cast<uint64_t>(bhi) << 32) | blo);   // (0 at 4-byte positions: the pairs carry the rank itself)
            ov = ((ovm >> rowbase) & 1u) != 0;
            c0 = n0; rk0 = base + static_cast<P>(a_v0);
            if (!two) { c1 = n1; rk1 = base + static_cast<P>(a_v1); ins1 = ((in1 >> (rowbase + n1 + 1u)) & 1u) != 0; }
        }
    }
    const uint64_t m_two = __ballot(two);
    if (!m_two) return;
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_two, j)) continue;
        const uint4 b = req[(rowbase + j) * NS + 1];
        va[j] = vec2{kRecNoPair, 0u};
        if (b.y & 2u) va[j] = as_global<vec2>(static_cast<const void *>(s_rec[(b.y >> 2) & 7u] + b.x))[sub];
    }
#pragma unroll
    for (int j = 0; j < kFan; ++j) {
        if (!round_has_owner(m_two, j)) continue;
        const uint4 b = req[(rowbase + j) * NS + 1];
        const uint32_t off = va[j].x, cl = va[j].y;
        const uint32_t len = row_next(cl) - cl;
        const uint32_t n1 = row_count(sub >= 2u && static_cast<int32_t>(off) < static_cast<int32_t>(b.z), rowbase);
        const uint32_t d1 = b.z - off;
        const uint32_t v1 = cl + (d1 < len ? d1 : len);
        const uint64_t in1 = __ballot(d1 <= len);
        const uint64_t ovm = __ballot(sub == 0u && (cl & kRecOverflow));
        const uint32_t a_v1 = row_pick(v1, rowbase, n1 + 1u);
        uint32_t blo = 0, bhi = 0;
        if (sizeof(P) == 8) { blo = row_pick(off, rowbase, 1u); bhi = row_pick(cl, rowbase, 1u); }
        if (static_cast<int>(sub) == j && two) {
            const P base = static_cast<P>((static_cast<uint64_t>(bhi) << 32) | blo);
            ov = ov || ((ovm >> rowbase) & 1u) != 0;
            c1 = n1; rk1 = base + static_cast<P>(a_v1); ins1 = ((in1 >> (rowbase + n1 + 1u)) & 1u) != 0;
        }
    }
}

// per-lane search of the staged top level, clamped like the levels below: entries [a, z) of the tree's slice of s_top
// are the ones inside the query's slice; returns # top entries below q
template <typename P>
__device__ __forceinline__ uint32_t top_count_clamped(const P *s_top, const DevTree &T, int top_sh, uint32_t lo_t, uint32_t hi_t, uint64_t q) {
    const uint64_t S1 = (uint64_t(1) << top_sh) - 1;
    uint64_t a = (static_cast<uint64_t>(lo_t) + S1) >> top_sh, z = (static_cast<uint64_t>(hi_t) + S1) >> top_sh;
    if (z > T.top_n) z = T.top_n;
    if (a > z) a = z;
    return static_cast<uint32_t>(a) + top_count<P>(s_top, T.top_off + static_cast<uint32_t>(a), static_cast<uint32_t>(z - a), q);
}


template <typename Kernel>
void raise_lds(Kernel kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> raised;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, reinterpret_cast<const void *>(kernel));
    std::lock_guard<std::mutex> g(mu);
    if (raised.insert(key).second) (void)hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

// ---- what a workgroup stages for the rank search, and one LF step of a whole wave -----------------------------------
template <typename P>
struct RunSearch {
    const DevTree *tree;         // [kMaxRunDepth]: one tree per k-mer depth
    const DevRunTab *tab;        // the tables' records (run_ntabs of them)
    const uint32_t *tab_first;   // [kMaxRunDepth + 1]: first record of each depth
    const RunRec *const *rec;    // [8]: bucket records per depth (nullptr: directories / descent)
    const P *top;                // staged top levels of the trees (descent only)
    uint4 *req;                  // this wave's request slots (kWave * ReqSlots<P>::v)
    int nlvl, top_sh;
};

// LDS a kernel that searches the run lists declares: static part (macro: the arrays must be __shared__ in the kernel)
// and dynamic part = run_ntabs records + the trees' top level.
#define RBG_RUN_SEARCH_SHARED(P, WAVES)                                   \
    __shared__ DevTree s_tree[kMaxRunDepth];                              \
    __shared__ uint32_t s_tab_first[kMaxRunDepth + 1];                    \
    __shared__ const RunRec *s_rec[8];                                    \
    __shared__ uint4 s_req[WAVES][kWave * ReqSlots<P>::v];                \
    extern __shared__ __align__(16) unsigned char s_dyn[]

inline size_t run_search_lds(const DevIndex &ix) {
    return static_cast<size_t>(ix.run_ntabs) * sizeof(DevRunTab) + static_cast<size_t>(ix.tree_top_n) * ix.pos_bytes + 16;
}

// fills the arrays of RBG_RUN_SEARCH_SHARED and returns the view of them; ends with __syncthreads()
template <typename P, int WAVES>
__device__ __forceinline__ RunSearch<P> stage_run_search(const DevIndex &ix, DevTree *s_tree, uint32_t *s_tab_first, const RunRec **s_rec,
                                                         uint4 (*s_req)[kWave * ReqSlots<P>::v], unsigned char *s_dyn) {
    DevRunTab *s_tab = reinterpret_cast<DevRunTab *>(s_dyn);
    P *s_top = reinterpret_cast<P *>(s_dyn + static_cast<size_t>(ix.run_ntabs) * sizeof(DevRunTab));
    if (threadIdx.x < 8) s_rec[threadIdx.x] = threadIdx.x < static_cast<uint32_t>(kMaxRunDepth) ? ix.run_rec[threadIdx.x] : nullptr;
    const uint32_t D = ix.run_ksteps;
    for (uint32_t t = threadIdx.x; t < D; t += blockDim.x) s_tree[t] = ix.trees[t];
    for (uint32_t t = threadIdx.x; t <= static_cast<uint32_t>(kMaxRunDepth); t += blockDim.x) s_tab_first[t] = ix.run_tab_first[t];
    for (uint32_t t = threadIdx.x; t < ix.run_ntabs; t += blockDim.x) s_tab[t] = ix.run_tabs[t];
    for (uint32_t t = threadIdx.x; t < ix.tree_top_n; t += blockDim.x) s_top[t] = static_cast<const P *>(ix.tree_top)[t];
    __syncthreads();
    RunSearch<P> S;
    S.tree = s_tree; S.tab = s_tab; S.tab_first = s_tab_first; S.rec = s_rec; S.top = s_top;
    S.req = s_req[(threadIdx.x >> 6) % WAVES];
    S.nlvl = static_cast<int>(ix.tree_nlvl);
    S.top_sh = 4 * (S.nlvl + 1);
    return S;
}

// the record of the step that consumes the k-mer `acc` (read as a base-nmajor number, least significant digit = the
// symbol next to the suffix) of `adv` symbols; adv == 1: acc is the symbol's slot
__device__ __forceinline__ uint32_t run_record(const uint32_t *s_tab_first, uint32_t adv, uint32_t acc) { return s_tab_first[adv - 1] + acc; }

// result of one lane's step
struct RunStep {
    uint64_t F = 0, c_before = 0, c_upto = 0;   // the table's first row; rank(lo, .), rank(hi + 1, .)
    bool inside = false;                        // row hi lies in a run of the table (LF_w_loc's fast branch, rowbowt.hpp:559-561)
    // else the predecessor run's sample: entry samp_run of depth d's arrays, or (samp_c > 0) entry e0 + samp_c - 1 of bucket record samp_run
    uint32_t samp_run = 0, samp_c = 0;
    uint64_t samp_e = 0;                        // format 2 (rbg_runs2_device.hpp): that entry's index in the depth's arrays
};
// its sample (one gather by the lane itself)
template <typename P>
__device__ __forceinline__ uint64_t run_step_sample(const DevIndex &ix, const RunRec *const *s_rec, uint32_t d, const RunStep &r) {
    uint32_t e = r.samp_run;
    if (r.samp_c) e = as_global(s_rec[d])[r.samp_run].e0 + r.samp_c - 1;
    return RunList<P>::samp(ix.run_samp[d], e);
}

// What the instrumented instantiations count on this layout (the same eight sums as SearchStat, other meanings:
// include/rbg.h): [kStSteps] search steps, [kStSlots] directory gathers (8 bytes: two neighbouring entries), [kStDense]
// run-list entries the probes needed (z + 1 each, at most 16; 2P bytes each), [kStSearch] narrowing rounds (16 pivot
// keys each), [kStFtab], [kStResample] (one sample gather), [kStChunks], [kStSymbols].

// Both ranks of one LF step for every lane of the wave: lanes with `stepping` carry (d = depth index, rec = record in
// S.tab, q0 = lo, q1 = hi + 1) and get the step's RunStep; every lane of the wave must call (the probes are
// cooperative).  rle_string::rank (rle_string.hpp:131-161) in the k-mer table: occurrences before the predecessor run
// plus the part of it below the position.
template <typename P, bool STATS = false>
__device__ __forceinline__ void coop_lf2(const DevIndex &ix, const RunSearch<P> &S, const bool stepping, const uint32_t d, const uint32_t rec,
                                         const uint64_t q0, const uint64_t q1, RunStep &out, unsigned long long *st = nullptr) {
    uint32_t lo_t = 0, hi_t = 0;
    uint32_t t0 = 0, t1 = 0;
    P pk0 = 0, pv0 = 0, nv0 = 0, pk1 = 0, pv1 = 0, nv1 = 0;
    bool descend = stepping;
    uint32_t s0 = 0, z0 = 0, s1 = 0, z1 = 0;
    bool direct = false, by_rec = false, ov = false;
    uint32_t rc0 = 0, rc1 = 0, o0 = 0, o1 = 0;
    if (stepping) {
        const DevRunTab r0 = S.tab[rec];
        out.F = r0.F;
        lo_t = static_cast<uint32_t>(r0.first);
        hi_t = static_cast<uint32_t>(S.tab[rec + 1].first) - 1;   // the slice's sentinel: never below a query
        const uint32_t *__restrict__ dir = ix.run_dir[d];
        if (S.rec[d]) {
            // the table's bucket records: one record per position answers its rank
            const uint64_t b0 = q0 >> r0.dir_shift, b1 = q1 >> r0.dir_shift;
            rc0 = r0.dir_off + static_cast<uint32_t>(b0);
            rc1 = r0.dir_off + static_cast<uint32_t>(b1);
            o0 = static_cast<uint32_t>(q0 - (b0 << r0.dir_shift));
            o1 = static_cast<uint32_t>(q1 - (b1 << r0.dir_shift));
            by_rec = true;
            descend = false;
        } else if (dir) {
            // the table's directory: # runs starting below the bucket of q and below the next bucket; the
            // candidates are those runs and the one before them
            dir += r0.dir_off;
            const uint64_t b0 = q0 >> r0.dir_shift, b1 = q1 >> r0.dir_shift;
            const uint32_t a0 = dir[b0], e0 = dir[b0 + 1];
            uint32_t a1 = a0, e1 = e0;
            if (b1 != b0) { a1 = dir[b1]; e1 = dir[b1 + 1]; }
            if (STATS) st[kStSlots] += b1 != b0 ? 2 : 1;
            s0 = lo_t + (a0 ? a0 - 1 : 0);
            s1 = lo_t + (a1 ? a1 - 1 : 0);
            z0 = lo_t + e0 - s0;
            z1 = lo_t + e1 - s1;
            // neighbouring buckets: when the second position's candidates end within 16 entries of the first's start, one
            // probe from there answers both (every entry before s1 is below q1 anyway)
            if (s1 != s0 && s1 + z1 - s0 <= static_cast<uint32_t>(kFan)) { z1 = s1 + z1 - s0; s1 = s0; }
            direct = true;   // (crowded buckets are narrowed below until one row probe covers their candidates)
            descend = false;
        }
    }
    P rk0 = 0, rk1 = 0;
    bool ins1 = false;
    uint32_t cn0 = 0, cn1 = 0;
    coop_rec2<P>(S.rec, S.req, d, by_rec, rc0, rc1, o0, o1, cn0, cn1, rk0, rk1, ins1, ov);
    if (by_rec && ov) {   // a bucket with more runs than a record holds: through the run list, like a directory's
        const RBG_GLOBAL RunRec *R = as_global(S.rec[d]);
        const uint32_t f0 = R[rc0].flags, f1 = R[rc1].flags;
        s0 = R[rc0].e0; z0 = (f0 & kRecOverflow) ? (f0 & 0x7FFFFFFFu) : (f0 & 0xFFu);
        s1 = R[rc1].e0; z1 = (f1 & kRecOverflow) ? (f1 & 0x7FFFFFFFu) : (f1 & 0xFFu);
        direct = true;
    }
    while (__ballot(direct && (z0 > static_cast<uint32_t>(kFan) || z1 > static_cast<uint32_t>(kFan)))) {
        if (STATS && direct) st[kStSearch] += (z0 > static_cast<uint32_t>(kFan) ? 1 : 0) + (z1 > static_cast<uint32_t>(kFan) ? 1 : 0);
        coop_narrow4<P>(S.tree, d, direct && z0 > static_cast<uint32_t>(kFan), s0, z0, static_cast<P>(q0));
        coop_narrow4<P>(S.tree, d, direct && z1 > static_cast<uint32_t>(kFan), s1, z1, static_cast<P>(q1));
    }
    if (STATS && direct) {   // entries the probe(s) need: the candidates and the one after the last
        const uint32_t m0 = z0 + 1 > static_cast<uint32_t>(kFan) ? kFan : z0 + 1, m1 = z1 + 1 > static_cast<uint32_t>(kFan) ? kFan : z1 + 1;
        st[kStDense] += s1 == s0 ? (m0 > m1 ? m0 : m1) : m0 + m1;
    }
    coop_probe2_rank4<P>(S.tree, d, direct, s0, z0, s1, z1, static_cast<P>(q0), static_cast<P>(q1), t0, t1, rk0, rk1, ins1);
    if (__ballot(descend)) {   // indexes without directories: the clamped descent
        uint32_t d0 = 0, d1 = 0;
        P ak0 = 0, av0 = 0, an0 = 0, ak1 = 0, av1 = 0, an1 = 0;
        if (descend) {
            d0 = top_count_clamped<P>(S.top, S.tree[d], S.top_sh, lo_t, hi_t, q0);
            d1 = top_count_clamped<P>(S.top, S.tree[d], S.top_sh, lo_t, hi_t, q1);
        }
        for (int l = S.nlvl - 1; l >= 0; --l) {
            const bool l0 = descend && d0 > 0, l1 = descend && d1 > 0;
            coop_level<P>(S.tree, l, d, lo_t, hi_t, l0, l1, d0, d1, static_cast<P>(q0), static_cast<P>(q1));
        }
        {
            const bool l0 = descend && d0 > 0, l1 = descend && d1 > 0;
            coop_leaf<P>(S.tree, d, lo_t, hi_t, l0, l1, d0, d1, static_cast<P>(q0), static_cast<P>(q1), ak0, av0, an0, ak1, av1, an1);
        }
        if (descend) { t0 = d0; t1 = d1; pk0 = ak0; pv0 = av0; nv0 = an0; pk1 = ak1; pv1 = av1; nv1 = an1; }
    }
    if (stepping) {
        // t <= lo_t: no run of this table starts before the position
        uint64_t c_before = 0, c_upto = 0;
        bool inside = false;
        if (direct) {   // (the quad computed the ranks)
            if (t0 > lo_t) c_before = rk0;
            if (t1 > lo_t) { c_upto = rk1; inside = ins1; }
        } else if (by_rec) {
            if (cn0) c_before = rk0;
            if (cn1) { c_upto = rk1; inside = ins1; }
        } else {
            if (t0 > lo_t) { const uint64_t len = static_cast<uint64_t>(nv0) - pv0, dd = q0 - pk0; c_before = pv0 + (dd < len ? dd : len); }
            if (t1 > lo_t) { const uint64_t len = static_cast<uint64_t>(nv1) - pv1, dd = q1 - pk1; c_upto = pv1 + (dd < len ? dd : len); inside = dd <= len; }
        }
        out.c_before = c_before;
        out.c_upto = c_upto;
        out.inside = inside;
        if (by_rec && !direct) { out.samp_run = rc1; out.samp_c = cn1; }   // entry e0 + cn1 - 1 of the record's header
        else { out.samp_run = t1 - 1; out.samp_c = 0; }
        if (STATS) st[kStSteps] += 1;
    }
}

}  // namespace
}  // namespace rbg
