// k_locate.hip -- K3: phi chains (ToeholdSA::locate_range, toehold_sa.hpp:37-49), their plan (counts + scan) and their order (radix sort of the toeholds)
#include <cstdlib>

#include "rbg_device.hpp"

namespace rbg {
namespace {

// ---- K3: locate (the counts of the plan are computed where its scan reads them: OccOf below) ---------
// ---- the locus order's sort key (rbg_dev.h order_docs): k <-> {offset >> low, document, offset & (2^low - 1)} -----------------------------
// A toehold outside the text (it wrapped below zero: k >= n) has no document: its key is all ones and K3 reads the toehold itself.
__device__ __forceinline__ uint64_t locus_key(const DevIndex &ix, const uint64_t k) {
    if (k >= ix.n) return ~uint64_t(0);
    uint32_t a = 0, z = ix.order_ndocs;          // the last document that starts at or before k (doclist.hpp:46-50: a predecessor query)
    while (z - a > 1u) {
        const uint32_t m = (a + z) >> 1;
        if (ix.order_docs[m] <= k) a = m; else z = m;
    }
    const uint64_t o = k - ix.order_docs[a];
    const uint32_t low = ix.order_lowbits, db = ix.order_dbits;
    return ((o >> low) << (db + low)) | (static_cast<uint64_t>(a) << low) | (o & ((uint64_t(1) << low) - 1u));
}
__device__ __forceinline__ uint64_t locus_toehold(const DevIndex &ix, const uint64_t key, const uint64_t *__restrict__ k, const uint64_t i) {
    if (key == ~uint64_t(0)) return k[i];
    const uint32_t low = ix.order_lowbits, db = ix.order_dbits;
    const uint64_t doc = (key >> low) & ((uint64_t(1) << db) - 1u);
    return ix.order_docs[doc] + (((key >> (db + low)) << low) | (key & ((uint64_t(1) << low) - 1u)));
}
// 4-byte positions, absolute order: the sort moves 32-bit keys (a third fewer bytes per radix pass than 64-bit ones).  A toehold outside the text
// (it wrapped below zero) travels as 0xFFFFFFFF -- no position of an index with 4-byte positions -- and K3 reads the toehold itself.
__global__ __launch_bounds__(256) void k_keys32(const DevIndex ix, const uint64_t *__restrict__ k, const uint64_t N, uint32_t *__restrict__ keys, uint32_t *__restrict__ iota) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint64_t v = k[i];
        keys[i] = v >= ix.n ? 0xFFFFFFFFu : static_cast<uint32_t>(v);
        iota[i] = static_cast<uint32_t>(i);
    }
}
__global__ __launch_bounds__(256) void k_locus_keys(const DevIndex ix, const uint64_t *__restrict__ k, const uint64_t N, uint64_t *__restrict__ keys,
                                                    uint32_t *__restrict__ iota) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        keys[i] = locus_key(ix, k[i]);
        iota[i] = static_cast<uint32_t>(i);
    }
}

template <typename P>
__device__ __forceinline__ uint64_t phi_step(const DevIndex &ix, uint64_t i, bool *searched = nullptr) {
    const PhiSlot<P> *__restrict__ slots = static_cast<const PhiSlot<P> *>(ix.phi_slots);
    if (i >= ix.n) {
        // Only a toehold that wrapped below zero gets here: a match at text position 0 leaves k - 1 =
        // 2^64 - 1 (LF_w_loc, rowbowt.hpp:561).  The reference's phi is outside its domain there
        // (toehold_sa.hpp:57-59 assert); its arithmetic on the last sampled position is followed, and no
        // table is indexed with the out-of-range value.
        const PhiEnt<P> e = static_cast<const PhiEnt<P> *>(ix.phi_ent)[ix.r - 1];
        const uint64_t sum = static_cast<uint64_t>(e.base) + (i - static_cast<uint64_t>(e.pos));  // wraps like uint64_t there
        return sum % ix.n;
    }
    const uint64_t b = i >> ix.phi_shift;
    // the slot as {dprev, d0, d1, off0, off1, cnt}: 4 x P, or the 16-byte packed form at 8-byte positions (rbg_dev.h)
    uint64_t dprev, d0, d1;
    uint32_t off0, off1, cnt;
    if (sizeof(P) == 8 && ix.phi_packed) {
        const uint4 raw = load_slot(static_cast<const uint4 *>(ix.phi_slots) + b);
        const uint64_t w0 = static_cast<uint64_t>(raw.x) | (static_cast<uint64_t>(raw.y) << 32);
        const uint64_t w1 = static_cast<uint64_t>(raw.z) | (static_cast<uint64_t>(raw.w) << 32);
        constexpr uint64_t M38 = (uint64_t(1) << 38) - 1;
        dprev = w0 & M38;
        d0 = (w0 >> 38) | ((w1 & 0xFFFull) << 26);
        d1 = (w1 >> 12) & M38;
        cnt = static_cast<uint32_t>(w1 >> 62);
        off0 = cnt >= 1 ? static_cast<uint32_t>(w1 >> 50) & 63u : 0xFFu;   // (absent offsets compare as "never above")
        off1 = cnt == 2 ? static_cast<uint32_t>(w1 >> 56) & 63u : 0xFFu;
    } else {
        const PhiSlot<P> sl = load_slot(slots + b);
        const uint32_t meta = static_cast<uint32_t>(sl.meta);
        dprev = sl.dprev; d0 = sl.d0; d1 = sl.d1;
        off0 = meta & 0xFFu; off1 = (meta >> 8) & 0xFFu; cnt = (meta >> 16) & 3u;
    }
    uint64_t s;
    if (cnt == kPhiOvf) {
        if (searched) *searched = true;
        const PhiEnt<P> *__restrict__ ent = static_cast<const PhiEnt<P> *>(ix.phi_ent);
        uint64_t a = ix.phi_ord[b], z = ix.phi_ord[b + 1];
        while (z - a > 4) {
            const uint64_t mid = a + ((z - a) >> 1);
            if (static_cast<uint64_t>(ent[mid].pos) < i) a = mid + 1; else z = mid;
        }
        while (a < z && static_cast<uint64_t>(ent[a].pos) < i) ++a;
        // a == pred_.rank(i); circular predecessor (sparse_sd_vector.hpp:141-143)
        const PhiEnt<P> e = ent[a ? a - 1 : ix.r - 1];
        const uint64_t j = e.pos;
        const uint64_t delta = j < i ? i - j : i + 1;  // toehold_sa.hpp:65
        s = static_cast<uint64_t>(e.base) + delta;
    } else {
        const uint32_t o = static_cast<uint32_t>(i - (b << ix.phi_shift));
        uint64_t D = dprev;
        if (o > off0) D = d0;
        if (o > off1) D = d1;
        s = D + i;  // D = (base - pos) mod n of the predecessor: base + (i - pos)
    }
    if (s >= ix.n) s -= ix.n;  // (prev_sample + delta) % n_ (toehold_sa.hpp:71); s < 2n
    return s;
}

// One lane walks one read's phi chain (toehold_sa.hpp:37-49); the chain is serial, the reads are not.  The values are staged per wave in LDS,
// kChunk steps at a time, and flushed as WINDOWS of kChunk locations on boundaries of the output array, kChunk lanes per read (rbg_device.hpp
// ChainStage: what is staged, why, and what it costs in LDS).  Eight steps = 64-byte windows at both position widths since the windows are
// aligned (profiles/r06_k3_ring_ab.txt); with unaligned segments sixteen steps were better at 4-byte positions (3.5 / 3.1 / 3.5 ms per 10 M
// reads at 8 / 16 / 32, round 3) -- a 64-byte store on a 64-byte boundary runs at 2.9 TB/s, a 128-byte one wherever it falls at 1.2.
#ifndef RBG_K3_CHUNK_U32
#define RBG_K3_CHUNK_U32 8
#endif
constexpr int kChunk = RBG_K3_CHUNK_U32;
#ifndef RBG_K3_CHUNK_U64
#define RBG_K3_CHUNK_U64 8
#endif

// STATS = the instrumented instantiation (rbg_locate_fill_stats_dev): the same walk plus the LocateStat sums.
// OUT = the width a location is stored at: uint64_t (the API's, toehold_sa.hpp:37-49 fills a vector<uint64_t>) or, for
// device pipelines on an index with 4-byte positions, uint32_t (rbg_locate_fill_dev32: half the write requests; the low
// 32 bits of the same values, so a toehold that wrapped below zero reads 0xFFFFFFFF).
// SUB = `sub` is subtracted from every location (locate_from_longest_seed, rowbowt.hpp:681-683).
template <typename P, bool STATS = false, typename OUT = uint64_t, bool SUB = false, bool HI8 = false, int CH = (sizeof(P) == 8 ? RBG_K3_CHUNK_U64 : kChunk), bool RING = true>
__global__ __launch_bounds__(256, (chain_stage_waves<ChainStage<P, CH, SUB, RING, HI8>>())) void k_locate_fill(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                        const uint64_t *__restrict__ hi, const uint64_t *__restrict__ k,
                                                        const uint64_t N, const uint64_t max_hits,
                                                        const uint64_t *__restrict__ loc_off, OUT *__restrict__ locs,
                                                        const uint64_t *__restrict__ sub, const uint32_t *__restrict__ order,
                                                        const uint64_t *__restrict__ skeys,
                                                        unsigned long long *__restrict__ stats = nullptr) {
    using Stage = ChainStage<P, CH, SUB, RING, HI8>;
    __shared__ Stage S;
    const uint64_t out_elem0 = reinterpret_cast<uintptr_t>(locs) / sizeof(OUT);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    unsigned long long c_locs = 0;
    unsigned long long st_phi = 0, st_ovf = 0, st_chains = 0;  // STATS only
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * blockDim.x + wv * kWave; base < N; base += stride) {
        // `order` (optional) lists the reads by toehold text position: neighbouring lanes then walk
        // neighbouring phi slots (same DRAM rows / L2 lines) for the whole chain, because the chains of
        // reads from nearby loci visit the haplotypes in the same order.  Results land at loc_off[i]
        // whatever the processing order.
        const uint64_t j = base + lane;
        uint64_t i = j;
        if (order && j < N) i = order[j];
        uint64_t occ = 0, k1 = 0, dst = 0;
        if (i < N) {
            dst = loc_off[i];
            if (skeys) {
                // ordered walk: the toehold travels with the sort (sequential read) and the count is the
                // planned one, loc_off[i+1] - loc_off[i] = min(occ, max_hits): one random 64-byte sector
                // per read instead of four (lo, hi, k, loc_off)
                if (ix.order_docs) k1 = locus_toehold(ix, skeys[j], k, i);
                else if (sizeof(P) == 4) { const uint32_t k32 = reinterpret_cast<const uint32_t *>(skeys)[j]; k1 = k32 == 0xFFFFFFFFu ? k[i] : k32; }   // (k_keys32)
                else k1 = skeys[j];
                occ = loc_off[i + 1] - dst;
            } else {
                const uint64_t l = lo[i], h = hi[i];
                occ = h >= l ? h - l + 1 : 0;  // toehold_sa.hpp:38-39
                if (occ > max_hits) occ = max_hits;
                k1 = k[i];
            }
        }
        const uint64_t minus = (SUB && i < N) ? sub[i] : 0;
        // the window grid of this read: its first location sits a elements past a CH-element boundary of the output array (RING; else a = 0)
        const uint32_t a = (RING && occ) ? static_cast<uint32_t>((out_elem0 + dst) & static_cast<uint64_t>(CH - 1)) : 0u;
        S.dst[wv][lane] = dst - a;   // (wraps for a read at the very start of a misaligned array; + v >= a brings it back)
        if (SUB) S.minus[wv][lane] = minus;
        // the toehold itself is not a text position when it wrapped (2^64 - 1): at 4-byte positions its owner stores that location (ChainStage)
        const bool off_text = Stage::kSentinel && k1 >= ix.n;
        if (off_text && occ) locs[dst] = static_cast<OUT>(k1 - minus);
        c_locs += occ;
        if (STATS && occ) st_chains += 1;
        uint64_t wmax = occ + a;      // the chain's extent in virtual columns
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(wmax, o, kWave);
            wmax = other > wmax ? other : wmax;
        }
        for (uint64_t t0 = 0; t0 < wmax; t0 += CH) {
            const uint32_t cnt = chain_round_count<CH>(occ, t0);
            S.bounds[wv][lane] = chain_window_bounds<CH>(occ, a, t0);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                if (static_cast<uint32_t>(e) < cnt) {
                    if (e || t0) {   // every location but a read's first is phi of the one before (toehold_sa.hpp:44)
                        if (STATS) {
                            bool searched = false;
                            k1 = phi_step<P>(ix, k1, &searched);
                            st_phi += 1;
                            st_ovf += searched ? 1 : 0;
                        } else {
                            k1 = phi_step<P>(ix, k1);
                        }
                    }
                    chain_put(S, wv, lane, a + static_cast<uint32_t>(t0) + e, k1, e == 0 && t0 == 0 && off_text);
                }
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
        st_ovf = wave_sum(st_ovf);
        st_chains = wave_sum(st_chains);
        if (lane == 0) {
            if (st_phi) atomicAdd(&stats[kLsPhiSteps], st_phi);
            if (st_ovf) atomicAdd(&stats[kLsPhiOvf], st_ovf);
            if (st_chains) atomicAdd(&stats[kLsChains], st_chains);
            if (c_locs) atomicAdd(&stats[kLsLocs], c_locs);
        }
    }
}

}  // namespace

// ---- chain ordering for locate: permutation of the reads by toehold value (radix sort) ----------
namespace {
__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ v, const uint64_t N) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) v[i] = static_cast<uint32_t>(i);
}
struct OrderWs {
    size_t perm, iota, keys, keys_in, sort, sort_bytes, total;   // keys_in: the locus keys the sort reads (the absolute order sorts the toeholds where they are)
};
OrderWs order_layout(uint64_t N) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    OrderWs w{};
    w.perm = 0;
    w.iota = up(w.perm + N * 4);
    w.keys = up(w.iota + N * 4);
    w.keys_in = up(w.keys + N * 8);
    w.sort = up(w.keys_in + N * 8);
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, static_cast<const uint64_t *>(nullptr), static_cast<uint64_t *>(nullptr),
                                             static_cast<const uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr),
                                             static_cast<int64_t>(N ? N : 1), 0, 64);
    w.sort_bytes = bytes;
    w.total = up(w.sort + bytes) + 256;
    return w;
}
}  // namespace

size_t locate_order_ws_bytes(uint64_t N) { return order_layout(N).total; }

int launch_locate_order(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *k, uint64_t N, void *ws, size_t ws_bytes,
                        void *stream) {
    if (N == 0) return 0;
    if (N >= 0xFFFFFFFFull) return static_cast<int>(hipErrorInvalidValue);  // permutation entries are 32-bit
    const OrderWs w = order_layout(N);
    if (ws_bytes < w.total || (reinterpret_cast<uintptr_t>(ws) & 255)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(ws);
    uint32_t *perm = reinterpret_cast<uint32_t *>(base + w.perm);
    uint32_t *iota = reinterpret_cast<uint32_t *>(base + w.iota);
    uint64_t *keys = reinterpret_cast<uint64_t *>(base + w.keys);
    if (ix.order_docs) {   // locus order: sort the keys {offset >> low, document, ...} on the bits above `low`
        uint64_t *keys_in = reinterpret_cast<uint64_t *>(base + w.keys_in);
        hipLaunchKernelGGL(k_locus_keys, dim3(grid_for(cfg, N)), dim3(256), 0, st, ix, k, N, keys_in, iota);
        int rc0 = static_cast<int>(hipGetLastError());
        if (rc0) return rc0;
        size_t bytes0 = w.sort_bytes;
        // (all 64 bits when a toehold may have wrapped: its all-ones key sorts last either way; the bits above obits + dbits are zero otherwise)
        const int lo_bit = static_cast<int>(ix.order_lowbits), hi_bit = static_cast<int>(ix.order_obits + ix.order_dbits) + 1;
        return static_cast<int>(hipcub::DeviceRadixSort::SortPairs(base + w.sort, bytes0, keys_in, keys, iota, perm, static_cast<int64_t>(N), lo_bit, hi_bit < 64 ? hi_bit : 64, st));
    }
    int end_bit = 1;
    while (end_bit < 64 && (ix.n >> end_bit)) ++end_bit;  // toeholds are text positions < n
    if (ix.pos_bytes == 4) {   // 32-bit keys (k_keys32): the key pass also makes the identity permutation
        uint32_t *k32_in = reinterpret_cast<uint32_t *>(base + w.keys_in), *k32_out = reinterpret_cast<uint32_t *>(base + w.keys);
        // (computing the keys where the sort reads them -- rocprim iterators, no key pass -- reads the 8-byte toeholds twice: 0.331 -> 0.337 ms, profiles/r06_experiments.txt)
        hipLaunchKernelGGL(k_keys32, dim3(grid_for(cfg, N)), dim3(256), 0, st, ix, k, N, k32_in, iota);
        int rc32 = static_cast<int>(hipGetLastError());
        if (rc32) return rc32;
        int b0 = static_cast<int>(ix.phi_shift) + 2;
        if (end_bit - b0 < 8) b0 = 0;
        size_t bytes32 = w.sort_bytes;   // (sized for 64-bit keys: enough for 32-bit ones)
        return static_cast<int>(hipcub::DeviceRadixSort::SortPairs(base + w.sort, bytes32, k32_in, k32_out, iota, perm, static_cast<int64_t>(N), b0, end_bit < 32 ? end_bit : 32, st));
    }
    hipLaunchKernelGGL(k_iota, dim3(grid_for(cfg, N)), dim3(256), 0, st, iota, N);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    // the order only has to bring chains of nearby text positions together: the low bits (positions
    // inside one 64-byte line of phi slots) need no sorting, which saves a radix pass
    int begin_bit = static_cast<int>(ix.phi_shift) + 2;
    if (end_bit - begin_bit < 8) begin_bit = 0;
    // (sorting on fewer, higher bits costs K3 what it saves the sort: profiles/r06_experiments.txt)
    size_t bytes = w.sort_bytes;
    return static_cast<int>(hipcub::DeviceRadixSort::SortPairs(base + w.sort, bytes, k, keys, iota, perm, static_cast<int64_t>(N),
                                                               begin_bit, end_bit, st));
}

size_t scan_tmp_bytes(uint64_t N) {
    size_t bytes = 0;
    uint64_t *p = nullptr;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, p, p, static_cast<int64_t>(N ? N : 1));
    return bytes + 256;
}

// occ(i) = min(hi - lo + 1, max_hits) or 0 (toehold_sa.hpp:38-39), computed where the scan reads it: the plan is ONE pass over (lo, hi) -- until
// round 6 a kernel wrote the counts and the scan read them back (0.126 -> 0.09 ms per 10 M reads)
struct OccOf {
    const uint64_t *lo, *hi;
    uint64_t max_hits;
    __host__ __device__ __forceinline__ uint64_t operator()(const uint64_t i) const {
        const uint64_t l = lo[i], h = hi[i];
        uint64_t occ = h >= l ? h - l + 1 : 0;
        return occ > max_hits ? max_hits : occ;
    }
};
__global__ void k_zero_u64(uint64_t *p) { *p = 0; }

int launch_locate_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                       uint64_t max_hits, uint64_t *loc_off, void *tmp, size_t tmp_bytes, void *stream) {
    (void)ix; (void)cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(1), 0, st, loc_off);
    int rc = static_cast<int>(hipGetLastError());
    if (rc || N == 0) return rc;
    hipcub::TransformInputIterator<uint64_t, OccOf, hipcub::CountingInputIterator<uint64_t>> in(hipcub::CountingInputIterator<uint64_t>(0), OccOf{lo, hi, max_hits});
    size_t need = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, need, in, loc_off + 1, static_cast<int64_t>(N));
    if (need > tmp_bytes) return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipcub::DeviceScan::InclusiveSum(tmp, need, in, loc_off + 1, static_cast<int64_t>(N), st));
}

int launch_locate_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi,
                       const uint64_t *k, uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs,
                       const uint64_t *sub, const void *order, void *stream, unsigned long long *stats, uint32_t *locs32) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (the grid: up to 32 workgroups per CU looping over the chains; 8 / 16 / 64 per CU and one workgroup per 256 chains without a loop measured
    //  level or worse within the processes' spread: profiles/r06_experiments.txt)
    const dim3 grid(grid_for(cfg, N)), block(cfg.block_threads);
    // the order workspace also holds the toeholds in sorted order (launch_locate_order's key output)
    const uint64_t *skeys = order ? reinterpret_cast<const uint64_t *>(static_cast<const char *>(order) + order_layout(N).keys) : nullptr;
    const uint32_t *perm = static_cast<const uint32_t *>(order);
    // (Line-aligned flush segments -- a lane waiting `dst mod CH` columns before its first step so that every flush writes whole lines -- lost twice
    //  and were retired with the round-3 staging: on the bench index in round 3, and at r = 1.2e8 in round 6 (11.1 -> 14.9 ms, profiles/
    //  r06_k3_align_ab.txt): lanes that wait a different number of columns fall out of PHASE, and chains of one locus walking in phase --
    //  neighbouring lanes asking for the same phi sector in the same instruction -- is what K3's speed rests on.)
    if ((stats || locs32) && sub) return static_cast<int>(hipErrorInvalidValue);
    if (ix.layout == 2 && !ix.phi_slots) {  // run-indexed layout (k_runs.hip); with phi slots (RBG_OPT_RUN_PHI) the slot kernels below answer its phi
        return launch_locate_fill_runs(ix, cfg, lo, hi, k, N, max_hits, loc_off, locs, sub, order, skeys, stream, stats, locs32);
    }
    // (rbg_device.hpp ChainStage: a value as its low word + one high byte; RBG_K3_HI8=0 -- tests -- stages whole 8-byte values at any n)
    const bool hi8 = ix.pos_bytes == 8 && ix.n < kChainHi8Limit && chain_hi8_enabled();
#define RBG_LAUNCH_K3(PT, STS, OUT, SB, DST) \
    do { \
        if (sizeof(PT) == 8 && hi8) hipLaunchKernelGGL((k_locate_fill<PT, STS, OUT, SB, sizeof(PT) == 8>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, DST, sub, perm, skeys, stats); \
        else hipLaunchKernelGGL((k_locate_fill<PT, STS, OUT, SB, false>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, DST, sub, perm, skeys, stats); \
    } while (0)
    if (locs32) {   // 4-byte locations (4-byte positions only; the caller checked)
        if (ix.pos_bytes != 4) return static_cast<int>(hipErrorInvalidValue);
        RBG_LAUNCH_K3(uint32_t, false, uint32_t, false, locs32);
    } else if (stats) {
        if (ix.pos_bytes == 4) RBG_LAUNCH_K3(uint32_t, true, uint64_t, false, locs); else RBG_LAUNCH_K3(uint64_t, true, uint64_t, false, locs);
    } else if (sub) {
        if (ix.pos_bytes == 4) RBG_LAUNCH_K3(uint32_t, false, uint64_t, true, locs); else RBG_LAUNCH_K3(uint64_t, false, uint64_t, true, locs);
    } else {
        if (ix.pos_bytes == 4) RBG_LAUNCH_K3(uint32_t, false, uint64_t, false, locs); else RBG_LAUNCH_K3(uint64_t, false, uint64_t, false, locs);
    }
#undef RBG_LAUNCH_K3
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
