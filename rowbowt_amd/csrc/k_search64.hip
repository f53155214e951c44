// k_search64.hip -- K1/K2 over 64-BYTE rank slots (rbg_dev.h RankSlot64; RBG_OPT_SLOT_BYTES = 64).
//
// RowBowt::find_range (rowbowt.hpp:121-131) / find_range_w_toehold (:169-184), the same k-mer steps as k_search.hip's
// k_find_range, but a step's slot is a whole 64-byte sector: what the fabric delivers for a 16-byte gather anyway
// (profiles/pmc_traffic.json: padding ratio 3.6), spent on 1024 rows instead of 256 -- fewer steps whose two positions
// straddle two buckets, fourteen inline runs, the run ordinal and the predecessor run's sample inline (rbg_dev.h).
// One lane still walks one read.  What is cooperative is only the FETCH: a lane reading its own 64 bytes with four
// 16-byte loads makes four requests (19.8 G records/s, tools/slot64_probe.hip), a QUAD reading the four quarters of one
// lane's record with one load instruction makes one (47-48 G records/s, the rate of 16-byte gathers).  So a step runs
// four rounds -- in round J the quad's lanes fetch the quarters of lane J's slot -- and the sixteen words are then handed
// to their owners by DPP quad permutes (a 4 x 4 transpose of uint4 registers across the quad: VALU moves, no LDS); the
// rank itself is decoded by the owner from its own registers.  The lanes of a wave therefore step together
// (`while (ballot(...))`), like the run-indexed kernels.
#include "rbg_device.hpp"

namespace rbg {
namespace {

template <int J> __device__ __forceinline__ uint32_t q_get(uint32_t v) {   // lane J of this lane's quad
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), J * 0x55, 0xF, 0xF, false));
}
template <int J> __device__ __forceinline__ uint64_t q_get(uint64_t v) {
    return (static_cast<uint64_t>(q_get<J>(static_cast<uint32_t>(v >> 32))) << 32) | q_get<J>(static_cast<uint32_t>(v));
}

// the quarter `sub` of lane J's record, when that lane wants one (addr != 0)
template <int J>
__device__ __forceinline__ u32x4 q_quarter(const uint64_t addr, const uint32_t sub) {
    const uint64_t oa = q_get<J>(addr);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (oa) v = *as_global<u32x4>(reinterpret_cast<const void *>(oa + 16ull * sub));
    return v;
}
// word k of quarter S of MY record: lane S of the quad holds it in the register set of round `sub`
template <int S>
__device__ __forceinline__ uint32_t q_word(const uint32_t sub, const uint32_t r0, const uint32_t r1, const uint32_t r2, const uint32_t r3) {
    const uint32_t a = q_get<S>(r0), b = q_get<S>(r1), c = q_get<S>(r2), d = q_get<S>(r3);
    return sub == 0 ? a : sub == 1 ? b : sub == 2 ? c : d;
}
#define RBG_Q_ROW(S, k, field) w[4 * S + k] = q_word<S>(sub, q.q0.field, q.q1.field, q.q2.field, q.q3.field)
// A fetch in two halves, so that the requests of BOTH slots of a step (lo's bucket and hi + 1's) are in flight before
// the first wait: issue (every lane of the wave calls; lanes with addr != 0 want the 64 bytes at addr) ...
struct Quarters { u32x4 q0, q1, q2, q3; };
__device__ __forceinline__ Quarters issue_slot64(const uint64_t addr) {
    const uint32_t sub = threadIdx.x & 3u;
    return Quarters{q_quarter<0>(addr, sub), q_quarter<1>(addr, sub), q_quarter<2>(addr, sub), q_quarter<3>(addr, sub)};
}
// ... and hand the sixteen words to their owners
__device__ __forceinline__ void take_slot64(const Quarters &q, uint32_t (&w)[16]) {
    const uint32_t sub = threadIdx.x & 3u;
    RBG_Q_ROW(0, 0, x); RBG_Q_ROW(0, 1, y); RBG_Q_ROW(0, 2, z); RBG_Q_ROW(0, 3, w);
    RBG_Q_ROW(1, 0, x); RBG_Q_ROW(1, 1, y); RBG_Q_ROW(1, 2, z); RBG_Q_ROW(1, 3, w);
    RBG_Q_ROW(2, 0, x); RBG_Q_ROW(2, 1, y); RBG_Q_ROW(2, 2, z); RBG_Q_ROW(2, 3, w);
    RBG_Q_ROW(3, 0, x); RBG_Q_ROW(3, 1, y); RBG_Q_ROW(3, 2, z); RBG_Q_ROW(3, 3, w);
}
#undef RBG_Q_ROW

// a[T..15] = keep ? a : b, with constant indices (a loop variable as index sent the arrays to scratch)
template <int T>
__device__ __forceinline__ void pick16(const bool keep, uint32_t (&a)[16], const uint32_t (&b)[16]) {
    if constexpr (T < 16) {
        a[T] = keep ? a[T] : b[T];
        pick16<T + 1>(keep, a, b);
    }
}

// STATS: the instrumented instantiation (rbg_find_range_stats_dev): [kStSlots] 64-byte slots fetched, [kStDense] 4-byte
// dense-table rows, [kStResample] re-samples that needed their one gather, the rest as in k_search.hip
template <typename P, bool TOEHOLD, bool USE_FTAB, bool STATS>
__global__ __launch_bounds__(512, 4) void k_find_range64(const DevIndex ix, const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                                         const uint64_t N, uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                         uint64_t *__restrict__ ss_out, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ nsel,
                                                         unsigned long long *__restrict__ stats) {
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_lut2[256];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    DevSym *s_tab = reinterpret_cast<DevSym *>(s_dyn);
    const uint64_t Neff = sel ? static_cast<uint64_t>(*nsel) : N;   // sel: only the reads the packed path hands back
    if (Neff == 0) return;
    stage_tables(ix, s_tab, s_lut, s_lut2, true);
    const uint32_t M = ix.nmajor, ksteps = ix.kmer_steps;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    unsigned long long c_occ = 0, c_reads = 0;
    uint32_t c_matched = 0;
    unsigned long long st[kStatSearchN] = {0, 0, 0, 0, 0, 0, 0, 0};   // STATS only
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t wave_first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + (threadIdx.x & ~(kWave - 1));
    for (uint64_t base = wave_first; base < Neff; base += stride) {   // the lanes of a wave iterate together
        const bool valid = base + lane < Neff;
        const uint64_t i = (sel && valid) ? static_cast<uint64_t>(sel[base + lane]) : base + lane;
        uint64_t beg = 0, p = 0;
        if (valid) { beg = off[i]; p = off[i + 1]; }
        const uint64_t p_end = p;
        uint64_t p_min = p;
        uint64_t lo = 0, hi = ix.n - 1;                       // full_range(), rowbowt.hpp:115-118
        uint64_t k = TOEHOLD ? ix.last_run_sample : 0;
        bool alive = valid;
        // deferred toehold re-sample (k_search.hip): only the last one of a read is ever used; here it is ONE gather -- the
        // sample of run pend_run of table pend_tab -- or none at all when the slot carried it (Rank64Aux::psamp)
        bool pend = false;
        uint32_t pend_tab = 0, pend_run = 0;
        ByteCursor rd{reinterpret_cast<const uint4 *>(seqs), ~uint64_t(0), make_uint4(0, 0, 0, 0)};
        if (USE_FTAB && valid && ix.ftab_k && p - beg >= ix.ftab_k) {   // rowbowt.hpp:124-125, :745-758 (k_search.hip)
            uint64_t idx = 0, pw = 1;
            bool all_major = true;
            for (uint32_t t = 1; t <= ix.ftab_k; ++t) {
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
        while (__ballot(alive && p > beg)) {                   // right-to-left over the reads (rowbowt.hpp:127-129, :175-181)
            bool stepping = alive && p > beg;
            uint32_t adv = 1, idx = 0;
            uint64_t slots_at = 0, F = 0;
            uint32_t bs = 0;
            if (stepping) {
                --p;
                const uint32_t c = rd.at(p);
                if (STATS && p < p_min) p_min = p;
                const uint32_t m0 = s_lut2[c];
                if (m0 != 0xFFu) {   // the longest run of k-mer symbols among the next five (k_search.hip)
                    uint32_t acc = m0, pw = M;
#pragma unroll 1
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
                const DevSym *rec = nullptr;
                if (adv == 1) {
                    const uint32_t slot = s_lut[c];
                    if (slot == 0xFFu) { alive = false; stepping = false; }   // symbol absent: f_[c] >= f_[c+1], rowbowt.hpp:76
                    else if (slot < static_cast<uint32_t>(kLdsSyms)) { idx = slot; rec = s_tab + slot; }
                    else { idx = kHbmRec | slot; rec = ix.syms + slot; }
                } else {
                    rec = s_tab + idx;
                }
                if (stepping) {
                    slots_at = reinterpret_cast<uint64_t>(rec->slots);
                    F = rec->F;
                    bs = rec->shift + kSlot64Extra;
                }
            }
            // both ranks of the step (rowbowt.hpp:79,83): the slot of lo's bucket, and of hi + 1's when it is another one
            const uint64_t q0 = lo, q1 = hi + 1;
            const uint64_t bl = q0 >> bs, bh = q1 >> bs;
            uint32_t w[16], wh[16];
            const bool two = stepping && bh != bl;
            const Quarters ql = issue_slot64(stepping ? slots_at + bl * sizeof(RankSlot64) : 0);
            const Quarters qh = issue_slot64(two ? slots_at + bh * sizeof(RankSlot64) : 0);   // (nearly every wave has such a lane)
            take_slot64(ql, w);
            take_slot64(qh, wh);
            pick16<0>(two, wh, w);
            // the inline runs are decoded up to the wave's largest count (dense buckets count as none)
            uint32_t mc = 0;
            if (stepping) {
                const uint32_t c0 = w[1] >> 28, c1 = wh[1] >> 28;
                mc = (c0 == kSlot64Ovf ? 0u : c0) > (c1 == kSlot64Ovf ? 0u : c1) ? (c0 == kSlot64Ovf ? 0u : c0) : (c1 == kSlot64Ovf ? 0u : c1);
            }
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(mc, o, kWave); mc = other > mc ? other : mc; }
            mc = __builtin_amdgcn_readfirstlane(mc);
            uint64_t c_before = 0, c_upto = 0;
            Rank64Aux a0, a1;
            a1.nbefore = 0; a1.ord = 0; a1.psamp = 0; a1.inside = false; a1.dense = false;
            if (stepping) {
                c_before = rank_in_slot64(w, bs, bl, q0, ix.dense, &a0, mc);
                c_upto = rank_in_slot64(wh, bs, bh, q1, ix.dense, &a1, mc);
                if (STATS) { st[kStSteps] += 1; st[kStSlots] += two ? 2 : 1; st[kStSymbols] += adv; st[kStDense] += (a0.dense ? 1 : 0) + (a1.dense ? 1 : 0); }
            }
            if (stepping) {
                const uint64_t c_inside = c_upto - c_before;
                if (c_inside == 0) {                            // rowbowt.hpp:85 (whichever of the nested steps emptied the range)
                    alive = false;
                } else {
                    if (TOEHOLD) {                              // LF_w_loc, rowbowt.hpp:559-566, `adv` times nested
                        if (a1.inside) k = k - adv;
                        else if (a1.nbefore == 0) { k = a1.psamp; pend = false; }   // the run before the bucket: its sample came with the slot
                        else { pend = true; pend_tab = idx; pend_run = a1.ord + a1.nbefore - 1; k = 0; }
                    }
                    lo = F + c_before;                          // rowbowt.hpp:86
                    hi = lo + c_inside - 1;                     // rowbowt.hpp:87
                    p -= adv - 1;                               // the left neighbours are consumed too
                }
            }
        }
        if (TOEHOLD && alive && pend) {
            const DevSym *rec = (pend_tab & kHbmRec) ? ix.syms + (pend_tab & ~kHbmRec) : s_tab + pend_tab;
            k += static_cast<uint64_t>(as_global<P>(rec->samp)[pend_run]);
            if (STATS) st[kStResample] += 1;
        }
        if (STATS && p_end > p_min) st[kStChunks] += ((p_end - 1) >> 4) - (p_min >> 4) + 1;
        if (!alive) { lo = 1; hi = 0; k = 0; }                 // {1,0}; LFData::clear rowbowt.hpp:153-159
        if (valid) {
            lo_out[i] = lo;
            hi_out[i] = hi;
            if (TOEHOLD) ss_out[i] = k;
            c_reads += 1;
            if (alive) { c_matched += 1; c_occ += hi - lo + 1; }
        }
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

}  // namespace

int launch_find_range64(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo, uint64_t *hi,
                        uint64_t *ssamp, const uint32_t *sel, const uint32_t *nsel, unsigned long long *stats, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = static_cast<size_t>(ix.kmer_steps >= 5 ? kTab5 : kTabMax) * sizeof(DevSym);
    LaunchCfg c = cfg;   // 512-thread workgroups, four waves per SIMD (two workgroups per CU share its LDS)
    c.block_threads = 512;
    c.max_blocks = cfg.max_blocks > 0 ? std::max(1, cfg.max_blocks / 2) : 256 * 16;
    const dim3 grid(sel ? std::min(grid_for(c, N), 256) : grid_for(c, N)), block(512);
    auto raise = [&](const void *kern) {
        if (lds <= 48 * 1024) return;
        static std::mutex mu;
        static std::set<std::pair<int, const void *>> raised;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(mu);
        if (raised.insert(std::make_pair(dev, kern)).second) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    };
#define RBG_L64(PT, TOE, FT, STS)                                                                                  \
    do {                                                                                                          \
        auto kern = k_find_range64<PT, TOE, FT, STS>;                                                             \
        raise(reinterpret_cast<const void *>(kern));                                                              \
        hipLaunchKernelGGL(kern, grid, block, lds, st, ix, seqs, off, N, lo, hi, ssamp, sel, nsel, stats);        \
    } while (0)
#define RBG_L64B(PT, TOE)                                                                                          \
    do {                                                                                                          \
        if (stats) { if (ix.ftab_k) RBG_L64(PT, TOE, true, true); else RBG_L64(PT, TOE, false, true); }           \
        else if (ix.ftab_k) RBG_L64(PT, TOE, true, false);                                                        \
        else RBG_L64(PT, TOE, false, false);                                                                      \
    } while (0)
    if (ix.pos_bytes == 4) { if (ssamp) RBG_L64B(uint32_t, true); else RBG_L64B(uint32_t, false); }
    else { if (ssamp) RBG_L64B(uint64_t, true); else RBG_L64B(uint64_t, false); }
#undef RBG_L64B
#undef RBG_L64
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
