// rbg_kernels.hip -- gfx950 (MI355X, wave64) kernels for the rb_align hot path.
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
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
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

template <typename SlotT>
__device__ __forceinline__ SlotT load_slot(const SlotT *p) {
    SlotT s;
    if constexpr (sizeof(SlotT) == 16) {
        const u32x4 t = *reinterpret_cast<const u32x4 *>(p);
        __builtin_memcpy(&s, &t, 16);
    } else {
        static_assert(sizeof(SlotT) == 32, "slot is 4 words of 4 or 8 bytes");
        const u64x2 a = *reinterpret_cast<const u64x2 *>(p);
        const u64x2 b = *(reinterpret_cast<const u64x2 *>(p) + 1);
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
};

// # runs of the symbol that start before i, searched in the run list of bucket b (overflow buckets without
// a dense table only)
template <typename P>
__device__ __forceinline__ uint64_t search_runs(const RunEnt<P> *__restrict__ ent, uint64_t a, uint64_t z, uint64_t i) {
    while (z - a > 4) {
        const uint64_t mid = a + ((z - a) >> 1);
        if (static_cast<uint64_t>(ent[mid].start) < i) a = mid + 1; else z = mid;
    }
    while (a < z && static_cast<uint64_t>(ent[a].start) < i) ++a;
    return a;
}
template <typename P>
__device__ __forceinline__ uint64_t runs_before(const DevSym &S, uint64_t b, uint64_t i) {
    return search_runs<P>(static_cast<const RunEnt<P> *>(S.ent), S.ord[b], S.ord[b + 1], i);
}

template <typename P>
__device__ __forceinline__ uint64_t rank_in_slot(const DevSym &S, const RankSlot &sl, uint64_t b, uint64_t i,
                                                 const uint8_t *__restrict__ dense, RankAux *aux) {
    const uint32_t w1 = sl.w1, w2 = sl.w2, w3 = sl.w3;
    const bool wide = S.shift > kMaxNarrowShift;
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
            const uint32_t e = reinterpret_cast<const uint16_t *>(dense + (static_cast<uint64_t>(w2) << 4))[o];
            aux->ovf = false;
            aux->nbefore = e >> 8;
            aux->inside = (e >> 8) == 255u;
            return (static_cast<uint64_t>(sl.r0) | (static_cast<uint64_t>(w3 >> 16) << 32)) + (e & 0xFFu);
        }
        aux->ovf = true;
        aux->nbefore = o;
        const RunEnt<P> *__restrict__ ent = static_cast<const RunEnt<P> *>(S.ent);
        const uint64_t a = runs_before<P>(S, b, i);
        if (a == 0) { aux->inside = false; return 0; }
        const RunEnt<P> e = ent[a - 1];
        const uint64_t len = static_cast<uint64_t>(ent[a].cum) - static_cast<uint64_t>(e.cum);
        const uint64_t d = i - static_cast<uint64_t>(e.start);
        aux->inside = d <= len;
        return static_cast<uint64_t>(e.cum) + (d < len ? d : len);
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

// both ranks of one LF step (rowbowt.hpp:79,83).  lo and hi+1 usually share a bucket late in the
// search (the range has narrowed to a few dozen rows), so the step is ONE 4-word load.
template <typename P>
__device__ __forceinline__ void rank_pair(const DevSym &S, const uint8_t *__restrict__ dense, uint64_t lo, uint64_t hi1,
                                          uint64_t *c_before, uint64_t *c_upto, uint64_t *bh_out, RankAux *qaux) {
    const RankSlot *__restrict__ slots = static_cast<const RankSlot *>(S.slots);
    const uint64_t bl = lo >> S.shift, bh = hi1 >> S.shift;
    const RankSlot sl = load_slot(slots + bl);
    RankSlot sh = sl;
    if (bh != bl) sh = load_slot(slots + bh);
    RankAux paux;
    *c_before = rank_in_slot<P>(S, sl, bl, lo, dense, &paux);
    *c_upto = rank_in_slot<P>(S, sh, bh, hi1, dense, qaux);
    *bh_out = bh;
}

// ordinal of the last run of the symbol that starts before the position a RankAux describes
template <typename P>
__device__ __forceinline__ uint64_t pred_run(const DevSym &S, uint64_t b, bool ovf, uint32_t v) {
    return (ovf ? runs_before<P>(S, b, (b << S.shift) + v) : static_cast<uint64_t>(S.ord[b]) + v) - 1;
}

// samples_last_ of the last run of the symbol that starts before i (LF_w_loc, rowbowt.hpp:563-566);
// only taken when position i-1 does not hold the symbol, which is the rare case.
template <typename P>
__device__ __forceinline__ uint64_t pred_sample(const DevSym &S, uint64_t b, const RankAux &aux) {
    return static_cast<uint64_t>(static_cast<const P *>(S.samp)[pred_run<P>(S, b, aux.ovf, aux.nbefore)]);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// ---- K1 / K2 ------------------------------------------------------------------------------------
// per-lane cursor over the read bytes, fetched as aligned 16-byte chunks while walking right to left
struct ByteCursor {
    const uint4 *__restrict__ chunks;
    uint64_t cur_ci;
    uint4 w;
    __device__ __forceinline__ uint32_t at(uint64_t p) {
        const uint64_t ci = p >> 4;
        if (ci != cur_ci) { w = chunks[ci]; cur_ci = ci; }
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

// The record table lives in dynamic LDS: kTabMax records (17 KB) up to 4-mer steps, kTab5 (66 KB) with the
// 5-mer level, then launched as 1024-thread workgroups so that two of them still give 8 waves per SIMD.
template <typename P, bool TOEHOLD, bool USE_FTAB>
__global__ __launch_bounds__(1024, 8) void k_find_range(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                    const uint64_t *__restrict__ off, const uint64_t N,
                                                    uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out,
                                                    uint64_t *__restrict__ ss_out, const uint32_t *__restrict__ sel,
                                                    const uint32_t *__restrict__ nsel) {
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
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t first_ = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint64_t j_ = first_; j_ < Neff; j_ += stride) {
        uint64_t beg, p;
        {
            const uint64_t i0 = sel ? static_cast<uint64_t>(sel[j_]) : j_;
            beg = off[i0];
            p = off[i0 + 1];
        }
        uint64_t lo = 0, hi = ix.n - 1;  // full_range(), rowbowt.hpp:115-118
        uint64_t k = TOEHOLD ? ix.last_run_sample : 0;
        // The toehold only flows forward through "k - adv" (row hi carries the symbol); a step that
        // re-samples overwrites it.  While the range is still wide almost every step re-samples
        // (bwt[hi] is a random symbol), so the two gathers of a re-sample (run ordinal, sample) are
        // deferred until a later step or the end of the read actually needs the value.
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
            return static_cast<uint64_t>(static_cast<const P *>(samp)[j]);
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
            if (all_major && ftab_lookup<P>(ix, idx, flo, fhi2, fk)) {
                lo = flo; hi = fhi2;
                if (TOEHOLD) k = fk;
                p -= ix.ftab_k;
                if (hi < lo) { alive = false; p = beg; }
            }
        }
        // one LF step (or several nested ones) through the record S; false = range emptied
        auto step = [&](const DevSym &S, uint32_t adv, uint32_t tab) -> bool {
            RankAux q;
            uint64_t c_before, c_upto, bh;
            rank_pair<P>(S, ix.dense, lo, hi + 1, &c_before, &c_upto, &bh, &q);      // rowbowt.hpp:79,83
            const uint64_t c_inside = c_upto - c_before;
            if (c_inside == 0) return false;                               // rowbowt.hpp:85
            if (TOEHOLD) {
                // LF_w_loc, rowbowt.hpp:559-566.  Either hi holds the symbol (bwt_[hi]==c -> k-1 per
                // nested step), or the last run starting before hi ends before hi and its last row
                // is select(rank(hi,c)-1,c), whose run-end sample is samples_last_[run] (resp. SA-adv).
                if (q.inside) {
                    if (pend) { k = resample(); pend = false; }
                    k = k - adv;
                } else {
                    pend = true;
                    pend_tab = tab;
                    pend_b = bh;
                    pend_abs = q.ovf;
                    pend_v = q.nbefore;
                }
            }
            lo = S.F + c_before;           // rowbowt.hpp:86
            hi = lo + c_inside - 1;        // rowbowt.hpp:87
            return true;
        };
        while (p > beg) {  // right-to-left over the read (rowbowt.hpp:127-129, :175-181)
            --p;
            const uint32_t c = rd.at(p);
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
        if (TOEHOLD && alive && pend) k = resample();
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
            return static_cast<uint64_t>(static_cast<const P *>(samp)[j]);
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
                    if (pend) { k = resample(); pend = false; }
                    k = k - adv;
                } else {
                    pend = true;
                    pend_tab = tab;
                    pend_b = bh;
                    pend_abs = q.ovf;
                    pend_v = q.nbefore;
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
        if (TOEHOLD && alive && pend) k = resample();
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

// ---- K3: locate ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_occ(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                             const uint64_t N, const uint64_t max_hits, uint64_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        uint64_t occ = hi[i] >= lo[i] ? hi[i] - lo[i] + 1 : 0;  // toehold_sa.hpp:38-39
        if (occ > max_hits) occ = max_hits;
        out[i + 1] = occ;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

template <typename P>
__device__ __forceinline__ uint64_t phi_step(const DevIndex &ix, uint64_t i) {
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
    const PhiSlot<P> sl = load_slot(slots + b);
    const uint32_t meta = static_cast<uint32_t>(sl.meta);
    uint64_t s;
    if (((meta >> 16) & 3u) == kPhiOvf) {
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
        uint64_t D = sl.dprev;
        if (o > (meta & 0xFFu)) D = sl.d0;
        if (o > ((meta >> 8) & 0xFFu)) D = sl.d1;
        s = D + i;  // D = (base - pos) mod n of the predecessor: base + (i - pos)
    }
    if (s >= ix.n) s -= ix.n;  // (prev_sample + delta) % n_ (toehold_sa.hpp:71); s < 2n
    return s;
}

// One lane walks one read's phi chain (toehold_sa.hpp:37-49); the chain is serial, the reads are
// not.  The locations of a read are contiguous in `locs`, but a lane storing its own values would
// make every store instruction touch 64 different lines, and stores are gather-class requests just
// like the slot loads (tools/gather_roof.hip).  So values are staged per wave in LDS, kChunk steps
// at a time, and flushed with kChunk lanes writing one read's (8 * kChunk)-byte segment: a store
// instruction then touches a handful of lines instead of 64.  Measured per 10M reads: unordered
// chains 10.5 / 8.6 / 7.6 ms at kChunk 8 / 16 / 32 (7.3 at 32 in a later build); with the chains in
// toehold order (the default) 3.5 / 3.1 / 3.5 ms, hence 16 (staged as uint64: 39 KB of LDS per workgroup;
// staged at the position width since: 24 KB at 4-byte positions, 3.1 -> 3.0 ms).
constexpr int kChunk = 16;

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving
    // the cross-lane reads above the writes (and vice versa)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename P>
__global__ __launch_bounds__(256) void k_locate_fill(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                     const uint64_t *__restrict__ hi, const uint64_t *__restrict__ k,
                                                     const uint64_t N, const uint64_t max_hits,
                                                     const uint64_t *__restrict__ loc_off, uint64_t *__restrict__ locs,
                                                     const uint64_t *__restrict__ sub, const uint32_t *__restrict__ order,
                                                     const uint64_t *__restrict__ skeys) {
    // staged at the position width: text positions fit P, and at 4 bytes the workgroup's LDS drops from
    // 39 KB to 24 KB (6 instead of 4 waves per SIMD); the per-read offset is applied when flushing
    __shared__ P s_val[4][kWave][kChunk + 1];  // +1: keeps the per-lane rows off the same banks
    __shared__ uint64_t s_dst[4][kWave];
    __shared__ uint64_t s_occ[4][kWave];
    __shared__ uint64_t s_minus[4][kWave];
    __shared__ uint64_t s_first[4][kWave];  // the toehold itself is not a text position when it wrapped (2^64 - 1)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    unsigned long long c_locs = 0;
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
                k1 = skeys[j];
                occ = loc_off[i + 1] - dst;
            } else {
                const uint64_t l = lo[i], h = hi[i];
                occ = h >= l ? h - l + 1 : 0;  // toehold_sa.hpp:38-39
                if (occ > max_hits) occ = max_hits;
                k1 = k[i];
            }
        }
        const uint64_t minus = (sub && i < N) ? sub[i] : 0;  // locate_from_longest_seed, rowbowt.hpp:681-683
        s_dst[wv][lane] = dst;
        s_occ[wv][lane] = occ;
        s_minus[wv][lane] = minus;
        s_first[wv][lane] = k1;
        c_locs += occ;
        uint64_t wmax = occ;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(wmax, o, kWave);
            wmax = other > wmax ? other : wmax;
        }
        for (uint64_t t0 = 0; t0 < wmax; t0 += kChunk) {
#pragma unroll
            for (int e = 0; e < kChunk; ++e) {
                const uint64_t t = t0 + e;
                if (t < occ) {
                    if (t) k1 = phi_step<P>(ix, k1);  // toehold_sa.hpp:44
                    s_val[wv][lane][e] = static_cast<P>(k1);
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int pass = 0; pass < kChunk; ++pass) {  // kWave/kChunk reads per pass, kChunk lanes each
                const int s = pass * (kWave / kChunk) + lane / kChunk;
                const int e = lane & (kChunk - 1);
                const uint64_t t = t0 + e;
                if (t < s_occ[wv][s]) locs[s_dst[wv][s] + t] = (t ? static_cast<uint64_t>(s_val[wv][s][e]) : s_first[wv][s]) - s_minus[wv][s];
            }
            wave_lds_sync();
        }
        wave_lds_sync();
    }
    c_locs = wave_sum(c_locs);
    if (lane == 0 && c_locs) atomicAdd(&ix.counters[3], c_locs);
}

// ---- K4: markers --------------------------------------------------------------------------------
// runs are disjoint, ascending inclusive SA-index intervals; at_range(lo,hi) = values of all runs
// with start <= hi && end >= lo, in run order.
__device__ __forceinline__ void marker_span(const DevIndex &ix, uint64_t lo, uint64_t hi, uint64_t *first, uint64_t *last) {
    if (ix.mk_bucket) {
        if (lo >= ix.n) { *first = *last = ix.mk_nruns; return; }  // caller-supplied rows beyond the BWT: nothing
        if (hi >= ix.n) hi = ix.n - 1;
        // first run with end >= lo: the bucket table gives the first run ending at or after the start of
        // lo's bucket; the answer is at most a bucket's worth of runs further on
        uint64_t a = ix.mk_bucket[lo >> ix.mk_shift];
        while (a < ix.mk_nruns && ix.mk_end[a] < lo) ++a;
        *first = a;
        // one past the last run with start <= hi: every run before the entry of hi's bucket ends, hence
        // starts, before hi
        uint64_t z = ix.mk_bucket[hi >> ix.mk_shift];
        if (z < a) z = a;
        while (z < ix.mk_nruns && ix.mk_start[z] <= hi) ++z;
        *last = z;
        return;
    }
    uint64_t a = 0, z = ix.mk_nruns;
    while (a < z) { const uint64_t m = a + ((z - a) >> 1); if (ix.mk_end[m] < lo) a = m + 1; else z = m; }
    *first = a;  // first run with end >= lo
    a = 0; z = ix.mk_nruns;
    while (a < z) { const uint64_t m = a + ((z - a) >> 1); if (ix.mk_start[m] <= hi) a = m + 1; else z = m; }
    *last = a;   // one past the last run with start <= hi
}

__global__ __launch_bounds__(256) void k_markers_count(const DevIndex ix, const uint64_t *__restrict__ lo,
                                                       const uint64_t *__restrict__ hi, const uint64_t N,
                                                       uint64_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        uint64_t cnt = 0;
        if (hi[i] >= lo[i]) {
            uint64_t f, l;
            marker_span(ix, lo[i], hi[i], &f, &l);
            if (l > f) cnt = ix.mk_off[l] - ix.mk_off[f];
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
        uint64_t f, l;
        marker_span(ix, lo[i], hi[i], &f, &l);
        if (l <= f) continue;
        const uint64_t src = ix.mk_off[f], cnt = ix.mk_off[l] - src;
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
                    uint64_t f, l;
                    marker_span(ix, lo, hi, &f, &l);
                    if (l > f) {
                        const uint64_t src = ix.mk_off[f], cnt = ix.mk_off[l] - src;
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
template <typename P, bool FILL>
__global__ __launch_bounds__(1024) void k_marker_seeds(const DevIndex ix, const uint8_t *__restrict__ seqs,
                                                      const uint64_t *__restrict__ off, const uint64_t N,
                                                      const uint64_t wsize, const uint64_t max_range, const uint64_t ftab_k,
                                                      uint64_t *__restrict__ seed_cnt, uint64_t *__restrict__ mk_cnt,
                                                      const uint64_t *__restrict__ seed_off,
                                                      const uint64_t *__restrict__ mk_off,
                                                      uint64_t *__restrict__ seeds, uint64_t *__restrict__ mk, const uint32_t max_k) {
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
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) {
        const uint64_t beg = off[i], m = off[i + 1] - beg;
        const uint64_t fhi = ix.n - 1;
        uint64_t lo = 0, hi = fhi, plo = 0, phi = fhi;    // range, prev_range (:427-428)
        uint64_t window_ei = m, seed_ei = m;              // :434
        uint64_t ns = 0, tot = 0, mb_begin = 0;           // mbuf == markers [mb_begin, tot) of this read
        uint64_t *srec = FILL ? seeds + 6 * seed_off[i] : nullptr;
        const uint64_t mbase = FILL ? mk_off[i] : 0;
        auto update_mbuf = [&](uint64_t l, uint64_t h) {  // :437-441
            if (!have_ma || h - l + 1 > max_range) return;
            uint64_t f, e;
            marker_span(ix, l, h, &f, &e);
            if (e <= f) return;
            const uint64_t src = ix.mk_off[f], cnt = ix.mk_off[e] - src;
            if (FILL) {
                uint64_t *d = mk + mbase + tot;
                for (uint64_t t = 0; t < cnt; ++t) d[t] = ix.mk_vals[src + t];
            }
            tot += cnt;
        };
        auto emit = [&](uint64_t l, uint64_t h, uint64_t qs, uint64_t qe) {  // fn(range, (qs, qe-1), mbuf)
            if (FILL) {
                uint64_t *d = srec + 6 * ns;
                d[0] = l; d[1] = h; d[2] = qs; d[3] = qe; d[4] = mbase + mb_begin; d[5] = mbase + tot;
            }
            ++ns;
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
                    for (; m - i2 - 1 >= K; ++i2) {        // :454-464 slide left until a k-mer is in the ftab
                        seed_ei = m - i2 - 1;
                        window_ei = m - i2 - 1;
                        if (ftab_hit(m - i2 - 1)) {
                            i2 += K;                       // :460, then the outer ++i2
                            plo = lo; phi = hi;
                            break;
                        }
                        lo = 0; hi = fhi;                  // :463
                    }
                }
            }
            if (hi >= lo && seed_ei - (m - i2) >= wsize) update_mbuf(lo, hi);   // :478-480
            emit(lo, hi, m - i2, seed_ei);                                      // :481
            if (!FILL) {
                seed_cnt[i + 1] = ns;
                mk_cnt[i + 1] = tot;
            }
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
        if (!FILL) {
            seed_cnt[i + 1] = ns;
            mk_cnt[i + 1] = tot;
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

int grid_for(const LaunchCfg &cfg, uint64_t N) {
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
}  // namespace

template <bool USE_FTAB>
int launch_find_range_impl(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                           uint64_t *lo, uint64_t *hi, uint64_t *ssamp, const uint32_t *sel, const uint32_t *nsel, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool toe = ssamp != nullptr;
    // sel mode: the number of reads is only known on the device; a fixed modest grid loops over it
    const int cap = sel ? 512 : 0;
#define RBG_LAUNCH_FR(PT, TOE)                                                                                      \
    do {                                                                                                            \
        auto kern = k_find_range<PT, TOE, USE_FTAB>;                                                                \
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, cap);                                                    \
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, lo, hi, ssamp, sel, nsel);           \
    } while (0)
    if (ix.pos_bytes == 4) {
        if (toe) RBG_LAUNCH_FR(uint32_t, true); else RBG_LAUNCH_FR(uint32_t, false);
    } else {
        if (toe) RBG_LAUNCH_FR(uint64_t, true); else RBG_LAUNCH_FR(uint64_t, false);
    }
#undef RBG_LAUNCH_FR
    return static_cast<int>(hipGetLastError());
}

int launch_find_range(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    // without a table the table-free instantiation runs (it is also the one that BUILDS the table,
    // so profiles show that one-off launch under its own kernel name)
    return ix.ftab_k ? launch_find_range_impl<true>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream)
                     : launch_find_range_impl<false>(ix, cfg, seqs, off, N, lo, hi, ssamp, nullptr, nullptr, stream);
}

static int scan_in_place(uint64_t *vals, uint64_t N, void *tmp, size_t tmp_bytes, hipStream_t st);

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

int launch_find_range_packed(const DevIndex &ix, const LaunchCfg &cfg, const void *ws, const uint8_t *seqs, const uint64_t *off,
                             uint64_t N, uint64_t total_bytes, uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream) {
    if (N == 0) return 0;
    const PackLayout L = pack_layout(N, total_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *b = static_cast<const char *>(ws);
    const uint2 *meta = reinterpret_cast<const uint2 *>(b + L.meta);
    const uint4 *chunks = reinterpret_cast<const uint4 *>(b + L.chunks);
    const bool toe = ssamp != nullptr;
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
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    // the reads the packed form cannot express (a symbol outside the major alphabet)
    const uint32_t *sel = reinterpret_cast<const uint32_t *>(b + L.sel), *nsel = reinterpret_cast<const uint32_t *>(b + L.nsel);
    return ix.ftab_k ? launch_find_range_impl<true>(ix, cfg, seqs, off, N, lo, hi, ssamp, sel, nsel, stream)
                     : launch_find_range_impl<false>(ix, cfg, seqs, off, N, lo, hi, ssamp, sel, nsel, stream);
}

// ---- slot tables built on the device ---------------------------------------------------------------
// The first-level tables (one RankSlot per 2^shift BWT positions per symbol or k-mer, 57 GB for the
// bench index) are a pure function of the run lists, which are uploaded anyway: building them here
// instead of on the host and copying them over PCIe took "slot tables + upload" from 13 s to the time
// of uploading the run lists.  One thread fills kBuildGroup consecutive buckets: one binary search,
// then a linear walk.  The encoding is the one rank_in_slot / phi_step decode (rbg_dev.h).
namespace {
constexpr int kBuildGroup = 8;

template <typename P>
__global__ __launch_bounds__(256) void k_build_rank_slots(const RunEnt<P> *__restrict__ ent, const uint64_t nruns, const uint64_t n,
                                                          const uint32_t shift, RankSlot *__restrict__ slots,
                                                          uint32_t *__restrict__ ord, unsigned long long *__restrict__ overflow,
                                                          unsigned long long *__restrict__ dense_cursor) {
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t S = uint64_t(1) << shift;
    const uint64_t ngroups = (nb + kBuildGroup - 1) / kBuildGroup;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long novf = 0;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        const uint64_t b0 = g * kBuildGroup, b1 = b0 + kBuildGroup < nb ? b0 + kBuildGroup : nb;
        // k = # runs with start < first row of the bucket (ent[nruns] is the sentinel {n, total})
        uint64_t k = 0, z = nruns;
        const uint64_t first = b0 << shift;
        while (k < z) {
            const uint64_t mid = k + ((z - k) >> 1);
            if (static_cast<uint64_t>(ent[mid].start) < first) k = mid + 1; else z = mid;
        }
        for (uint64_t b = b0; b < b1; ++b) {
            const uint64_t B0 = b << shift;
            while (k < nruns && static_cast<uint64_t>(ent[k].start) < B0) ++k;
            ord[b] = static_cast<uint32_t>(k);
            uint64_t r0 = 0, ext = 0, prev_is_c = 0;
            if (k > 0) {
                const uint64_t ps = ent[k - 1].start, pc = ent[k - 1].cum;
                const uint64_t pl = static_cast<uint64_t>(ent[k].cum) - pc;
                r0 = pc + (pl < B0 - ps ? pl : B0 - ps);
                if (ps + pl > B0) ext = ps + pl - B0 < S ? ps + pl - B0 : S;
                prev_is_c = ps + pl >= B0 ? 1 : 0;
            }
            const bool wide = shift > kMaxNarrowShift;
            const uint64_t inline_runs = wide ? kSlotRunsWide : kSlotRuns;
            const uint32_t absent = wide ? 0xFFFFFFu : 0xFFFFu;
            uint32_t run[kSlotRuns] = {absent, absent, absent, absent};
            uint64_t cnt = 0;
            while (k + cnt < nruns && static_cast<uint64_t>(ent[k + cnt].start) < B0 + S) {
                if (cnt < inline_runs) {
                    const uint64_t st = ent[k + cnt].start;
                    const uint64_t off = st - B0;
                    const uint64_t full = static_cast<uint64_t>(ent[k + cnt + 1].cum) - static_cast<uint64_t>(ent[k + cnt].cum);
                    const uint64_t len = full < B0 + S - st ? full : B0 + S - st;
                    const uint32_t v = static_cast<uint32_t>(off | ((len - 1) << (wide ? 12 : 8)));
                    if (cnt == 0) run[0] = v; else if (cnt == 1) run[1] = v; else if (cnt == 2) run[2] = v; else run[3] = v;
                }
                ++cnt;
            }
            uint32_t code = static_cast<uint32_t>(cnt);
            if (cnt > inline_runs) { code = kSlotOvf; ++novf; }
            RankSlot s;
            s.r0 = static_cast<uint32_t>(r0);
            if (wide) {  // flatten() guarantees n < 2^40 for wide buckets
                s.w1 = static_cast<uint32_t>(r0 >> 32) | (static_cast<uint32_t>(ext) << 8) | (code << 21) | (static_cast<uint32_t>(prev_is_c) << 24);
                s.w2 = run[0];
                s.w3 = run[1];
            } else {
                s.w1 = static_cast<uint32_t>(ext) | (code << 9) | (static_cast<uint32_t>(prev_is_c) << 12) | (run[0] << 16);
                s.w2 = run[1] | (run[2] << 16);
                s.w3 = run[3] | (static_cast<uint32_t>(r0 >> 32) << 16);  // flatten() guarantees n < 2^48
                // overflow bucket: w2 = where its dense table will live (16-byte units; k_fill_dense writes it
                // once the total is known and the pool exists)
                if (code == kSlotOvf && dense_cursor)
                    s.w2 = static_cast<uint32_t>(atomicAdd(dense_cursor, static_cast<unsigned long long>(S < 8 ? 1 : S >> 3)));
            }
            slots[b] = s;
        }
    }
    novf = wave_sum(novf);
    if ((threadIdx.x & (kWave - 1)) == 0 && novf) atomicAdd(overflow, novf);
}

// Dense tables of the overflow buckets (rbg_dev.h): two bytes per row o -- # of the symbol in [B0, B0 + o), and
// 255 if row B0 + o - 1 holds the symbol, else # runs of it that start in [B0, B0 + o) (at most 128).
// One thread per bucket; only the few overflow buckets do any work (0.05-1 % of them).
template <typename P>
__global__ __launch_bounds__(256) void k_fill_dense(const RunEnt<P> *__restrict__ ent, const uint64_t n, const uint32_t shift,
                                                    const RankSlot *__restrict__ slots, const uint32_t *__restrict__ ord,
                                                    uint8_t *__restrict__ dense) {
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t S = uint64_t(1) << shift;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b + 1 < nb; b += stride) {
        const uint32_t w1 = slots[b].w1;
        if (((w1 >> 9) & 7u) != kSlotOvf) continue;
        uint32_t *out = reinterpret_cast<uint32_t *>(dense + (static_cast<uint64_t>(slots[b].w2) << 4));
        uint64_t k = ord[b];
        const uint64_t kend = ord[b + 1];
        const uint64_t B0 = b << shift;
        uint64_t cur_end = 0;  // end (exclusive) of the last run of the symbol that began at or before the current row
        if (k > 0) cur_end = static_cast<uint64_t>(ent[k - 1].start) + (static_cast<uint64_t>(ent[k].cum) - static_cast<uint64_t>(ent[k - 1].cum));
        uint64_t next_start = k < kend ? static_cast<uint64_t>(ent[k].start) : ~uint64_t(0);
        bool prev_c = (w1 >> 12) & 1u;  // row B0 - 1 holds the symbol
        uint32_t d = 0, starts = 0;
        for (uint64_t o = 0; o < S; o += 2) {
            uint32_t word = 0;
            for (uint32_t t = 0; t < 2; ++t) {
                const uint64_t pos = B0 + o + t;
                word |= ((d & 0xFFu) | ((prev_c ? 255u : starts) << 8)) << (16 * t);
                if (pos == next_start) {
                    cur_end = pos + (static_cast<uint64_t>(ent[k + 1].cum) - static_cast<uint64_t>(ent[k].cum));
                    ++k;
                    ++starts;
                    next_start = k < kend ? static_cast<uint64_t>(ent[k].start) : ~uint64_t(0);
                }
                prev_c = pos < cur_end;
                if (prev_c) ++d;
            }
            out[o >> 1] = word;
        }
    }
}

template <typename P>
__global__ __launch_bounds__(256) void k_build_phi_slots(const PhiEnt<P> *__restrict__ ent, const uint64_t r, const uint64_t n,
                                                         const uint32_t shift, PhiSlot<P> *__restrict__ slots,
                                                         uint32_t *__restrict__ ord, unsigned long long *__restrict__ overflow) {
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t S = uint64_t(1) << shift;
    const uint64_t ngroups = (nb + kBuildGroup - 1) / kBuildGroup;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    // D = base - pos (mod n); with no predecessor the reference uses the LAST record with
    // delta = i + 1 (toehold_sa.hpp:59,65), i.e. pos = -1
    auto D_of = [&](uint64_t j) { return (static_cast<uint64_t>(ent[j].base) + n - static_cast<uint64_t>(ent[j].pos)) % n; };
    unsigned long long novf = 0;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        const uint64_t b0 = g * kBuildGroup, b1 = b0 + kBuildGroup < nb ? b0 + kBuildGroup : nb;
        uint64_t k = 0, z = r;
        const uint64_t first = b0 << shift;
        while (k < z) {
            const uint64_t mid = k + ((z - k) >> 1);
            if (static_cast<uint64_t>(ent[mid].pos) < first) k = mid + 1; else z = mid;
        }
        for (uint64_t b = b0; b < b1; ++b) {
            const uint64_t B0 = b << shift;
            while (k < r && static_cast<uint64_t>(ent[k].pos) < B0) ++k;
            ord[b] = static_cast<uint32_t>(k);
            PhiSlot<P> s;
            s.dprev = static_cast<P>(k ? D_of(k - 1) : (static_cast<uint64_t>(ent[r - 1].base) + 1) % n);
            uint64_t cnt = 0;
            uint32_t off0 = 0xFFu, off1 = 0xFFu;
            uint64_t d0 = 0, d1 = 0;
            while (k + cnt < r && static_cast<uint64_t>(ent[k + cnt].pos) < B0 + S) {
                if (cnt == 0) { off0 = static_cast<uint32_t>(static_cast<uint64_t>(ent[k].pos) - B0); d0 = D_of(k); }
                else if (cnt == 1) { off1 = static_cast<uint32_t>(static_cast<uint64_t>(ent[k + 1].pos) - B0); d1 = D_of(k + 1); }
                ++cnt;
            }
            uint32_t code = static_cast<uint32_t>(cnt);
            if (cnt > 2) { code = kPhiOvf; ++novf; }
            s.d0 = static_cast<P>(d0);
            s.d1 = static_cast<P>(d1);
            s.meta = static_cast<P>(off0 | (off1 << 8) | (code << 16));
            slots[b] = s;
        }
    }
    novf = wave_sum(novf);
    if ((threadIdx.x & (kWave - 1)) == 0 && novf) atomicAdd(overflow, novf);
}
}  // namespace

int launch_build_rank_slots(uint32_t pos_bytes, const void *ent, uint64_t nruns, uint64_t n, uint32_t shift, void *slots,
                            uint32_t *ord, unsigned long long *overflow, unsigned long long *dense_cursor, void *stream) {
    if (shift > kMaxNarrowShift) dense_cursor = nullptr;  // wide buckets keep the run-list search
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t groups = (nb + kBuildGroup - 1) / kBuildGroup;
    const int grid = static_cast<int>(std::min<uint64_t>((groups + 255) / 256, 256ull * 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4)
        hipLaunchKernelGGL((k_build_rank_slots<uint32_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint32_t> *>(ent), nruns, n, shift,
                           static_cast<RankSlot *>(slots), ord, overflow, dense_cursor);
    else
        hipLaunchKernelGGL((k_build_rank_slots<uint64_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint64_t> *>(ent), nruns, n, shift,
                           static_cast<RankSlot *>(slots), ord, overflow, dense_cursor);
    return static_cast<int>(hipGetLastError());
}

int launch_fill_dense(uint32_t pos_bytes, const void *ent, uint64_t n, uint32_t shift, const void *slots, const uint32_t *ord,
                      uint8_t *dense, void *stream) {
    if (shift > kMaxNarrowShift) return 0;
    const uint64_t nb = (n >> shift) + 2;
    const int grid = static_cast<int>(std::min<uint64_t>((nb + 255) / 256, 256ull * 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4)
        hipLaunchKernelGGL((k_fill_dense<uint32_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint32_t> *>(ent), n, shift,
                           static_cast<const RankSlot *>(slots), ord, dense);
    else
        hipLaunchKernelGGL((k_fill_dense<uint64_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint64_t> *>(ent), n, shift,
                           static_cast<const RankSlot *>(slots), ord, dense);
    return static_cast<int>(hipGetLastError());
}

int launch_build_phi_slots(uint32_t pos_bytes, const void *ent, uint64_t r, uint64_t n, uint32_t shift, void *slots, uint32_t *ord,
                           unsigned long long *overflow, void *stream) {
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t groups = (nb + kBuildGroup - 1) / kBuildGroup;
    const int grid = static_cast<int>(std::min<uint64_t>((groups + 255) / 256, 256ull * 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4)
        hipLaunchKernelGGL((k_build_phi_slots<uint32_t>), dim3(grid), dim3(256), 0, st, static_cast<const PhiEnt<uint32_t> *>(ent), r, n, shift,
                           static_cast<PhiSlot<uint32_t> *>(slots), ord, overflow);
    else
        hipLaunchKernelGGL((k_build_phi_slots<uint64_t>), dim3(grid), dim3(256), 0, st, static_cast<const PhiEnt<uint64_t> *>(ent), r, n, shift,
                           static_cast<PhiSlot<uint64_t> *>(slots), ord, overflow);
    return static_cast<int>(hipGetLastError());
}

// ---- ftab construction: search every word of k major symbols with the step kernel itself ---------
namespace {
// words base .. base + W of the table, as a batch of W reads
__global__ __launch_bounds__(256) void k_ftab_words(const DevIndex ix, const uint32_t k, const uint64_t base, const uint64_t W,
                                                    uint8_t *__restrict__ seqs, uint64_t *__restrict__ off,
                                                    const uint8_t *__restrict__ major_byte) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w <= W; w += stride) {
        off[w] = w * k;
        if (w == W) break;
        uint64_t x = base + w;
        for (uint32_t t = k; t > 0; --t) {  // least significant digit = rightmost symbol
            seqs[w * k + t - 1] = major_byte[x % ix.nmajor];
            x /= ix.nmajor;
        }
    }
}
template <typename P>
__global__ __launch_bounds__(256) void k_ftab_pack(const uint64_t W, const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                                   const uint64_t *__restrict__ ss, void *__restrict__ tab) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < W; w += stride) {
        const uint64_t k = ss ? ss[w] : 0;
        if constexpr (sizeof(P) == 4) {
            uint4 e = make_uint4(static_cast<uint32_t>(lo[w]), static_cast<uint32_t>(hi[w]), static_cast<uint32_t>(k), 0u);
            if (k == ~uint64_t(0)) e.z = 0xFFFFFFFFu;
            else if (k >= 0xFFFFFFF0ull) e = make_uint4(2u, 0u, 0u, 0u);  // not expressible: search this word step by step
            static_cast<uint4 *>(tab)[w] = e;
        } else {
            static_cast<ulonglong4 *>(tab)[w] = make_ulonglong4(lo[w], hi[w], k, 0);
        }
    }
}
}  // namespace

// scratch of one build chunk: word bytes, offsets, three result arrays
constexpr uint64_t kFtabChunk = uint64_t(1) << 26;
size_t ftab_build_scratch_bytes(uint64_t words, uint32_t k) {
    const uint64_t C = std::min<uint64_t>(words, kFtabChunk);
    return static_cast<size_t>(C * (k + 32ull) + 1024);
}

int launch_build_ftab(const DevIndex &ix, const LaunchCfg &cfg, uint32_t k, void *tab, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t W = 1;
    for (uint32_t t = 0; t < k; ++t) W *= ix.nmajor;
    const uint64_t C = std::min<uint64_t>(W, kFtabChunk);  // words per pass: bounds the scratch whatever the table size
    uint8_t *seqs = nullptr, *mb = nullptr;
    uint64_t *off = nullptr, *lo = nullptr, *hi = nullptr, *ss = nullptr;
    hipError_t e = hipMalloc(&seqs, C * k + 64);
    if (e == hipSuccess) e = hipMalloc(&off, (C + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&lo, C * 8);
    if (e == hipSuccess) e = hipMalloc(&hi, C * 8);
    if (e == hipSuccess && ix.has_tsa) e = hipMalloc(&ss, C * 8);
    if (e == hipSuccess) e = hipMalloc(&mb, 256);
    if (e == hipSuccess) {
        // major index -> byte, recovered from lut2 on the host side of the caller would need another
        // argument; derive it here from the device lut2 with a tiny copy
        uint8_t lut2[256], inv[256] = {0};
        e = hipMemcpy(lut2, ix.lut2, 256, hipMemcpyDeviceToHost);
        for (int b = 0; b < 256; ++b)
            if (lut2[b] != 0xFF) inv[lut2[b]] = static_cast<uint8_t>(b);
        if (e == hipSuccess) e = hipMemcpy(mb, inv, 256, hipMemcpyHostToDevice);
    }
    int rc = static_cast<int>(e);
    DevIndex plain = ix;  // the words are searched WITHOUT a table
    plain.ftab = nullptr;
    plain.ftab_k = 0;
    const size_t entry = ix.pos_bytes == 4 ? 16 : 32;
    for (uint64_t base = 0; !rc && base < W; base += C) {
        const uint64_t cnt = std::min<uint64_t>(C, W - base);
        void *dst = static_cast<char *>(tab) + base * entry;
        hipLaunchKernelGGL(k_ftab_words, dim3(grid_for(cfg, cnt + 1)), dim3(256), 0, st, plain, k, base, cnt, seqs, off, mb);
        rc = static_cast<int>(hipGetLastError());
        if (!rc) rc = launch_find_range(plain, cfg, seqs, off, cnt, lo, hi, ss, st);
        if (!rc) {
            if (ix.pos_bytes == 4) hipLaunchKernelGGL(k_ftab_pack<uint32_t>, dim3(grid_for(cfg, cnt)), dim3(256), 0, st, cnt, lo, hi, ss, dst);
            else hipLaunchKernelGGL(k_ftab_pack<uint64_t>, dim3(grid_for(cfg, cnt)), dim3(256), 0, st, cnt, lo, hi, ss, dst);
            rc = static_cast<int>(hipGetLastError());
        }
    }
    if (!rc) rc = static_cast<int>(hipStreamSynchronize(st));
    (void)hipFree(seqs); (void)hipFree(off); (void)hipFree(lo); (void)hipFree(hi); (void)hipFree(ss); (void)hipFree(mb);
    return rc;
}

// ---- chain ordering for locate: permutation of the reads by toehold value (radix sort) ----------
namespace {
__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ v, const uint64_t N) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) v[i] = static_cast<uint32_t>(i);
}
struct OrderWs {
    size_t perm, iota, keys, sort, sort_bytes, total;
};
OrderWs order_layout(uint64_t N) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    OrderWs w{};
    w.perm = 0;
    w.iota = up(w.perm + N * 4);
    w.keys = up(w.iota + N * 4);
    w.sort = up(w.keys + N * 8);
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
    hipLaunchKernelGGL(k_iota, dim3(grid_for(cfg, N)), dim3(256), 0, st, iota, N);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    int end_bit = 1;
    while (end_bit < 64 && (ix.n >> end_bit)) ++end_bit;  // toeholds are text positions < n
    // the order only has to bring chains of nearby text positions together: the low bits (positions
    // inside one 64-byte line of phi slots) need no sorting, which saves a radix pass
    int begin_bit = static_cast<int>(ix.phi_shift) + 2;
    if (end_bit - begin_bit < 8) begin_bit = 0;
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

static int scan_in_place(uint64_t *vals, uint64_t N, void *tmp, size_t tmp_bytes, hipStream_t st) {
    if (N == 0) return 0;
    size_t need = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, need, vals, vals, static_cast<int64_t>(N));
    if (need > tmp_bytes) return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipcub::DeviceScan::InclusiveSum(tmp, need, vals, vals, static_cast<int64_t>(N), st));
}

int launch_locate_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                       uint64_t max_hits, uint64_t *loc_off, void *tmp, size_t tmp_bytes, void *stream) {
    (void)ix;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_occ, dim3(grid_for(cfg, N)), dim3(cfg.block_threads), 0, st, lo, hi, N, max_hits, loc_off);
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    return scan_in_place(loc_off + 1, N, tmp, tmp_bytes, st);
}

int launch_locate_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi,
                       const uint64_t *k, uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs,
                       const uint64_t *sub, const void *order, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(cfg, N)), block(cfg.block_threads);
    // the order workspace also holds the toeholds in sorted order (launch_locate_order's key output)
    const uint64_t *skeys = order ? reinterpret_cast<const uint64_t *>(static_cast<const char *>(order) + order_layout(N).keys) : nullptr;
    if (ix.pos_bytes == 4) hipLaunchKernelGGL((k_locate_fill<uint32_t>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, locs, sub, static_cast<const uint32_t *>(order), skeys);
    else hipLaunchKernelGGL((k_locate_fill<uint64_t>), grid, block, 0, st, ix, lo, hi, k, N, max_hits, loc_off, locs, sub, static_cast<const uint32_t *>(order), skeys);
    return static_cast<int>(hipGetLastError());
}

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
                             size_t tmp_bytes, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ix.pos_bytes == 4) {
        auto kern = k_marker_seeds<uint32_t, false>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, seed_off, mk_off, nullptr, nullptr, nullptr, nullptr, kSeedKmerLevel);
    } else {
        auto kern = k_marker_seeds<uint64_t, false>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, seed_off, mk_off, nullptr, nullptr, nullptr, nullptr, kSeedKmerLevel);
    }
    int rc = static_cast<int>(hipGetLastError());
    if (rc) return rc;
    rc = scan_in_place(seed_off + 1, N, tmp, tmp_bytes, st);
    if (rc) return rc;
    return scan_in_place(mk_off + 1, N, tmp, tmp_bytes, st);
}

int launch_marker_seeds_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range, uint64_t ftab_k, const uint64_t *seed_off, const uint64_t *mk_off,
                             uint64_t *seeds, uint64_t *mk, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ix.pos_bytes == 4) {
        auto kern = k_marker_seeds<uint32_t, true>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, nullptr, nullptr, seed_off, mk_off, seeds, mk, kSeedKmerLevel);
    } else {
        auto kern = k_marker_seeds<uint64_t, true>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, wsize, max_range, ftab_k, nullptr, nullptr, seed_off, mk_off, seeds, mk, kSeedKmerLevel);
    }
    return static_cast<int>(hipGetLastError());
}

int launch_greedy_seed(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                       uint64_t min_length, uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, void *stream) {
    if (N == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ix.pos_bytes == 4) {
        auto kern = k_greedy_seed<uint32_t>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, min_length, lo, hi, qs, qe, ss, kSeedKmerLevel);
    } else {
        auto kern = k_greedy_seed<uint64_t>;
        const KmerLaunch L = kmer_launch(ix, cfg, N, kern, 0, kSeedKmerLevel);
        hipLaunchKernelGGL(kern, L.grid, L.block, L.lds, st, ix, seqs, off, N, min_length, lo, hi, qs, qe, ss, kSeedKmerLevel);
    }
    return static_cast<int>(hipGetLastError());
}

int launch_lf(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym,
              uint64_t N, uint64_t *lo_out, uint64_t *hi_out, void *stream) {
    if (N == 0) return 0;
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
