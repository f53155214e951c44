// k_build.hip -- load-time kernels: first-level slot tables and dense overflow tables from the uploaded run lists, the ftab.
#include "rbg_device.hpp"

namespace rbg {

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

template <typename P, bool PACKED = false>
__global__ __launch_bounds__(256) void k_build_phi_slots(const PhiEnt<P> *__restrict__ ent, const uint64_t r, const uint64_t n,
                                                         const uint32_t shift, void *__restrict__ slots_v,
                                                         uint32_t *__restrict__ ord, unsigned long long *__restrict__ overflow) {
    PhiSlot<P> *__restrict__ slots = static_cast<PhiSlot<P> *>(slots_v);
    PhiSlotPacked *__restrict__ packed = static_cast<PhiSlotPacked *>(slots_v);
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
            if (PACKED) {   // rbg_dev.h PhiSlotPacked (offsets below 64: shift <= 6; D values below n < 2^38)
                const uint64_t dp = static_cast<uint64_t>(s.dprev);
                PhiSlotPacked q;
                q.w0 = dp | ((d0 & ((uint64_t(1) << 26) - 1)) << 38);
                q.w1 = (d0 >> 26) | (d1 << 12) | (static_cast<uint64_t>(off0 & 63u) << 50) | (static_cast<uint64_t>(off1 & 63u) << 56) |
                       (static_cast<uint64_t>(code) << 62);
                packed[b] = q;
            } else {
                slots[b] = s;
            }
        }
    }
    novf = wave_sum(novf);
    if ((threadIdx.x & (kWave - 1)) == 0 && novf) atomicAdd(overflow, novf);
}
// ---- run-indexed layout built from run lists that are already on the device (the k-mer levels of k_compose.hip) -----------
// dir[doff[t] + b] = # runs of table t that start below b << dshift[t] (rbg_dev.h DevRunTab2): one thread per directory entry
template <typename P>
__global__ __launch_bounds__(256) void k_run_dirs(const RunEnt<P> *__restrict__ ent, const uint64_t *__restrict__ first, const uint64_t *__restrict__ nruns,
                                                  const uint64_t *__restrict__ doff, const uint32_t *__restrict__ dshift, const uint32_t T,
                                                  const uint64_t total, uint32_t *__restrict__ dir) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint32_t a = 0, z = T;                       // the table: last t with doff[t] <= i
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (doff[mid] <= i) a = mid; else z = mid;
        }
        const uint64_t lim = (i - doff[a]) << dshift[a];
        const RunEnt<P> *__restrict__ e = ent + first[a];
        uint64_t lo = 0, hi = nruns[a];
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (static_cast<uint64_t>(e[mid].start) < lim) lo = mid + 1; else hi = mid;
        }
        dir[i] = static_cast<uint32_t>(lo);
    }
}
// 8-byte samples -> the 6-byte form of the run-indexed layout (rbg_dev.h Samp48)
__global__ __launch_bounds__(256) void k_pack_samp48(const uint64_t *__restrict__ in, const uint64_t n, uint16_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint64_t v = in[j];
        out[3 * j] = static_cast<uint16_t>(v);
        out[3 * j + 1] = static_cast<uint16_t>(v >> 16);
        out[3 * j + 2] = static_cast<uint16_t>(v >> 32);
    }
}
}  // namespace

// ---- format 2 of the run-indexed layout at 8-byte positions (rbg_dev.h DevRunTab2): fillers, low-word pairs, {count, hi}
// directories, the phi list's 12-byte entries, its directory and super counts -- all from {key, value} u64 pairs on the device
namespace {
typedef unsigned long long u64pair __attribute__((ext_vector_type(2)));

// arr[j + 1] = 1 + # fillers entry j needs (arr[0] = 0; an inclusive scan then gives every entry's place); *total += the
// fillers.  ent: m pairs sorted by key within a table, every table closed by a sentinel with key n (phi: one table)
__global__ __launch_bounds__(256) void k_fill_count(const u64pair *__restrict__ ent, const uint64_t m, const uint64_t n, const uint32_t fs, uint64_t *__restrict__ arr,
                                                    unsigned long long *__restrict__ total) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m; j += stride) {
        const uint64_t key = ent[j].x;
        uint64_t nf = 0;
        if (key != n && j + 1 < m) {
            const uint64_t gap = ent[j + 1].x - key;
            nf = gap ? (gap - 1) >> fs : 0;
        }
        if (arr) { arr[j + 1] = 1 + nf; if (j == 0) arr[0] = 0; }
        mine += nf;
    }
    mine = wave_sum(mine);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(total, mine);
}
// the list with its fillers: entry j goes to pos[j], its fillers behind it.  PHI: the value of a filler at key + f * G is
// (value + f * G) mod n (phi(i) = base + (i - pos) unchanged); runs: value + min(f * G, run length) -- a continuation of
// the run, or an empty run once it has ended -- and the run's sample again
template <bool PHI>
__global__ __launch_bounds__(256) void k_fill_expand(const u64pair *__restrict__ ent, const uint64_t *__restrict__ samp, const uint64_t m, const uint64_t n,
                                                     const uint32_t fs, const uint64_t *__restrict__ pos, u64pair *__restrict__ ent_out, uint64_t *__restrict__ samp_out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m; j += stride) {
        const u64pair e = ent[j];
        const uint64_t o = pos[j], nf = pos[j + 1] - o - 1;
        ent_out[o] = e;
        const uint64_t sv = samp ? samp[j] : 0;
        if (samp_out) samp_out[o] = sv;
        if (!nf) continue;
        const uint64_t len = PHI ? 0 : ent[j + 1].y - e.y;
        for (uint64_t f = 1; f <= nf; ++f) {
            const uint64_t adv = f << fs;
            const uint64_t v = PHI ? (e.y + adv) % n : e.y + (adv < len ? adv : len);
            ent_out[o + f] = u64pair{e.x + adv, v};
            if (samp_out) samp_out[o + f] = sv;
        }
    }
}
__global__ __launch_bounds__(256) void k_gather_u64(const uint64_t *__restrict__ src, const uint64_t *__restrict__ idx, const uint64_t count, uint64_t *__restrict__ out) {
    const uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < count) out[j] = src[idx[j]];
}
// {start, cum} -> their low words (m pairs, then `spare` copies of the last)
__global__ __launch_bounds__(256) void k_pack_pairs32(const u64pair *__restrict__ ent, const uint64_t m, const uint64_t spare, uint2 *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m + spare; j += stride) {
        const u64pair e = ent[j < m ? j : m - 1];
        out[j] = make_uint2(static_cast<uint32_t>(e.x), static_cast<uint32_t>(e.y));
    }
}
// {pos, base} -> PhiEnt12 (m entries incl. the sentinel, then `spare` copies of it)
__global__ __launch_bounds__(256) void k_pack_phi12(const u64pair *__restrict__ ent, const uint64_t m, const uint64_t spare, PhiEnt12 *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m + spare; j += stride) {
        const u64pair e = ent[j < m ? j : m - 1];
        out[j] = PhiEnt12{static_cast<uint32_t>(e.x), static_cast<uint32_t>(e.y), static_cast<uint32_t>(e.y >> 32)};
    }
}
// RunDir64 directories of a depth's tables (k_run_dirs with the rank's high part beside the count)
__global__ __launch_bounds__(256) void k_run_dirs2(const u64pair *__restrict__ ent, const uint64_t *__restrict__ first, const uint64_t *__restrict__ nruns,
                                                   const uint64_t *__restrict__ doff, const uint32_t *__restrict__ dshift, const uint32_t T,
                                                   const uint64_t total, RunDir64 *__restrict__ dir) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint32_t a = 0, z = T;                       // the table: last t with doff[t] <= i
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (doff[mid] <= i) a = mid; else z = mid;
        }
        const uint64_t lim = (i - doff[a]) << dshift[a];
        const u64pair *__restrict__ e = ent + first[a];
        uint64_t lo = 0, hi = nruns[a];
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (e[mid].x < lim) lo = mid + 1; else hi = mid;
        }
        dir[i] = RunDir64{static_cast<uint32_t>(lo), static_cast<uint32_t>(e[lo ? lo - 1 : 0].y >> 31)};   // (cums carry the table's F: below its first run the rank is F, the first entry's cum)
    }
}
// cum += F of the entry's table (rbg_dev.h: the run lists hold rows of the F column, not counts)
template <typename P>
__global__ __launch_bounds__(256) void k_fold_F(RunEnt<P> *__restrict__ ent, const uint64_t *__restrict__ first, const uint64_t *__restrict__ F, const uint32_t T,
                                               const uint64_t total) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint32_t a = 0, z = T;                       // the table: last t with first[t] <= i
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (first[mid] <= i) a = mid; else z = mid;
        }
        ent[i].cum = static_cast<P>(static_cast<uint64_t>(ent[i].cum) + F[a]);
    }
}
// bucket records of a depth's tables (rbg_dev.h RunRec2): record i belongs to bucket i - roff[t] of table t
template <typename P>
__global__ __launch_bounds__(256) void k_run_recs2(const RunEnt<P> *__restrict__ ent, const uint64_t *__restrict__ first, const uint64_t *__restrict__ nruns,
                                                   const uint64_t *__restrict__ roff, const uint32_t *__restrict__ rshift, const uint32_t T,
                                                   const uint64_t total, RunRec2 *__restrict__ recs, unsigned long long *__restrict__ overflow) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long novf = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint32_t a = 0, z = T;                       // the table: last t with roff[t] <= i
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (roff[mid] <= i) a = mid; else z = mid;
        }
        const uint32_t sh = rshift[a];
        const uint64_t lim = (i - roff[a]) << sh, lim2 = lim + (uint64_t(1) << sh);
        const RunEnt<P> *__restrict__ e = ent + first[a];
        const uint64_t nr = nruns[a];
        uint64_t lo = 0, hi = nr;
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (static_cast<uint64_t>(e[mid].start) < lim) lo = mid + 1; else hi = mid; }
        uint64_t up = lo;                            // # entries starting below the next bucket
        constexpr uint64_t kScan = kRec2CompactIn + 2;   // (a compact record holds eleven; one more and the bucket overflows whatever its runs are)
        while (up < nr && up < lo + kScan && static_cast<uint64_t>(e[up].start) < lim2) ++up;
        if (up == lo + kScan) {                      // more than a record holds: count them all
            uint64_t l2 = up, h2 = nr;
            while (l2 < h2) { const uint64_t mid = l2 + ((h2 - l2) >> 1); if (static_cast<uint64_t>(e[mid].start) < lim2) l2 = mid + 1; else h2 = mid; }
            up = l2;
        }
        const uint64_t e0 = lo ? lo - 1 : 0, cnt = up - e0;
        RunRec2 r;
        r.e0 = static_cast<uint32_t>(e0);
        r.hi = sizeof(P) == 8 ? static_cast<uint32_t>(static_cast<uint64_t>(e[lo ? lo - 1 : 0].cum) >> 31) : 0u;   // (cums carry the table's F: entry 0's is F itself)
#pragma unroll
        for (uint32_t k = 0; k < kRec2Pivots; ++k) r.ent[k] = 0u;
        // COMPACT (rbg_dev.h): the first entry held in full, up to ten more as {offset into the bucket : sh bits, length : 32 - sh bits}
        const uint64_t len_lim = sh < 32 ? (uint64_t(1) << (32 - sh)) : 0;
        bool compact = sh < 32 && cnt <= kRec2CompactIn + 1;
        for (uint64_t j = 1; compact && j < cnt; ++j) {
            const uint64_t len = static_cast<uint64_t>(e[e0 + j + 1].cum) - static_cast<uint64_t>(e[e0 + j].cum);
            compact = static_cast<uint64_t>(e[e0 + j].start) >= lim && len < len_lim;   // (entries e0 + 1 .. start inside the bucket: lo <= e0 + 1)
        }
        if (compact) {
            const uint32_t unused = static_cast<uint32_t>((uint64_t(1) << sh) - 1);                   // (offset = the bucket's last row, length 0: never below a position, adds nothing)
            r.meta = static_cast<uint32_t>(cnt) | kRec2Compact;
            r.cum_end = static_cast<uint32_t>(static_cast<uint64_t>(e[e0].cum));                      // cum of the first entry held (none held: the table's first, or its sentinel -- F)
            r.ent[0] = cnt ? static_cast<uint32_t>(static_cast<uint64_t>(e[e0].start)) : 0u;
            r.ent[1] = cnt ? static_cast<uint32_t>(static_cast<uint64_t>(e[e0 + 1].cum) - static_cast<uint64_t>(e[e0].cum)) : 0u;   // its length (< 2^32: fillers)
            for (uint64_t j = 1; j <= kRec2CompactIn; ++j) {
                uint32_t v = unused;
                if (j < cnt) {
                    const uint64_t off = static_cast<uint64_t>(e[e0 + j].start) - lim;
                    const uint64_t len = static_cast<uint64_t>(e[e0 + j + 1].cum) - static_cast<uint64_t>(e[e0 + j].cum);
                    v = static_cast<uint32_t>(off) | static_cast<uint32_t>(len << sh);
                }
                r.ent[1 + j] = v;
            }
        } else {
            // more entries than a record holds, or a run too long for its length field: twelve PIVOTS a stride apart (rbg_runs2_device.hpp
            // LaneRec: the starts of candidates stride, 2 x stride, ...), then one scan of the run list
            r.meta = kRec2Overflow;
            r.cum_end = static_cast<uint32_t>(cnt > 0xFFFFFFFFull ? 0xFFFFFFFFull : cnt);
            const uint64_t z = r.cum_end, stride = (z + 12) / 13;
            for (uint64_t j = 0; j < kRec2Pivots; ++j)
                r.ent[j] = (j + 1) * stride < z ? static_cast<uint32_t>(static_cast<uint64_t>(e[e0 + (j + 1) * stride].start)) : 0xFFFFFFFFu;
            ++novf;
        }
        recs[i] = r;
    }
    novf = wave_sum(novf);
    if ((threadIdx.x & (kWave - 1)) == 0 && novf && overflow) atomicAdd(overflow, novf);
}

// the phi directory: dir[b] = low word of # entries with pos < b << shift; super[b >> ss] = that count in full (ss == 0: none)
template <typename KeyAt>
__device__ __forceinline__ uint64_t count_below(KeyAt key_at, uint64_t m, uint64_t lim) {
    uint64_t lo = 0, hi = m;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (key_at(mid) < lim) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <typename P>
__global__ __launch_bounds__(256) void k_phi_dir(const void *__restrict__ ent, const uint64_t m, const uint32_t shift, const uint64_t nb, uint32_t *__restrict__ dir,
                                                 const uint32_t ss, uint64_t *__restrict__ super) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nb; b += stride) {
        const uint64_t lim = b << shift;
        uint64_t g;
        if constexpr (sizeof(P) == 8) g = count_below([&](uint64_t j) { return static_cast<const u64pair *>(ent)[j].x; }, m, lim);
        else g = count_below([&](uint64_t j) { return static_cast<uint64_t>(static_cast<const PhiEnt<uint32_t> *>(ent)[j].pos); }, m, lim);
        dir[b] = static_cast<uint32_t>(g);
        if (ss && (b & ((uint64_t(1) << ss) - 1)) == 0) super[b >> ss] = g;
    }
}
inline int grid_of(uint64_t n) { return static_cast<int>(std::min<uint64_t>((n + 255) / 256, 256ull * 64)); }
}  // namespace

int launch_fill_count(const void *ent, uint64_t m, uint64_t n, uint32_t fill_shift, uint64_t *arr, unsigned long long *total, void *stream) {
    if (!m) return 0;
    hipLaunchKernelGGL(k_fill_count, dim3(grid_of(m)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u64pair *>(ent), m, n, fill_shift, arr, total);
    return static_cast<int>(hipGetLastError());
}
int launch_scan_u64(uint64_t *vals, uint64_t N, void *tmp, size_t tmp_bytes, void *stream) { return scan_in_place(vals, N, tmp, tmp_bytes, static_cast<hipStream_t>(stream)); }
int launch_fill_expand(bool phi, const void *ent, const uint64_t *samp, uint64_t m, uint64_t n, uint32_t fill_shift, const uint64_t *pos, void *ent_out, uint64_t *samp_out,
                       void *stream) {
    if (!m) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (phi) hipLaunchKernelGGL(k_fill_expand<true>, dim3(grid_of(m)), dim3(256), 0, st, static_cast<const u64pair *>(ent), samp, m, n, fill_shift, pos, static_cast<u64pair *>(ent_out), samp_out);
    else hipLaunchKernelGGL(k_fill_expand<false>, dim3(grid_of(m)), dim3(256), 0, st, static_cast<const u64pair *>(ent), samp, m, n, fill_shift, pos, static_cast<u64pair *>(ent_out), samp_out);
    return static_cast<int>(hipGetLastError());
}
int launch_gather_u64(const uint64_t *src, const uint64_t *idx, uint64_t count, uint64_t *out, void *stream) {
    if (!count) return 0;
    hipLaunchKernelGGL(k_gather_u64, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src, idx, count, out);
    return static_cast<int>(hipGetLastError());
}
int launch_pack_pairs32(const void *ent, uint64_t m, uint64_t spare, void *out, void *stream) {
    if (!m) return 0;
    hipLaunchKernelGGL(k_pack_pairs32, dim3(grid_of(m + spare)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u64pair *>(ent), m, spare, static_cast<uint2 *>(out));
    return static_cast<int>(hipGetLastError());
}
int launch_pack_phi12(const void *ent, uint64_t m, uint64_t spare, void *out, void *stream) {
    if (!m) return 0;
    hipLaunchKernelGGL(k_pack_phi12, dim3(grid_of(m + spare)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u64pair *>(ent), m, spare, static_cast<PhiEnt12 *>(out));
    return static_cast<int>(hipGetLastError());
}
int launch_run_dirs2(const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *doff, const uint32_t *dshift, uint32_t T, uint64_t total,
                     void *dir, void *stream) {
    if (!total) return 0;
    hipLaunchKernelGGL(k_run_dirs2, dim3(grid_of(total)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u64pair *>(ent), first, nruns, doff, dshift, T, total,
                       static_cast<RunDir64 *>(dir));
    return static_cast<int>(hipGetLastError());
}
int launch_fold_F(uint32_t pos_bytes, void *ent, const uint64_t *first, const uint64_t *F, uint32_t T, uint64_t total, void *stream) {
    if (total == 0 || T == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4) hipLaunchKernelGGL((k_fold_F<uint32_t>), dim3(grid_of(total)), dim3(256), 0, st, static_cast<RunEnt<uint32_t> *>(ent), first, F, T, total);
    else hipLaunchKernelGGL((k_fold_F<uint64_t>), dim3(grid_of(total)), dim3(256), 0, st, static_cast<RunEnt<uint64_t> *>(ent), first, F, T, total);
    return static_cast<int>(hipGetLastError());
}
int launch_run_recs2(uint32_t pos_bytes, const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *roff, const uint32_t *rshift, uint32_t T, uint64_t total,
                     void *recs, unsigned long long *overflow, void *stream) {
    if (!total) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4) hipLaunchKernelGGL((k_run_recs2<uint32_t>), dim3(grid_of(total)), dim3(256), 0, st, static_cast<const RunEnt<uint32_t> *>(ent), first, nruns, roff, rshift, T, total,
                                           static_cast<RunRec2 *>(recs), overflow);
    else hipLaunchKernelGGL((k_run_recs2<uint64_t>), dim3(grid_of(total)), dim3(256), 0, st, static_cast<const RunEnt<uint64_t> *>(ent), first, nruns, roff, rshift, T, total,
                            static_cast<RunRec2 *>(recs), overflow);
    return static_cast<int>(hipGetLastError());
}
int launch_phi_dir(uint32_t pos_bytes, const void *ent, uint64_t m, uint32_t shift, uint64_t nb, uint32_t *dir, uint32_t ss, uint64_t *super, void *stream) {
    if (!nb) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4) hipLaunchKernelGGL(k_phi_dir<uint32_t>, dim3(grid_of(nb)), dim3(256), 0, st, ent, m, shift, nb, dir, 0u, static_cast<uint64_t *>(nullptr));
    else hipLaunchKernelGGL(k_phi_dir<uint64_t>, dim3(grid_of(nb)), dim3(256), 0, st, ent, m, shift, nb, dir, ss, super);
    return static_cast<int>(hipGetLastError());
}

int launch_run_dirs(uint32_t pos_bytes, const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *doff, const uint32_t *dshift,
                    uint32_t T, uint64_t total, uint32_t *dir, void *stream) {
    if (!total) return 0;
    const int grid = static_cast<int>(std::min<uint64_t>((total + 255) / 256, 256ull * 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4)
        hipLaunchKernelGGL((k_run_dirs<uint32_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint32_t> *>(ent), first, nruns, doff, dshift, T, total, dir);
    else
        hipLaunchKernelGGL((k_run_dirs<uint64_t>), dim3(grid), dim3(256), 0, st, static_cast<const RunEnt<uint64_t> *>(ent), first, nruns, doff, dshift, T, total, dir);
    return static_cast<int>(hipGetLastError());
}
int launch_pack_samp48(const uint64_t *in, uint64_t n, void *out, void *stream) {
    if (!n) return 0;
    const int grid = static_cast<int>(std::min<uint64_t>((n + 255) / 256, 256ull * 64));
    hipLaunchKernelGGL(k_pack_samp48, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), in, n, static_cast<uint16_t *>(out));
    return static_cast<int>(hipGetLastError());
}

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

int launch_build_phi_slots(uint32_t pos_bytes, bool packed, const void *ent, uint64_t r, uint64_t n, uint32_t shift, void *slots, uint32_t *ord,
                           unsigned long long *overflow, void *stream) {
    const uint64_t nb = (n >> shift) + 2;
    const uint64_t groups = (nb + kBuildGroup - 1) / kBuildGroup;
    const int grid = static_cast<int>(std::min<uint64_t>((groups + 255) / 256, 256ull * 64));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pos_bytes == 4)
        hipLaunchKernelGGL((k_build_phi_slots<uint32_t, false>), dim3(grid), dim3(256), 0, st, static_cast<const PhiEnt<uint32_t> *>(ent), r, n, shift,
                           slots, ord, overflow);
    else if (packed)
        hipLaunchKernelGGL((k_build_phi_slots<uint64_t, true>), dim3(grid), dim3(256), 0, st, static_cast<const PhiEnt<uint64_t> *>(ent), r, n, shift,
                           slots, ord, overflow);
    else
        hipLaunchKernelGGL((k_build_phi_slots<uint64_t, false>), dim3(grid), dim3(256), 0, st, static_cast<const PhiEnt<uint64_t> *>(ent), r, n, shift,
                           slots, ord, overflow);
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

// ---- synthetic reads generated on the device (BASELINE.json configs[3]: "1B synthetic 150 bp reads" -- the host
// cannot feed 150 GB).  Counter-based: read g (a GLOBAL read index) is a pure function of (seed, g),
// so any rank can generate its shard of any batch: haplotype = r0 % H, offset = r1 % (L - m + 1) inside the
// haplotype (never crossing the pads), and with probability sub_ppm / 1e6 one substitution to a different base.
namespace {
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_sample_reads(const uint8_t *__restrict__ text, const uint64_t unit, const uint64_t H,
                                                      const uint64_t L, const uint64_t m, const uint64_t seed, const uint64_t first,
                                                      const uint64_t N, const uint32_t sub_ppm, uint8_t *__restrict__ seqs,
                                                      uint64_t *__restrict__ off, uint64_t *__restrict__ start_out) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    for (uint64_t i = wave; i < N; i += nwaves) {  // one wave copies one read: coalesced loads and stores
        const uint64_t g = first + i;
        const uint64_t r0 = splitmix64(seed ^ (g * 0xD1B54A32D192ED03ull));
        const uint64_t r1 = splitmix64(r0), r2 = splitmix64(r1), r3 = splitmix64(r2);
        const uint64_t start = (r0 % H) * unit + r1 % (L - m + 1);
        const bool mutate = (r2 % 1000000ull) < sub_ppm;
        const uint64_t mpos = (r2 >> 32) % m;
        for (uint64_t j = lane; j < m; j += kWave) {
            uint32_t c = text[start + j];
            if (mutate && j == mpos) {
                const uint32_t code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
                c = "ACGT"[(code + 1u + static_cast<uint32_t>(r3 % 3)) & 3u];
            }
            seqs[i * m + j] = static_cast<uint8_t>(c);
        }
        if (lane == 0) {
            off[i] = i * m;
            if (i + 1 == N) off[N] = N * m;
            if (start_out) start_out[i] = start;
        }
    }
}
// the same reads from the pangenome's STRUCTURE instead of its text (n = 3e11 symbols do not fit the HBM they would be
// sampled from): symbol (h, p) of the text is base[p], or alt[j] when p is variant site j and haplotype h carries the
// alternative allele (G[j * H + h] != 0).  Read g is the same function of (seed, g) as above.
constexpr int kPgMaxSites = 12;   // variant sites of one read kept in LDS (1.5 on average at a site rate of 1 %); a read with more takes the direct path
__global__ __launch_bounds__(256) void k_sample_reads_pg(const uint8_t *__restrict__ base, const uint64_t *__restrict__ sites, const uint8_t *__restrict__ alt,
                                                         const uint8_t *__restrict__ G, const uint64_t S, const uint32_t *__restrict__ site_dir,
                                                         const uint32_t site_dir_shift, const uint64_t unit, const uint64_t H,
                                                         const uint64_t L, const uint64_t m, const uint64_t seed, const uint64_t first,
                                                         const uint64_t N, const uint32_t sub_ppm, uint8_t *__restrict__ seqs,
                                                         uint64_t *__restrict__ off, uint64_t *__restrict__ start_out) {
    // Two phases per 64 reads of a wave.  (1) every LANE looks up ITS read: haplotype, offset, the variant sites inside it and
    // whether the haplotype carries them -- a handful of dependent loads, made for 64 reads at once instead of one after the
    // other (the first version walked one read per wave through them: 7 microseconds of latency per read, 9.6 ms per 10 M).
    // (2) the wave copies the 64 reads one after the other, coalesced, patching the sites from LDS.
    __shared__ uint64_t s_p0[4][kWave];
    __shared__ uint32_t s_mut[4][kWave];                      // position of the substitution | 0x80000000, or 0
    __shared__ uint8_t s_mc[4][kWave];                        // r3 % 3
    __shared__ uint32_t s_ns[4][kWave];
    __shared__ uint32_t s_so[4][kWave][kPgMaxSites];          // offset in the read << 8 | the symbol there
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
    for (uint64_t i0 = wave * kWave; i0 < N; i0 += nwaves * kWave) {
        const uint64_t i = i0 + lane;
        if (i < N) {
            const uint64_t g = first + i;
            const uint64_t r0 = splitmix64(seed ^ (g * 0xD1B54A32D192ED03ull));
            const uint64_t r1 = splitmix64(r0), r2 = splitmix64(r1), r3 = splitmix64(r2);
            const uint64_t h = r0 % H, p0 = r1 % (L - m + 1);
            const bool mutate = (r2 % 1000000ull) < sub_ppm;
            uint64_t a = 0, z = S;                         // first site at or after p0 (site_dir[b] = # sites below b << shift, when given)
            if (site_dir) { a = site_dir[p0 >> site_dir_shift]; z = site_dir[(p0 >> site_dir_shift) + 1]; }
            while (a < z) { const uint64_t mid = a + ((z - a) >> 1); if (sites[mid] < p0) a = mid + 1; else z = mid; }
            uint32_t ns = 0;
            for (uint64_t t = a; t < S; ++t) {
                const uint64_t sp = sites[t];
                if (sp >= p0 + m) break;
                if (G[t * H + h]) {
                    if (ns < static_cast<uint32_t>(kPgMaxSites)) s_so[wv][lane][ns] = (static_cast<uint32_t>(sp - p0) << 8) | alt[t];
                    ++ns;
                }
            }
            s_p0[wv][lane] = p0;
            s_mut[wv][lane] = mutate ? (static_cast<uint32_t>((r2 >> 32) % m) | 0x80000000u) : 0u;
            s_mc[wv][lane] = static_cast<uint8_t>(r3 % 3);
            s_ns[wv][lane] = ns;
            off[i] = i * m;
            if (i + 1 == N) off[N] = N * m;
            if (start_out) start_out[i] = h * unit + p0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t cnt = static_cast<uint32_t>(N - i0 < static_cast<uint64_t>(kWave) ? N - i0 : kWave);
        for (uint32_t q = 0; q < cnt; ++q) {
            const uint64_t p0 = s_p0[wv][q];
            const uint32_t mut = s_mut[wv][q], ns = s_ns[wv][q];
            for (uint64_t j = lane; j < m; j += kWave) {
                uint32_t c = base[p0 + j];
                if (ns <= static_cast<uint32_t>(kPgMaxSites)) {
                    for (uint32_t t = 0; t < ns; ++t) {
                        const uint32_t so = s_so[wv][q][t];
                        if ((so >> 8) == j) c = so & 0xFFu;
                    }
                } else {   // more sites than the list holds: look them up again (every lane for its own symbol)
                    const uint64_t g = first + i0 + q;
                    const uint64_t h = splitmix64(seed ^ (g * 0xD1B54A32D192ED03ull)) % H;
                    uint64_t a = 0, z = S;
                    while (a < z) { const uint64_t mid = a + ((z - a) >> 1); if (sites[mid] < p0 + j) a = mid + 1; else z = mid; }
                    if (a < S && sites[a] == p0 + j && G[a * H + h]) c = alt[a];
                }
                if ((mut & 0x80000000u) && j == (mut & 0x7FFFFFFFu)) {
                    const uint32_t code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
                    c = "ACGT"[(code + 1u + s_mc[wv][q]) & 3u];
                }
                seqs[(i0 + q) * m + j] = static_cast<uint8_t>(c);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
}  // namespace

int launch_sample_reads_pg(const uint8_t *base, const uint64_t *sites, const uint8_t *alt, const uint8_t *G, uint64_t S, const uint32_t *site_dir, uint32_t site_dir_shift,
                           uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first, uint64_t N, uint32_t sub_ppm, uint8_t *seqs, uint64_t *off,
                           uint64_t *start_out, void *stream) {
    if (N == 0) return 0;
    const uint64_t blocks = std::min<uint64_t>((N + 255) / 256, 256ull * 16);   // (a wave takes 64 reads at a time)
    hipLaunchKernelGGL(k_sample_reads_pg, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), base, sites, alt, G, S, site_dir,
                       site_dir_shift, unit, H, L, m,
                       seed, first, N, sub_ppm, seqs, off, start_out);
    return static_cast<int>(hipGetLastError());
}

int launch_sample_reads(const uint8_t *text, uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first,
                        uint64_t N, uint32_t sub_ppm, uint8_t *seqs, uint64_t *off, uint64_t *start_out, void *stream) {
    if (N == 0) return 0;
    const uint64_t blocks = std::min<uint64_t>((N + 3) / 4, 256ull * 64);
    hipLaunchKernelGGL(k_sample_reads, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), text, unit, H, L, m,
                       seed, first, N, sub_ppm, seqs, off, start_out);
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
