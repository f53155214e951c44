// k_compose.hip -- the k-mer tables of DESIGN.md 2b composed ON THE DEVICE at load time.
//
// A depth-(d+1) table T[x_d .. x_1 x_0] lists, as runs, the rows whose d + 1 preceding text characters spell the k-mer.
// It follows from depth d by one sweep per symbol c: LF-map every c-run of the BWT onto the row-ordered segmentation G_d
// of [0, n) by depth-d k-mer and cut it at the segment boundaries; every piece is one depth-(d+1) run, and the SA value at
// its last row is either the sample of the G_d segment it ends with or the c-run's own sample minus d (rbg_host.cpp
// compose() is the same statement as serial host code and stays as the reference: RBG_HOST_COMPOSE=1, and indexes built
// without a device).  On the host these sweeps were 4.1 of the 8.2 s a load of the bench index took and 33 of 69 s at
// n = 5e10; they are merges, sorts and scans over O(r) elements, i.e. what a GPU does at memory speed:
//   per symbol:  boundaries = run images (Q_k = F_c + cum[k]) merged with the G boundaries inside the symbol's image
//                (merge by mutual binary-search ranks, duplicates dropped) -> pieces {row start, length, sample, table}
//   per level:   stable radix sort of the pieces by table -> run lists in table order, cum by one scan; radix sort of
//                the pieces by row -> G_{d+1} (gaps = rows whose context leaves the k-mer alphabet).
// Outputs per level: the tables' {start, cum} pairs back to back in table order, each table closed by its sentinel
// {n, total} -- the array shape both layouts consume (DevSym::ent slices / DevTree::ent) -- their run-end samples, and per
// table (runs, total, F).
#include "rbg_device.hpp"

#include <chrono>
#include <cstdlib>
#include <vector>

namespace rbg {
namespace {

constexpr uint32_t kNoTab = 0xFFFFFFFFu;

#define CK(expr)                                                                         \
    do {                                                                                 \
        const hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "rbg: %s failed in compose: %s\n", #expr, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? -5 : -3;                                  \
        }                                                                                \
    } while (0)

// The sweeps' temporaries come from a pool that lives for one composition: a block that is given back is handed out again
// to the next request it fits (blocks are allocated a quarter larger than asked, so the next depth's slightly longer arrays
// fit the previous depth's blocks), and everything goes back to the driver at the end.  hipMalloc / hipFree of multi-GB
// blocks stall unpredictably on this platform (tools/alloc_probe.py: seconds), and the first version of this file spent
// 4.5 of its 5 s at n = 5e10 in them.
struct Pool {
    struct Blk { void *p; size_t bytes; bool used; };
    std::vector<Blk> blks;
    ~Pool() { for (Blk &b : blks) (void)hipFree(b.p); }
    hipError_t get(size_t bytes, void **out) {
        if (bytes == 0) bytes = 16;
        Blk *best = nullptr;
        for (Blk &b : blks)
            if (!b.used && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
        if (best) { best->used = true; *out = best->p; return hipSuccess; }
        // (no idle block is large enough.  Before the pool grows by a big block the idle ones go back to the driver: the depths' requests
        //  grow from depth to depth, so the previous depth's blocks would never be handed out again -- they were 60-90 GB of the 230 GB the
        //  composition held at r = 5e8)
        if (bytes >= (size_t(256) << 20)) purge();
        const size_t want = bytes + (bytes < (size_t(1) << 30) ? bytes / 4 : bytes / 64) + 256;   // (big blocks: the slack is gigabytes at r = 1e9)
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {   // give the idle blocks back and ask for exactly what is needed
            (void)hipGetLastError();
            for (size_t i = 0; i < blks.size();)
                if (!blks[i].used) { (void)hipFree(blks[i].p); blks.erase(blks.begin() + static_cast<std::ptrdiff_t>(i)); } else ++i;
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return e;
            blks.push_back({p, bytes, true});
        } else {
            blks.push_back({p, want, true});
        }
        *out = p;
        return hipSuccess;
    }
    void put(void *p) {
        for (Blk &b : blks)
            if (b.p == p) { b.used = false; return; }
    }
    void purge() {   // idle blocks back to the driver (before an allocation outside the pool that may need their space)
        for (size_t i = 0; i < blks.size();)
            if (!blks[i].used) { (void)hipFree(blks[i].p); blks.erase(blks.begin() + static_cast<std::ptrdiff_t>(i)); } else ++i;
    }
};
thread_local Pool *t_pool = nullptr;

// RAII device temporary (from the composition's pool)
struct Tmp {
    void *p = nullptr;
    ~Tmp() { release(); }
    hipError_t alloc(size_t bytes) { release(); return t_pool->get(bytes, &p); }
    void release() { if (p) t_pool->put(p); p = nullptr; }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

template <typename T>
__device__ __forceinline__ uint64_t lower_bound_dev(const T *a, uint64_t n, uint64_t v) {   // # elements < v
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (static_cast<uint64_t>(a[mid]) < v) lo = mid + 1; else hi = mid; }
    return lo;
}
template <typename T>
__device__ __forceinline__ uint64_t upper_bound_dev(const T *a, uint64_t n, uint64_t v) {   // # elements <= v
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (static_cast<uint64_t>(a[mid]) <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
// run image k starts at row F + cum[k]; # images starting <= v (binary search over the cum column)
template <typename P>
__device__ __forceinline__ uint64_t images_le(const RunEnt<P> *ent, uint64_t nruns, uint64_t F, uint64_t v) {
    uint64_t lo = 0, hi = nruns;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (F + static_cast<uint64_t>(ent[mid].cum) <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

// 1. the G boundaries strictly inside the symbol's image: is B_j = g_start[g_lo + 1 + j] also a run-image start?
template <typename P>
__global__ __launch_bounds__(256) void k_flag_bounds(const RunEnt<P> *__restrict__ ent, const uint64_t nruns, const uint64_t F,
                                                     const P *__restrict__ g_start, const uint64_t g_lo, const uint64_t nB,
                                                     uint32_t *__restrict__ keep /* nB + 1 */) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j <= nB; j += stride) {
        if (j == nB) { keep[j] = 0; continue; }
        const uint64_t b = g_start[g_lo + 1 + j];
        const uint64_t la = images_le<P>(ent, nruns, F, b);
        keep[j] = (la > 0 && F + static_cast<uint64_t>(ent[la - 1].cum) == b) ? 0u : 1u;
    }
}

// 2. the merged boundary list of one symbol: boundary r -> (row of the image space, run k, segment g)
template <typename P>
__global__ __launch_bounds__(256) void k_merge_bounds(const RunEnt<P> *__restrict__ ent, const uint64_t nruns, const uint64_t F,
                                                      const P *__restrict__ g_start, const uint64_t g_lo, const uint64_t nB,
                                                      const uint32_t *__restrict__ keep, const uint64_t *__restrict__ posB /* exclusive scan of keep, nB + 1 */,
                                                      uint64_t *__restrict__ bnd, uint32_t *__restrict__ bk, uint32_t *__restrict__ bg) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const P *B = g_start + g_lo + 1;
    for (uint64_t x = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < nruns + nB; x += stride) {
        if (x < nruns) {
            const uint64_t k = x, b = F + static_cast<uint64_t>(ent[k].cum);
            const uint64_t j0 = lower_bound_dev<P>(B, nB, b);          // # G boundaries below b (none of them equals a later image start)
            const uint64_t j1 = upper_bound_dev<P>(B, nB, b);          // # G boundaries at or below b: the segment that holds row b
            const uint64_t r = k + posB[j0];
            bnd[r] = b; bk[r] = static_cast<uint32_t>(k); bg[r] = static_cast<uint32_t>(g_lo + j1);
        } else {
            const uint64_t j = x - nruns;
            if (!keep[j]) continue;
            const uint64_t b = B[j];
            const uint64_t la = images_le<P>(ent, nruns, F, b);         // all of them start strictly below b
            const uint64_t r = la + posB[j];
            bnd[r] = b; bk[r] = static_cast<uint32_t>(la - 1); bg[r] = static_cast<uint32_t>(g_lo + 1 + j);
        }
    }
}

// 3. boundaries -> pieces of symbol m (appended at `base` of the level's piece arrays)
template <typename P>
__global__ __launch_bounds__(256) void k_make_pieces(const RunEnt<P> *__restrict__ ent, const P *__restrict__ samp, const uint64_t F,
                                                     const uint64_t total, const P *__restrict__ g_start, const uint32_t *__restrict__ g_id,
                                                     const P *__restrict__ g_samp, const uint64_t *__restrict__ bnd, const uint32_t *__restrict__ bk,
                                                     const uint32_t *__restrict__ bg, const uint64_t np, const uint32_t depth, const uint32_t M,
                                                     const uint32_t m, const uint32_t T, const bool with_samples, uint32_t *__restrict__ p_tab,
                                                     P *__restrict__ p_start, P *__restrict__ p_len, P *__restrict__ p_samp, int *__restrict__ err) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < np; r += stride) {
        const uint64_t b = bnd[r], e = r + 1 < np ? bnd[r + 1] : F + total;
        const uint32_t k = bk[r], g = bg[r];
        const uint32_t id = g_id[g];
        const uint64_t Q = F + static_cast<uint64_t>(ent[k].cum);
        p_start[r] = static_cast<P>(static_cast<uint64_t>(ent[k].start) + (b - Q));
        p_len[r] = static_cast<P>(e - b);
        p_tab[r] = id == kNoTab ? T : id * M + m;       // T: rows whose context leaves the k-mer alphabet (sorted to the end, dropped)
        if (with_samples) {
            uint64_t v = 0;
            if (id != kNoTab) {
                if (e == static_cast<uint64_t>(g_start[g + 1])) v = g_samp[g];
                else {
                    const uint64_t sv = samp[k];
                    if (sv < depth) *err = 1;           // would need the terminator inside the k-mer
                    v = sv - depth;
                }
            }
            p_samp[r] = static_cast<P>(v);
        }
    }
}

// 4. first index of every table's pieces in the table-sorted order (T + 2 entries: [T] = first dropped piece, [T + 1] = np)
__global__ __launch_bounds__(256) void k_table_firsts(const uint32_t *__restrict__ sorted_tab, const uint64_t np, const uint32_t T,
                                                      uint64_t *__restrict__ first) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > T + 1) return;
    first[t] = t == T + 1 ? np : lower_bound_dev<uint32_t>(sorted_tab, np, t);
}

template <typename P>
__global__ __launch_bounds__(256) void k_gather_len(const uint32_t *__restrict__ perm, const P *__restrict__ p_len, const uint64_t np,
                                                    uint64_t *__restrict__ out /* np + 1; out[np] = 0 */) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j <= np; j += stride) out[j] = j < np ? static_cast<uint64_t>(p_len[perm[j]]) : 0;
}

// 5. the level's arrays: table t's runs at [first[t] + t, first[t + 1] + t), its sentinel behind them
template <typename P>
__global__ __launch_bounds__(256) void k_write_tables(const uint32_t *__restrict__ sorted_tab, const uint32_t *__restrict__ perm,
                                                      const uint64_t *__restrict__ first, const uint64_t *__restrict__ cumlen /* exclusive scan, np + 1 */,
                                                      const P *__restrict__ p_start, const P *__restrict__ p_samp, const uint64_t nkept, const uint32_t T,
                                                      const uint64_t n, const bool with_samples, RunEnt<P> *__restrict__ ent, P *__restrict__ samp) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < nkept + T; j += stride) {
        if (j < nkept) {
            const uint32_t t = sorted_tab[j], src = perm[j];
            RunEnt<P> e;
            e.start = p_start[src];
            e.cum = static_cast<P>(cumlen[j] - cumlen[first[t]]);
            ent[j + t] = e;
            if (with_samples) samp[j + t] = p_samp[src];
        } else {
            const uint32_t t = static_cast<uint32_t>(j - nkept);
            RunEnt<P> e;
            e.start = static_cast<P>(n);
            e.cum = static_cast<P>(cumlen[first[t + 1]] - cumlen[first[t]]);
            ent[first[t + 1] + t] = e;
            if (with_samples) samp[first[t + 1] + t] = 0;
        }
    }
}

// 6. F_{d+1}[id, m] = F_d[id] + rank_d(F[m], id): rows of the id-interval followed by a smaller symbol (prev table id's run list)
template <typename P>
__global__ __launch_bounds__(256) void k_table_F(const RunEnt<P> *const *__restrict__ prev_ent, const uint64_t *__restrict__ prev_nruns,
                                                 const uint64_t *__restrict__ prev_F, const uint64_t *__restrict__ sym_F, const uint32_t n_ids,
                                                 const uint32_t M, uint64_t *__restrict__ out_F) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ids * M) return;
    const uint32_t id = t / M, m = t % M;
    const RunEnt<P> *e = prev_ent[id];
    const uint64_t nr = prev_nruns[id], i = sym_F[m];
    uint64_t lo = 0, hi = nr;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (static_cast<uint64_t>(e[mid].start) < i) lo = mid + 1; else hi = mid; }
    uint64_t rk = 0;
    if (lo > 0) {
        const uint64_t d = i - static_cast<uint64_t>(e[lo - 1].start), len = static_cast<uint64_t>(e[lo].cum) - static_cast<uint64_t>(e[lo - 1].cum);
        rk = static_cast<uint64_t>(e[lo - 1].cum) + (d < len ? d : len);
    }
    out_F[t] = prev_F[id] + rk;
}

// 7. the next segmentation from the kept pieces in row order: a gap segment (no table) wherever rows are missing
template <typename P>
__global__ __launch_bounds__(256) void k_gap_flags(const uint32_t *__restrict__ rperm, const P *__restrict__ p_start, const P *__restrict__ p_len,
                                                   const uint64_t nkept, uint64_t *__restrict__ gap /* nkept + 1 */, const uint64_t n) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q <= nkept; q += stride) {
        const uint64_t prev_end = q ? static_cast<uint64_t>(p_start[rperm[q - 1]]) + static_cast<uint64_t>(p_len[rperm[q - 1]]) : 0;
        const uint64_t here = q < nkept ? static_cast<uint64_t>(p_start[rperm[q]]) : n;
        gap[q] = here > prev_end ? 1 : 0;
    }
}
template <typename P>
__global__ __launch_bounds__(256) void k_write_segments(const uint32_t *__restrict__ rperm, const uint32_t *__restrict__ p_tab, const P *__restrict__ p_start,
                                                        const P *__restrict__ p_len, const P *__restrict__ p_samp, const uint64_t nkept,
                                                        const uint64_t *__restrict__ gapx /* exclusive scan of gap, nkept + 2 */, const uint64_t n,
                                                        const bool with_samples, P *__restrict__ g_start, uint32_t *__restrict__ g_id, P *__restrict__ g_samp) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q <= nkept; q += stride) {
        const uint64_t prev_end = q ? static_cast<uint64_t>(p_start[rperm[q - 1]]) + static_cast<uint64_t>(p_len[rperm[q - 1]]) : 0;
        const uint64_t at = q + gapx[q];
        const bool gap = gapx[q + 1] != gapx[q];
        if (gap) { g_start[at] = static_cast<P>(prev_end); g_id[at] = kNoTab; if (with_samples) g_samp[at] = 0; }
        if (q < nkept) {
            const uint32_t src = rperm[q];
            const uint64_t o = at + (gap ? 1 : 0);
            g_start[o] = p_start[src]; g_id[o] = p_tab[src]; if (with_samples) g_samp[o] = p_samp[src];
        } else {
            g_start[at + (gap ? 1 : 0)] = static_cast<P>(n);   // sentinel
        }
    }
}

__global__ __launch_bounds__(256) void k_widen_flags(const uint32_t *__restrict__ in, const uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_gather_u64(const uint64_t *__restrict__ src, const uint64_t *__restrict__ idx, const uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
template <typename P>
__global__ __launch_bounds__(256) void k_gather_pos(const uint32_t *__restrict__ perm, const P *__restrict__ src, const uint64_t n, P *__restrict__ dst) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = src[perm[j]];
}
__global__ __launch_bounds__(256) void k_iota32(uint32_t *__restrict__ v, const uint64_t N) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) v[i] = static_cast<uint32_t>(i);
}

inline dim3 grid_n(uint64_t n) { return dim3(static_cast<unsigned>(std::min<uint64_t>((n + 255) / 256 ? (n + 255) / 256 : 1, 256ull * 64))); }

int exclusive_scan_u64(const uint64_t *in, uint64_t *out, uint64_t n, hipStream_t st) {
    size_t bytes = 0;
    CK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, static_cast<int64_t>(n), st));
    Tmp t;
    CK(t.alloc(bytes));
    CK(hipcub::DeviceScan::ExclusiveSum(t.p, bytes, in, out, static_cast<int64_t>(n), st));
    CK(hipStreamSynchronize(st));
    return 0;
}

template <typename K>
int sort_pairs(const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, uint64_t n, int end_bit, hipStream_t st) {
    size_t bytes = 0;
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, static_cast<int64_t>(n), 0, end_bit, st));
    Tmp t;
    CK(t.alloc(bytes));
    CK(hipcub::DeviceRadixSort::SortPairs(t.p, bytes, kin, kout, vin, vout, static_cast<int64_t>(n), 0, end_bit, st));
    CK(hipStreamSynchronize(st));
    return 0;
}

int bits_for(uint64_t v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

template <typename P>
int compose_impl(const uint64_t n, const uint32_t M, const ComposeTable *major, const void *g1_start, const uint32_t *g1_id, const void *g1_samp,
                 const uint64_t g1_n, const uint32_t kmax, const bool with_samples, std::vector<ComposedLevel> &out, hipStream_t st, const uint32_t keep_mask,
                 bool *inputs_released) {
    out.clear();
    // the segmentation of the current depth (depth 1: the caller's arrays, not owned)
    const P *g_start = static_cast<const P *>(g1_start);
    const uint32_t *g_id = g1_id;
    const P *g_samp = static_cast<const P *>(g1_samp);
    uint64_t g_n = g1_n;
    Tmp own_start, own_id, own_samp;
    // the previous depth's tables (for F): depth 1 = the major symbols' own tables
    std::vector<const void *> prev_ent(M);
    std::vector<uint64_t> prev_nruns(M), prev_F(M), sym_F(M);
    for (uint32_t m = 0; m < M; ++m) { prev_ent[m] = major[m].ent; prev_nruns[m] = major[m].nruns; prev_F[m] = major[m].F; sym_F[m] = major[m].F; }
    Tmp d_err;
    CK(d_err.alloc(sizeof(int)));
    CK(hipMemsetAsync(d_err.p, 0, sizeof(int), st));
    uint32_t n_ids = M;
    Tmp d_F_next;
    bool have_F_next = false;
    // kept depths that went to host memory while deeper ones were composed (see spill_kept below)
    struct Spilled { size_t level; void *h_ent, *h_samp; size_t ent_bytes, samp_bytes; };
    std::vector<Spilled> spilled;
    struct SpillGuard { std::vector<Spilled> &v; ~SpillGuard() { for (Spilled &x : v) { std::free(x.h_ent); std::free(x.h_samp); } } } spill_guard{spilled};
    const bool verbose = std::getenv("RBG_VERBOSE") != nullptr;
    auto used_gb = [] { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return static_cast<double>(t - f) / 1e9; };
    double gb_pieces = 0, gb_tables = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    for (uint32_t depth = 1; depth < kmax; ++depth, n_ids *= M) {
        const uint32_t T = n_ids * M;
        const auto t_lvl = now();
        // ---- F of the new tables first: it is the only thing that reads the PREVIOUS depth's run lists, and a depth the caller
        // does not keep (keep_mask: the run-indexed layout's depth set) gives its arrays back before this depth's temporaries
        // are made -- at r = 1e9 that is 50 GB of the 300 the composition would otherwise want at once
        Tmp d_F;
        if (have_F_next) {   // computed at the end of the previous iteration, before that depth's arrays could be given back
            std::swap(d_F.p, d_F_next.p);
            have_F_next = false;
        } else {
            Tmp d_prev_ent, d_prev_nruns, d_prev_F, d_sym_F;
            CK(d_prev_ent.alloc(n_ids * 8)); CK(d_prev_nruns.alloc(n_ids * 8)); CK(d_prev_F.alloc(n_ids * 8)); CK(d_sym_F.alloc(M * 8)); CK(d_F.alloc(T * 8));
            CK(hipMemcpy(d_prev_ent.p, prev_ent.data(), n_ids * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_prev_nruns.p, prev_nruns.data(), n_ids * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_prev_F.p, prev_F.data(), n_ids * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_sym_F.p, sym_F.data(), M * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_table_F<P>), dim3((T + 255) / 256), dim3(256), 0, st, d_prev_ent.as<const RunEnt<P> *>(), d_prev_nruns.as<uint64_t>(),
                               d_prev_F.as<uint64_t>(), d_sym_F.as<uint64_t>(), n_ids, M, d_F.as<uint64_t>());
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
        }
        if (depth >= 2 && keep_mask && !(keep_mask >> (depth - 1) & 1u)) {   // depth `depth` = out[depth - 2]: not kept, no longer read
            ComposedLevel &old = out[depth - 2];
            if (old.ent) (void)hipFree(old.ent);
            if (old.samp) (void)hipFree(old.samp);
            old.ent = old.samp = nullptr;
        }
        const bool last_level = depth + 1 == kmax;
        // SPILL: the sweeps of this depth take about 80 bytes per piece beside what is resident; when that is more than the device has
        // free, the kept depths made so far (24 bytes per entry, read by nothing until the composition is over) wait in host memory and
        // come back at the end.  r = 1e9: depth 3's 56 GB, without which depths 4 and 5 do not fit beside their own temporaries.
        if (depth >= 2) {
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            const double need = 84.0 * 1.3 * static_cast<double>(g_n);
            if (need > static_cast<double>(free_b)) {
                t_pool->purge();
                (void)hipMemGetInfo(&free_b, &total_b);
            }
            if (need > static_cast<double>(free_b) || std::getenv("RBG_COMPOSE_SPILL")) {   // (the switch: tests take the path at their size)
                for (size_t li = 0; li < out.size(); ++li) {
                    ComposedLevel &L = out[li];
                    if (!L.ent) continue;
                    const size_t eb = (L.entries + 2) * sizeof(RunEnt<P>), sb = L.samp ? (L.entries + 2) * sizeof(P) : 0;
                    if (static_cast<double>(eb + sb) > 0.5 * host_memory_available()) break;   // (the host must not be what runs out instead)
                    Spilled x{li, std::malloc(eb), sb ? std::malloc(sb) : nullptr, eb, sb};
                    if (!x.h_ent || (sb && !x.h_samp)) { std::free(x.h_ent); std::free(x.h_samp); break; }
                    spilled.push_back(x);
                    CK(hipMemcpy(x.h_ent, L.ent, eb, hipMemcpyDeviceToHost));
                    if (sb) CK(hipMemcpy(x.h_samp, L.samp, sb, hipMemcpyDeviceToHost));
                    (void)hipFree(L.ent);
                    if (L.samp) (void)hipFree(L.samp);
                    L.ent = L.samp = nullptr;
                    if (verbose) std::fprintf(stderr, "rbg:   depth %zu (%.1f GB) waits in host memory while depth %u is composed\n", li + 2, (eb + sb) / 1e9, depth + 1);
                }
            }
        }
        // ---- pieces of every symbol -------------------------------------------------------------------------------------
        // which G segments a symbol's image [F, F + total) meets: two binary searches per symbol, made from the host over the
        // device array (a few dozen 8-byte copies per level)
        std::vector<uint64_t> g_lo(M), nB(M), np_m(M), base(M + 1, 0);
        // (the keep flags and their scan -- 12 bytes per segment boundary -- are made twice per symbol, once to count and once to merge, instead
        //  of being held for all symbols from the first pass to the second: 26 GB at r = 1e9)
        std::vector<Tmp> keep(M), posB(M);
        for (uint32_t m = 0; m < M; ++m) {
            const ComposeTable &tc = major[m];
            if (tc.nruns == 0 || tc.total == 0) { g_lo[m] = 0; nB[m] = 0; np_m[m] = 0; base[m + 1] = base[m]; continue; }
            // g_lo = last segment starting <= F; g_hi = last segment starting <= F + total - 1 (host-side binary search over the device array)
            auto seg_of = [&](uint64_t row, uint64_t *res) -> int {
                uint64_t lo = 0, hi = g_n;     // # starts <= row, minus one
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    P v;
                    CK(hipMemcpy(&v, g_start + mid, sizeof(P), hipMemcpyDeviceToHost));
                    if (static_cast<uint64_t>(v) <= row) lo = mid + 1; else hi = mid;
                }
                *res = lo ? lo - 1 : 0;
                return 0;
            };
            uint64_t ghi = 0;
            int rc = seg_of(tc.F, &g_lo[m]);
            if (rc) return rc;
            if ((rc = seg_of(tc.F + tc.total - 1, &ghi))) return rc;
            nB[m] = ghi - g_lo[m];
            CK(keep[m].alloc((nB[m] + 1) * 4));
            CK(posB[m].alloc((nB[m] + 2) * 8));
            hipLaunchKernelGGL((k_flag_bounds<P>), grid_n(nB[m] + 1), dim3(256), 0, st, static_cast<const RunEnt<P> *>(tc.ent), tc.nruns, tc.F, g_start,
                               g_lo[m], nB[m], keep[m].as<uint32_t>());
            CK(hipGetLastError());
            // exclusive scan of the keep flags -> rank of every kept G boundary among the kept ones
            {
                Tmp k64;
                CK(k64.alloc((nB[m] + 1) * 8));
                hipLaunchKernelGGL(k_widen_flags, grid_n(nB[m] + 1), dim3(256), 0, st, keep[m].as<uint32_t>(), nB[m] + 1, k64.as<uint64_t>());
                CK(hipGetLastError());
                const int rc2 = exclusive_scan_u64(k64.as<uint64_t>(), posB[m].as<uint64_t>(), nB[m] + 1, st);
                if (rc2) return rc2;
            }
            uint64_t kept = 0;
            CK(hipMemcpy(&kept, posB[m].as<uint64_t>() + nB[m], 8, hipMemcpyDeviceToHost));
            np_m[m] = tc.nruns + kept;
            base[m + 1] = base[m] + np_m[m];
            keep[m].release();
            posB[m].release();
        }
        const uint64_t np = base[M];
        if (np >= 0xFFFFFFF0ull) return -4;   // piece and run ordinals are 32-bit
        Tmp p_tab, p_start, p_len, p_samp;
        CK(p_tab.alloc((np + 1) * 4));
        CK(p_start.alloc((np + 1) * sizeof(P)));
        CK(p_len.alloc((np + 1) * sizeof(P)));
        if (with_samples) CK(p_samp.alloc((np + 1) * sizeof(P)));
        for (uint32_t m = 0; m < M; ++m) {
            if (np_m[m] == 0) continue;
            const ComposeTable &tc = major[m];
            {   // the symbol's keep flags and their scan again (see above)
                CK(keep[m].alloc((nB[m] + 1) * 4));
                CK(posB[m].alloc((nB[m] + 2) * 8));
                hipLaunchKernelGGL((k_flag_bounds<P>), grid_n(nB[m] + 1), dim3(256), 0, st, static_cast<const RunEnt<P> *>(tc.ent), tc.nruns, tc.F, g_start,
                                   g_lo[m], nB[m], keep[m].as<uint32_t>());
                CK(hipGetLastError());
                Tmp k64;
                CK(k64.alloc((nB[m] + 1) * 8));
                hipLaunchKernelGGL(k_widen_flags, grid_n(nB[m] + 1), dim3(256), 0, st, keep[m].as<uint32_t>(), nB[m] + 1, k64.as<uint64_t>());
                CK(hipGetLastError());
                const int rc2 = exclusive_scan_u64(k64.as<uint64_t>(), posB[m].as<uint64_t>(), nB[m] + 1, st);
                if (rc2) return rc2;
            }
            Tmp bnd, bk, bg;
            CK(bnd.alloc(np_m[m] * 8));
            CK(bk.alloc(np_m[m] * 4));
            CK(bg.alloc(np_m[m] * 4));
            hipLaunchKernelGGL((k_merge_bounds<P>), grid_n(tc.nruns + nB[m]), dim3(256), 0, st, static_cast<const RunEnt<P> *>(tc.ent), tc.nruns, tc.F, g_start,
                               g_lo[m], nB[m], keep[m].as<uint32_t>(), posB[m].as<uint64_t>(), bnd.as<uint64_t>(), bk.as<uint32_t>(), bg.as<uint32_t>());
            CK(hipGetLastError());
            hipLaunchKernelGGL((k_make_pieces<P>), grid_n(np_m[m]), dim3(256), 0, st, static_cast<const RunEnt<P> *>(tc.ent), static_cast<const P *>(tc.samp), tc.F,
                               tc.total, g_start, g_id, g_samp, bnd.as<uint64_t>(), bk.as<uint32_t>(), bg.as<uint32_t>(), np_m[m], depth, M, m, T, with_samples,
                               p_tab.as<uint32_t>() + base[m], p_start.as<P>() + base[m], p_len.as<P>() + base[m],
                               with_samples ? p_samp.as<P>() + base[m] : nullptr, d_err.as<int>());
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            keep[m].release();
            posB[m].release();
        }
        const auto t_pieces = now();
        if (verbose) gb_pieces = used_gb();
        own_start.release(); own_id.release(); own_samp.release();   // (the segmentation is read by the pieces only: the next one is made below)
        // inputs_released != nullptr: the caller hands over its inputs -- the depth-1 segmentation (20 bytes per run) is read by the
        // first depth's pieces only, the major symbols' own tables (24 bytes per run) by every depth's pieces and no later: 45 GB
        // at r = 1e9 that the last depth's tables need
        if (inputs_released && depth == 1) {
            (void)hipFree(const_cast<void *>(g1_start)); (void)hipFree(const_cast<uint32_t *>(g1_id));
            if (g1_samp) (void)hipFree(const_cast<void *>(g1_samp));
            inputs_released[0] = true;
        }
        if (inputs_released && last_level) {
            for (uint32_t m = 0; m < M; ++m) {
                if (major[m].ent) (void)hipFree(const_cast<void *>(major[m].ent));
                if (major[m].samp) (void)hipFree(const_cast<void *>(major[m].samp));
            }
            inputs_released[1] = true;
        }
        // ---- tables: stable sort by table, cum by one scan ------------------------------------------------------------------
        Tmp iota, s_tab, perm;
        CK(iota.alloc((np + 1) * 4));
        CK(s_tab.alloc((np + 1) * 4));
        CK(perm.alloc((np + 1) * 4));
        hipLaunchKernelGGL(k_iota32, grid_n(np), dim3(256), 0, st, iota.as<uint32_t>(), np);
        CK(hipGetLastError());
        int rc = sort_pairs<uint32_t>(p_tab.as<uint32_t>(), s_tab.as<uint32_t>(), iota.as<uint32_t>(), perm.as<uint32_t>(), np, bits_for(T), st);
        if (rc) return rc;
        iota.release();
        if (last_level) p_tab.release();   // (the next segmentation reads it)
        Tmp first;
        CK(first.alloc((T + 2) * 8));
        hipLaunchKernelGGL(k_table_firsts, dim3((T + 2 + 255) / 256), dim3(256), 0, st, s_tab.as<uint32_t>(), np, T, first.as<uint64_t>());
        CK(hipGetLastError());
        std::vector<uint64_t> h_first(T + 2);
        CK(hipMemcpy(h_first.data(), first.p, (T + 2) * 8, hipMemcpyDeviceToHost));
        const uint64_t nkept = h_first[T];
        Tmp lens, cumlen;
        CK(lens.alloc((np + 1) * 8));
        CK(cumlen.alloc((np + 1) * 8));
        hipLaunchKernelGGL((k_gather_len<P>), grid_n(np + 1), dim3(256), 0, st, perm.as<uint32_t>(), p_len.as<P>(), np, lens.as<uint64_t>());
        CK(hipGetLastError());
        if ((rc = exclusive_scan_u64(lens.as<uint64_t>(), cumlen.as<uint64_t>(), np + 1, st))) return rc;
        lens.release();
        if (last_level) p_len.release();
        t_pool->purge();   // (the level's own arrays come straight from the driver)
        ComposedLevel L;
        L.entries = nkept + T;
        void *d_ent = nullptr, *d_samp = nullptr;
        CK(hipMalloc(&d_ent, (L.entries + 2) * sizeof(RunEnt<P>)));    // (+ spare entries: the run-indexed kernels' two-entry loads may touch one past the last sentinel)
        L.ent = d_ent;
        if (with_samples) {
            const hipError_t e2 = hipMalloc(&d_samp, (L.entries + 2) * sizeof(P));
            if (e2 != hipSuccess) { (void)hipFree(d_ent); return e2 == hipErrorOutOfMemory ? -5 : -3; }
        }
        L.samp = d_samp;
        out.push_back(L);   // (from here on the caller owns the arrays, also on an error return)
        ComposedLevel &Lv = out.back();
        CK(hipMemsetAsync(static_cast<char *>(d_ent) + L.entries * sizeof(RunEnt<P>), 0xFF, 2 * sizeof(RunEnt<P>), st));
        hipLaunchKernelGGL((k_write_tables<P>), grid_n(nkept + T), dim3(256), 0, st, s_tab.as<uint32_t>(), perm.as<uint32_t>(), first.as<uint64_t>(),
                           cumlen.as<uint64_t>(), p_start.as<P>(), p_samp.as<P>(), nkept, T, n, with_samples, static_cast<RunEnt<P> *>(d_ent),
                           static_cast<P *>(d_samp));
        CK(hipGetLastError());
        // per table: runs, total, F
        std::vector<uint64_t> h_cum_at(T + 1);
        {
            Tmp d_cum_at;   // cumlen at the tables' first pieces
            CK(d_cum_at.alloc((T + 1) * 8));
            hipLaunchKernelGGL(k_gather_u64, dim3((T + 1 + 255) / 256), dim3(256), 0, st, cumlen.as<uint64_t>(), first.as<uint64_t>(), T + 1, d_cum_at.as<uint64_t>());
            CK(hipGetLastError());
            CK(hipMemcpy(h_cum_at.data(), d_cum_at.p, (T + 1) * 8, hipMemcpyDeviceToHost));
        }
        Lv.nruns.resize(T); Lv.total.resize(T); Lv.F.resize(T); Lv.first.resize(T);
        CK(hipMemcpy(Lv.F.data(), d_F.p, T * 8, hipMemcpyDeviceToHost));
        for (uint32_t t = 0; t < T; ++t) {
            Lv.nruns[t] = h_first[t + 1] - h_first[t];
            Lv.total[t] = h_cum_at[t + 1] - h_cum_at[t];
            Lv.first[t] = h_first[t] + t;
        }
        CK(hipStreamSynchronize(st));
        const auto t_tables = now();
        if (verbose) gb_tables = used_gb();
        // ---- the next depth's segmentation (not needed after the last level) ----------------------------------------------
        s_tab.release(); cumlen.release(); first.release();   // (the tables are written: 16 bytes per piece less under the segmentation's arrays)
        // the tables just made are the next level's "previous" ones: its F is computed NOW, so that a depth the caller does not keep gives its
        // arrays (24 bytes per entry) back before the segmentation's are made, not after
        prev_ent.assign(T, nullptr); prev_nruns.assign(T, 0); prev_F.assign(T, 0);
        for (uint32_t t = 0; t < T; ++t) {
            prev_ent[t] = static_cast<const char *>(Lv.ent) + Lv.first[t] * sizeof(RunEnt<P>);
            prev_nruns[t] = Lv.nruns[t];
            prev_F[t] = Lv.F[t];
        }
        if (depth + 1 < kmax) {
            const uint32_t T2 = T * M;
            Tmp d_prev_ent, d_prev_nruns, d_prev_F, d_sym_F;
            CK(d_prev_ent.alloc(T * 8)); CK(d_prev_nruns.alloc(T * 8)); CK(d_prev_F.alloc(T * 8)); CK(d_sym_F.alloc(M * 8)); CK(d_F_next.alloc(T2 * 8));
            CK(hipMemcpy(d_prev_ent.p, prev_ent.data(), T * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_prev_nruns.p, prev_nruns.data(), T * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_prev_F.p, prev_F.data(), T * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_sym_F.p, sym_F.data(), M * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_table_F<P>), dim3((T2 + 255) / 256), dim3(256), 0, st, d_prev_ent.as<const RunEnt<P> *>(), d_prev_nruns.as<uint64_t>(),
                               d_prev_F.as<uint64_t>(), d_sym_F.as<uint64_t>(), T, M, d_F_next.as<uint64_t>());
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            have_F_next = true;
            if (keep_mask && !(keep_mask >> depth & 1u)) {   // this depth (depth + 1) is not kept and nothing reads it any more
                (void)hipFree(Lv.ent);
                if (Lv.samp) (void)hipFree(Lv.samp);
                Lv.ent = Lv.samp = nullptr;
            }
        }
        if (depth + 1 < kmax) {
            Tmp rkeys, rperm;
            CK(rkeys.alloc((nkept + 1) * sizeof(P)));
            CK(rperm.alloc((nkept + 1) * 4));
            // kept pieces = perm[0 .. nkept): sort THEM by row start.  Keys = p_start gathered through perm.
            Tmp kin;
            CK(kin.alloc((nkept + 1) * sizeof(P)));
            hipLaunchKernelGGL((k_gather_pos<P>), grid_n(nkept), dim3(256), 0, st, perm.as<uint32_t>(), p_start.as<P>(), nkept, kin.as<P>());
            CK(hipGetLastError());
            if ((rc = sort_pairs<P>(kin.as<P>(), rkeys.as<P>(), perm.as<uint32_t>(), rperm.as<uint32_t>(), nkept, bits_for(n), st))) return rc;
            kin.release(); rkeys.release();
            Tmp gap, gapx;
            CK(gap.alloc((nkept + 2) * 8));
            CK(gapx.alloc((nkept + 2) * 8));
            hipLaunchKernelGGL((k_gap_flags<P>), grid_n(nkept + 1), dim3(256), 0, st, rperm.as<uint32_t>(), p_start.as<P>(), p_len.as<P>(), nkept, gap.as<uint64_t>(), n);
            CK(hipGetLastError());
            CK(hipMemsetAsync(gap.as<uint64_t>() + nkept + 1, 0, 8, st));
            if ((rc = exclusive_scan_u64(gap.as<uint64_t>(), gapx.as<uint64_t>(), nkept + 2, st))) return rc;
            uint64_t ngaps = 0;
            CK(hipMemcpy(&ngaps, gapx.as<uint64_t>() + nkept + 1, 8, hipMemcpyDeviceToHost));
            const uint64_t nseg = nkept + ngaps;
            Tmp n_start, n_id, n_samp;
            CK(n_start.alloc((nseg + 2) * sizeof(P)));
            CK(n_id.alloc((nseg + 2) * 4));
            if (with_samples) CK(n_samp.alloc((nseg + 2) * sizeof(P)));
            hipLaunchKernelGGL((k_write_segments<P>), grid_n(nkept + 1), dim3(256), 0, st, rperm.as<uint32_t>(), p_tab.as<uint32_t>(), p_start.as<P>(),
                               p_len.as<P>(), p_samp.as<P>(), nkept, gapx.as<uint64_t>(), n, with_samples, n_start.as<P>(), n_id.as<uint32_t>(), n_samp.as<P>());
            CK(hipGetLastError());
            CK(hipStreamSynchronize(st));
            std::swap(own_start.p, n_start.p);
            std::swap(own_id.p, n_id.p);
            std::swap(own_samp.p, n_samp.p);
            g_start = own_start.as<P>();
            g_id = own_id.as<uint32_t>();
            g_samp = own_samp.as<P>();
            g_n = nseg;
        }
        CK(hipStreamSynchronize(st));
        int h_err = 0;
        CK(hipMemcpy(&h_err, d_err.p, sizeof(int), hipMemcpyDeviceToHost));
        if (h_err) return -2;   // RBG_EFORMAT: a sample below the depth (the terminator inside a k-mer)
        if (verbose)
            std::fprintf(stderr, "rbg:   depth %u: %llu pieces, %llu runs in %u tables: pieces %.3f s, tables %.3f s, next segmentation %.3f s; HBM in use %.1f GB after the "
                                 "pieces, %.1f after the tables, %.1f at the end\n", depth + 1,
                         static_cast<unsigned long long>(np), static_cast<unsigned long long>(nkept), T, secs(t_lvl, t_pieces), secs(t_pieces, t_tables),
                         secs(t_tables, now()), gb_pieces, gb_tables, used_gb());
    }
    // the depths that waited in host memory come back (everything the sweeps held has been given back by now)
    if (!spilled.empty()) t_pool->purge();
    for (Spilled &x : spilled) {
        ComposedLevel &L = out[x.level];
        void *de = nullptr, *ds = nullptr;
        CK(hipMalloc(&de, x.ent_bytes));
        L.ent = de;
        CK(hipMemcpy(de, x.h_ent, x.ent_bytes, hipMemcpyHostToDevice));
        if (x.samp_bytes) {
            CK(hipMalloc(&ds, x.samp_bytes));
            L.samp = ds;
            CK(hipMemcpy(ds, x.h_samp, x.samp_bytes, hipMemcpyHostToDevice));
        }
    }
    return 0;
}

}  // namespace

// RBG_* codes: 0 ok, -2 EFORMAT, -3 ENODEV, -4 EARG, -5 ENOMEM (include/rbg.h)
int compose_levels_device(uint32_t pos_bytes, uint64_t n, uint32_t M, const ComposeTable *major, const void *g_start, const uint32_t *g_id,
                          const void *g_samp, uint64_t g_n, uint32_t kmax, bool with_samples, std::vector<ComposedLevel> &out, void *stream, uint32_t keep_mask,
                          bool *inputs_released) {
    if (inputs_released) inputs_released[0] = inputs_released[1] = false;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Pool pool;
    struct Scope { Pool *prev; explicit Scope(Pool *p) : prev(t_pool) { t_pool = p; } ~Scope() { t_pool = prev; } } scope(&pool);
    return pos_bytes == 4 ? compose_impl<uint32_t>(n, M, major, g_start, g_id, g_samp, g_n, kmax, with_samples, out, st, keep_mask, inputs_released)
                          : compose_impl<uint64_t>(n, M, major, g_start, g_id, g_samp, g_n, kmax, with_samples, out, st, keep_mask, inputs_released);
}

}  // namespace rbg
