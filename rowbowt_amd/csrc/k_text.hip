// k_text.hip -- the text of `rb_align -s` made on the device (rbg_align_text, include/rbg.h).
//
// rb_report (rb_align.cpp:118-145) prints per read
//     <name> (<lo>,<hi>), count=<hi - lo + 1>\n
//     \tlocs: <pos>/<doc>:<pos - doc start> ... \n            (locs_at + resolve_offset, rowbowt.hpp:613-625, doclist.hpp:46-79)
// -- 975 bytes per read on the bench index (41 locations), 9.75 GB per 10 M reads, and writing those decimals was what
// bound the tool: 0.76 s of a 0.92 s loop on the box's 16 CPUs while the kernels that FIND the locations take 8 ms.
// Here the text is laid out by a scan and written by kernels; the host copies it out once.
//
// Elements in output order: read i's HEAD (name line + "\tlocs: ", + "\n" when it has no location) is element
// i + loc_off[i], its locations follow (the last one carries the line's "\n").  Five passes over the batch:
//   k_text_mark      head elements get their read's number                      (then an inclusive max-scan: every element knows its read)
//   k_text_len       length of every element; the document of every location
//   (exclusive sum)  where every element starts
//   k_text_write     256 elements per workgroup are formatted into LDS at their offsets and the workgroup's stretch of the
//                    text leaves in 16-byte pieces (a lane writing its own 20-byte element byte by byte costs a request per byte)
#include <hipcub/hipcub.hpp>

#include "rbg_device.hpp"

namespace rbg {
namespace {

__device__ __forceinline__ uint32_t dec_len(uint64_t v) {
    uint32_t n = 1;
    if (v >= 10000000000000000ull) { v /= 10000000000000000ull; n += 16; }
    if (v >= 100000000ull) { v /= 100000000ull; n += 8; }
    if (v >= 10000ull) { v /= 10000ull; n += 4; }
    if (v >= 100ull) { v /= 100ull; n += 2; }
    if (v >= 10ull) n += 1;
    return n;
}
// the n = dec_len(v) digits of v at p[0 .. n)
template <typename Ptr>
__device__ __forceinline__ void put_dec(Ptr p, uint64_t v, uint32_t n) {
    for (uint32_t j = n; j-- > 0;) {
        const uint64_t q = v / 10;
        p[j] = static_cast<char>('0' + static_cast<uint32_t>(v - q * 10));
        v = q;
    }
}

struct TextArgs {
    const uint64_t *lo, *hi, *loc_off, *locs;      // of the batch (device)
    uint64_t N, E;                                 // reads; elements = N + loc_off[N]
    const char *names;                             // the reads' names back to back
    const uint32_t *name_off;                      // [N + 1]
    const uint64_t *doc_start;                     // [ndocs] sorted (DocList::doc_offsets_, doclist.hpp:57-73)
    const char *doc_names;                         // back to back
    const uint32_t *doc_name_off;                  // [ndocs + 1]
    uint64_t ndocs, text_size;                     // text_size: DocList::size (the n of the index)
    bool with_locs;                                // false: the count-only report (no "\tlocs:" line; loc_off is all zero)
    const uint64_t *mk_off, *mk;                   // -m: markers_at of every read's range (rowbowt.hpp:282-285); nullptr = no markers line
    uint64_t per_read;                             // fixed elements per read: the head, + the markers line when there is one
};
// Elements of read i: its head at i * per_read + loc_off[i], its locations behind it, then (with -m) its markers line.
__device__ __forceinline__ uint64_t head_of(const TextArgs &a, uint64_t i) { return i * a.per_read + a.loc_off[i]; }
constexpr uint32_t kNoMarkersLen = 73;
__device__ const char kNoMarkers[] = "no markers (consider building the marker array with a larger window size)";
// rowbowt_gpu.hpp get_pos / get_allele (MarkerT fields)
__device__ __forceinline__ uint64_t mk_pos(uint64_t m) { return m & ((uint64_t(1) << 48) - 1); }
__device__ __forceinline__ uint64_t mk_allele(uint64_t m) { return (m >> 60) & 0xF; }

__global__ __launch_bounds__(256) void k_text_mark(const uint64_t *__restrict__ loc_off, const uint64_t N, const uint64_t per_read, uint32_t *__restrict__ mark) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += stride) mark[i * per_read + loc_off[i]] = static_cast<uint32_t>(i);
}

// doclist.hpp:46-50, :77-79: the document of text position pos = the last one starting at or before it
__device__ __forceinline__ uint64_t doc_of(const TextArgs &a, uint64_t pos) {
    const uint64_t q = pos + 1 > a.text_size ? a.text_size : pos + 1;   // (pos + 1 wraps like the reference's: then k = 0 below)
    uint64_t lo = 0, hi = a.ndocs;
    while (lo < hi) {                                                    // lower_bound(starts, q)
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (a.doc_start[mid] < q) lo = mid + 1; else hi = mid;
    }
    return lo;   // 0: no document (the reference indexes doc_names_[-1] there; the caller reports RBG_EARG)
}

__global__ __launch_bounds__(256) void k_text_len(const TextArgs a, const uint32_t *__restrict__ eread, uint32_t *__restrict__ len,
                                                  uint32_t *__restrict__ doc, unsigned int *__restrict__ bad) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < a.E; e += stride) {
        const uint64_t i = eread[e];
        const uint64_t l0 = a.loc_off[i], l1 = a.loc_off[i + 1];
        const uint64_t eh = head_of(a, i);
        if (e == eh) {   // head
            const uint64_t lo = a.lo[i], hi = a.hi[i];
            len[e] = (a.name_off[i + 1] - a.name_off[i]) + 2 + dec_len(lo) + 1 + dec_len(hi) + 9 + dec_len(hi - lo + 1) + 1 + (a.with_locs ? 7 + (l1 == l0 ? 1u : 0u) : 0u);
        } else if (a.mk_off && e == eh + (l1 - l0) + 1) {   // the markers line (rb_align.cpp:134-142)
            const uint64_t m0 = a.mk_off[i], m1 = a.mk_off[i + 1];
            uint32_t n = 10 + 1;
            if (m1 == m0) n += kNoMarkersLen;
            for (uint64_t t = m0; t < m1; ++t) n += dec_len(mk_pos(a.mk[t])) + 1 + dec_len(mk_allele(a.mk[t])) + 1;
            len[e] = n;
            doc[e] = 0;
        } else {
            const uint64_t t = l0 + (e - eh - 1), pos = a.locs[t];
            const uint64_t k = doc_of(a, pos);
            if (k == 0) { atomicOr(bad, 1u); len[e] = 0; doc[e] = 0; continue; }
            const uint64_t d = k - 1;
            doc[e] = static_cast<uint32_t>(d);
            len[e] = dec_len(pos) + 1 + (a.doc_name_off[d + 1] - a.doc_name_off[d]) + 1 + dec_len(pos - a.doc_start[d]) + 1 + (t + 1 == l1 ? 1u : 0u);
        }
    }
}

// one element's text at p[0 .. its length)
template <typename Ptr>
__device__ __forceinline__ void put_element(const TextArgs &a, const uint64_t e, const uint64_t i, const uint32_t d, Ptr p) {
    const uint64_t l0 = a.loc_off[i], l1 = a.loc_off[i + 1];
    const uint64_t eh = head_of(a, i);
    if (e == eh) {
        const uint32_t nb = a.name_off[i], nl = a.name_off[i + 1] - nb;
        for (uint32_t j = 0; j < nl; ++j) p[j] = a.names[nb + j];
        p += nl;
        p[0] = ' '; p[1] = '(';
        p += 2;
        const uint64_t lo = a.lo[i], hi = a.hi[i], c = hi - lo + 1;   // unsigned wrap for the empty range, like the reference
        uint32_t n = dec_len(lo);
        put_dec(p, lo, n); p += n;
        *p = ','; p += 1;
        n = dec_len(hi);
        put_dec(p, hi, n); p += n;
        const char lit[] = "), count=";
        for (int j = 0; j < 9; ++j) p[j] = lit[j];
        p += 9;
        n = dec_len(c);
        put_dec(p, c, n); p += n;
        *p = '\n'; p += 1;
        if (a.with_locs) {
            const char lit2[] = "\tlocs: ";
            for (int j = 0; j < 7; ++j) p[j] = lit2[j];
            p += 7;
            if (l1 == l0) *p = '\n';
        }
    } else if (a.mk_off && e == eh + (l1 - l0) + 1) {
        const char lit[] = "\tmarkers: ";
        for (int j = 0; j < 10; ++j) p[j] = lit[j];
        p += 10;
        const uint64_t m0 = a.mk_off[i], m1 = a.mk_off[i + 1];
        if (m1 == m0) {
            for (uint32_t j = 0; j < kNoMarkersLen; ++j) p[j] = kNoMarkers[j];
            p += kNoMarkersLen;
        }
        for (uint64_t t = m0; t < m1; ++t) {
            const uint64_t mp = mk_pos(a.mk[t]), ma = mk_allele(a.mk[t]);
            uint32_t n = dec_len(mp);
            put_dec(p, mp, n); p += n;
            *p = '/'; p += 1;
            n = dec_len(ma);
            put_dec(p, ma, n); p += n;
            *p = ' '; p += 1;
        }
        *p = '\n';
    } else {
        const uint64_t t = l0 + (e - eh - 1), pos = a.locs[t];
        uint32_t n = dec_len(pos);
        put_dec(p, pos, n); p += n;
        *p = '/'; p += 1;
        const uint32_t db = a.doc_name_off[d], dl = a.doc_name_off[d + 1] - db;
        for (uint32_t j = 0; j < dl; ++j) p[j] = a.doc_names[db + j];
        p += dl;
        *p = ':'; p += 1;
        const uint64_t off = pos - a.doc_start[d];                    // doclist.hpp:48
        n = dec_len(off);
        put_dec(p, off, n); p += n;
        *p = ' '; p += 1;
        if (t + 1 == l1) *p = '\n';
    }
}

constexpr uint32_t kTextLds = 40 * 1024;   // bytes of text a workgroup stages (256 elements: 10 KB at 40 bytes each; long names take the slow path)

__global__ __launch_bounds__(256) void k_text_write(const TextArgs a, const uint32_t *__restrict__ eread, const uint32_t *__restrict__ doc,
                                                    const uint64_t *__restrict__ at, const uint64_t total, char *__restrict__ text) {
    __shared__ __align__(16) char s_buf[kTextLds + 16];
    const uint64_t nblocks = (a.E + 255) / 256;
    for (uint64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint64_t e0 = blk * 256, e1 = e0 + 256 < a.E ? e0 + 256 : a.E;
        const uint64_t t0 = at[e0], t1 = e1 < a.E ? at[e1] : total;
        const uint64_t e = e0 + threadIdx.x;
        const uint32_t shift = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(text) + t0) & 15u);   // LDS and memory addresses congruent mod 16
        if (t1 - t0 <= kTextLds) {
            if (e < e1) put_element(a, e, eread[e], doc[e], s_buf + shift + (at[e] - t0));
            __syncthreads();
            const uint32_t nbytes = static_cast<uint32_t>(t1 - t0);
            char *dst = text + t0;
            const char *src = s_buf + shift;
            const uint32_t head = (16u - shift) & 15u;                      // bytes before the first 16-byte boundary
            const uint32_t h = head < nbytes ? head : nbytes;
            if (threadIdx.x < h) dst[threadIdx.x] = src[threadIdx.x];
            const uint32_t body = (nbytes - h) >> 4;
            for (uint32_t j = threadIdx.x; j < body; j += 256)
                reinterpret_cast<uint4 *>(dst + h)[j] = reinterpret_cast<const uint4 *>(src + h)[j];
            const uint32_t done = h + (body << 4);
            if (threadIdx.x < nbytes - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
            __syncthreads();
        } else if (e < e1) {
            put_element(a, e, eread[e], doc[e], text + at[e]);              // (long names: straight to memory, byte by byte)
        }
    }
}

// the call's inputs come in by a kernel that reads the pinned block over PCIe, not by the copy engine: that one is busy
// with the previous batch's text on its way out, and a 9 MB copy queued behind it waited 3 ms
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const uint64_t n16) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

struct MaxOp {
    __device__ __forceinline__ uint32_t operator()(uint32_t x, uint32_t y) const { return x > y ? x : y; }
};

}  // namespace

// workspace of a batch of E elements: eread / len / doc (4 bytes each), at (8), the two scans' temporaries
size_t text_ws_bytes(uint64_t E) {
    size_t s1 = 0, s2 = 0;
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, s1, static_cast<uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), MaxOp(), static_cast<int64_t>(E ? E : 1));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, s2, static_cast<uint32_t *>(nullptr), static_cast<uint64_t *>(nullptr), static_cast<int64_t>(E ? E : 1));
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    return 3 * up(E * 4) + up(E * 8 + 8) + up(std::max(s1, s2)) + 1024;
}

int launch_copy16(const void *pinned_src, void *dst, uint64_t bytes, void *stream) {   // both 16-byte aligned; bytes rounded up to 16
    const uint64_t n16 = (bytes + 15) / 16;
    if (!n16) return 0;
    hipLaunchKernelGGL(k_copy16, dim3(static_cast<int>(std::min<uint64_t>((n16 + 255) / 256, 2048))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(pinned_src), static_cast<uint4 *>(dst), n16);
    return static_cast<int>(hipGetLastError());
}

// phase 1: element lengths and offsets.  *d_total (device, 8 bytes) = bytes of text; *d_bad != 0: a location outside every document
int launch_text_plan(const uint64_t *lo, const uint64_t *hi, const uint64_t *loc_off, const uint64_t *locs, uint64_t N, uint64_t E, const char *names,
                     const uint32_t *name_off, const uint64_t *doc_start, const char *doc_names, const uint32_t *doc_name_off, uint64_t ndocs,
                     uint64_t text_size, bool with_locs, const uint64_t *mk_off, const uint64_t *mk, void *ws, size_t ws_bytes, unsigned int *d_bad, void *stream) {
    if (ws_bytes < text_ws_bytes(E) || (reinterpret_cast<uintptr_t>(ws) & 255)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    char *b = static_cast<char *>(ws);
    uint32_t *eread = reinterpret_cast<uint32_t *>(b), *len = reinterpret_cast<uint32_t *>(b + up(E * 4)), *doc = reinterpret_cast<uint32_t *>(b + 2 * up(E * 4));
    uint64_t *at = reinterpret_cast<uint64_t *>(b + 3 * up(E * 4));
    void *tmp = b + 3 * up(E * 4) + up(E * 8 + 8);
    size_t tmp_bytes = ws_bytes - (3 * up(E * 4) + up(E * 8 + 8));
    const TextArgs a{lo, hi, loc_off, locs, N, E, names, name_off, doc_start, doc_names, doc_name_off, ndocs, text_size, with_locs, mk_off, mk, mk_off ? uint64_t(2) : uint64_t(1)};
    hipError_t e = hipMemsetAsync(eread, 0, E * 4, st);
    if (e != hipSuccess) return static_cast<int>(e);
    const int gN = static_cast<int>(std::min<uint64_t>((N + 255) / 256, 256ull * 32)), gE = static_cast<int>(std::min<uint64_t>((E + 255) / 256, 256ull * 32));
    hipLaunchKernelGGL(k_text_mark, dim3(gN), dim3(256), 0, st, loc_off, N, a.per_read, eread);
    size_t tb = tmp_bytes;
    e = hipcub::DeviceScan::InclusiveScan(tmp, tb, eread, eread, MaxOp(), static_cast<int64_t>(E), st);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(k_text_len, dim3(gE), dim3(256), 0, st, a, eread, len, doc, d_bad);
    tb = tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, len, at, static_cast<int64_t>(E), st);
    if (e != hipSuccess) return static_cast<int>(e);
    return static_cast<int>(hipGetLastError());
}
// where the total is: at[E - 1] + len[E - 1] (both in the workspace)
void text_total_ptrs(void *ws, uint64_t E, const uint64_t **last_at, const uint32_t **last_len) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    char *b = static_cast<char *>(ws);
    *last_len = reinterpret_cast<const uint32_t *>(b + up(E * 4)) + (E - 1);
    *last_at = reinterpret_cast<const uint64_t *>(b + 3 * up(E * 4)) + (E - 1);
}
// phase 2: the text itself (total bytes at `text`)
int launch_text_fill(const uint64_t *lo, const uint64_t *hi, const uint64_t *loc_off, const uint64_t *locs, uint64_t N, uint64_t E, const char *names,
                     const uint32_t *name_off, const uint64_t *doc_start, const char *doc_names, const uint32_t *doc_name_off, uint64_t ndocs,
                     uint64_t text_size, bool with_locs, const uint64_t *mk_off, const uint64_t *mk, void *ws, uint64_t total, char *text, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    char *b = static_cast<char *>(ws);
    const uint32_t *eread = reinterpret_cast<const uint32_t *>(b), *doc = reinterpret_cast<const uint32_t *>(b + 2 * up(E * 4));
    const uint64_t *at = reinterpret_cast<const uint64_t *>(b + 3 * up(E * 4));
    const TextArgs a{lo, hi, loc_off, locs, N, E, names, name_off, doc_start, doc_names, doc_name_off, ndocs, text_size, with_locs, mk_off, mk, mk_off ? uint64_t(2) : uint64_t(1)};
    const int g = static_cast<int>(std::min<uint64_t>((E + 255) / 256, 256ull * 16));
    hipLaunchKernelGGL(k_text_write, dim3(g), dim3(256), 0, st, a, eread, doc, at, total, text);
    return static_cast<int>(hipGetLastError());
}

}  // namespace rbg
