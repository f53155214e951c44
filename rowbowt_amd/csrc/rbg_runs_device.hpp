// rbg_runs_device.hpp -- what the kernels of the run-indexed layout share beside the search itself (rbg_runs2_device.hpp):
// the wave-local LDS fence, the one-off raise of a kernel's dynamic LDS limit, the step's record lookup and result.
// (Until round 4 this file held the wave-cooperative probes of the layout's first format -- 16-lane rows, quads, 128-byte
// bucket records; they lost every A/B to one lane per query and were retired: profiles/r04_fmt_ab.txt, DESIGN_HISTORY.md.)
// Everything lives in an anonymous namespace: each unit gets its own inlined copy.
#pragma once

#include <map>

#include "rbg_device.hpp"

namespace rbg {
namespace {

// (wave_lds_sync -- the wave-local LDS fence -- is rbg_device.hpp's)

// `bytes` = the launch's DYNAMIC LDS (what the attribute limits); `static_bytes` = the kernel's static __shared__ arrays, which count
// towards the 48 KB a kernel gets without asking but are not part of the attribute's value
template <typename Kernel>
void raise_lds(Kernel kernel, size_t bytes, size_t static_bytes = 0) {
    if (bytes + static_bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> raised;   // the largest size raised so far per (device, kernel)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, reinterpret_cast<const void *>(kernel));
    std::lock_guard<std::mutex> g(mu);
    size_t &have = raised[key];
    if (bytes <= have) return;
    // (a later index on the same device may need more than the first one did: raise again, and say so when the runtime refuses --
    //  the launch that follows would fail with an unrelated-looking error)
    const hipError_t e = hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e == hipSuccess) have = bytes;
    else std::fprintf(stderr, "rbg: raising a kernel's dynamic LDS limit to %zu bytes failed: %s\n", bytes, hipGetErrorString(e));
}

// ---- STAGE: the wave's reads as 2-bit codes in LDS, made by the wave itself -----------------------------------------------------------
// A lane that walks its read through a cursor of 16-byte chunks fetches a chunk every second step: by then the memory system has turned its
// caches over (a step is a random sector per lane), so every chunk is an L2 miss of its own -- 6.8 per 100 bp read, a third of K2's misses
// (profiles/r06_k2_sectors.md).  Here every lane fetches ALL chunks of its read back to back at the top of the wave's iteration -- neighbouring
// lanes hold neighbouring reads, so a line is asked for by all its readers within a few hundred cycles and leaves HBM once -- and keeps them
// as 2-bit codes of the k-mer alphabet in consumption order (the layout of k_pack_reads: symbol q[m - 1 - t] at bits [2t, 2t + 2)), word w of
// lane l at codes[w * 64 + l]: a step's table index is then the next 2 * adv bits, two LDS words and a funnel shift, no per-symbol lookup.
// Four symbols per instruction: a byte's code and the byte it has to be come out of two 8-byte register tables by v_perm_b32 (rbg_dev.h
// stage_*).  A wave with a read longer than kStageCap symbols, or with a symbol outside the k-mer alphabet, walks its reads as bytes.
constexpr uint32_t kStageCap = 256u;                         // symbols per read the staging holds
constexpr uint32_t kStageWords = kStageCap / 16u + 1u;       // (+ 1: the word after the last one is read, never used)
constexpr uint32_t kStageWaveBytes = kStageWords * 64u * 4u;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct StageTab { uint32_t code_lo, code_hi, byte_lo, byte_hi, shift; };
// one aligned 16-byte chunk -> its sixteen codes in consumption order (byte 15 first) + the bytes that are not symbols of the alphabet (diff != 0)
__device__ __forceinline__ uint32_t stage_chunk(const u32x4 w, const StageTab &T, uint32_t (&diff)[4]) {
    const uint32_t x[4] = {w.x, w.y, w.z, w.w};
    uint32_t c8[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t idx = (x[t] >> T.shift) & 0x07070707u;
        const uint32_t code = __builtin_amdgcn_perm(T.code_hi, T.code_lo, idx);
        diff[t] = x[t] ^ __builtin_amdgcn_perm(T.byte_hi, T.byte_lo, idx);
        c8[t] = (code * 0x40100401u) >> 24;                 // code(byte 3) | code(byte 2) << 2 | code(byte 1) << 4 | code(byte 0) << 6
    }
    return c8[3] | (c8[2] << 8) | (c8[1] << 16) | (c8[0] << 24);
}
// bytes of a 16-bit per-byte mask's nibble as a 32-bit byte mask
__device__ __forceinline__ uint32_t nibble_bytes(const uint32_t nib) { return (((nib & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu; }
// the read [beg, end) of this lane -> codes column `col` (stride 64 words); returns true when a byte is no symbol of the k-mer alphabet.
// m = end - beg <= kStageCap.  chunks_out: 16-byte chunks fetched (STATS).
__device__ __forceinline__ bool stage_read(const uint4 *__restrict__ chunks16, const uint64_t beg, const uint64_t end, const StageTab &T, lds_u32 *col, uint32_t &chunks_out) {
    chunks_out = 0;
    if (end <= beg) return false;
    uint64_t ci = (end - 1) >> 4;
    const uint64_t ci_lo = beg >> 4;
    const uint32_t hi_b = static_cast<uint32_t>(end - 1) & 15u, lo_b = static_cast<uint32_t>(beg) & 15u;
    const uint32_t skip2 = 2u * (15u - hi_b);                   // the top chunk's first symbols lie beyond the read's end
    uint32_t nwords = static_cast<uint32_t>((end - beg + 15) >> 4);
    uint32_t bad = 0;
    uint32_t d[4];
    u32x4 w = as_global<u32x4>(static_cast<const void *>(chunks16))[ci];
    uint32_t prev = stage_chunk(w, T, d);
    {   // the top chunk: bytes above hi_b (and, for a read inside one chunk, below lo_b) are not the read's
        uint32_t vm = (2u << hi_b) - 1u;
        if (ci == ci_lo) vm &= ~((1u << lo_b) - 1u);
        bad |= (d[0] & nibble_bytes(vm)) | (d[1] & nibble_bytes(vm >> 4)) | (d[2] & nibble_bytes(vm >> 8)) | (d[3] & nibble_bytes(vm >> 12));
    }
    chunks_out = static_cast<uint32_t>(ci - ci_lo) + 1u;
    while (nwords) {
        uint32_t next = 0;
        if (ci > ci_lo) {
            --ci;
            w = as_global<u32x4>(static_cast<const void *>(chunks16))[ci];
            next = stage_chunk(w, T, d);
            if (ci == ci_lo) {                                  // the bottom chunk: bytes below lo_b are not the read's
                const uint32_t vm = ~((1u << lo_b) - 1u);
                bad |= (d[0] & nibble_bytes(vm)) | (d[1] & nibble_bytes(vm >> 4)) | (d[2] & nibble_bytes(vm >> 8)) | (d[3] & nibble_bytes(vm >> 12));
            } else {
                bad |= d[0] | d[1] | d[2] | d[3];
            }
        }
        *col = __builtin_amdgcn_alignbit(next, prev, skip2);   // ({next, prev} >> skip2): skip2 <= 30
        col += 64;
        prev = next;
        --nwords;
    }
    return bad != 0;
}

// the next `nsym` (<= 16) symbols of a staged read from consumption index t on (t = symbols already consumed from the read's end), as a base-4 number
// with the first of them in the low bits: the k-mer table index of a step, or the ftab word
__device__ __forceinline__ uint32_t staged_bits(const lds_u32 *col, const uint32_t t, const uint32_t nsym) {
    const lds_u32 *cw = col + (t >> 4) * 64u;
    const uint32_t v = __builtin_amdgcn_alignbit(cw[64], cw[0], (t & 15u) * 2u);
    return nsym >= 16u ? v : (v & ((1u << (2u * nsym)) - 1u));
}

// the record of the step that consumes the k-mer `acc` (read as a base-nmajor number, least significant digit = the
// symbol next to the suffix) of `adv` symbols; adv == 1: acc is the symbol's slot
__device__ __forceinline__ uint32_t run_record(const uint32_t *s_tab_first, uint32_t adv, uint32_t acc) { return s_tab_first[adv - 1] + acc; }

// result of one lane's step
struct RunStep {
    uint64_t F = 0, c_before = 0, c_upto = 0;   // F + rank(lo, .), F + rank(hi + 1, .) with F the table's first row: the run lists' cums carry it (F itself stays 0)
    bool inside = false;                        // row hi lies in a run of the table (LF_w_loc's fast branch, rowbowt.hpp:559-561)
    uint64_t samp_e = 0;                        // else the predecessor run's sample: that entry's index relative to the table's first entry
};

// What the instrumented instantiations count on this layout (the same eight sums as SearchStat, other meanings:
// include/rbg.h): [kStSteps] search steps, [kStSlots] bucket records fetched (64 bytes) or directory gathers (two neighbouring
// entries: 8 or 16 bytes), [kStDense] run-list entries the scans needed (8 bytes each), [kStSearch] narrowing rounds (seven
// 4-byte pivots each), [kStFtab], [kStResample] (one sample gather), [kStChunks], [kStSymbols].

}  // namespace
}  // namespace rbg
