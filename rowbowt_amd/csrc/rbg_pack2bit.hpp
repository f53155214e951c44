// rbg_pack2bit.hpp -- host-side 2-bit packing of reads for the packed search kernel (k_find_range_packed,
// k_search.hip): plain C++, no HIP.  Used by the host-pointer pipeline (rbg_hostpath.hpp) and the FASTQ front end
// of the command-line tools; unit-tested on the CPU (tests/cpp/pack2bit_check.cpp).
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rbg_hostpath {

#if defined(__x86_64__)
// 32 symbols per step with AVX2 (chosen at run time; the 64-bit SWAR loop below is the portable path).  Packs the whole
// read when m >= 32: full blocks from the read's end, then ONE overlapping block over the read's first 32 characters for
// the (m mod 32) symbols that are left -- its packed word shifted down so that only the missing symbols remain -- which
// replaces a byte-by-byte tail whose data-dependent branches cost more than the three full blocks of a 100 bp read.
// Returns the number of bytes of dst written (whole symbols of the read; the caller zero-fills the rest of the
// chunks); *bad is OR-ed with a non-zero value if any symbol is not A, C, G or T.
// the 32 characters at p, last one first, as 64 bits of codes; *ok keeps all-ones only where the characters are ACGT
__attribute__((target("avx2"))) inline uint64_t pack_block32_avx2(const uint8_t *p, __m256i *ok) {
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
    const __m256i letters = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
    v = _mm256_permute2x128_si256(_mm256_shuffle_epi8(v, rev), _mm256_shuffle_epi8(v, rev), 1);   // p[31] first
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);                             // A 0, C 1, T 2, G 3
    const __m256i code = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), one));        // A 0, C 1, G 2, T 3
    *ok = _mm256_and_si256(*ok, _mm256_cmpeq_epi8(_mm256_shuffle_epi8(letters, code), v));
    const __m256i p4 = _mm256_maddubs_epi16(code, w14);      // s[2i] + 4 s[2i+1]
    const __m256i p8 = _mm256_madd_epi16(p4, w116);          // ... + 16 (s[2i+2] + 4 s[2i+3]): one byte per 32-bit lane
    const __m256i by = _mm256_shuffle_epi8(p8, pick);
    const uint32_t lo4 = static_cast<uint32_t>(_mm256_extract_epi32(by, 0)), hi4 = static_cast<uint32_t>(_mm256_extract_epi32(by, 4));
    return static_cast<uint64_t>(lo4) | (static_cast<uint64_t>(hi4) << 32);
}
__attribute__((target("avx2"))) inline uint64_t pack_read_avx2(const uint8_t *q, uint64_t m, uint8_t *dst_bytes, uint64_t *bad) {
    __m256i ok = _mm256_set1_epi8(-1);
    uint64_t t = 0;
    while (t + 32 <= m) {
        const uint64_t out = pack_block32_avx2(q + (m - t - 32), &ok);             // symbols t .. t+31 (symbol t = q[m-1-t])
        std::memcpy(dst_bytes + (t >> 2), &out, 8);
        t += 32;
    }
    uint64_t written = t >> 2;
    if (t < m) {                                                  // symbols t .. m-1 = the read's first m - t characters
        const uint64_t out = pack_block32_avx2(q, &ok) >> (2 * (32 - (m - t)));    // q[0..31] holds symbols m-32 .. m-1
        std::memcpy(dst_bytes + (t >> 2), &out, 8);               // (t is a multiple of 32 below m: these 8 bytes lie inside the chunks)
        written += 8;
    }
    if (_mm256_movemask_epi8(ok) != -1) *bad |= 1;
    return written;
}
inline bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}
#endif

// ---- 2-bit packing of one read, in the order the search consumes it --------------------------------------------------
// Symbol t of the stream is q[m-1-t] (the search runs right to left, rowbowt.hpp:127-129) at bits [2t, 2t+2): code
// 0..3 = A, C, G, T.  dst receives ceil(m / 64) 16-byte chunks (zero padded).  Returns false when the read holds
// anything else (it is then searched from its bytes).
inline bool pack_read_acgt(const uint8_t *q, uint64_t m, uint32_t *dst) {
    const uint64_t nwords = ((m + 63) / 64) * 4;
    uint64_t bad = 0;
#if defined(__x86_64__)
    if (m >= 32 && have_avx2()) {
        uint8_t *d8 = reinterpret_cast<uint8_t *>(dst);
        const uint64_t written = pack_read_avx2(q, m, d8, &bad);
        std::memset(d8 + written, 0, nwords * 4 - written);
        return bad == 0;
    }
#endif
    uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
    const uint64_t nhalf = nwords * 2;
    uint64_t t = 0;        // symbols packed so far
    while (t + 8 <= m) {   // eight symbols per step, SWAR
        uint64_t w;
        std::memcpy(&w, q + (m - t - 8), 8);
        w = __builtin_bswap64(w);  // q[m-1-t] into the lowest byte
        const uint64_t x = (w >> 1) & 0x0303030303030303ull;            // A 0, C 1, T 2, G 3
        const uint64_t hi = (x >> 1) & 0x0101010101010101ull;
        const uint64_t code = x ^ hi;                                     // A 0, C 1, G 2, T 3
        // the character each code stands for: 0x41 + {0, 2, 6, 0x13}; anything else in the input shows up as a difference
        const uint64_t lo1 = code & 0x0101010101010101ull, hi1 = (code >> 1) & 0x0101010101010101ull, both = lo1 & hi1;
        const uint64_t expect = 0x4141414141414141ull + (lo1 << 1) + (hi1 << 1) + (hi1 << 2) + both + (both << 1) + (both << 3);
        bad |= expect ^ w;
        uint64_t y = code;
        y = (y | (y >> 6)) & 0x000F000F000F000Full;
        y = (y | (y >> 12)) & 0x000000FF000000FFull;
        y = (y | (y >> 24)) & 0xFFFFull;
        d16[t >> 3] = static_cast<uint16_t>(y);
        t += 8;
    }
    uint64_t filled = t >> 3;
    if (t < m) {  // the read's first (m mod 8) symbols, without data-dependent branches
        static const char kLetters[4] = {'A', 'C', 'G', 'T'};
        uint32_t y = 0;
        for (uint64_t u = 0; t + u < m; ++u) {
            const uint32_t c = q[m - 1 - (t + u)];
            const uint32_t x = (c >> 1) & 3u, code = x ^ (x >> 1);
            bad |= static_cast<uint64_t>(c ^ static_cast<uint32_t>(kLetters[code]));
            y |= code << (2 * u);
        }
        d16[filled++] = static_cast<uint16_t>(y);
    }
    for (; filled < nhalf; ++filled) d16[filled] = 0;
    return bad == 0;
}

// the same for an arbitrary 4-symbol major alphabet (lut2: byte -> 0..3 or 0xFF)
inline bool pack_read_lut(const uint8_t *q, uint64_t m, const uint8_t *lut2, uint32_t *dst) {
    const uint64_t nwords = ((m + 63) / 64) * 4;
    for (uint64_t w = 0; w < nwords; ++w) dst[w] = 0;
    bool ok = true;
    for (uint64_t t = 0; t < m; ++t) {
        const uint32_t code = lut2[q[m - 1 - t]];
        if (code > 3u) { ok = false; break; }
        dst[t >> 4] |= code << (2 * (t & 15));
    }
    return ok;
}

}  // namespace rbg_hostpath
