// rbg_pack2bit.hpp -- host-side 2-bit packing of reads for the packed search kernel (k_find_range_packed,
// k_search.hip): plain C++, no HIP.  Used by the host-pointer pipeline (rbg_hostpath.hpp) and the FASTQ front end
// of the command-line tools; unit-tested on the CPU (tests/cpp/pack2bit_check.cpp).
#pragma once

#include <cstdint>
#include <cstring>

namespace rbg_hostpath {

// ---- 2-bit packing of one read, in the order the search consumes it --------------------------------------------------
// Symbol t of the stream is q[m-1-t] (the search runs right to left, rowbowt.hpp:127-129) at bits [2t, 2t+2): code
// 0..3 = A, C, G, T.  dst receives ceil(m / 64) 16-byte chunks (zero padded).  Returns false when the read holds
// anything else (it is then searched from its bytes).
inline bool pack_read_acgt(const uint8_t *q, uint64_t m, uint32_t *dst) {
    const uint64_t nwords = ((m + 63) / 64) * 4;
    uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
    const uint64_t nhalf = nwords * 2;
    uint64_t t = 0;        // symbols packed so far
    uint64_t bad = 0;
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
    if (t < m) {  // the read's first (m mod 8) symbols
        uint32_t y = 0;
        for (uint64_t u = 0; t + u < m; ++u) {
            const uint8_t c = q[m - 1 - (t + u)];
            const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
            if (code > 3u) bad = 1;
            y |= (code & 3u) << (2 * u);
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
