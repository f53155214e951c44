// CPU test of rowbowt_amd/csrc/rbg_pack2bit.hpp: the SWAR ACGT packer against the table-driven one and against a
// bit-by-bit restatement of the layout k_find_range_packed reads (symbol t = q[m-1-t] at bits [2t, 2t+2) of the
// read's little-endian stream, whole 16-byte chunks, zero padded), on random reads of every length 0..300 with and
// without symbols outside ACGT.  Prints "pack2bit ok <reads checked>".
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../rowbowt_amd/csrc/rbg_pack2bit.hpp"

int main() {
    std::mt19937_64 rng(12345);
    uint8_t lut2[256];
    for (int c = 0; c < 256; ++c) lut2[c] = 0xFF;
    lut2['A'] = 0; lut2['C'] = 1; lut2['G'] = 2; lut2['T'] = 3;
    const char alphabet[] = "ACGTNacgt\x01\xff@EB";
    uint64_t checked = 0, n_bad = 0;
    for (int rep = 0; rep < 40; ++rep)
        for (uint64_t m = 0; m <= 300; ++m) {
            std::vector<uint8_t> q(m + 16, 'X');   // (slack: the packer must not read outside [0, m))
            const bool dirty = rng() % 4 == 0;
            for (uint64_t i = 0; i < m; ++i) q[8 + i] = dirty && rng() % 23 == 0 ? alphabet[4 + rng() % 11] : alphabet[rng() % 4];
            const uint64_t nw = ((m + 63) / 64) * 4;
            std::vector<uint32_t> a(nw + 4, 0xDEADBEEF), b(nw + 4, 0xDEADBEEF), want(nw, 0);
            bool want_ok = true;
            for (uint64_t t = 0; t < m; ++t) {
                const uint8_t c = q[8 + m - 1 - t];
                if (lut2[c] > 3) { want_ok = false; break; }
                want[t / 16] |= static_cast<uint32_t>(lut2[c]) << (2 * (t % 16));
            }
            const bool oka = rbg_hostpath::pack_read_acgt(q.data() + 8, m, a.data());
            const bool okb = rbg_hostpath::pack_read_lut(q.data() + 8, m, lut2, b.data());
            if (oka != want_ok || okb != want_ok) { std::printf("flag mismatch at m=%llu\n", (unsigned long long)m); return 1; }
            for (uint64_t w = nw; w < nw + 4; ++w)
                if (a[w] != 0xDEADBEEF || b[w] != 0xDEADBEEF) { std::printf("wrote past the chunks at m=%llu\n", (unsigned long long)m); return 1; }
            if (want_ok)
                for (uint64_t w = 0; w < nw; ++w)
                    if (a[w] != want[w] || b[w] != want[w]) { std::printf("word %llu differs at m=%llu\n", (unsigned long long)w, (unsigned long long)m); return 1; }
            n_bad += !want_ok;
            ++checked;
        }
    std::printf("pack2bit ok %llu reads (%llu with other symbols)\n", (unsigned long long)checked, (unsigned long long)n_bad);
    return 0;
}
