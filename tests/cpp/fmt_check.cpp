// Test harness (CPU only): rbg_cli::fmt_u64 (fastx.hpp, the tools' number writer) against snprintf on boundary values and
// random ones of every length, written back to back into a TextBuf through FastOut (growth keeps what was written).
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../rowbowt_amd/csrc/fastx.hpp"

int main() {
    std::mt19937_64 rng(7);
    rbg_cli::TextBuf buf;
    rbg_cli::FastOut out(buf);
    std::string want;
    auto put = [&](uint64_t v) {
        char *p = out.room(24);
        char *e = rbg_cli::fmt_u64(p, v);
        *e++ = ' ';
        out.len += static_cast<size_t>(e - p);
        char tmp[32];
        std::snprintf(tmp, sizeof(tmp), "%" PRIu64 " ", v);
        want += tmp;
    };
    uint64_t pow10 = 1;
    for (int d = 0; d < 20; ++d) {
        for (uint64_t v : {pow10 - 1, pow10, pow10 + 1, pow10 * 9, pow10 * 9 + (pow10 - 1)}) put(v);
        if (d < 19) pow10 *= 10;
    }
    put(0); put(~uint64_t(0)); put(uint64_t(1) << 32); put((uint64_t(1) << 32) - 1);
    for (int bits = 1; bits <= 64; ++bits)
        for (int t = 0; t < 3000; ++t) put(bits == 64 ? rng() : rng() & ((uint64_t(1) << bits) - 1));
    out.finish();
    if (buf.size() != want.size() || std::memcmp(buf.data(), want.data(), want.size()) != 0) {
        std::printf("MISMATCH\n");
        return 1;
    }
    std::printf("ok %zu bytes\n", buf.size());
    return 0;
}
