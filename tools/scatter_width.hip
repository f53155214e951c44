// scatter_width.hip -- what K3's location stores cost the memory system, apart from the walk: every group of LANES lanes stores one contiguous segment of
// LANES x 8 bytes at a random place of a 16 GiB buffer, the place aligned to ALIGN bytes (8 = wherever a read's locations happen to start, as K3's flush does;
// 64 / 128 = whole sectors / lines).  Independent stores (no chain), so the rate is the store path's own.  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 tools/scatter_width.hip -o tools/scatter_width      usage: scatter_width [GiB = 16] [steps = 256]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

template <int LANES, int ALIGN>
__global__ __launch_bounds__(256) void scatter(u64 *__restrict__ buf, u64 nwords, int steps) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    const u64 grp = tid / LANES, l = tid % LANES;
    u64 x = grp * 0x9E3779B97F4A7C15ull + 12345;
    for (int s = 0; s < steps; ++s) {
        x = mix(x + s);
        u64 w = x % (nwords - 2 * LANES);
        w &= ~static_cast<u64>(ALIGN / 8 - 1);
        buf[w + l] = x + l;
    }
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    u64 *buf = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", bytes); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%s, %d CUs, buffer %.1f GiB, %d stores per lane, 8 waves per SIMD\n", prop.gcnArchName, cus, gib, steps);
    auto run = [&](auto kern, int lanes, int align) {
        const int blocks = cus * 8;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, (u64)(bytes / 8), steps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double segs = (double)blocks * 256 / lanes * steps;
        printf("  segments of %3d bytes (%2d lanes x 8) at %3d-byte aligned random places : %8.2f ms  %6.2f G segments/s  %5.2f TB/s stored\n", lanes * 8, lanes, align, best,
               segs / best / 1e6, segs * lanes * 8 / best / 1e9);
    };
    run(scatter<1, 8>, 1, 8);
    run(scatter<4, 8>, 4, 8);
    run(scatter<4, 32>, 4, 32);
    run(scatter<8, 8>, 8, 8);
    run(scatter<8, 64>, 8, 64);
    run(scatter<16, 8>, 16, 8);
    run(scatter<16, 32>, 16, 32);
    run(scatter<16, 64>, 16, 64);
    run(scatter<16, 128>, 16, 128);
    run(scatter<32, 8>, 32, 8);
    run(scatter<32, 128>, 32, 128);
    run(scatter<64, 8>, 64, 8);
    run(scatter<64, 128>, 64, 128);
    return 0;
}
