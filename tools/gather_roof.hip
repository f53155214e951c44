// gather_roof.hip -- empirical random-gather ceiling of one MI355X, in the access shape of the
// backward-search kernels: every lane follows its own dependent chain of aligned W-byte loads at
// uniformly random offsets of a T-byte table (no reuse between lanes).  Reported as accesses/s and
// as GB/s of W-byte payload; compare with the 8 TB/s streaming peak to see what "HBM roofline"
// means for a latency/transaction-bound integer gather.  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_roof.hip -o tools/gather_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// variant: 16-byte accesses with a non-temporal hint, and two independent chains per lane
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: nt loads, 1: two chains per lane (plain loads), 2: two chains + nt
__global__ __launch_bounds__(256) void chase16x(const uint4 *__restrict__ tab, unsigned long long nslots, int steps,
                                                unsigned long long *out) {
    unsigned long long x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned long long y = x ^ 0xD1B54A32D192ED03ull;
    unsigned long long acc = 0, acc2 = 0;
    for (int s = 0; s < steps; ++s) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint4 *p = tab + x % nslots;
        uint4 v;
        if (MODE == 0 || MODE == 2) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); v = make_uint4(t.x, t.y, t.z, t.w); } else v = *p;
        if (MODE >= 1) {
            y ^= y >> 29; y *= 0xBF58476D1CE4E5B9ull; y ^= y >> 32;
            const uint4 *p2 = tab + y % nslots;
            uint4 v2;
            if (MODE == 2) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p2)); v2 = make_uint4(t.x, t.y, t.z, t.w); } else v2 = *p2;
            acc2 += v2.x;
            y += acc2;
        }
        acc += v.x;
        x += acc;
    }
    out[blockIdx.x * 256ull + threadIdx.x] = acc + acc2;
}

template <int W>  // bytes per access: 8, 16, 32, 64
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tab, unsigned long long nslots, int steps,
                                             unsigned long long *out) {
    unsigned long long x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned long long acc = 0;
    for (int s = 0; s < steps; ++s) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const unsigned long long slot = x % nslots;
        const uint4 *p = tab + slot * (W / 16 ? W / 16 : 1);
        if (W == 8) { acc += reinterpret_cast<const unsigned long long *>(tab)[slot]; }
        else {
            uint4 v = p[0];
            acc += v.x;
            if (W >= 32) { uint4 w = p[1]; acc += w.y; }
            if (W >= 64) { uint4 w2 = p[2], w3 = p[3]; acc += w2.z + w3.w; }
        }
        x += acc;  // next address depends on the loaded data: one outstanding access per lane
    }
    out[blockIdx.x * 256ull + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const int waves_per_simd = argc > 3 ? atoi(argv[3]) : 8;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    uint4 *tab;
    hipMalloc(&tab, bytes);
    hipMemset(tab, 1, bytes);
    int cus = 256;
    const int blocks = cus * waves_per_simd;  // 256-lane blocks: one block per CU = 1 wave per SIMD
    unsigned long long *out;
    hipMalloc(&out, blocks * 256ull * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("table %.3f GiB, %d blocks x 256 lanes (%d waves/SIMD), %d dependent accesses per lane\n", gib, blocks, waves_per_simd, steps);
    auto run = [&](int W) {
        const unsigned long long nslots = bytes / (W < 16 ? 8 : W);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (W == 8) hipLaunchKernelGGL(chase<8>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            if (W == 16) hipLaunchKernelGGL(chase<16>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            if (W == 32) hipLaunchKernelGGL(chase<32>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            if (W == 64) hipLaunchKernelGGL(chase<64>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double acc = (double)blocks * 256 * steps;
        printf("W=%2d B: %.2f ms  %.2f G accesses/s  payload %.1f GB/s  (64B-sector traffic %.1f GB/s)\n", W, ms,
               acc / ms / 1e6, acc * W / ms / 1e6, acc * (W > 64 ? W : 64) / ms / 1e6);
    };
    for (int W : {8, 16, 32, 64}) run(W);
    for (int mode = 0; mode < 3; ++mode) {
        const unsigned long long nslots = bytes / 16;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(chase16x<0>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            if (mode == 1) hipLaunchKernelGGL(chase16x<1>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            if (mode == 2) hipLaunchKernelGGL(chase16x<2>, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double acc = (double)blocks * 256 * steps * (mode >= 1 ? 2 : 1);
        printf("16 B %s: %.2f ms  %.2f G accesses/s\n", mode == 0 ? "nt hint" : mode == 1 ? "two chains per lane" : "two chains + nt", ms, acc / ms / 1e6);
    }
    return 0;
}
