// overlap_probe.hip -- do H2D copy, kernel and D2H copy of successive chunks on different streams overlap on this box?
// Ten chunks (40 MB in, a kernel of about 0.8 ms, 16 MB out), issued round-robin on 1 / 2 / 4 streams from pinned
// memory; the kernel either fills every wave slot of the chip or a quarter of them.  Prints the wall time per variant.
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void spin(const uint32_t *in, uint32_t *out, long long cycles) {
    const long long t0 = wall_clock64();
    uint32_t v = in[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF];
    while (wall_clock64() - t0 < cycles) v = v * 1664525u + 1013904223u;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t in_b = 40u << 20, out_b = 16u << 20;
    const int chunks = 10, S = 4;
    void *h_in[S], *h_out[S], *d_in[S], *d_out[S];
    hipStream_t st[S];
    for (int s = 0; s < S; ++s) {
        CK(hipHostMalloc(&h_in[s], in_b, hipHostMallocDefault));
        CK(hipHostMalloc(&h_out[s], out_b, hipHostMallocDefault));
        CK(hipMalloc(&d_in[s], in_b));
        CK(hipMalloc(&d_out[s], out_b));
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        std::memset(h_in[s], 1, in_b);
    }
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const long long cycles = static_cast<long long>(clk_khz) * 8 / 10;  // 0.8 ms of the wall clock
    for (int full = 1; full >= 0; --full) {
        const int blocks = full ? 256 * 8 : 256 * 2;  // 1024 threads per block: 8 blocks per CU fill 8 waves/SIMD... 2 fill a quarter
        for (int ns : {1, 2, 4}) {
            for (int mode = 0; mode < 3; ++mode) {  // 0: copy in + kernel + copy out, 1: copies only, 2: kernel only
                double best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int c = 0; c < chunks; ++c) {
                        const int s = c % ns;
                        if (mode != 2) CK(hipMemcpyAsync(d_in[s], h_in[s], in_b, hipMemcpyHostToDevice, st[s]));
                        if (mode != 1) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, st[s], static_cast<const uint32_t *>(d_in[s]), static_cast<uint32_t *>(d_out[s]), cycles);
                        if (mode != 2) CK(hipMemcpyAsync(h_out[s], d_out[s], out_b, hipMemcpyDeviceToHost, st[s]));
                    }
                    for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
                    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                }
                std::printf("kernel grid %5d blocks, %d stream(s), %s: %7.2f ms for %d chunks\n", blocks, ns,
                            mode == 0 ? "in + kernel + out" : mode == 1 ? "copies only     " : "kernel only     ", best * 1e3, chunks);
            }
        }
    }
    return 0;
}
