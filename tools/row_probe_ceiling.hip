// row_probe_ceiling.hip -- what the chip serves when a 16-lane ROW reads one short stretch of consecutive 8-byte
// entries at a random place (the run-indexed layout's probe, k_runs.hip coop_probe2_at), swept over the number of
// entries the row actually loads (lanes beyond it re-read its last entry) and over the alignment of the stretch's start
// (128-byte line, 64-byte sector, any 8-byte entry): is the ceiling one of sectors, of lines or of requests?
// Sixteen probes per lane are in flight, as in the kernel (one round per owner lane).  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 tools/row_probe_ceiling.hip -o tools/row_probe_ceiling
// usage: row_probe_ceiling [table GiB = 16] [steps = 64]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

// WIDTH entries per probe, start aligned to ALIGN entries
template <int WIDTH, int ALIGN>
__global__ __launch_bounds__(256) void probe(const uint2 *__restrict__ tab, u64 nent, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    const unsigned sub = threadIdx.x & 15u;
    const u64 row = tid >> 4;
    u64 acc = 0;
    u64 x = row * 0x9E3779B97F4A7C15ull + 12345;
    for (int s = 0; s < steps; ++s) {
        uint2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = mix(x + 0x632BE59BD9B4E019ull);
            const u64 start = (x % (nent - 64)) / ALIGN * ALIGN;
            v[j] = tab[start + (sub < WIDTH ? sub : WIDTH - 1)];   // lanes beyond the stretch re-read its last entry: no branch, no further sector
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j].x;
    }
    out[tid] = acc;
}

// the other shape: every LANE reads one aligned record of BYTES bytes (BYTES / 16 loads of 16 bytes, all of one sector
// or line), C dependent chains per lane (the next record's address needs this one's data, like an LF step)
template <int BYTES, int C>
__global__ __launch_bounds__(256) void lane_record(const uint4 *__restrict__ tab, u64 nrec, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    u64 x[C];
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = (tid * C + c) * 0x9E3779B97F4A7C15ull + 12345;
    for (int s = 0; s < steps; ++s) {
        uint4 v[C][BYTES / 16];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            x[c] = mix(x[c]);
            const uint4 *r = tab + (x[c] % nrec) * (BYTES / 16);
#pragma unroll
            for (int k = 0; k < BYTES / 16; ++k) v[c][k] = r[k];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            u64 sum = 0;
#pragma unroll
            for (int k = 0; k < BYTES / 16; ++k) sum += v[c][k].x ^ v[c][k].w;
            acc += sum;
            x[c] += sum;
        }
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 512;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    uint2 *tab = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", bytes); return 1; }
    hipMemset(tab, 1, bytes);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int waves = 4;
    const int blocks = cus * waves;
    u64 *out = nullptr;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const u64 nent = bytes / 8;
    printf("%s, %d CUs, table %.1f GiB of 8-byte entries, %d waves/SIMD, 16 probes per lane in flight, %d rounds\n", prop.gcnArchName, cus, gib, waves, steps);
    printf("  entries loaded  start aligned to   ms      G probes/s   sectors/probe   G sectors/s   lines/probe  G lines/s\n");
    auto run = [&](auto kern, int width, int align) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nent, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double probes = (double)blocks * 16 * steps * 16;   // rows x rounds x 16 probes
        // expected 64-byte sectors / 128-byte lines a stretch of `width` entries touches when its start is uniform over multiples of `align` entries
        auto touched = [&](int unit) {
            double sum = 0;
            int cnt = 0;
            for (int o = 0; o < unit; o += align) { sum += (o + width - 1) / unit + 1; ++cnt; }
            return cnt ? sum / cnt : 1.0;
        };
        const double spp = align >= 8 ? (width + 7) / 8 : touched(8), lpp = align >= 16 ? 1.0 : touched(16);
        printf("  %14d  %9d bytes  %7.2f  %10.2f  %14.2f  %12.2f  %12.2f  %9.2f\n", width, align * 8, best, probes / best / 1e6, spp, probes * spp / best / 1e6, lpp,
               probes * lpp / best / 1e6);
        fflush(stdout);
    };
    run(probe<16, 16>, 16, 16);
    run(probe<16, 8>, 16, 8);
    run(probe<16, 1>, 16, 1);
    run(probe<8, 8>, 8, 8);
    run(probe<8, 1>, 8, 1);
    run(probe<6, 1>, 6, 1);
    run(probe<4, 1>, 4, 1);
    run(probe<2, 1>, 2, 1);
    run(probe<1, 1>, 1, 1);
    printf("  one aligned record per lane, dependent chains:\n  record bytes  chains/lane  waves/SIMD     ms     G records/s   G sectors/s\n");
    auto runl = [&](auto kern, int rb, int C, int wv) {
        float best = 1e30f;
        const int blk = cus * wv;
        const int st = steps / 2;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blk), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(tab), (u64)(bytes / rb), st, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double recs = (double)blk * 256 * st * C;
        printf("  %12d  %11d  %10d  %7.2f  %12.2f  %12.2f\n", rb, C, wv, best, recs / best / 1e6, recs * (rb > 64 ? rb / 64 : 1) / best / 1e6);
        fflush(stdout);
    };
    runl(lane_record<16, 2>, 16, 2, 4);
    runl(lane_record<16, 2>, 16, 2, 8);
    runl(lane_record<32, 2>, 32, 2, 4);
    runl(lane_record<32, 2>, 32, 2, 8);
    runl(lane_record<64, 1>, 64, 1, 4);
    runl(lane_record<64, 2>, 64, 2, 4);
    runl(lane_record<64, 1>, 64, 1, 8);
    runl(lane_record<64, 2>, 64, 2, 8);
    runl(lane_record<128, 1>, 128, 1, 4);
    runl(lane_record<128, 2>, 128, 2, 4);
    return 0;
}
