// gather_ceiling.hip -- the request ceiling bench.py's roofline.request_roof is priced against: the highest rate
// at which this MI355X serves random aligned 16-byte gathers from a table far larger than its caches, swept over
// the memory-level parallelism a lane offers (C independent chains per lane, each chain's next address depending
// on the data it just loaded, like the LF steps of one read) and over occupancy, plus the limit case of
// addresses that depend on nothing (a counter hash: every load of a lane independent).  The ceiling is the
// maximum over the sweep; a search kernel issuing one dependent gather per lane and step cannot exceed it.
// Not part of the library.   build: hipcc -O3 --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling
// usage: gather_ceiling [table GiB = 16] [steps = 256] [out.json]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

template <int C, bool DEPENDENT>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ tab, u64 nslots, int steps, u64 *out) {
    u64 x[C];
    u64 acc = 0;
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = (tid * C + c) * 0x9E3779B97F4A7C15ull + 12345;
    for (int s = 0; s < steps; ++s) {
        uint4 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            x[c] = mix(x[c]);
            v[c] = tab[x[c] % nslots];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            acc += v[c].x;
            if (DEPENDENT) x[c] += v[c].x;  // the chain's next address needs this load
            else x[c] += 0x632BE59BD9B4E019ull;
        }
    }
    out[tid] = acc;
}

struct Row { int chains, waves, dependent; double gps, ms; };

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 256;
    const char *json = argc > 3 ? argv[3] : nullptr;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    uint4 *tab = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc %zu failed\n", bytes); return 1; }
    hipMemset(tab, 1, bytes);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    u64 *out = nullptr;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const u64 nslots = bytes / 16;
    std::vector<Row> rows;
    printf("%s, %d CUs, table %.1f GiB of 16-byte slots, %d gathers per chain\n", prop.gcnArchName, cus, gib, steps);
    auto run = [&](auto kern, int C, bool dep, int waves) {
        const int blocks = cus * waves;  // 256-lane blocks: `waves` waves per SIMD
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nslots, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double acc = (double)blocks * 256 * steps * C;
        rows.push_back({C, waves, dep ? 1 : 0, acc / best / 1e6, best});
        printf("  %s chains/lane %2d  waves/SIMD %d : %8.2f ms  %6.2f G gathers/s  (64-B sectors: %.2f TB/s)\n", dep ? "dependent  " : "independent", C,
               waves, best, acc / best / 1e6, acc * 64 / best / 1e9);
    };
    for (int waves : {2, 4, 8}) {
        run(gather<1, true>, 1, true, waves);
        run(gather<2, true>, 2, true, waves);
        run(gather<4, true>, 4, true, waves);
        run(gather<8, true>, 8, true, waves);
    }
    for (int waves : {4, 8}) {
        run(gather<4, false>, 4, false, waves);
        run(gather<8, false>, 8, false, waves);
        run(gather<16, false>, 16, false, waves);
    }
    double peak = 0;
    for (const Row &r : rows) peak = r.gps > peak ? r.gps : peak;
    printf("ceiling: %.2f G random 16-byte gathers/s\n", peak);
    if (json) {
        FILE *f = fopen(json, "w");
        if (!f) return 1;
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"table_gib\": %.1f, \"gathers_per_chain\": %d, \"peak_G_gathers_per_s\": %.3f,\n \"rows\": [", prop.gcnArchName, cus, gib, steps, peak);
        for (size_t i = 0; i < rows.size(); ++i)
            fprintf(f, "%s\n  {\"chains_per_lane\": %d, \"waves_per_simd\": %d, \"dependent\": %s, \"G_gathers_per_s\": %.3f, \"ms\": %.3f}", i ? "," : "",
                    rows[i].chains, rows[i].waves, rows[i].dependent ? "true" : "false", rows[i].gps, rows[i].ms);
        fprintf(f, "\n ]}\n");
        fclose(f);
    }
    return 0;
}
