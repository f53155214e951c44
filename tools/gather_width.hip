// gather_width.hip -- what the random-gather ceiling (tools/gather_ceiling.hip: 49 G aligned 16-byte gathers/s from a table beyond the caches) is a
// ceiling OF: the same dependent random gathers at growing width -- every lane reads W contiguous bytes at a W-aligned random address, W = 16 ... 512 --
// and, for two sectors per step, the two halves of ONE 128-byte line against two sectors of DIFFERENT lines.  If the rate in accesses/s holds from
// 16 to 128 bytes, the ceiling is DRAM accesses of 128 bytes (49 G/s x 128 B = 6.3 TB/s: the HBM rate of a sequential sweep), and a 64-byte record uses
// half of what its access moves.  Not part of the library.   build: hipcc -O3 --offload-arch=gfx950 tools/gather_width.hip -o tools/gather_width
// usage: gather_width [table GiB = 16] [steps = 128] [out.json]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

// V uint4s (16 V bytes) per access; SPLIT: the second half of the access comes from another random place
template <int V, bool SPLIT>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ tab, u64 nunits, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const uint4 *p = tab + (x % nunits) * V;
        const uint4 *q = SPLIT ? tab + (mix(x + 1) % nunits) * V + V / 2 : p + V / 2;
        uint4 v[V];
#pragma unroll
        for (int c = 0; c < V; ++c) v[c] = (c < V / 2 || V == 1) ? p[c] : q[c - V / 2];
        unsigned sum = 0;
#pragma unroll
        for (int c = 0; c < V; ++c) sum += v[c].x;
        acc += sum;
        x += sum;   // the chain's next address needs every piece of this access
    }
    out[tid] = acc;
}

// One dependent 16-byte gather from the big table per step (a miss) + HITS independent 8-byte gathers from a SMALL table (512 KB: resident in every
// XCD's L2 -- the run-indexed search's hot words of depth 8): what do cache-served scattered requests cost a kernel that sits at the miss ceiling?
template <int HITS>
__global__ __launch_bounds__(256) void gather_mix(const uint4 *__restrict__ tab, u64 nunits, const u64 *__restrict__ small, u64 nsmall, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, y = tid * 0xD6E8FEB86659FD93ull + 7, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const uint4 v = tab[x % nunits];
        u64 h = 0;
#pragma unroll
        for (int c = 0; c < HITS; ++c) { y = mix(y + c); h += small[y % nsmall]; }
        acc += v.x + h;
        x += v.x;
    }
    out[tid] = acc;
}

struct Row { int bytes, split, waves; double gps, tbs, ms; };

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 128;
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
    std::vector<Row> rows;
    printf("%s, %d CUs, table %.1f GiB, %d dependent accesses per lane\n", prop.gcnArchName, cus, gib, steps);
    auto run = [&](auto kern, int V, bool split, int waves) {
        const int blocks = cus * waves;
        const u64 nunits = bytes / (16ull * V);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nunits, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double acc = (double)blocks * 256 * steps;
        rows.push_back({16 * V, split ? 1 : 0, waves, acc / best / 1e6, acc * 16 * V / best / 1e9, best});
        printf("  %3d bytes per access%s  waves/SIMD %d : %8.2f ms  %6.2f G accesses/s  %5.2f TB/s of requested bytes\n", 16 * V,
               split ? " (two halves from different places)" : "                                   ", waves, best, acc / best / 1e6, acc * 16 * V / best / 1e9);
    };
    for (int waves : {4, 8}) {
        run(gather<1, false>, 1, false, waves);
        run(gather<2, false>, 2, false, waves);
        run(gather<4, false>, 4, false, waves);
        run(gather<8, false>, 8, false, waves);
        run(gather<8, true>, 8, true, waves);
        run(gather<16, false>, 16, false, waves);
        run(gather<32, false>, 32, false, waves);
    }
    {
        const u64 nsmall = (512u << 10) / 8;
        u64 *small = nullptr;
        hipMalloc(&small, nsmall * 8);
        hipMemset(small, 1, nsmall * 8);
        const u64 nunits = bytes / 16;
        auto runm = [&](auto kern, int hits, int waves) {
            const int blocks = cus * waves;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nunits, small, nsmall, steps, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double acc = (double)blocks * 256 * steps;
            rows.push_back({-hits, 0, waves, acc / best / 1e6, 0.0, best});
            printf("  one 16-byte miss + %d 8-byte gathers from a 512 KB table per step, waves/SIMD %d : %8.2f ms  %6.2f G steps/s  (%6.2f G requests/s)\n", hits, waves, best,
                   acc / best / 1e6, acc * (1 + hits) / best / 1e6);
        };
        for (int waves : {4, 8}) {
            runm(gather_mix<0>, 0, waves);
            runm(gather_mix<1>, 1, waves);
            runm(gather_mix<2>, 2, waves);
            runm(gather_mix<4>, 4, waves);
        }
    }
    if (json) {
        FILE *f = fopen(json, "w");
        if (!f) return 1;
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"table_gib\": %.1f, \"accesses_per_lane\": %d,\n \"rows\": [", prop.gcnArchName, cus, gib, steps);
        for (size_t i = 0; i < rows.size(); ++i)
            fprintf(f, "%s\n  {\"bytes_per_access (negative: one 16-byte miss + that many 8-byte gathers from a 512 KB table)\": %d, \"two_places\": %s, \"waves_per_simd\": %d, \"G_accesses_per_s\": %.3f, \"TB_per_s\": %.3f, \"ms\": %.3f}", i ? "," : "",
                    rows[i].bytes, rows[i].split ? "true" : "false", rows[i].waves, rows[i].gps, rows[i].tbs, rows[i].ms);
        fprintf(f, "\n ]}\n");
        fclose(f);
    }
    return 0;
}
