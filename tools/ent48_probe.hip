// ent48_probe.hip -- what the sixteen-entry probe of the run-indexed layout (rbg_runs_device.hpp quad_load) costs as a
// function of the ENTRY FORMAT at 8-byte positions, on this MI355X.  A quad of lanes fetches sixteen consecutive entries
// starting at a random entry index (the directory's answer), four entries per lane; four owners per quad take turns.
//   e16      16-byte entries {key, value} of u64: four 16-byte loads per lane, 16-byte aligned          (round 2's format)
//   e12x4    12-byte entries, the lane's 48 bytes as three 16-byte loads at a 4-byte boundary
//   e12x3    12-byte entries, one 12-byte load (dwordx3) per entry: four requests per lane
//   soa      {key_lo, value_lo} pairs (8 bytes) in one array, {key_hi, value_hi} 16-bit halves (4 bytes) in another:
//            two 16-byte loads at an 8-byte boundary (what the 4-byte-position path does) + one 16-byte load at a 4-byte boundary
//   soa8     the same, the halves as two 8-byte loads
//   e8       8-byte entries (the 4-byte-position format): two 16-byte loads at an 8-byte boundary          (reference point)
// Dependent chains (the next start needs the data), one chain per lane.  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 tools/ent48_probe.hip -o /tmp/ent48_probe ; usage: ent48_probe [GiB = 8] [steps = 128]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 v4a16 __attribute__((ext_vector_type(4)));
typedef u32 v4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef u32 v4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef u32 v3a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef u32 v2a4 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
template <int CTRL> __device__ __forceinline__ u32 dpp(u32 v) { return static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false)); }
__device__ __forceinline__ u32 quad_sum(u32 v) { v += dpp<0xB1>(v); v += dpp<0x4E>(v); return v; }
template <int J> __device__ __forceinline__ u64 quad_get64(u64 v) {
    return (static_cast<u64>(dpp<J * 0x55>(static_cast<u32>(v >> 32))) << 32) | dpp<J * 0x55>(static_cast<u32>(v));
}

enum { E16, E12X4, E12X3, SOA, SOA8, E8 };
// (every word that is loaded is used, so that the compiler keeps the loads as written)
__device__ __forceinline__ u32 fold(v4a16 v) { return v.x + 3u * v.y + 5u * v.z + 7u * v.w; }
__device__ __forceinline__ u32 fold(v3a4 v) { return v.x + 3u * v.y + 5u * v.z; }
__device__ __forceinline__ u32 fold(v2a4 v) { return v.x + 3u * v.y; }   // (the alignment is not part of a vector's type: one overload per length)

// the lane's four entries of the stretch that starts at entry `s`, folded into one word
template <int F>
__device__ __forceinline__ u32 fetch(const unsigned char *__restrict__ tab, const unsigned char *__restrict__ tab2, u64 s, u32 sub) {
    const u64 g = s + 4u * sub;
    if (F == E16) {
        const v4a16 *p = reinterpret_cast<const v4a16 *>(tab) + g;
        const v4a16 a = p[0], b = p[1], c = p[2], d = p[3];
        return fold(a) + fold(b) + fold(c) + fold(d);
    } else if (F == E12X4) {
        const v4a4 *p = reinterpret_cast<const v4a4 *>(tab + 12 * g);
        const v4a4 a = p[0], b = p[1], c = p[2];
        return fold(a) + fold(b) + fold(c);
    } else if (F == E12X3) {
        const v3a4 *p = reinterpret_cast<const v3a4 *>(tab + 12 * g);
        const v3a4 a = p[0], b = p[1], c = p[2], d = p[3];
        return fold(a) + fold(b) + fold(c) + fold(d);
    } else if (F == SOA) {
        const v4a8 *p = reinterpret_cast<const v4a8 *>(tab + 8 * g);
        const v4a8 a = p[0], b = p[1];
        const v4a4 h = *reinterpret_cast<const v4a4 *>(tab2 + 4 * g);
        return fold(a) + fold(b) + fold(h);
    } else if (F == SOA8) {
        const v4a8 *p = reinterpret_cast<const v4a8 *>(tab + 8 * g);
        const v4a8 a = p[0], b = p[1];
        const v2a4 *q = reinterpret_cast<const v2a4 *>(tab2 + 4 * g);
        const v2a4 h0 = q[0], h1 = q[1];
        return fold(a) + fold(b) + fold(h0) + fold(h1);
    } else {
        const v4a8 *p = reinterpret_cast<const v4a8 *>(tab + 8 * g);
        const v4a8 a = p[0], b = p[1];
        return fold(a) + fold(b);
    }
}

template <int F>
__global__ __launch_bounds__(256) void probe(const unsigned char *__restrict__ tab, const unsigned char *__restrict__ tab2, u64 nent, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    const u32 sub = threadIdx.x & 3u;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const u64 st = x % (nent - 32);
        const u32 q0 = fetch<F>(tab, tab2, quad_get64<0>(st), sub);
        const u32 q1 = fetch<F>(tab, tab2, quad_get64<1>(st), sub);
        const u32 q2 = fetch<F>(tab, tab2, quad_get64<2>(st), sub);
        const u32 q3 = fetch<F>(tab, tab2, quad_get64<3>(st), sub);
        const u32 s0 = quad_sum(q0), s1 = quad_sum(q1), s2 = quad_sum(q2), s3 = quad_sum(q3);
        const u32 t = sub == 0 ? s0 : sub == 1 ? s1 : sub == 2 ? s2 : s3;
        acc += t;
        x += t;
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 128;
    const size_t bytes = (size_t)(gib * (1ull << 30));
    unsigned char *tab = nullptr, *tab2 = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&tab2, bytes / 2) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(tab, 1, bytes);
    (void)hipMemset(tab2, 1, bytes / 2);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    u64 *out = nullptr;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // the same NUMBER of entries in every format (so the same spread of starts): what a 16-byte-entry table of `gib` holds
    const u64 nent = bytes / 16;
    printf("%s, %d CUs, %llu entries (%.1f GiB at 16 bytes each), %d dependent probes per lane, 16 entries per probe\n", prop.gcnArchName, cus, nent, gib, steps);
    auto run = [&](const char *name, auto kern, int ent_bytes, int waves) {
        const int blocks = cus * waves;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, tab2, nent, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double probes = (double)blocks * 256 * steps;
        printf("  %-6s waves/SIMD %d : %8.2f ms  %6.2f G probes/s  (%d-byte entries: %.2f TB/s of entries)\n", name, waves, best, probes / best / 1e6, ent_bytes,
               probes * 16 * ent_bytes / best / 1e9);
    };
    for (int waves : {2, 4, 8}) {
        run("e16", probe<E16>, 16, waves);
        run("e12x4", probe<E12X4>, 12, waves);
        run("e12x3", probe<E12X3>, 12, waves);
        run("soa", probe<SOA>, 12, waves);
        run("soa8", probe<SOA8>, 12, waves);
        run("e8", probe<E8>, 8, waves);
    }
    return 0;
}
