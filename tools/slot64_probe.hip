// slot64_probe.hip -- what a random ALIGNED 64-byte record costs on this MI355X, next to the 16-byte gather the rank
// slots are built around (rbg_dev.h RankSlot): the measurement behind the "use the 48 bytes the fabric already delivers"
// experiment.  A 16-byte slot arrives as a 64-byte sector anyway (profiles/pmc_traffic.json: padding
// ratio 3.6); a 64-byte slot over four times the rows would hold three times the inline runs, the run ordinal and the
// predecessor's sample -- IF fetching all 64 bytes costs no more requests than fetching 16.  Variants, all dependent
// chains (the next address needs the data, like the LF steps of a read), one chain per lane:
//   lane16    one 16-byte load per lane and step                        (the baseline: 49 G/s, gather_ceiling.hip)
//   lane64    the lane loads its own 64 bytes as four 16-byte loads     (four requests per record from one lane)
//   quad64    four rounds per step: in round J the four lanes of a quad load the four quarters of lane J's record
//             (one coalesced 64-byte request per record), every lane ends up with one quarter of four records and the
//             quad reduces them by DPP                                   (what a cooperative slot kernel would do)
//   pair32    two rounds, lane pairs, 32-byte records
// Not part of the library.  build: hipcc -O3 --offload-arch=gfx950 tools/slot64_probe.hip -o tools/slot64_probe
// usage: slot64_probe [table GiB = 16] [steps = 256]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
template <int CTRL> __device__ __forceinline__ unsigned dpp(unsigned v) {
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned quad_sum(unsigned v) { v += dpp<0xB1>(v); v += dpp<0x4E>(v); return v; }
template <int J> __device__ __forceinline__ u64 quad_get64(u64 v) {
    return (static_cast<u64>(dpp<J * 0x55>(static_cast<unsigned>(v >> 32))) << 32) | dpp<J * 0x55>(static_cast<unsigned>(v));
}

__global__ __launch_bounds__(256) void lane16(const uint4 *__restrict__ tab, u64 nrec, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const uint4 v = tab[(x % nrec) * 4];
        acc += v.x;
        x += v.x;
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(256) void lane64(const uint4 *__restrict__ tab, u64 nrec, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const uint4 *r = tab + (x % nrec) * 4;
        const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        const unsigned t = a.x + b.y + c.z + d.w;
        acc += t;
        x += t;
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(256) void quad64(const uint4 *__restrict__ tab, u64 nrec, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    const unsigned sub = threadIdx.x & 3u;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const u64 rec = x % nrec;
        // round J: every lane of the quad loads quarter `sub` of lane J's record (all four loads in flight together)
        const uint4 q0 = tab[quad_get64<0>(rec) * 4 + sub];
        const uint4 q1 = tab[quad_get64<1>(rec) * 4 + sub];
        const uint4 q2 = tab[quad_get64<2>(rec) * 4 + sub];
        const uint4 q3 = tab[quad_get64<3>(rec) * 4 + sub];
        // each owner needs something of its whole record: quad sums of the quarters, picked by owner
        const unsigned s0 = quad_sum(q0.x + q0.w), s1 = quad_sum(q1.x + q1.w), s2 = quad_sum(q2.x + q2.w), s3 = quad_sum(q3.x + q3.w);
        const unsigned t = sub == 0 ? s0 : sub == 1 ? s1 : sub == 2 ? s2 : s3;
        acc += t;
        x += t;
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(256) void pair32(const uint4 *__restrict__ tab, u64 nrec, int steps, u64 *out) {
    const u64 tid = blockIdx.x * 256ull + threadIdx.x;
    const unsigned sub = threadIdx.x & 1u;
    u64 x = tid * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
    for (int s = 0; s < steps; ++s) {
        x = mix(x);
        const u64 rec = x % nrec;   // 32-byte records
        const u64 r0 = (static_cast<u64>(dpp<0xA0>(static_cast<unsigned>(rec >> 32))) << 32) | dpp<0xA0>(static_cast<unsigned>(rec));   // quad_perm [0,0,2,2]
        const u64 r1 = (static_cast<u64>(dpp<0xF5>(static_cast<unsigned>(rec >> 32))) << 32) | dpp<0xF5>(static_cast<unsigned>(rec));   // quad_perm [1,1,3,3]
        const uint4 q0 = tab[r0 * 2 + sub], q1 = tab[r1 * 2 + sub];
        unsigned a = q0.x + q0.w, b = q1.x + q1.w;
        a += dpp<0xB1>(a);
        b += dpp<0xB1>(b);
        const unsigned t = sub == 0 ? a : b;
        acc += t;
        x += t;
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 256;
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
    printf("%s, %d CUs, table %.1f GiB, %d dependent steps per lane\n", prop.gcnArchName, cus, gib, steps);
    auto run = [&](const char *name, auto kern, size_t rec_bytes, size_t stride_bytes, int waves) {
        const int blocks = cus * waves;
        const u64 nrec = bytes / stride_bytes;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, nrec, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double recs = (double)blocks * 256 * steps;
        printf("  %-7s waves/SIMD %d : %8.2f ms  %6.2f G records/s  (%zu-byte records: %.2f TB/s of payload)\n", name, waves, best, recs / best / 1e6,
               rec_bytes, recs * rec_bytes / best / 1e9);
    };
    for (int waves : {2, 4, 8}) {
        run("lane16", lane16, 16, 64, waves);     // (one 16-byte load at the start of a random 64-byte line)
        run("lane64", lane64, 64, 64, waves);
        run("quad64", quad64, 64, 64, waves);
        run("pair32", pair32, 32, 32, waves);
    }
    return 0;
}
