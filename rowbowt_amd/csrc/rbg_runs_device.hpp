// rbg_runs_device.hpp -- what the kernels of the run-indexed layout share beside the search itself (rbg_runs2_device.hpp):
// the wave-local LDS fence, the one-off raise of a kernel's dynamic LDS limit, the step's record lookup and result.
// (Until round 4 this file held the wave-cooperative probes of the layout's first format -- 16-lane rows, quads, 128-byte
// bucket records; they lost every A/B to one lane per query and were retired: profiles/r04_fmt_ab.txt, DESIGN_HISTORY.md.)
// Everything lives in an anonymous namespace: each unit gets its own inlined copy.
#pragma once

#include <map>

#include "rbg_device.hpp"

namespace rbg {
namespace {

__device__ __forceinline__ void wave_lds_sync() {  // as in k_locate.hip: orders the wave's own LDS writes and cross-lane reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// `bytes` = the launch's DYNAMIC LDS (what the attribute limits); `static_bytes` = the kernel's static __shared__ arrays, which count
// towards the 48 KB a kernel gets without asking but are not part of the attribute's value
template <typename Kernel>
void raise_lds(Kernel kernel, size_t bytes, size_t static_bytes = 0) {
    if (bytes + static_bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> raised;   // the largest size raised so far per (device, kernel)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, reinterpret_cast<const void *>(kernel));
    std::lock_guard<std::mutex> g(mu);
    size_t &have = raised[key];
    if (bytes <= have) return;
    // (a later index on the same device may need more than the first one did: raise again, and say so when the runtime refuses --
    //  the launch that follows would fail with an unrelated-looking error)
    const hipError_t e = hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (e == hipSuccess) have = bytes;
    else std::fprintf(stderr, "rbg: raising a kernel's dynamic LDS limit to %zu bytes failed: %s\n", bytes, hipGetErrorString(e));
}

// the record of the step that consumes the k-mer `acc` (read as a base-nmajor number, least significant digit = the
// symbol next to the suffix) of `adv` symbols; adv == 1: acc is the symbol's slot
__device__ __forceinline__ uint32_t run_record(const uint32_t *s_tab_first, uint32_t adv, uint32_t acc) { return s_tab_first[adv - 1] + acc; }

// result of one lane's step
struct RunStep {
    uint64_t F = 0, c_before = 0, c_upto = 0;   // F + rank(lo, .), F + rank(hi + 1, .) with F the table's first row: the run lists' cums carry it (F itself stays 0)
    bool inside = false;                        // row hi lies in a run of the table (LF_w_loc's fast branch, rowbowt.hpp:559-561)
    uint64_t samp_e = 0;                        // else the predecessor run's sample: that entry's index relative to the table's first entry
};

// What the instrumented instantiations count on this layout (the same eight sums as SearchStat, other meanings:
// include/rbg.h): [kStSteps] search steps, [kStSlots] bucket records fetched (64 bytes) or directory gathers (two neighbouring
// entries: 8 or 16 bytes), [kStDense] run-list entries the scans needed (8 bytes each), [kStSearch] narrowing rounds (seven
// 4-byte pivots each), [kStFtab], [kStResample] (one sample gather), [kStChunks], [kStSymbols].

}  // namespace
}  // namespace rbg
