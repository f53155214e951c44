// rbg_hostpath.hpp -- the host-pointer search calls (rbg_find_range, rbg_count, rbg_find_range_w_toehold and the
// span variants the command-line tools use) as a pipeline that keeps up with the kernels (next-row f2):
//   * no allocation per call: device buffers and pinned staging live in a workspace kept with the index (one per
//     concurrent caller, grown on demand, released by rbg_free);
//   * reads cross PCIe as 2-bit codes when the index's k-mer alphabet is ACGT: worker threads pack each read
//     straight from the caller's bytes into the layout k_find_range_packed consumes (a quarter of the bytes, and no
//     pack kernel: the bytes are touched once, on the host, where the copy into pinned memory had to touch them
//     anyway); reads with any other symbol go through the byte kernel afterwards;
//   * chunks of the batch go through kSlots buffers on as many streams: the host packs ahead while up to kSlots - 1
//     chunks are in flight (copy in, search and copy back of successive chunks overlap one another on the device), and
//     finished chunks are handed to the caller between two packing passes without ever waiting for the GPU unless
//     every buffer is busy; results leave through pinned memory.
// Included by rbg_capi.hip only (it needs rbg_index).
#pragma once

#include "rbg_numa.hpp"
#include "rbg_pack2bit.hpp"
#include "rbg_thread_team.hpp"

namespace rbg_hostpath {

// ---- pinned + device buffers of one in-flight chunk ------------------------------------------------------------------
struct Slot {
    void *h_in = nullptr, *d_in = nullptr;     // packed: [meta uint2[C] | chunks];  raw: [off u64[C+1] | bytes]
    size_t in_cap = 0;
    void *h_out = nullptr, *d_out = nullptr;   // lo | hi | ssamp (or count), C entries each
    size_t out_cap = 0;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    // what is in flight in it
    uint64_t begin = 0, cnt = 0;
    bool busy = false;
};

constexpr unsigned kSlots = 4;

struct Workspace {
    int device = -1;
    Slot slot[kSlots];
    std::unique_ptr<ThreadTeam> team;
    std::vector<std::vector<uint64_t>> bad;  // per team member: reads (batch indices) the packed form cannot express
    ~Workspace() { release(); }
    void release() {
        for (Slot &s : slot) {
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.h_out) (void)hipHostFree(s.h_out);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.st) (void)hipStreamDestroy(s.st);
            s = Slot();
        }
    }
    // 0 on success
    int ensure(Slot &s, size_t in_bytes, size_t out_bytes) {
        if (!s.st && hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) return 1;
        if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return 1;
        if (in_bytes > s.in_cap) {
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.d_in) (void)hipFree(s.d_in);
            s.h_in = s.d_in = nullptr;
            s.in_cap = 0;
            const size_t cap = in_bytes + in_bytes / 4 + 4096;
            if (rbg_numa::host_malloc_near(&s.h_in, cap, hipHostMallocDefault, device) != hipSuccess) return 2;
            if (hipMalloc(&s.d_in, cap) != hipSuccess) return 2;
            s.in_cap = cap;
        }
        if (out_bytes > s.out_cap) {
            if (s.h_out) (void)hipHostFree(s.h_out);
            if (s.d_out) (void)hipFree(s.d_out);
            s.h_out = s.d_out = nullptr;
            s.out_cap = 0;
            const size_t cap = out_bytes + out_bytes / 4 + 4096;
            if (rbg_numa::host_malloc_near(&s.h_out, cap, hipHostMallocDefault, device) != hipSuccess) return 2;
            if (hipMalloc(&s.d_out, cap) != hipSuccess) return 2;
            s.out_cap = cap;
        }
        return 0;
    }
};

}  // namespace rbg_hostpath
