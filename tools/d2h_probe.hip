// Probe (GPU box): ways of getting a 3 GB ragged result from HBM into caller-visible host memory.
// build: hipcc -O2 --offload-arch=gfx950 tools/d2h_probe.hip -o /tmp/d2h_probe -pthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const size_t bytes = size_t(3) << 30;
    void *d = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        char *h = static_cast<char *>(malloc(bytes));
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        double t1 = now();
        printf("malloc + hipMemcpy D2H (first touch by the copy):      %.3f s  %.1f GB/s\n", t1 - t0, bytes / 1e9 / (t1 - t0));
        t0 = now();
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        t1 = now();
        printf("hipMemcpy D2H into already-touched pageable memory:     %.3f s  %.1f GB/s\n", t1 - t0, bytes / 1e9 / (t1 - t0));
        free(h);
        t0 = now();
        h = static_cast<char *>(aligned_alloc(size_t(2) << 20, bytes));
        int mrc = madvise(h, bytes, MADV_HUGEPAGE);
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        t1 = now();
        printf("aligned_alloc(2MB) + MADV_HUGEPAGE (rc %d) + hipMemcpy D2H:  %.3f s  %.1f GB/s\n", mrc, t1 - t0, bytes / 1e9 / (t1 - t0));
        free(h);
        t0 = now();
        void *p = nullptr;
        CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        double t1a = now();
        CK(hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost));
        t1 = now();
        printf("hipHostMalloc %.3f s + hipMemcpy D2H %.3f s:              %.3f s  %.1f GB/s overall\n", t1a - t0, t1 - t1a, t1 - t0, bytes / 1e9 / (t1 - t0));
        double t2 = now();
        CK(hipHostFree(p));
        printf("hipHostFree %.3f s\n", now() - t2);
        // pinned ring + threaded memcpy into fresh malloc memory
        for (int nthreads : {1, 4, 8}) {
            const size_t chunk = size_t(64) << 20;
            void *ring[2];
            CK(hipHostMalloc(&ring[0], chunk, hipHostMallocDefault));
            CK(hipHostMalloc(&ring[1], chunk, hipHostMallocDefault));
            hipStream_t st;
            CK(hipStreamCreate(&st));
            hipEvent_t ev[2];
            CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
            t0 = now();
            char *dst = static_cast<char *>(malloc(bytes));
            const size_t nchunks = (bytes + chunk - 1) / chunk;
            CK(hipMemcpyAsync(ring[0], d, chunk, hipMemcpyDeviceToHost, st));
            CK(hipEventRecord(ev[0], st));
            for (size_t c = 0; c < nchunks; ++c) {
                if (c + 1 < nchunks) {
                    CK(hipMemcpyAsync(ring[(c + 1) & 1], static_cast<char *>(d) + (c + 1) * chunk, chunk, hipMemcpyDeviceToHost, st));
                    CK(hipEventRecord(ev[(c + 1) & 1], st));
                }
                CK(hipEventSynchronize(ev[c & 1]));
                std::vector<std::thread> th;
                const size_t part = chunk / nthreads;
                for (int t = 0; t < nthreads; ++t)
                    th.emplace_back([&, t] { memcpy(dst + c * chunk + t * part, static_cast<char *>(ring[c & 1]) + t * part, part); });
                for (auto &x : th) x.join();
            }
            t1 = now();
            printf("pinned 2 x 64 MB ring + %d-thread memcpy into malloc:      %.3f s  %.1f GB/s\n", nthreads, t1 - t0, bytes / 1e9 / (t1 - t0));
            free(dst);
            CK(hipHostFree(ring[0])); CK(hipHostFree(ring[1]));
            CK(hipStreamDestroy(st));
        }
    }
    return 0;
}
