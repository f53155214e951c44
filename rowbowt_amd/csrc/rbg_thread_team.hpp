// rbg_thread_team.hpp -- the worker threads of the host-pointer pipeline (rbg_hostpath.hpp).  Plain C++17, no HIP:
// unit-tested on the CPU under ThreadSanitizer (tests/cpp/thread_team_check.cpp).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

namespace rbg_hostpath {

// CPUs this process may actually use: the smaller of the hardware's count, the affinity mask and the container's CPU
// bandwidth quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us).  A team larger than the quota runs in bursts and is then
// throttled for the rest of the scheduling period (50-90 ms stalls on a 16-CPU quota with 256 visible CPUs: measured).
inline unsigned cpu_budget() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) n = std::min(n, static_cast<unsigned>(c));
    }
    auto quota = [](const char *path, const char *path_period) -> double {
        FILE *f = std::fopen(path, "r");
        if (!f) return 0;
        char a[64] = {0}, b[64] = {0};
        const int got = std::fscanf(f, "%63s %63s", a, b);
        std::fclose(f);
        if (got < 1 || a[0] == 'm' || a[0] == '-') return 0;   // "max" / -1: no quota
        double q = std::atof(a), per = got >= 2 ? std::atof(b) : 0;
        if (path_period) {
            FILE *g = std::fopen(path_period, "r");
            if (g) { if (std::fscanf(g, "%63s", b) == 1) per = std::atof(b); std::fclose(g); }
        }
        return (q > 0 && per > 0) ? q / per : 0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (q > 0) n = std::min(n, std::max(1u, static_cast<unsigned>(q + 0.5)));
#endif
    return n;
}

// ---- a team of worker threads that runs one function on every member and waits ------------------------------------
// A host call runs some thirty short passes (sizing, packing, handing results back: a fraction of a millisecond each)
// in a burst, so waking the members through a condition variable every time cost more than the passes' work (about
// 0.1 ms per pass with 64 members: measured).  Members therefore spin on the generation counter for a short while
// (60 us: the passes of one chunk follow one another within that) after a pass before they go to sleep, and the caller
// spins on the completion counter: a pass that follows within the window starts within a microsecond, and an idle team
// sleeps -- spinning is CPU time too, and a container's quota counts it.
class ThreadTeam {
   public:
    explicit ThreadTeam(unsigned n) : n_(n ? n : 1) {
        for (unsigned t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~ThreadTeam() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_.store(true, std::memory_order_release);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned size() const { return n_; }
    void run(const std::function<void(unsigned)> &fn) {  // fn(member) on every member; returns when all are done
        if (n_ == 1) { fn(0); return; }
        fn_ = &fn;
        left_.store(n_ - 1, std::memory_order_relaxed);
        gen_.fetch_add(1);                       // (sequentially consistent, like the members' sleepers_ / gen_ pair below:
        if (sleepers_.load() > 0) {              //  either the member sees the new generation or the caller sees the sleeper)
            { std::lock_guard<std::mutex> g(mu_); }   // a member between its last check and its wait sees the new generation
            cv_.notify_all();
        }
        fn(0);
        unsigned spins = 0;
        while (left_.load(std::memory_order_acquire) != 0) {
            if (++spins < 4096) cpu_relax(); else std::this_thread::yield();
        }
        fn_ = nullptr;
    }

   private:
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void loop(unsigned t) {
        uint64_t seen = 0;
        while (true) {
            // spin for a while, then sleep
            bool got = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; ++it) {
                if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
                cpu_relax();
                if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(60)) break;
            }
            if (!got) {
                std::unique_lock<std::mutex> g(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(g, [&] { return gen_.load() != seen; });
                sleepers_.fetch_sub(1);
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire)) return;
            (*fn_)(t);
            left_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    unsigned n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    std::atomic<unsigned> left_{0}, sleepers_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace rbg_hostpath
