// thread_team_check.cpp -- rbg_hostpath::ThreadTeam (rbg_thread_team.hpp): every member runs every pass exactly once,
// passes in quick succession (members still spinning) and after pauses (members asleep), teams of one, teams
// created and destroyed while idle or asleep.  Built with -fsanitize=thread by tests/test_capi_host.py.
#include <cstdio>
#include <numeric>

#include "../../rowbowt_amd/csrc/rbg_thread_team.hpp"

int main() {
    using rbg_hostpath::ThreadTeam;
    for (unsigned n : {1u, 2u, 7u, 16u}) {
        ThreadTeam team(n);
        std::vector<uint64_t> hits(n, 0);
        uint64_t plain = 0;  // written by member 0 only, read by the caller after the pass: ordered by run()
        for (int pass = 0; pass < 3000; ++pass) {
            if (pass % 500 == 499) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // members go to sleep
            const std::function<void(unsigned)> fn = [&](unsigned t) {
                hits[t] += 1;
                if (t == 0) plain += static_cast<uint64_t>(pass);
            };
            team.run(fn);
            if (plain != static_cast<uint64_t>(pass) * (pass + 1) / 2) { std::printf("pass %d: member 0 out of step\n", pass); return 1; }
        }
        for (unsigned t = 0; t < n; ++t)
            if (hits[t] != 3000) { std::printf("team of %u: member %u ran %llu passes\n", n, t, static_cast<unsigned long long>(hits[t])); return 1; }
    }
    for (int k = 0; k < 50; ++k) {   // construction / destruction with members spinning or asleep
        ThreadTeam team(5);
        if (k & 1) std::this_thread::sleep_for(std::chrono::microseconds(700));
        std::atomic<int> c{0};
        const std::function<void(unsigned)> fn = [&](unsigned) { c.fetch_add(1); };
        if (k % 3) team.run(fn);
        if (k % 3 && c.load() != 5) { std::printf("short-lived team: %d of 5\n", c.load()); return 1; }
    }
    const unsigned budget = rbg_hostpath::cpu_budget();   // hardware, affinity mask and cgroup quota: at least one, never more than the hardware's
    if (budget < 1 || budget > std::max(1u, std::thread::hardware_concurrency())) { std::printf("cpu_budget() = %u\n", budget); return 1; }
    std::printf("thread team ok (cpu budget %u)\n", budget);
    return 0;
}
