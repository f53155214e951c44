// rbg_numa.hpp -- pinned host buffers on the GPU's own NUMA node, and a trace of where they are.
// The copy engines move the results of a batch (3 GB of locations, 10 GB of rb_align -s text per 10 M reads) into pinned host
// memory; on a two-socket host a copy into the far socket's memory crosses the socket interconnect.  Round 3 left the
// process-to-process bimodality of `rb_align -s -m` (3.4e7 against 2.0e7 reads/s) at "a property of the process's memory", and
// this was the suspect.  Measured (profiles/r04_numa_probe.txt, 44 fresh processes started on either socket): NOT the cause -- on
// this platform hipHostMalloc already puts the pages on the GPU's node wherever the calling thread runs (every buffer of every
// process: node 1 = the GPU's).  The cause was a fourth 0.7 GB text buffer allocated inside the timed loop whenever the writer had
// not yet given one of the three reserved ones back (rb_align.cpp reserves four now: 7 % spread).  What stays of the suspicion:
// the allocating thread is moved onto the CPUs of the GPU's node for the duration of the allocation (sched_setaffinity; no
// libnuma), so that the placement does not depend on the runtime doing it; RBG_PIN_NUMA=0 turns that off; RBG_NUMA_TRACE=1 says on
// stderr where the process, the GPU and the first page of each pinned buffer are -- the trace that found the real cause.
#pragma once

#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace rbg_numa {

inline int read_int_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    int v = -1;
    if (std::fscanf(f, "%d", &v) != 1) v = -1;
    std::fclose(f);
    return v;
}

// NUMA node of HIP device `device` (-1: unknown / no NUMA)
inline int gpu_node(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; ++c) *c = static_cast<char>(std::tolower(static_cast<unsigned char>(*c)));
    return read_int_file(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
}

// the CPUs of a node as a cpu_set_t (false: unknown)
inline bool node_cpus(int node, cpu_set_t *set) {
    CPU_ZERO(set);
    FILE *f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool ok = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!ok) return false;
    // "0-63,128-191": parsed by hand -- no strtok, whose hidden state is shared by every thread of the process
    int n = 0;
    for (const char *c = buf; *c;) {
        if (!std::isdigit(static_cast<unsigned char>(*c))) { ++c; continue; }
        char *end = nullptr;
        long a = std::strtol(c, &end, 10), b = a;
        if (*end == '-' && std::isdigit(static_cast<unsigned char>(end[1]))) b = std::strtol(end + 1, &end, 10);
        for (long v = a; v <= b && v < CPU_SETSIZE; ++v) { CPU_SET(static_cast<int>(v), set); ++n; }
        c = end;
    }
    return n > 0;
}

// what host_malloc_near needs to know about a device, read from the system once per device and process (the library allocates
// pinned buffers from many threads at once: per-replica threads, workers)
struct DeviceNode {
    std::once_flag once;
    int node = -1;
    bool have_cpus = false;
    cpu_set_t cpus;
};
inline DeviceNode &device_node(int device) {
    static DeviceNode tab[64];
    DeviceNode &d = tab[device >= 0 && device < 64 ? device : 0];
    std::call_once(d.once, [&] {
        d.node = gpu_node(device);
        d.have_cpus = d.node >= 0 && node_cpus(d.node, &d.cpus);
    });
    return d;
}

// node of the CPU this thread runs on right now (-1: unknown)
inline int cpu_node() {
    unsigned cpu = 0, node = 0;
    if (syscall(SYS_getcpu, &cpu, &node, nullptr) != 0) return -1;
    return static_cast<int>(node);
}

// node that holds the page at p (-1: unknown); the page must have been touched
inline int page_node(const void *p) {
    void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095));
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0) return -1;
    return status;
}

inline bool enabled() {
    const char *e = std::getenv("RBG_PIN_NUMA");
    return !(e && e[0] == '0');
}

// pinned host memory for copies to / from `device`, on that device's NUMA node when the machine says which it is
inline hipError_t host_malloc_near(void **p, size_t bytes, unsigned flags, int device) {
    cpu_set_t before;
    bool moved = false;
    DeviceNode *dn = enabled() ? &device_node(device) : nullptr;
    if (dn && dn->have_cpus && sched_getaffinity(0, sizeof(before), &before) == 0) {
        cpu_set_t both;
        CPU_AND(&both, &before, &dn->cpus);      // (never outside what the process may use)
        if (CPU_COUNT(&both) > 0 && sched_setaffinity(0, sizeof(both), &both) == 0) moved = true;
    }
    const int pnode = cpu_node();
    const hipError_t e = hipHostMalloc(p, bytes, flags);
    if (moved) (void)sched_setaffinity(0, sizeof(before), &before);
    if (e == hipSuccess && std::getenv("RBG_NUMA_TRACE")) {
        static_cast<volatile char *>(*p)[0] = 0;
        std::fprintf(stderr, "rbg: pinned buffer of %.0f MB: allocating thread on NUMA node %d, GPU %d on node %d, first page on node %d (RBG_PIN_NUMA %s)\n",
                     bytes / 1e6, pnode, device, device_node(device).node, page_node(*p), enabled() ? "on" : "off");
    }
    return e;
}

}  // namespace rbg_numa
