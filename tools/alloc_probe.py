#!/usr/bin/env python3
"""What a large hipMalloc costs on this platform (measurement for DESIGN.md: the 218 GB arena of the bench replica):
fresh process vs. right after a free, one allocation vs. several threads allocating at once.  usage: alloc_probe.py [serial|parallel]"""
import ctypes as C, sys, threading, time
hip = C.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = C.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes))
    assert rc == 0, rc
    return p, time.perf_counter() - t
def free(p):
    t = time.perf_counter(); hip.hipFree(p); return time.perf_counter() - t
mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
hip.hipInit(0); hip.hipSetDevice(0)
GB = 10**9
if mode == "serial":
    p, t = malloc(200 * GB); print(f"fresh process: hipMalloc(200 GB) {t:.2f} s")
    hip.hipMemset(p, 1, C.c_size_t(200 * GB)); hip.hipDeviceSynchronize()
    print(f"hipFree {free(p):.2f} s")
    p, t = malloc(200 * GB); print(f"again right away: hipMalloc(200 GB) {t:.2f} s"); free(p)
    for gb in (1, 5, 25, 100):
        p, t = malloc(gb * GB); tf = free(p); print(f"hipMalloc({gb} GB) {t:.3f} s, hipFree {tf:.3f} s")
else:
    out = [None] * 8
    def work(i): out[i] = malloc(25 * GB)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"fresh process: 8 threads x hipMalloc(25 GB): {time.perf_counter() - t0:.2f} s wall; each {[round(o[1], 2) for o in out]}")
