// capi/replicas.ipp -- several GPUs in one process: peer-copied replicas, sharded search, counters and their RCCL all-reduce.  Part of rbg_capi.hip.
// ---- more than one GPU in one process (index replicated, reads sharded, no data-path collective) ----
// The replica is built ONCE (load / build on the primary's device) and copied to the other devices peer to peer
// (xGMI); records that hold device pointers (DevSym arrays, DevIndex) are re-pointed into the copy.



namespace {

// one target of a fan-out: the new handle, its stream (on the target device) and the relocation map of its copy
struct ReplicaJob {
    rbg_index *r = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the job's copies on its stream: the per-target duration (rbg_replicate_stats)
    Reloc reloc;
};

// allocate on `device` and ENQUEUE the peer copies of every allocation of `src` on the job's own stream: nothing here
// waits, so the copies of several targets run side by side (each target pulls over its own xGMI link to the source)
int replicate_begin(rbg_index *src, int device, ReplicaJob &job) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RBG_ENODEV;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RBG_ENODEV;
    rbg_index *r = new (std::nothrow) rbg_index();
    if (!r) return RBG_ENOMEM;
    job.r = r;
    r->primary = src;
    r->device = device;
    r->cfg = src->cfg;
    r->cfg.max_blocks = prop.multiProcessorCount * 32;
    r->runs_layout = src->runs_layout;
    r->run_depth_mask = src->run_depth_mask;
    r->runs_report = src->runs_report;
    r->rank_slots = src->rank_slots; r->rank_slots_overflow = src->rank_slots_overflow;
    r->phi_slots = src->phi_slots; r->phi_slots_overflow = src->phi_slots_overflow;
    r->kmer_steps_requested = src->kmer_steps_requested;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return RBG_ENODEV;
    r->hbm_free_at_load = free_b;
    r->hbm_budget = src->hbm_budget;
    if (device != src->device) {
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, device, src->device);
        if (can) (void)hipDeviceEnablePeerAccess(src->device, 0);  // already enabled is fine
        (void)hipGetLastError();
        r->replicate_peer = can ? 1 : 0;
        if (!can)   // (still correct: hipMemcpyPeerAsync stages through the host; said so that a slow fan-out has its reason on stderr)
            std::fprintf(stderr, "rbg: device %d has no peer access to device %d: the replica's %.1f GB are copied through the host\n", device, src->device, src->hbm_bytes / 1e9);
    }
    HIP_TRY(hipStreamCreateWithFlags(&job.st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&job.ev0));
    HIP_TRY(hipEventCreate(&job.ev1));
    HIP_TRY(hipEventRecord(job.ev0, job.st));
    for (const DevAlloc &a : src->allocs) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, a.bytes);
        if (e == hipSuccess) {
            r->allocs.push_back({p, a.bytes});
            r->hbm_bytes += a.bytes;
            e = hipMemcpyPeerAsync(p, device, a.p, src->device, a.bytes, job.st);
        }
        if (e != hipSuccess) {
            std::fprintf(stderr, "rbg: replicating %.1f GB to device %d failed: %s\n", src->hbm_bytes / 1e9, device, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV;
        }
        job.reloc.from.push_back(a);
        job.reloc.to.push_back({p, a.bytes});
    }
    HIP_TRY(hipEventRecord(job.ev1, job.st));
    return RBG_OK;
}

// wait for the job's copies, then re-point the records that hold device pointers
int replicate_finish(rbg_index *src, ReplicaJob &job) {
    rbg_index *r = job.r;
    const Reloc &reloc = job.reloc;
    DeviceScope scope(r->device);
    if (scope.rc) return scope.rc;
    if (hipError_t e = hipStreamSynchronize(job.st); e != hipSuccess) {
        std::fprintf(stderr, "rbg: the peer copies of the replica to device %d (from device %d) failed: %s\n", r->device, src->device, hipGetErrorString(e));
        return RBG_ENODEV;
    }
    float ms = 0.f;
    if (job.ev0 && job.ev1 && hipEventElapsedTime(&ms, job.ev0, job.ev1) == hipSuccess) r->replicate_ms = ms;
    for (const PtrTable &t : src->ptr_tables) {
        std::vector<char> buf(t.count * t.stride);
        void *dst = const_cast<void *>(reloc(t.d_ptr));
        if (!dst) return RBG_ENODEV;
        if (hipMemcpy(buf.data(), dst, buf.size(), hipMemcpyDeviceToHost) != hipSuccess) return RBG_ENODEV;
        for (size_t i = 0; i < t.count; ++i)
            for (size_t o : t.ptr_offsets) {
                const void *old;
                std::memcpy(&old, buf.data() + i * t.stride + o, sizeof(old));
                const void *nw = reloc(old);
                std::memcpy(buf.data() + i * t.stride + o, &nw, sizeof(nw));
            }
        if (hipMemcpy(dst, buf.data(), buf.size(), hipMemcpyHostToDevice) != hipSuccess) return RBG_ENODEV;
        r->ptr_tables.push_back({dst, t.count, t.stride, t.ptr_offsets});
    }
    DevIndex d = src->dev;
    reloc.fix(d.syms); reloc.fix(d.phi_ent); reloc.fix(d.phi_slots); reloc.fix(d.phi_ord);
    reloc.fix(d.mk_start); reloc.fix(d.mk_end); reloc.fix(d.mk_off); reloc.fix(d.mk_vals); reloc.fix(d.mk_bucket); reloc.fix(d.mk_rec);
    reloc.fix(d.counters); reloc.fix(d.lut); reloc.fix(d.pairs); reloc.fix(d.triples); reloc.fix(d.quads); reloc.fix(d.quints);
    reloc.fix(d.lut2); reloc.fix(d.ftab); reloc.fix(d.dense);
    reloc.fix(d.phi_dir); reloc.fix(d.order_docs);
    for (int t = 0; t < kMaxRunDepth; ++t) reloc.fix(d.run_samp[t]);
    reloc.fix(d.run_tabs2); reloc.fix(d.run_hot); reloc.fix(d.phi_super);
    for (int t = 0; t < kMaxRunDepth; ++t) { reloc.fix(d.run_ent2[t]); reloc.fix(d.run_dir2[t]); reloc.fix(d.run_rec2[t]); }
    r->dev = d;
    if (hipMemset(d.counters, 0, 4 * sizeof(uint64_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RBG_ENODEV;
    return RBG_OK;
}

}  // namespace

extern "C" {

// Replicas on `devices[0..G)` from ONE finished index: every target's copies are enqueued before any is waited for, so
// the G transfers overlap -- on MI355X's fully connected xGMI each target has its own link to the source, and the
// fan-out takes about the time of one copy instead of G (a chain 0 -> 1 -> ... would use each link once as well but
// serialise on the first hop; with a direct link per pair the star is the better shape).
int rbg_replicate_many(rbg_index *src, const int *devices, int G, rbg_index **out) {
    return guarded([&]() -> int {
    if (!src || !out || !devices || G <= 0) return RBG_EARG;
    for (int g = 0; g < G; ++g) out[g] = nullptr;
    if (!queryable(src)) return RBG_ENODEV;
    if (src->primary) return RBG_EARG;  // replicate the primary, not a replica
    {
        DeviceScope s0(src->device);
        if (s0.rc) return s0.rc;
        HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<ReplicaJob> jobs(G);
    int rc = RBG_OK;
    for (int g = 0; g < G && !rc; ++g) rc = replicate_begin(src, devices[g], jobs[g]);
    for (int g = 0; g < G; ++g) {
        if (!jobs[g].r) continue;
        if (!rc && jobs[g].st) rc = replicate_finish(src, jobs[g]);
        else if (jobs[g].st) (void)hipStreamSynchronize(jobs[g].st);   // never free memory a copy is still writing
        if (jobs[g].st) {
            DeviceScope scope(jobs[g].r->device);
            (void)hipStreamDestroy(jobs[g].st);
            if (jobs[g].ev0) (void)hipEventDestroy(jobs[g].ev0);
            if (jobs[g].ev1) (void)hipEventDestroy(jobs[g].ev1);
        }
    }
    if (rc) {
        for (int g = 0; g < G; ++g)
            if (jobs[g].r) rbg_free(jobs[g].r);
        return rc;
    }
    for (int g = 0; g < G; ++g) out[g] = jobs[g].r;
    return RBG_OK;
    });
}

// What the fan-out cost THIS replica: out = {milliseconds of its copies on its own stream, bytes copied, peer access (1 / 0; -1: same device)}.
int rbg_replicate_stats(const rbg_index *replica, double out[3]) {
    if (!replica || !out) return RBG_EARG;
    if (!replica->primary) return RBG_EARG;   // a primary was built, not copied
    out[0] = replica->replicate_ms;
    out[1] = static_cast<double>(replica->hbm_bytes);
    out[2] = replica->replicate_peer;
    return RBG_OK;
}

int rbg_replicate(rbg_index *src, int device, rbg_index **out) {
    if (!out) return RBG_EARG;
    return rbg_replicate_many(src, &device, 1, out);
}

int rbg_shard_bounds(uint64_t n_items, int rank, int world, uint64_t *begin, uint64_t *end) {
    if (world <= 0 || rank < 0 || rank >= world || !begin || !end) return RBG_EARG;
    *begin = static_cast<uint64_t>((static_cast<unsigned __int128>(n_items) * static_cast<unsigned>(rank)) / static_cast<unsigned>(world));
    *end = static_cast<uint64_t>((static_cast<unsigned __int128>(n_items) * (static_cast<unsigned>(rank) + 1u)) / static_cast<unsigned>(world));
    return RBG_OK;
}

int rbg_find_range_sharded(rbg_index *const *replicas, int G, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                           uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (!replicas || G <= 0) return RBG_EARG;
    for (int g = 0; g < G; ++g)
        if (!queryable(replicas[g])) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!off || !lo || !hi || (!seqs && off[N])) return RBG_EARG;
    int rc0 = check_offsets(off, N);
    if (rc0) return rc0;
    std::vector<int> rcs(G, RBG_OK);
    auto work = [&](int g) {
        uint64_t b, e;
        (void)rbg_shard_bounds(N, g, G, &b, &e);
        if (e == b) return;
        std::vector<uint64_t> o(e - b + 1);  // the shard's offsets, re-based
        for (uint64_t i = b; i <= e; ++i) o[i - b] = off[i] - off[b];
        rcs[g] = find_range_host(replicas[g], seqs + off[b], o.data(), e - b, lo + b, hi + b, ssamp ? ssamp + b : nullptr, nullptr);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
    for (int rc : rcs)
        if (rc) return rc;
    return RBG_OK;
    });
}

// ---- counters --------------------------------------------------------------------------------------

int rbg_counters(rbg_index *ix, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, ix->dev.counters, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RBG_OK;
    });
}

int rbg_combine_stats(rbg_index *ix, uint64_t out[2]) {
    if (!ix || !out) return RBG_EARG;
    rbg_index *root = ix;
    out[0] = root->comb_launches.load(std::memory_order_relaxed);
    out[1] = root->comb_requests.load(std::memory_order_relaxed);
    return RBG_OK;
}

int rbg_counters_reset(rbg_index *ix) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ix->dev.counters, 0, 4 * sizeof(uint64_t)));
    return RBG_OK;
    });
}

}  // extern "C"

namespace {

// RCCL is needed by two optional calls only (the counters' all-reduce), so the library does not link it: it is opened
// on first use, and a process that never reduces counters loads librbg.so on a machine without RCCL.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
    static Rccl &get() {
        static Rccl r = [] {
            Rccl x;
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.h) break;
            }
            if (!x.h) {
                std::fprintf(stderr, "rbg: RCCL not found (%s): the counters' all-reduce is unavailable\n", dlerror());
                return x;
            }
            x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.h, "ncclAllReduce"));
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.h, "ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.h, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.h, "ncclGroupEnd"));
            x.ok = x.AllReduce && x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd;
            return x;
        }();
        return r;
    }
};

// One communicator clique per set of devices, made on first use and kept for the life of the process: a reduce per
// batch must not pay ncclCommInitAll (hundreds of milliseconds on 8 GPUs) every time.  rbg_comm_cache_clear() drops them.
struct CliqueCache {
    std::mutex mu;
    std::map<std::vector<int>, std::vector<ncclComm_t>> cliques;
    static CliqueCache &get() {
        static CliqueCache *c = new CliqueCache();   // never destroyed: communicators must not be torn down at exit time
        return *c;
    }
};

}  // namespace

extern "C" {

int rbg_comm_cache_clear(void) {
    return guarded([&]() -> int {
    CliqueCache &cc = CliqueCache::get();
    std::lock_guard<std::mutex> lk(cc.mu);
    if (!cc.cliques.empty() && Rccl::get().ok)
        for (auto &kv : cc.cliques)
            for (ncclComm_t c : kv.second) (void)Rccl::get().CommDestroy(c);
    cc.cliques.clear();
    return RBG_OK;
    });
}

// One RCCL all-reduce (sum) of the four 64-bit counters over the communicator's ranks: the run's only collective
// (the reference has none).  `nccl_comm` is the caller's ncclComm_t for this replica's device.
int rbg_counters_allreduce(rbg_index *ix, void *nccl_comm, void *stream, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!nccl_comm || !out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DevBuf sum;
    int rc = sum.alloc(4 * sizeof(uint64_t));
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());  // every query launched so far has added its counts
    if (!Rccl::get().ok) return RBG_ENODEV;
    if (Rccl::get().AllReduce(ix->dev.counters, sum.p, 4, ncclUint64, ncclSum, static_cast<ncclComm_t>(nccl_comm), st) != ncclSuccess) return RBG_ENODEV;
    HIP_TRY(hipMemcpyAsync(out, sum.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RBG_OK;
    });
}

// The same for G replicas held by ONE process: a communicator clique over their devices (ncclCommInitAll once per
// device set, kept: CliqueCache), one grouped all-reduce, every replica ends with the same sums.  Needs G distinct devices.
int rbg_counters_allreduce_local(rbg_index *const *replicas, int G, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!replicas || G <= 0 || !out) return RBG_EARG;
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) {
        if (!queryable(replicas[g])) return RBG_ENODEV;
        devs[g] = replicas[g]->device;
        for (int h = 0; h < g; ++h)
            if (devs[h] == devs[g]) return RBG_EARG;
    }
    Rccl &nc = Rccl::get();
    if (!nc.ok) return RBG_ENODEV;
    CliqueCache &cc = CliqueCache::get();
    std::lock_guard<std::mutex> lk(cc.mu);   // one reduce at a time per process: the clique is shared
    auto it = cc.cliques.find(devs);
    if (it == cc.cliques.end()) {
        std::vector<ncclComm_t> fresh(G);
        if (ncclResult_t e = nc.CommInitAll(fresh.data(), G, devs.data()); e != ncclSuccess) {
            std::fprintf(stderr, "rbg: ncclCommInitAll over %d devices failed (ncclResult %d): no counters clique\n", G, static_cast<int>(e));
            return RBG_ENODEV;
        }
        it = cc.cliques.emplace(devs, std::move(fresh)).first;
    }
    const std::vector<ncclComm_t> &comms = it->second;
    std::vector<void *> sums(G, nullptr);
    int rc = RBG_OK;
    for (int g = 0; g < G && !rc; ++g) {
        DeviceScope scope(devs[g]);
        if (hipDeviceSynchronize() != hipSuccess || hipMalloc(&sums[g], 32) != hipSuccess) rc = RBG_ENODEV;
    }
    if (!rc) {
        (void)nc.GroupStart();
        for (int g = 0; g < G; ++g) {
            DeviceScope scope(devs[g]);
            if (ncclResult_t e = nc.AllReduce(replicas[g]->dev.counters, sums[g], 4, ncclUint64, ncclSum, comms[g], nullptr); e != ncclSuccess) {
                std::fprintf(stderr, "rbg: ncclAllReduce of the counters on device %d failed (ncclResult %d)\n", devs[g], static_cast<int>(e));
                rc = RBG_ENODEV;
            }
        }
        if (ncclResult_t e = nc.GroupEnd(); e != ncclSuccess) {
            std::fprintf(stderr, "rbg: ncclGroupEnd of the counters' all-reduce over %d devices failed (ncclResult %d)\n", G, static_cast<int>(e));
            rc = RBG_ENODEV;
        }
    }
    for (int g = 0; g < G; ++g) {
        DeviceScope scope(devs[g]);
        if (!rc && hipDeviceSynchronize() != hipSuccess) rc = RBG_ENODEV;
        if (!rc && g == 0 && hipMemcpy(out, sums[0], 32, hipMemcpyDeviceToHost) != hipSuccess) rc = RBG_ENODEV;
        if (sums[g]) (void)hipFree(sums[g]);
    }
    return rc;
    });
}

}  // extern "C"

