// capi/hostpath.ipp -- the host-pointer entry points: the chunked pipeline over pinned staging (rbg_hostpath.hpp), micro-batching of one-read calls,
// rbg_lf / rbg_find_range* / rbg_locs_at.  Part of rbg_capi.hip.
extern "C" {
// ---- host-buffer entry points ----------------------------------------------------------------------

// Reads of a host batch: either the C-ABI's packed layout (read i = seqs[off[i], off[i+1])) or spans of a larger
// buffer (read i = base[begin[i], begin[i] + len[i]): what a parser that leaves the bytes in its input buffer has).
struct HostReads {
    const uint8_t *base = nullptr;
    const uint64_t *off = nullptr;     // packed layout (N + 1), or
    const uint64_t *begin = nullptr;   // spans
    const uint32_t *len = nullptr;
    const uint8_t *ptr(uint64_t i) const { return base + (off ? off[i] : begin[i]); }
    uint64_t length(uint64_t i) const { return off ? off[i + 1] - off[i] : len[i]; }
};

// a workspace of the index for the duration of one call
struct WsLease {
    rbg_index *ix;
    std::unique_ptr<rbg_hostpath::Workspace> ws;
    explicit WsLease(rbg_index *ix_) : ix(ix_) {
        std::lock_guard<std::mutex> g(ix->ws_mu);
        if (!ix->ws_free.empty()) { ws = std::move(ix->ws_free.back()); ix->ws_free.pop_back(); }
        if (!ws) { ws.reset(new rbg_hostpath::Workspace()); ws->device = ix->device; }
    }
    ~WsLease() {
        std::lock_guard<std::mutex> g(ix->ws_mu);
        ix->ws_free.push_back(std::move(ws));
    }
};

constexpr uint64_t kHostChunkReads = uint64_t(1) << 20;    // reads per in-flight chunk ...
constexpr uint64_t kHostChunkBytes = uint64_t(384) << 20;  // ... and symbols per chunk (long reads)

static int find_range_host_core(rbg_index *ix, const HostReads &R, uint64_t N, uint64_t *lo, uint64_t *hi, uint64_t *ssamp,
                                uint64_t *count, bool allow_pack);

// ---- micro-batching of one-read calls ---------------------------------------------------------------------------------
// A caller written against the reference asks one read at a time (RowBowt::find_range(query), rowbowt.hpp:121-131);
// when several of its threads do so concurrently (the reference's only parallel dispatcher is rb_markers' thread pool,
// rb_markers.cpp:318-535) their calls are combined: whoever arrives while no launch is being prepared becomes the
// leader, takes everything that has queued up, runs ONE batched call for it and hands the answers back; whoever arrives
// in the meantime queues for the next round.  No timer and no added latency: a lone caller's request is a batch of
// one, and the batch size follows the concurrency by itself.  RBG_HOST_COMBINE=0 switches it off (A/B).
struct CombineReq {
    bool done = false;
    int rc = RBG_OK;
};
// exec(batch) answers every request of the batch (sets rc); match(a, b): may b ride in a's batch?
extern "C++" {
template <typename Req, typename Match, typename Exec>
int combine_submit(rbg_index *ix, rbg_index::Combiner &C, Req &mine, Match match, Exec exec) {
    std::unique_lock<std::mutex> lk(C.mu);
    C.pending.push_back(&mine);
    while (!mine.done) {
        if (C.leader) { C.cv.wait(lk); continue; }
        C.leader = true;
        std::vector<Req *> batch;
        std::vector<void *> rest;
        for (void *p : C.pending) {
            Req *r = static_cast<Req *>(p);
            if (r == &mine || match(mine, *r)) batch.push_back(r); else rest.push_back(p);
        }
        C.pending.swap(rest);
        lk.unlock();
        int rc_all = RBG_OK;
        try {
            exec(batch);
        } catch (const std::bad_alloc &) {
            rc_all = RBG_ENOMEM;
        } catch (...) {
            rc_all = RBG_EFORMAT;
        }
        ix->comb_launches.fetch_add(1, std::memory_order_relaxed);
        ix->comb_requests.fetch_add(batch.size(), std::memory_order_relaxed);
        lk.lock();
        for (Req *r : batch) { if (rc_all) r->rc = rc_all; r->done = true; }
        C.leader = false;
        C.cv.notify_all();
    }
    return mine.rc;
}
}  // extern "C++"
inline bool combine_enabled() {
    static const bool on = [] { const char *e = std::getenv("RBG_HOST_COMBINE"); return !(e && e[0] == '0'); }();
    return on;
}

struct RangeReq : CombineReq {
    const uint8_t *seq = nullptr;
    uint64_t len = 0;
    bool want_ss = false;
    uint64_t lo = 1, hi = 0, ss = 0;
};

// one read through the combiner: find_range / count / find_range_w_toehold with N = 1
static int find_range_one(rbg_index *ix, const uint8_t *seq, uint64_t len, uint64_t *lo, uint64_t *hi, uint64_t *ssamp, uint64_t *count) {
    RangeReq mine;
    mine.seq = seq;
    mine.len = len;
    mine.want_ss = ssamp != nullptr;
    const int rc = combine_submit(ix, ix->comb_range, mine, [](const RangeReq &a, const RangeReq &b) { return a.len <= 0xFFFFFFFFull && b.len <= 0xFFFFFFFFull; },
        [&](std::vector<RangeReq *> &batch) {
            const uint64_t K = batch.size();
            bool any_ss = false;
            const uint8_t *base = nullptr;
            for (RangeReq *r : batch) {
                any_ss = any_ss || r->want_ss;
                if (r->len && (!base || r->seq < base)) base = r->seq;
            }
            std::vector<uint64_t> begin(K), blo(K), bhi(K), bss(any_ss ? K : 0);
            std::vector<uint32_t> blen(K);
            HostReads R;
            int rc2;
            if (K == 1 && batch[0]->len > 0xFFFFFFFFull) {   // (a read beyond 4 GB: the packed layout takes any length)
                const uint64_t off[2] = {0, batch[0]->len};
                R.base = batch[0]->seq;
                R.off = off;
                rc2 = find_range_host_core(ix, R, 1, blo.data(), bhi.data(), any_ss ? bss.data() : nullptr, nullptr, true);
            } else {
                static const uint8_t kNone = 0;
                if (!base) base = &kNone;
                for (uint64_t i = 0; i < K; ++i) {
                    begin[i] = batch[i]->len ? static_cast<uint64_t>(batch[i]->seq - base) : 0;
                    blen[i] = static_cast<uint32_t>(batch[i]->len);
                }
                R.base = base;
                R.begin = begin.data();
                R.len = blen.data();
                rc2 = find_range_host_core(ix, R, K, blo.data(), bhi.data(), any_ss ? bss.data() : nullptr, nullptr, true);
            }
            for (uint64_t i = 0; i < K; ++i) {
                batch[i]->rc = rc2;
                batch[i]->lo = blo[i];
                batch[i]->hi = bhi[i];
                if (batch[i]->want_ss) batch[i]->ss = bss[i];
            }
        });
    if (rc) return rc;
    if (lo) { *lo = mine.lo; *hi = mine.hi; }
    if (ssamp) *ssamp = mine.ss;
    if (count) *count = mine.hi >= mine.lo ? mine.hi - mine.lo + 1 : 0;
    return RBG_OK;
}

static int find_range_host(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo, uint64_t *hi,
                           uint64_t *ssamp, uint64_t *count) {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!off || (!seqs && off[N])) return RBG_EARG;
    if (N == 1 && off[0] == 0 && combine_enabled()) return find_range_one(ix, seqs, off[1], lo, hi, ssamp, count);
    if (off[0] != 0) return RBG_EARG;   // (the rest of check_offsets() is done by the staging passes, chunk by chunk, before any byte is read)
    HostReads R;
    R.base = seqs;
    R.off = off;
    return find_range_host_core(ix, R, N, lo, hi, ssamp, count, true);
}

// The pipeline of rbg_hostpath.hpp.  Outputs: lo/hi (both or neither), ssamp (toehold search), count.
static int find_range_host_core(rbg_index *ix, const HostReads &R, uint64_t N, uint64_t *lo, uint64_t *hi, uint64_t *ssamp,
                                uint64_t *count, bool allow_pack) {
    using rbg_hostpath::Slot;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    WsLease lease(ix);
    rbg_hostpath::Workspace &W = *lease.ws;
    const int64_t pk = g_opt_packed_reads.load();
    const HostIndex &h = ix->H();
    // 2-bit transfer: needs the packed search kernel's alphabet (four k-mer symbols) and the slot-table layout
    const bool pack = allow_pack && ix->dev.nmajor == 4 && (pk == 2 || (pk == 1 && N >= 4096));
    const bool acgt = h.major_byte[0] == 'A' && h.major_byte[1] == 'C' && h.major_byte[2] == 'G' && h.major_byte[3] == 'T';
    if (!W.team) {
        // a quarter of the hardware's CPUs, at most 64 and at most what the container's CPU quota lets run at once
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned nt = std::min({64u, std::max(1u, hw / 4), rbg_hostpath::cpu_budget()});
        if (const char *e = std::getenv("RBG_HOST_THREADS")) nt = static_cast<unsigned>(std::max(1, std::min(256, std::atoi(e))));
        W.team.reset(new rbg_hostpath::ThreadTeam(nt));
        W.bad.resize(W.team->size());
    }
    rbg_hostpath::ThreadTeam &team = *W.team;
    const unsigned T = team.size();
    // small batches (the shim's one-read calls among them) stay on the calling thread: waking the team costs more
    auto par = [&](uint64_t work_items, const std::function<void(unsigned)> &fn) {
        if (work_items < 16384) { for (unsigned t = 0; t < T; ++t) fn(t); }
        else team.run(fn);
    };
    for (auto &v : W.bad) v.clear();
    const int nout = (lo ? 2 : 0) + (ssamp ? 1 : 0) + (count ? 1 : 0);
    const bool need_lohi_dev = true;  // the kernels always write lo/hi
    (void)need_lohi_dev;

    int rc = RBG_OK;
    const bool trace = std::getenv("RBG_HOST_TRACE") != nullptr;   // per-call breakdown on stderr
    double t_pack = 0, t_wait = 0, t_out = 0, t_enq = 0;
    // RBG_HOST_TRACE=2: also the device-side timeline of every chunk (timing events around copy in / search / copy out)
    const bool timeline = trace && std::atoi(std::getenv("RBG_HOST_TRACE")) >= 2;
    const char *e_direct = std::getenv("RBG_HOST_DIRECT_OUT");
    const bool direct_out = !(e_direct && e_direct[0] == '0');
    struct ChunkEvents { hipEvent_t e[4]; double host_ms; };
    std::vector<ChunkEvents> tl;
    auto mark = [&](int which, hipStream_t st) {
        if (!timeline) return;
        if (which == 0) { tl.emplace_back(); for (hipEvent_t &e : tl.back().e) (void)hipEventCreate(&e); }
        (void)hipEventRecord(tl.back().e[which], st);
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_call = now();
    auto drain = [&](Slot &s) -> int {  // wait for the chunk in flight in `s`, hand its results to the caller
        if (!s.busy) return RBG_OK;
        const auto tw0 = now();
        if (hipEventSynchronize(s.done) != hipSuccess) return RBG_ENODEV;
        const auto tw1 = now();
        t_wait += secs(tw0, tw1);
        const uint64_t *src = static_cast<const uint64_t *>(s.h_out);
        uint64_t *dsts[4];
        int ncol = 0;
        if (lo) { dsts[ncol++] = lo + s.begin; dsts[ncol++] = hi + s.begin; }
        if (ssamp) dsts[ncol++] = ssamp + s.begin;
        if (count) dsts[ncol++] = count + s.begin;
        const uint64_t cnt = s.cnt;
        par(cnt, [&](unsigned t) {  // pinned -> the caller's (pageable) arrays, every member a slice of every column
            const uint64_t i0 = cnt * t / T, i1 = cnt * (t + 1) / T;
            for (int c = 0; c < ncol; ++c) std::memcpy(dsts[c] + i0, src + static_cast<uint64_t>(c) * cnt + i0, (i1 - i0) * 8);
        });
        t_out += secs(tw1, now());
        s.busy = false;
        return RBG_OK;
    };

    uint64_t chunk_reads = kHostChunkReads;
    if (const char *e = std::getenv("RBG_HOST_CHUNK_READS")) chunk_reads = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));   // (tests: many chunks from a small batch)
    std::vector<uint64_t> part(T + 1), psym(T);
    std::atomic<bool> bad_offsets{false};   // an offset smaller than its predecessor (check_offsets), found by a sizing pass
    uint64_t b = 0;
    unsigned seq = 0;   // chunks enqueued so far
    while (b < N && !rc) {
        // the chunk [b, e): bounded in reads and in symbols (no pass over the reads just to find the bound: the packed
        // layout has the symbol count in its offsets, spans are sampled and measured by the staging pass itself)
        uint64_t e = std::min<uint64_t>(N, b + chunk_reads), sym = 0;
        if (R.off) {
            while (e > b + 1 && R.off[e] - R.off[b] > kHostChunkBytes) e = b + (e - b) / 2;
        } else {
            uint64_t mx = 0;
            for (uint64_t i = b; i < e; i += 1 + (e - b) / 64) mx = std::max<uint64_t>(mx, R.len[i]);
            while (e > b + 1 && (e - b) * std::max<uint64_t>(mx, 64) > 4 * kHostChunkBytes) e = b + (e - b) / 2;   // long reads: fewer per chunk
        }
        // sizing pass: every member measures its slice (symbols, or 16-byte chunks of the 2-bit form) and checks that the
        // offsets ascend (check_offsets): nothing of the caller's read bytes is touched, and nothing sized, before that
        const auto ts0 = now();
        par(e - b, [&](unsigned t) {
            const uint64_t i0 = b + (e - b) * t / T, i1 = b + (e - b) * (t + 1) / T;
            uint64_t c = 0, sy = 0;
            bool bad = false;
            if (R.off) {
                for (uint64_t i = i0; i < i1; ++i) { const uint64_t m = R.off[i + 1] - R.off[i]; bad |= R.off[i + 1] < R.off[i]; sy += m; c += (m + 63) >> 6; }
            } else {
                for (uint64_t i = i0; i < i1; ++i) { const uint64_t m = R.len[i]; sy += m; c += (m + 63) >> 6; }
            }
            if (bad) bad_offsets = true;
            part[t + 1] = pack ? c : sy;
            psym[t] = sy;
        });
        t_pack += secs(ts0, now());
        if (bad_offsets) { rc = RBG_EARG; break; }
        for (uint64_t v : psym) sym += v;
        part[0] = 0;
        for (unsigned t = 0; t < T; ++t) part[t + 1] += part[t];
        if (pack && part[T] >= (uint64_t(1) << 32)) { rc = RBG_EARG; break; }   // chunk indices are 32-bit
        const uint64_t cnt = e - b;
        Slot &s = W.slot[seq % rbg_hostpath::kSlots];
        ++seq;
        if ((rc = drain(s))) break;   // (waits only when every buffer is in flight: normally drained below)
        // device columns: lo, hi, [ssamp], [count]
        const uint64_t dev_cols = 2 + (ssamp ? 1 : 0) + (count ? 1 : 0);
        size_t in_bytes;
        if (pack) in_bytes = cnt * 8 + 16 + (sym / 64 + cnt + 1) * 16;
        else in_bytes = (cnt + 1) * 8 + 16 + sym + 32;
        const int er = W.ensure(s, in_bytes, dev_cols * cnt * 8);
        if (er) { rc = er == 2 ? RBG_ENOMEM : RBG_ENODEV; break; }
        const auto tp0 = now();
        char *hin = static_cast<char *>(s.h_in);
        char *din = static_cast<char *>(s.d_in);
        uint64_t *dout = static_cast<uint64_t *>(s.d_out);
        uint64_t *d_lo = dout, *d_hi = dout + cnt, *d_ss = ssamp ? dout + 2 * cnt : nullptr;
        uint64_t *d_cnt = count ? dout + (ssamp ? 3 : 2) * cnt : nullptr;
        // Results leave without a copy engine: the kernels store the columns the caller wants straight into the pinned
        // buffer (device-visible host memory; 8 bytes per lane, whole lines per wave, posted writes over PCIe).  A
        // device-to-host copy enqueued behind the search of chunk c holds up the copy IN of chunk c + 1 on this
        // platform until that search has finished (one engine serves both directions, in order: measured with
        // RBG_HOST_TRACE=2), which serialised copy in / search / copy out of successive chunks.  RBG_HOST_DIRECT_OUT=0
        // keeps the copies (A/B measurements).
        if (direct_out) {
            uint64_t *hcol = static_cast<uint64_t *>(s.h_out);
            if (lo) { d_lo = hcol; d_hi = hcol + cnt; hcol += 2 * cnt; }
            if (ssamp) { d_ss = hcol; hcol += cnt; }
            if (count) { d_cnt = hcol; hcol += cnt; }
        }
        size_t used = 0;
        if (pack) {
            uint2 *meta = reinterpret_cast<uint2 *>(hin);
            const size_t chunks_at = (cnt * 8 + 15) & ~size_t(15);
            uint32_t *chunks = reinterpret_cast<uint32_t *>(hin + chunks_at);
            // every member packs from its own prefix of 16-byte chunks (sizing pass above)
            par(cnt, [&](unsigned t) {
                const uint64_t i0 = b + cnt * t / T, i1 = b + cnt * (t + 1) / T;
                uint64_t c = part[t];
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint64_t m = R.length(i);
                    uint32_t *dst = chunks + c * 4;
                    const bool ok = m < 0x80000000ull &&
                                    (acgt ? rbg_hostpath::pack_read_acgt(R.ptr(i), m, dst) : rbg_hostpath::pack_read_lut(R.ptr(i), m, h.major_of, dst));
                    meta[i - b] = make_uint2(static_cast<uint32_t>(c), ok ? static_cast<uint32_t>(m) : 0x80000000u);
                    if (!ok) W.bad[t].push_back(i);
                    c += (m + 63) >> 6;
                }
            });
            used = chunks_at + part[T] * 16;
            t_pack += secs(tp0, now());
            mark(0, s.st);
            if (timeline) tl.back().host_ms = secs(t_call, now()) * 1e3;
            if (hipMemcpyAsync(din, hin, used, hipMemcpyHostToDevice, s.st) != hipSuccess) rc = RBG_ENODEV;
            mark(1, s.st);
            if (!rc && launch_find_range_packed_only(ix->dev, ix->cfg, reinterpret_cast<const uint2 *>(din), reinterpret_cast<const uint4 *>(din + chunks_at), cnt,
                                                     d_lo, d_hi, d_ss, s.st))
                rc = RBG_ENODEV;
            mark(2, s.st);
        } else {
            uint64_t *off2 = reinterpret_cast<uint64_t *>(hin);
            const size_t bytes_at = ((cnt + 1) * 8 + 15) & ~size_t(15);
            char *bytes = hin + bytes_at;
            par(cnt, [&](unsigned t) {
                const uint64_t i0 = b + cnt * t / T, i1 = b + cnt * (t + 1) / T;
                uint64_t c = part[t];
                if (R.off && i1 > i0) {  // contiguous in the source: one copy per slice
                    std::memcpy(bytes + c, R.ptr(i0), R.off[i1] - R.off[i0]);
                    for (uint64_t i = i0; i < i1; ++i) off2[i - b] = c + (R.off[i] - R.off[i0]);
                } else {
                    for (uint64_t i = i0; i < i1; ++i) {
                        const uint64_t m = R.length(i);
                        std::memcpy(bytes + c, R.ptr(i), m);
                        off2[i - b] = c;
                        c += m;
                    }
                }
            });
            off2[cnt] = part[T];
            used = bytes_at + part[T];
            t_pack += secs(tp0, now());
            mark(0, s.st);
            if (timeline) tl.back().host_ms = secs(t_call, now()) * 1e3;
            if (hipMemcpyAsync(din, hin, used, hipMemcpyHostToDevice, s.st) != hipSuccess) rc = RBG_ENODEV;
            mark(1, s.st);
            if (!rc && launch_find_range(ix->dev, ix->cfg, reinterpret_cast<const uint8_t *>(din + bytes_at), reinterpret_cast<const uint64_t *>(din), cnt, d_lo,
                                         d_hi, d_ss, s.st))
                rc = RBG_ENODEV;
            mark(2, s.st);
        }
        if (rc) break;
        if (count && launch_count_from_ranges(d_lo, d_hi, cnt, d_cnt, s.st)) { rc = RBG_ENODEV; break; }
        // results through pinned memory: [lo | hi] (when asked for) | ssamp | count, the columns the caller wants
        {
            char *hout = static_cast<char *>(s.h_out);
            size_t at = 0;
            hipError_t e2 = hipSuccess;
            if (!direct_out) {
                if (lo) { e2 = hipMemcpyAsync(hout, d_lo, 2 * cnt * 8, hipMemcpyDeviceToHost, s.st); at += 2 * cnt * 8; }
                if (e2 == hipSuccess && ssamp) { e2 = hipMemcpyAsync(hout + at, d_ss, cnt * 8, hipMemcpyDeviceToHost, s.st); at += cnt * 8; }
                if (e2 == hipSuccess && count) { e2 = hipMemcpyAsync(hout + at, d_cnt, cnt * 8, hipMemcpyDeviceToHost, s.st); at += cnt * 8; }
            }
            mark(3, s.st);
            if (e2 == hipSuccess) e2 = hipEventRecord(s.done, s.st);
            if (e2 != hipSuccess) { rc = RBG_ENODEV; break; }
        }
        s.begin = b;
        s.cnt = cnt;
        s.busy = true;
        b = e;
        t_enq = secs(t_call, now()) - t_pack - t_wait - t_out;
        // while the GPU works: hand the chunks that have finished to the caller, oldest first, without waiting
        for (unsigned j = 1; j < rbg_hostpath::kSlots && !rc; ++j) {
            Slot &o = W.slot[(seq - 1 + j) % rbg_hostpath::kSlots];
            if (!o.busy) continue;
            if (hipEventQuery(o.done) != hipSuccess) break;
            rc = drain(o);
        }
    }
    (void)nout;
    for (unsigned j = 0; j < rbg_hostpath::kSlots; ++j) {   // what is still in flight, oldest first
        Slot &s = W.slot[(seq + j) % rbg_hostpath::kSlots];
        const int r2 = rc ? RBG_OK : drain(s);
        if (!rc) rc = r2;
        // on an error EVERY stream that exists is drained, marked busy or not: a chunk whose copy or search was enqueued
        // before a later step of the same chunk failed is in flight without the mark, and the workspace (its pinned
        // buffers, which a direct-out kernel writes) goes back to the pool when this call returns
        if (rc && s.st) { (void)hipStreamSynchronize(s.st); s.busy = false; }
    }
    if (rc) {
        for (ChunkEvents &c : tl) for (hipEvent_t &e : c.e) (void)hipEventDestroy(e);
        return rc;
    }
    if (trace)
        std::fprintf(stderr, "rbg host call: %llu reads, %s, %u threads: %.2f ms = stage %.2f + enqueue/other %.2f + wait for the GPU %.2f + copy out %.2f\n",
                     static_cast<unsigned long long>(N), pack ? "2-bit" : "bytes", T, secs(t_call, now()) * 1e3, t_pack * 1e3, t_enq * 1e3, t_wait * 1e3,
                     t_out * 1e3);
    if (timeline && !tl.empty()) {
        for (size_t c = 0; c < tl.size(); ++c) {
            float t[4] = {0, 0, 0, 0};
            for (int j = 0; j < 4; ++j) (void)hipEventElapsedTime(&t[j], tl[0].e[0], tl[c].e[j]);
            std::fprintf(stderr, "  chunk %2zu: enqueued at %7.2f ms (host clock); device clock from the first copy: copy in %7.2f..%7.2f, search ..%7.2f, copy out ..%7.2f\n",
                         c, tl[c].host_ms, t[0], t[1], t[2], t[3]);
        }
        for (ChunkEvents &c : tl) for (hipEvent_t &e : c.e) (void)hipEventDestroy(e);
    }
    // reads the 2-bit form cannot express (any symbol outside the k-mer alphabet): searched from their bytes
    std::vector<uint64_t> bad;
    for (auto &v : W.bad) bad.insert(bad.end(), v.begin(), v.end());
    if (!bad.empty()) {
        std::sort(bad.begin(), bad.end());
        std::vector<uint64_t> bb(bad.size());
        std::vector<uint32_t> bl(bad.size());
        bool fits = true;
        for (size_t j = 0; j < bad.size(); ++j) {
            bb[j] = static_cast<uint64_t>(R.ptr(bad[j]) - R.base);
            const uint64_t m = R.length(bad[j]);
            if (m > 0xFFFFFFFFull) fits = false;
            bl[j] = static_cast<uint32_t>(m);
        }
        std::vector<uint64_t> t_lo(bad.size()), t_hi(bad.size()), t_ss(ssamp ? bad.size() : 0), t_cnt(count ? bad.size() : 0);
        HostReads Rb;
        Rb.base = R.base;
        std::vector<uint64_t> off3;
        std::string flat;
        if (fits) {
            Rb.begin = bb.data();
            Rb.len = bl.data();
        } else {  // a read beyond 4 GB: gather into the packed layout
            off3.assign(1, 0);
            for (uint64_t i : bad) { flat.append(reinterpret_cast<const char *>(R.ptr(i)), R.length(i)); off3.push_back(flat.size()); }
            Rb.base = reinterpret_cast<const uint8_t *>(flat.data());
            Rb.off = off3.data();
        }
        // (the lease is still held: the recursive call takes another workspace)
        rc = find_range_host_core(ix, Rb, bad.size(), t_lo.data(), t_hi.data(), ssamp ? t_ss.data() : nullptr, count ? t_cnt.data() : nullptr, false);
        if (rc) return rc;
        for (size_t j = 0; j < bad.size(); ++j) {
            if (lo) { lo[bad[j]] = t_lo[j]; hi[bad[j]] = t_hi[j]; }
            if (ssamp) ssamp[bad[j]] = t_ss[j];
            if (count) count[bad[j]] = t_cnt[j];
        }
    }
    return RBG_OK;
}

int rbg_lf(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym, uint64_t N, uint64_t *lo_out,
           uint64_t *hi_out) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!lo || !hi || !sym || !lo_out || !hi_out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    DevBuf dlo, dhi, dsym, dlo2, dhi2;
    int rc;
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = dsym.alloc(N)) || (rc = dlo2.alloc(N * 8)) || (rc = dhi2.alloc(N * 8)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dlo.p, lo, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dhi.p, hi, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dsym.p, sym, N, hipMemcpyHostToDevice, st));
    if (launch_lf(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), dsym.as<uint8_t>(), N, dlo2.as<uint64_t>(), dhi2.as<uint64_t>(), st))
        return RBG_ENODEV;
    HIP_TRY(hipMemcpyAsync(lo_out, dlo2.p, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hi_out, dhi2.p, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RBG_OK;
    });
}

int rbg_find_range(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo, uint64_t *hi) {
    return guarded([&]() -> int {
    if (N && (!lo || !hi)) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, lo, hi, nullptr, nullptr);
    });
}

int rbg_count(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *count) {
    return guarded([&]() -> int {
    if (N && !count) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, nullptr, nullptr, nullptr, count);
    });
}

int rbg_find_range_spans(rbg_index *ix, const uint8_t *base, const uint64_t *begin, const uint32_t *len, uint64_t N, uint64_t *lo,
                         uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (ssamp && !ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N == 0) return RBG_OK;
    if (!base || !begin || !len || !lo || !hi) return RBG_EARG;
    HostReads R;
    R.base = base;
    R.begin = begin;
    R.len = len;
    return find_range_host_core(ix, R, N, lo, hi, ssamp, nullptr, true);
    });
}

int rbg_find_range_w_toehold(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                             uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (ix && !ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!lo || !hi || !ssamp)) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, lo, hi, ssamp, nullptr);
    });
}

int rbg_locs_at(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N,
                uint64_t max_hits, uint64_t *loc_off, uint64_t **locs) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!loc_off || !locs || (N && (!lo || !hi || !k))) return RBG_EARG;
    *locs = nullptr;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    DevBuf dlo, dhi, dk, doff, dtmp;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    int rc;
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = dk.alloc(N * 8)) || (rc = doff.alloc((N + 1) * 8)) ||
        (rc = dtmp.alloc(tmp_bytes)))
        return rc;
    if (N) {
        HIP_TRY(hipMemcpyAsync(dlo.p, lo, N * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dhi.p, hi, N * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dk.p, k, N * 8, hipMemcpyHostToDevice, st));
    }
    if (launch_locate_plan(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, max_hits, doff.as<uint64_t>(), dtmp.p, tmp_bytes, st))
        return RBG_ENODEV;
    DevBuf dord;
    const void *order = nullptr;
    if ((rc = make_order(ix, dk.as<uint64_t>(), N, dord, st, &order))) return rc;
    return ragged_finish(N, doff, loc_off, locs, st, [&](uint64_t *d_vals) {
        return launch_locate_fill(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), dk.as<uint64_t>(), N, max_hits,
                                  doff.as<uint64_t>(), d_vals, nullptr, order, st) ? RBG_ENODEV : RBG_OK;
    });
    });
}

}  // extern "C"
