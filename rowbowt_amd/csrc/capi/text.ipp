// capi/text.ipp -- rbg_align_text and its pinned text buffers (rb_align's output lines, written on the device).  Part of rbg_capi.hip.
namespace {
// the document table on the handle's device (rbg_align_text): sorted starts, names back to back
int ensure_text_docs(rbg_index *ix) {
    std::lock_guard<std::mutex> g(ix->text_mu);
    if (ix->text_docs.start) return RBG_OK;
    const RawDocs &d = ix->H().dl;
    const uint64_t n = d.names.size();
    if (n == 0 || d.sorted.size() != n) return RBG_ENOTLOADED;
    std::vector<uint32_t> off(n + 1, 0);
    std::string blob;
    for (uint64_t j = 0; j < n; ++j) { blob += d.names[j]; off[j + 1] = static_cast<uint32_t>(blob.size()); }
    const void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    int rc;
    if ((rc = dev_upload(ix, d.sorted.data(), n * 8, &p0)) || (rc = dev_upload(ix, blob.data(), blob.size() ? blob.size() : 1, &p1)) ||
        (rc = dev_upload(ix, off.data(), (n + 1) * 4, &p2)))
        return rc;
    ix->text_docs.names = static_cast<const char *>(p1);
    ix->text_docs.name_off = static_cast<const uint32_t *>(p2);
    ix->text_docs.n = n;
    ix->text_docs.size = d.starts.back() + 1;   // (what rbg_doc_table reports as the collection's size)
    ix->text_docs.start = static_cast<const uint64_t *>(p0);
    return RBG_OK;
}
// RBG_TEXT_TRACE=1: where rbg_align_text spends its time, summed over the process's calls and printed at exit
struct TextTrace {
    bool on = std::getenv("RBG_TEXT_TRACE") != nullptr;
    std::mutex mu;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t calls = 0, bytes = 0;
    ~TextTrace() {
        if (on && calls)
            std::fprintf(stderr, "rbg_align_text: %llu calls, %.1f MB of text: names %.3f s, buffers + copy in %.3f, locate plan %.3f, locate fill %.3f, "
                                 "text plan %.3f, text fill %.3f, copy out %.3f\n", static_cast<unsigned long long>(calls), static_cast<double>(bytes) / 1e6,
                         t[0], t[1], t[2], t[3], t[4], t[5], t[6]);
    }
};
TextTrace g_text_trace;
// a pinned buffer of at least `bytes` from the handle's pool
int take_text_out(rbg_index *ix, size_t bytes, char **out) {
    std::lock_guard<std::mutex> g(ix->text_mu);
    if (!ix->text_copy_stream && hipStreamCreateWithFlags(&ix->text_copy_stream, hipStreamNonBlocking) != hipSuccess) return RBG_ENODEV;
    for (auto &t : ix->text_out)
        if (!t.busy && t.cap >= bytes) { t.busy = true; *out = t.p; return RBG_OK; }
    {   // every idle buffer is too small: ONE of them, the smallest, is replaced -- the others stay for the usual batches (one
        // oversized batch used to discard all the buffers rbg_reserve_text had made before the clock started)
        rbg_index::TextOut *smallest = nullptr;
        for (auto &t : ix->text_out)
            if (!t.busy && t.p && (!smallest || t.cap < smallest->cap)) smallest = &t;
        if (smallest) { (void)hipHostFree(smallest->p); smallest->p = nullptr; smallest->cap = 0; }
    }
    const size_t cap = std::max<size_t>(size_t(1) << 20, bytes + bytes / 4);
    void *p = nullptr;
    if (rbg_numa::host_malloc_near(&p, cap, hipHostMallocDefault, ix->device) != hipSuccess) { (void)hipGetLastError(); return RBG_ENOMEM; }
    for (auto &t : ix->text_out)
        if (!t.p) { t.p = static_cast<char *>(p); t.cap = cap; t.busy = true; *out = t.p; return RBG_OK; }
    rbg_index::TextOut t;
    t.p = static_cast<char *>(p);
    t.cap = cap;
    t.busy = true;
    if (hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p); return RBG_ENODEV; }
    ix->text_out.push_back(t);
    *out = static_cast<char *>(p);
    return RBG_OK;
}
// a typed view of a piece of a device block (like DevBuf::as)
struct DevView { char *b; template <typename T> T *as() { return reinterpret_cast<T *>(b); } };
// pinned staging for one call's inputs; given back by the guard
struct TextInHold {
    rbg_index *ix;
    size_t slot = ~size_t(0);
    char *p = nullptr;
    explicit TextInHold(rbg_index *i) : ix(i) {}
    int take(size_t bytes) {
        std::lock_guard<std::mutex> g(ix->text_mu);
        for (size_t j = 0; j < ix->text_in.size(); ++j)
            if (!ix->text_in[j].busy && ix->text_in[j].cap >= bytes) { slot = j; break; }
        if (slot == ~size_t(0)) {
            for (size_t j = 0; j < ix->text_in.size(); ++j)
                if (!ix->text_in[j].busy) { (void)hipHostFree(ix->text_in[j].p); ix->text_in[j].p = nullptr; ix->text_in[j].cap = 0; slot = j; break; }
            if (slot == ~size_t(0)) { ix->text_in.emplace_back(); slot = ix->text_in.size() - 1; }
            const size_t cap = bytes + bytes / 4 + 4096;
            void *q = nullptr;
            if (rbg_numa::host_malloc_near(&q, cap, hipHostMallocDefault, ix->device) != hipSuccess) { (void)hipGetLastError(); slot = ~size_t(0); return RBG_ENOMEM; }
            ix->text_in[slot].p = static_cast<char *>(q);
            ix->text_in[slot].cap = cap;
        }
        ix->text_in[slot].busy = true;
        p = ix->text_in[slot].p;
        return RBG_OK;
    }
    ~TextInHold() {
        if (slot == ~size_t(0)) return;
        std::lock_guard<std::mutex> g(ix->text_mu);
        ix->text_in[slot].busy = false;
    }
};
// the record of a buffer handed out (text_mu held by the caller)
rbg_index::TextOut *find_text_out(rbg_index *ix, const char *p) {
    for (auto &t : ix->text_out)
        if (t.p == p) return &t;
    return nullptr;
}
}  // namespace
extern "C" {

int rbg_align_text(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N, uint64_t max_hits, uint32_t flags,
                   const char *name_base, const uint64_t *name_begin, const uint32_t *name_len, const char **text, uint64_t *text_len) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    const bool with_locs = k != nullptr;   // k == NULL: the count-only report (rb_align without -s): one line per read
    const bool with_markers = (flags & RBG_TEXT_MARKERS) != 0;   // the "\tmarkers: ..." line of -m behind every read
    if (flags & ~static_cast<uint32_t>(RBG_TEXT_MARKERS)) return RBG_EARG;
    if (with_locs && (!ix->H().has_tsa || !ix->H().has_dl)) return RBG_ENOTLOADED;
    if (with_markers && !ix->H().has_ma) return RBG_ENOTLOADED;
    if (!text || !text_len || (N >> 32) || (N && (!lo || !hi || !name_base || !name_begin || !name_len))) return RBG_EARG;
    *text = nullptr;
    *text_len = 0;
    if (N == 0) return RBG_OK;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    int rc = with_locs ? ensure_text_docs(ix) : RBG_OK;
    if (rc) return rc;
    hipStream_t st = hipStreamPerThread;
    double lap_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto lap_from = std::chrono::steady_clock::now();
    auto lap = [&](int slot, bool sync) {
        if (!g_text_trace.on) return;
        if (sync) (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        lap_t[slot] += std::chrono::duration<double>(now - lap_from).count();
        lap_from = now;
    };
    // the call's inputs go through ONE pinned block -- [lo | hi | k | name offsets | names] filled by the worker threads,
    // one copy at PCIe rate (five copies out of pageable memory ran at 3 GB/s: 0.1 s per 10 M reads)
    const size_t o_lo = 0, o_hi = N * 8, o_k = 2 * N * 8, o_noff = 3 * N * 8, o_names = (o_noff + (N + 1) * 4 + 15) & ~size_t(15);
    uint64_t name_bytes = 0;
    for (uint64_t i = 0; i < N; ++i) name_bytes += name_len[i];
    if (name_bytes >> 32) return RBG_EARG;
    TextInHold in(ix);
    if ((rc = in.take(o_names + name_bytes + 16))) return rc;
    uint32_t *noff = reinterpret_cast<uint32_t *>(in.p + o_noff);
    {   // offsets: a serial prefix over the lengths (10 M additions: 10 ms), then everything else in parallel
        uint32_t acc = 0;
        for (uint64_t i = 0; i < N; ++i) { noff[i] = acc; acc += name_len[i]; }
        noff[N] = acc;
    }
    parallel_for(N, [&](uint64_t a, uint64_t b, unsigned) {
        std::memcpy(in.p + o_lo + a * 8, lo + a, (b - a) * 8);
        std::memcpy(in.p + o_hi + a * 8, hi + a, (b - a) * 8);
        if (with_locs) std::memcpy(in.p + o_k + a * 8, k + a, (b - a) * 8);
        for (uint64_t i = a; i < b; ++i) std::memcpy(in.p + o_names + noff[i], name_base + name_begin[i], name_len[i]);
    });
    lap(0, false);
    DevBuf din, doff, dtmp, dbad;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    if ((rc = din.alloc(o_names + name_bytes + 16)) || (rc = doff.alloc((N + 1) * 8)) || (rc = dtmp.alloc(tmp_bytes)) || (rc = dbad.alloc(16))) return rc;
    if (launch_copy16(in.p, din.p, o_names + name_bytes, st)) return RBG_ENODEV;
    HIP_TRY(hipMemsetAsync(dbad.p, 0, 16, st));
    DevView dlo{din.as<char>() + o_lo}, dhi{din.as<char>() + o_hi}, dk{din.as<char>() + o_k}, dnoff{din.as<char>() + o_noff}, dnames{din.as<char>() + o_names};
    lap(1, true);
    // locs_at (rowbowt.hpp:613-621) on the device, as rbg_locs_at does it -- the locations never leave it
    uint64_t nlocs = 0;
    if (with_locs) {
        if (launch_locate_plan(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, max_hits, doff.as<uint64_t>(), dtmp.p, tmp_bytes, st)) return RBG_ENODEV;
        HIP_TRY(hipMemcpyAsync(&nlocs, doff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        HIP_TRY(hipMemsetAsync(doff.p, 0, (N + 1) * 8, st));   // no locations: every read is one element
    }
    lap(2, false);
    DevBuf dlocs, dord, dws, dtext;
    if ((rc = dlocs.alloc(nlocs * 8))) return rc;
    const void *order = nullptr;
    if (with_locs && (rc = make_order(ix, dk.as<uint64_t>(), N, dord, st, &order))) return rc;
    if (nlocs && launch_locate_fill(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), dk.as<uint64_t>(), N, max_hits, doff.as<uint64_t>(),
                                    dlocs.as<uint64_t>(), nullptr, order, st))
        return RBG_ENODEV;
    lap(3, true);
    // markers_at (rowbowt.hpp:282-285) of every range, as rbg_markers_at does it -- they stay on the device too
    DevBuf dmoff, dmk;
    const uint64_t *d_mk_off = nullptr, *d_mk = nullptr;
    if (with_markers) {
        if ((rc = dmoff.alloc((N + 1) * 8))) return rc;
        if (launch_markers_plan(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, dmoff.as<uint64_t>(), dtmp.p, tmp_bytes, st)) return RBG_ENODEV;
        uint64_t nmk = 0;
        HIP_TRY(hipMemcpyAsync(&nmk, dmoff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = dmk.alloc(nmk * 8))) return rc;
        if (nmk && launch_markers_fill(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, dmoff.as<uint64_t>(), dmk.as<uint64_t>(), st)) return RBG_ENODEV;
        d_mk_off = dmoff.as<uint64_t>();
        d_mk = dmk.as<uint64_t>();
    }
    // the text: lengths, offsets, bytes (k_text.hip)
    const uint64_t E = N * (with_markers ? 2 : 1) + nlocs;
    const size_t ws_bytes = text_ws_bytes(E);
    if ((rc = dws.alloc(ws_bytes))) return rc;
    const auto &D = ix->text_docs;
    if (launch_text_plan(dlo.as<uint64_t>(), dhi.as<uint64_t>(), doff.as<uint64_t>(), dlocs.as<uint64_t>(), N, E, dnames.as<char>(), dnoff.as<uint32_t>(),
                              D.start, D.names, D.name_off, D.n, D.size, with_locs, d_mk_off, d_mk, dws.p, ws_bytes, dbad.as<unsigned int>(), st))
        return RBG_ENODEV;
    const uint64_t *p_at = nullptr;
    const uint32_t *p_len = nullptr;
    text_total_ptrs(dws.p, E, &p_at, &p_len);
    uint64_t last_at = 0;
    uint32_t last_len = 0, bad = 0;
    HIP_TRY(hipMemcpyAsync(&last_at, p_at, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&last_len, p_len, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    lap(4, false);
    if (bad) return RBG_EARG;   // a location before every document: rbg_resolve_offset's error (the reference indexes doc_names_[-1])
    const uint64_t total = last_at + last_len;
    if ((rc = dtext.alloc(total))) return rc;
    if (launch_text_fill(dlo.as<uint64_t>(), dhi.as<uint64_t>(), doff.as<uint64_t>(), dlocs.as<uint64_t>(), N, E, dnames.as<char>(), dnoff.as<uint32_t>(),
                              D.start, D.names, D.name_off, D.n, D.size, with_locs, d_mk_off, d_mk, dws.p, total, dtext.as<char>(), st))
        return RBG_ENODEV;
    lap(5, true);
    char *out = nullptr;
    if ((rc = take_text_out(ix, total, &out))) return rc;
    // the copy-out runs on the handle's copy stream, behind the fill kernel; the caller returns at once and the text's
    // reader waits (rbg_wait_text): the next batch's search and kernels run under this batch's 5 ms of PCIe
    {
        std::lock_guard<std::mutex> g(ix->text_mu);
        rbg_index::TextOut *t = find_text_out(ix, out);
        hipError_t e = hipEventRecord(t->done, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(ix->text_copy_stream, t->done, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(out, dtext.p, total, hipMemcpyDeviceToHost, ix->text_copy_stream);
        if (e == hipSuccess) e = hipEventRecord(t->done, ix->text_copy_stream);
        if (e != hipSuccess) { t->busy = false; HIP_TRY(e); }
        t->pending = true;
        t->d_text = dtext.p; t->d_cls = dtext.cls; t->d_dev = dtext.dev;
        dtext.p = nullptr;   // (the record owns the device block until the copy has been waited for)
    }
    if (g_text_trace.on) (void)rbg_wait_text(ix, out);
    lap(6, false);
    if (g_text_trace.on) {
        std::lock_guard<std::mutex> g(g_text_trace.mu);
        for (int j = 0; j < 8; ++j) g_text_trace.t[j] += lap_t[j];
        g_text_trace.calls += 1;
        g_text_trace.bytes += total;
    }
    *text = out;
    *text_len = total;
    return RBG_OK;
    });
}

int rbg_reserve_text(rbg_index *ix, uint64_t bytes, int count) {
    return guarded([&]() -> int {
    if (!queryable(ix) || count < 0 || count > 64) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    std::vector<char *> got;
    int rc = RBG_OK;
    for (int j = 0; j < count && !rc; ++j) {
        char *p = nullptr;
        rc = take_text_out(ix, bytes, &p);
        if (!rc) got.push_back(p);
    }
    for (char *p : got) (void)rbg_release_text(ix, p);
    return rc;
    });
}

int rbg_wait_text(rbg_index *ix, const char *text) {
    if (!ix) return RBG_EARG;
    if (!text) return RBG_OK;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->text_mu);
        rbg_index::TextOut *t = find_text_out(ix, text);
        if (!t || !t->busy) return RBG_EARG;
        if (!t->pending) return RBG_OK;
        ev = t->done;
    }
    const hipError_t e = hipEventSynchronize(ev);   // (outside the lock: other texts are being made meanwhile)
    std::lock_guard<std::mutex> g(ix->text_mu);
    rbg_index::TextOut *t = find_text_out(ix, text);
    if (t && t->pending) {
        t->pending = false;
        if (t->d_text) { DevPool::get().release(t->d_text, t->d_cls, t->d_dev); t->d_text = nullptr; }
    }
    return e == hipSuccess ? RBG_OK : RBG_ENODEV;
}

int rbg_release_text(rbg_index *ix, const char *text) {
    if (!ix) return RBG_EARG;
    if (!text) return RBG_OK;
    const int rc = rbg_wait_text(ix, text);   // (a text given back unread: its copy must not land in a buffer that has a new owner)
    std::lock_guard<std::mutex> g(ix->text_mu);
    rbg_index::TextOut *t = find_text_out(ix, text);
    if (!t) return RBG_EARG;
    t->busy = false;
    return rc;
}

}  // extern "C"
