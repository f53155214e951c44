// capi/load.ipp -- upload() (layout and budget rules, the slot layout's tables), options_for(), finish(); the loading / converting / building
// entry points, rbg_info and the host-array accessors.  Part of rbg_capi.hip.
namespace {
int upload(rbg_index *ix) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ix->device < 0 || ix->device >= ndev) {
        std::fprintf(stderr, "rbg: no usable HIP device %d (found %d); this library has no CPU path\n", ix->device, ndev);
        return RBG_ENODEV;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "rbg: device %d is %s; kernels are built for gfx950 (MI355X) only\n", ix->device, prop.gcnArchName);
        return RBG_ENODEV;
    }
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    HostIndex &h = ix->H();
    DevIndex &d = ix->dev;
    d = DevIndex{};
    d.n = h.n;
    d.r = h.r;
    d.sigma = h.sigma;
    d.pos_bytes = h.pos_bytes;
    d.has_tsa = h.has_tsa ? 1 : 0;
    d.last_run_sample = h.last_run_sample;
    d.phi_shift = h.phi_shift;
    if (h.kmer_deferred) {   // depths 2.. composed on the device (flatten() only chose the k-mer alphabet)
        const int rcc = h.pos_bytes == 4 ? compose_on_device<uint32_t>(ix) : compose_on_device<uint64_t>(ix);
        if (rcc) return rcc;
    }
    // The k-mer tables buy speed with memory: keep the deepest level that fits a quarter of the free HBM
    // (default_budget above) or RBG_OPT_HBM_BUDGET_MB.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const int64_t opt_mb = g_opt_hbm_budget_mb.load();
    // (the budget options_for() fixed before the composition put its levels on the device, where it was taken: VERDICT r4 item 8)
    const size_t budget = ix->plan_budget ? static_cast<size_t>(ix->plan_budget) : opt_mb > 0 ? static_cast<size_t>(opt_mb) << 20 : default_budget(free_b);
    if (ix->plan_free) free_b = static_cast<size_t>(ix->plan_free);
    auto need = [&] { return h.pos_bytes == 4 ? replica_bytes<uint32_t>(h) : replica_bytes<uint64_t>(h); };
    // Layout: the slot tables cost n/16 bytes per table + n/2 (n at 8-byte positions) for phi, whatever r is.  When
    // even the single-symbol level does not fit the budget -- or on request -- the run-indexed layout takes over
    // (space proportional to r; wave-cooperative predecessor search, k_runs.hip).
    bool runs_layout = g_opt_rank_layout.load() == RBG_LAYOUT_RUNS || ix->auto_runs;
    if (!runs_layout && layout_automatic()) {
        size_t lvl1 = 0;  // the single-symbol level alone
        {
            std::vector<SymTable> held[kMaxKmerDepth - 1];
            for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d) held[d - 2].swap(h.kmer(d));
            lvl1 = need();
            for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d) held[d - 2].swap(h.kmer(d));
        }
        runs_layout = lvl1 > budget;
    }
    // RBG_LAYOUT_AUTO, second look (options_for's was an estimate from n alone, before anything was composed): if the slot tables of the
    // levels at hand exceed the budget, the rule below would give the deep levels wider buckets and then drop levels -- the run-indexed
    // layout keeps every level at full speed instead, while it fits (about 110 bytes per run at its leanest).  Measured on the bench
    // index at the default budget: 1.13e9 reads/s from the 70 GB of five symbols in wide buckets, 1.14e9 from the 59 GB of four
    // symbols, 1.23e9 from the 8.7 GB of this layout (profiles/r04_bench.json space_speed / value_library_default).
    if (!runs_layout && g_opt_rank_layout.load() == RBG_LAYOUT_AUTO && !h.kmer(2).empty() && h.sigma <= static_cast<uint32_t>(kLdsSyms) &&
        need() > budget && 110.0 * static_cast<double>(h.r) <= static_cast<double>(budget)) {
        if (std::getenv("RBG_VERBOSE"))
            std::fprintf(stderr, "rbg: device %d: the slot tables of all k-mer levels (%.1f GB) exceed the %.1f GB replica budget: the run-indexed layout instead of "
                                 "wider buckets or fewer symbols per step (RBG_LAYOUT_PREFER_SLOTS keeps slot tables)\n", ix->device, need() / 1e9, budget / 1e9);
        runs_layout = true;
        ix->auto_runs = true;
        // the levels at hand are the slot layout's (at most kMaxSlotKmerDepth, composed before this look could be taken); the run-indexed
        // layout steps by as many symbols as were asked for: compose again, that deep (rare: options_for's estimate usually decides first)
        const uint32_t asked = static_cast<uint32_t>(std::min<int64_t>(g_opt_kmer_steps.load(), kMaxKmerDepth));
        if (asked > h.kmer_levels() && compose_deferred(ix->device) && h.nmajor >= 1) {
            for (SymTable &t : h.sym) {
                free_tracked(ix, const_cast<void *>(t.dev_ent));
                free_tracked(ix, const_cast<void *>(t.dev_samp));
                t.dev_ent = t.dev_samp = nullptr;
            }
            h.kmer_deferred = asked;
            ix->kmer_steps_requested = asked;
            const int rcc = h.pos_bytes == 4 ? compose_on_device<uint32_t>(ix) : compose_on_device<uint64_t>(ix);
            if (rcc) return rcc;
        }
    }
    if (ix->runs_forced) runs_layout = true;
    if (runs_layout && h.sigma > static_cast<uint32_t>(kLdsSyms)) {
        std::fprintf(stderr, "rbg: %u distinct symbols: the run-indexed layout serves at most %d; keeping the slot tables\n", h.sigma, kLdsSyms);
        runs_layout = false;
    }
    ix->runs_layout = runs_layout;
    if (!runs_layout)   // (the slot layout stages at most kMaxSlotKmerDepth symbols per gather)
        while (levels_of(h) > kMaxSlotKmerDepth) drop_kmer_level(ix, h.kmer(h.kmer_levels()));
    auto levels = [&] { return static_cast<int>(h.kmer_levels()); };
    if (runs_layout) {
        // the k-mer depths stay (their run lists are O(r) too: DevRunTab2, rbg_dev.h); the deepest goes while the replica
        // exceeds the budget
        if (ix->kmer_steps_requested == 0) ix->kmer_steps_requested = static_cast<uint64_t>(levels());
        // RBG_OPT_RUN_DEPTHS: a step needs no table of every depth below the deepest -- a stretch of 4 symbols is a depth-3
        // step and a single one where depth 4 is left out -- and the deepest lists are the largest (2.4 entries
        // per run at depth 5 of the H = 200 pangenome, 9.3 over the five).  Over budget the depths between the first and the
        // deepest go first (deepest of them first), then the deepest itself.
        // Default: every other depth counted from the deepest (1, 3, 5 of five) -- two thirds of the space and the same
        // rate on whole reads, a few per cent more steps where stretches are ragged (marker seeds); 0x1F keeps them all.
        ix->runs_report = rbg_index::RunsReport();
        ix->runs_report.depth_mask_asked = static_cast<uint32_t>(g_opt_run_depths.load());
        uint32_t mask = g_opt_run_depths.load() ? static_cast<uint32_t>(g_opt_run_depths.load()) | 1u : default_depth_mask(static_cast<uint32_t>(levels()));
        auto deepest_of = [&]() -> std::vector<SymTable> & { return h.kmer(static_cast<uint32_t>(std::max(2, levels()))); };
        while (levels() > 1 && !(mask >> (levels() - 1) & 1u)) drop_kmer_level(ix, deepest_of());   // (nothing steps by a depth above the deepest kept)
        mask &= (1u << levels()) - 1u;
        auto need_runs = [&] { return h.pos_bytes == 4 ? runs_replica_bytes<uint32_t>(h, mask) : runs_replica_bytes<uint64_t>(h, mask); };
        while (need_runs() > budget && levels() > 1) {
            uint32_t mid = 0;
            for (int d = levels() - 1; d >= 2 && !mid; --d)
                if (mask >> (d - 1) & 1u) mid = static_cast<uint32_t>(d);
            if (mid) {
                std::fprintf(stderr, "rbg: run-indexed replica of %.1f GB exceeds the %.1f GB budget: leaving out the run lists of depth %u\n", need_runs() / 1e9, budget / 1e9, mid);
                mask &= ~(1u << (mid - 1));
                ix->runs_report.depths_dropped_budget |= 1u << (mid - 1);
                continue;
            }
            std::fprintf(stderr, "rbg: run-indexed replica of %.1f GB exceeds the %.1f GB budget: dropping the %zu-table k-mer level\n",
                         need_runs() / 1e9, budget / 1e9, deepest_of().size());
            ix->runs_report.depths_dropped_budget |= 1u << (levels() - 1);
            drop_kmer_level(ix, deepest_of());
            // (the new deepest depth must still have its lists: a depth the composition gave back early goes too)
            while (levels() > 1 && !level_has_data(ix, static_cast<uint32_t>(levels()))) drop_kmer_level(ix, deepest_of());
            mask = (mask & ((1u << levels()) - 1u)) | (1u << (levels() - 1));   // (the new deepest level is stepped by again)
        }
        // Records everywhere before depths in between: a depth whose ranks go through a directory pays narrowing rounds in crowded buckets (n = 5e10 under the
        // default budget: 2.7 rounds and 78 scanned entries per read, K2 13.8 ms against 7 with records; profiles/r05_pangenome_stream_n5e10_default.json),
        // the depths between the first and the deepest save a step per read.  So when the automatic rules decide (no RBG_OPT_RUN_DEPTHS, no RBG_OPT_RUN_REC) and
        // bucket records for every kept depth do not fit the budget, but would with the first and the deepest depth alone, the depths between them go.
        if (g_opt_run_depths.load() == 0 && g_opt_run_rec.load() == 0 && levels() > 2) {
            const uint32_t ends = 1u | (1u << (levels() - 1));
            const uint32_t max_shift = h.pos_bytes == 8 ? static_cast<uint32_t>(env_opt("RBG_RUN_FILL_SHIFT", kRunFillShift, 4, kRunFillShift)) : 31u;
            auto with_records = [&](uint32_t m) {
                double bytes = static_cast<double>(h.pos_bytes == 4 ? runs_replica_bytes<uint32_t>(h, m) : runs_replica_bytes<uint64_t>(h, m));
                if (g_opt_run_phi.load() != 1) bytes += runs_phi_slot_bytes(h);
                for (int d = 0; d < levels(); ++d)
                    if (m >> d & 1u) bytes += runs_record_count(h, static_cast<uint32_t>(d), 6.0, max_shift) * 64.0;
                return bytes;
            };
            if (mask != ends && with_records(mask) > static_cast<double>(budget) && with_records(ends) <= static_cast<double>(budget)) {
                std::fprintf(stderr, "rbg: bucket records for every kept depth (%.1f GB with them) exceed the %.1f GB budget: leaving out the depths between 1 and %d "
                                     "(%.1f GB with records for both)\n", with_records(mask) / 1e9, budget / 1e9, levels(), with_records(ends) / 1e9);
                ix->runs_report.depths_dropped_budget |= mask & ~ends;
                mask = ends;
            }
        }
        for (int d = 2; d < levels(); ++d)   // the depths left out give their device arrays back now
            if (!(mask >> (d - 1) & 1u)) {
                release_kmer_level(ix, static_cast<uint32_t>(d));
                for (SymTable &st : kmer_level_tables(h, static_cast<uint32_t>(d))) st.dev_ent = st.dev_samp = nullptr;
            }
        ix->run_depth_mask = mask;
    }
    ix->kmer_steps_requested = std::max<uint64_t>(ix->kmer_steps_requested, static_cast<uint64_t>(levels()));  // options_for() may have capped the depth already
    ix->hbm_free_at_load = free_b;
    ix->hbm_budget = budget;
    // Over budget: first give the k-mer levels wider buckets, deepest level first (their runs are sparse: a table
    // goes to the widest bucket that still holds about half a run start on average, at most 4096 rows, in the
    // wide-slot encoding of rbg_dev.h -- a few per cent slower per step), then drop the deepest level
    // and try again.  At pangenome scale this keeps a level more than dropping alone.
    auto widen = [&](std::vector<SymTable> &lvl) {
        if (h.n >> 40) return;  // wide slots carry 40-bit ranks
        for (SymTable &t : lvl) {
            const double rows_per_run = static_cast<double>(h.n) / static_cast<double>(std::max<uint64_t>(1, t.nruns));
            uint32_t want = 0;
            while (want < kMaxWideShift && static_cast<double>(uint64_t(2) << want) <= rows_per_run) ++want;   // 2^want <= rows_per_run / 2
            if (want > t.shift) t.shift = want;
        }
    };
    bool widened = false;
    while (need() > budget && !h.kmer(2).empty() && !runs_layout) {
        if (!widened && g_opt_deep_shift.load() < 0 && g_opt_rank_shift.load() < 0) {
            widened = true;
            for (uint32_t wd = static_cast<uint32_t>(kMaxSlotKmerDepth); wd >= 2; --wd) {
                std::vector<SymTable> *lvl = &h.kmer(wd);
                if (!lvl->empty() && need() > budget) {
                    const size_t before = need();
                    widen(*lvl);
                    if (need() != before)
                        std::fprintf(stderr, "rbg: replica of %.1f GB exceeds the %.1f GB budget: wider buckets for the %zu-table k-mer level (%.1f GB)\n",
                                     before / 1e9, budget / 1e9, lvl->size(), need() / 1e9);
                }
            }
            continue;
        }
        std::vector<SymTable> &deepest = h.kmer(static_cast<uint32_t>(levels()));
        std::fprintf(stderr, "rbg: replica of %.1f GB exceeds the %.1f GB budget: dropping the %zu-table k-mer level\n",
                     need() / 1e9, budget / 1e9, deepest.size());
        drop_kmer_level(ix, deepest);
    }
    // (the slot layout stages at most kMaxSlotKmerDepth symbols per gather: a default request of eight that became five there is not news)
    const uint64_t asked_here = runs_layout ? ix->kmer_steps_requested : std::min<uint64_t>(ix->kmer_steps_requested, static_cast<uint64_t>(kMaxSlotKmerDepth));
    if (std::getenv("RBG_VERBOSE") || static_cast<uint64_t>(levels()) != asked_here)
        std::fprintf(stderr, "rbg: device %d: %.1f GB free, replica budget %.1f GB: keeping %d of %llu symbol(s) per %s (%.1f GB)\n", ix->device,
                     free_b / 1e9, budget / 1e9, levels(), static_cast<unsigned long long>(ix->kmer_steps_requested), runs_layout ? "search step" : "gather",
                     (runs_layout ? (h.pos_bytes == 4 ? runs_replica_bytes<uint32_t>(h, ix->run_depth_mask) : runs_replica_bytes<uint64_t>(h, ix->run_depth_mask)) : need()) / 1e9);
    if (runs_layout && std::getenv("RBG_VERBOSE")) std::fprintf(stderr, "rbg: device %d: k-mer depths with run lists: mask 0x%x\n", ix->device, ix->run_depth_mask);
    int rc;
    d.layout = RBG_LAYOUT_SLOTS;
    if (runs_layout) {
        if (std::getenv("RBG_VERBOSE")) std::fprintf(stderr, "rbg: device %d: run-indexed layout (space proportional to r)\n", ix->device);
        rc = h.pos_bytes == 4 ? upload_tables_runs2<uint32_t>(ix) : upload_tables_runs2<uint64_t>(ix);
        if (rc) return rc;
    } else {
        ix->arena_bytes = h.pos_bytes == 4 ? replica_bytes<uint32_t>(h, true) : replica_bytes<uint64_t>(h, true);   // (without the lists that are on the device already)
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (ix->arena_bytes > free_b) {
            std::fprintf(stderr, "rbg: index needs %.1f GB of HBM, %.1f GB free\n", ix->arena_bytes / 1e9, free_b / 1e9);
            return RBG_ENOMEM;
        }
        {
            VStage vs("arena hipMalloc");
            // (a platform cost: fresh VRAM is mapped and cleared at some 45-70 GB/s when the memory was freed shortly before --
            //  2.7-7 s for the bench replica's 218 GB -- and next to nothing when it has been idle: tools/alloc_probe.py,
            //  profiles/r03_load_time.txt.  Asking for the block from a helper thread while the host flattens and the
            //  device composes was tried: the driver serialises the composition's own allocations behind it, no gain.)
            HIP_TRY(hipMalloc(&ix->arena, ix->arena_bytes));
        }
        ix->allocs.push_back({ix->arena, ix->arena_bytes});
        ix->hbm_bytes += ix->arena_bytes;
        ix->arena_used = 0;
        rc = h.pos_bytes == 4 ? upload_tables<uint32_t>(ix) : upload_tables<uint64_t>(ix);
        if (rc) return rc;
    }
    const void *p = nullptr;
    if ((rc = dev_upload(ix, h.lut, 256, &p))) return rc;
    d.lut = static_cast<const uint8_t *>(p);
    const unsigned long long zero[4] = {0, 0, 0, 0};
    if ((rc = dev_upload(ix, zero, sizeof(zero), &p))) return rc;
    d.counters = const_cast<unsigned long long *>(static_cast<const unsigned long long *>(p));
    if (h.has_ma) {
        VStage vs("markers");
        if ((rc = upload_markers(ix))) return rc;
    }
    ix->cfg.block_threads = static_cast<int>(g_opt_block_threads.load());
    ix->cfg.max_blocks = prop.multiProcessorCount * 32;
    // ftab (next-row f3): built last, with the finished replica, by searching every word on the GPU
    d.ftab = nullptr;
    d.ftab_k = 0;
    int64_t fk = g_opt_ftab_k.load();
    if (fk < 0) {  // automatic: the longest word <= 12 with nmajor^k <= n/16 (4^12 words x 16 B = 268 MB; DESIGN.md 4 on why not longer)
        fk = 0;
        double w = 1;
        while (d.nmajor >= 2 && fk < 12 && w * d.nmajor <= static_cast<double>(h.n) / 16) { w *= d.nmajor; ++fk; }
    }
    if (fk > 0 && d.nmajor >= 2) {
        VStage vs("ftab");
        double words = 1;
        for (int64_t t = 0; t < fk; ++t) words *= d.nmajor;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // the table plus the scratch of building it (in chunks) must leave half of the free memory to the queries
        const size_t entry = h.pos_bytes == 4 ? 16 : 32;
        if (words < 4.0e9 &&
            words * static_cast<double>(entry) + static_cast<double>(ftab_build_scratch_bytes(static_cast<uint64_t>(words), static_cast<uint32_t>(fk))) <
                0.5 * static_cast<double>(free_b)) {
            const uint64_t W = static_cast<uint64_t>(words);
            void *tab = nullptr;
            HIP_TRY(hipMalloc(&tab, W * entry));
            ix->allocs.push_back({tab, static_cast<size_t>(W * entry)});
            ix->hbm_bytes += W * entry;
            if (launch_build_ftab(d, ix->cfg, static_cast<uint32_t>(fk), tab, nullptr)) return RBG_ENODEV;
            d.ftab = tab;
            d.ftab_k = static_cast<uint32_t>(fk);
            HIP_TRY(hipMemset(d.counters, 0, 4 * sizeof(uint64_t)));  // the build's own searches are not user queries
        }
    }
    return RBG_OK;
}

FlattenOptions current_options() {
    FlattenOptions o;
    o.rank_bucket_shift = static_cast<int>(g_opt_rank_shift.load());
    o.deep_bucket_shift = static_cast<int>(g_opt_deep_shift.load());
    o.phi_bucket_shift = static_cast<int>(g_opt_phi_shift.load());
    o.force_pos_bytes = static_cast<int>(g_opt_pos_bytes.load());
    o.kmer_steps = static_cast<int>(g_opt_kmer_steps.load());
    return o;
}

// The options of a load that is going to `device`: k-mer levels that cannot fit the replica budget even in their
// smallest form (every table at the widest bucket, kMaxWideShift) are not composed at all -- upload() would drop them
// anyway, and composing the deepest level is the most expensive part of flatten() (47 of 76 s at n = 5e10).  The bound
// is conservative: a level upload() could keep is never excluded.  *requested = the depth asked for when it was
// capped here (else 0: upload() reports what flatten() composed).
// the k-mer tables of an index that goes to a device are composed there (RBG_HOST_COMPOSE=1: on the host, the reference
// statement -- A/B measurements and the test that compares the two)
bool compose_deferred(int device) {
    const char *e = std::getenv("RBG_HOST_COMPOSE");
    return device != RBG_DEVICE_NONE && !(e && e[0] == '1');
}

FlattenOptions options_for(int device, const RawRle &rle, uint64_t *requested, bool *auto_runs, rbg_index *ix) {
    FlattenOptions o = current_options();
    *requested = 0;
    *auto_runs = false;
    // A host-only handle (conversion, cache writing, tests) answers no query -- the library has no CPU search path -- so nothing reads k-mer
    // tables there: compose none (eight depths on the host cost 3 GB and a second per 3e6 runs, hundreds of GB at pangenome r: ADVICE r5).
    // (RBG_HOST_COMPOSE=1 keeps the host composition -- the serial reference statement of k_compose.hip -- reachable without a device: tests)
    if (device == RBG_DEVICE_NONE) {
        const char *e = std::getenv("RBG_HOST_COMPOSE");
        if (!(e && e[0] == '1')) o.kmer_steps = 1;
        return o;
    }
    if (o.kmer_steps < 2) return o;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return o;
    DeviceScope scope(device);
    if (scope.rc) return o;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return o;
    free_b = assumed_free_hbm(free_b);
    const int64_t opt_mb = g_opt_hbm_budget_mb.load();
    double budget = static_cast<double>(opt_mb > 0 ? static_cast<size_t>(opt_mb) << 20 : default_budget(free_b));
    // (the budget of this load is fixed HERE, while nothing of it is on the device: upload() measures again after the composition has
    //  taken its share, and a quarter of what is left then is not a quarter of the device)
    ix->plan_free = free_b;
    ix->plan_budget = static_cast<uint64_t>(budget);
    bool seen[256] = {};
    unsigned sigma = 0;
    for (uint8_t c : rle.heads)
        if (!seen[c]) { seen[c] = true; ++sigma; }
    const double major = static_cast<double>(std::min(4u, sigma > 1 ? sigma - 1 : 0u));  // at least this many k-mer symbols
    if (major < 2) return o;
    // the run-indexed layout keeps its k-mer depths as run lists (space proportional to r): nothing to cap when it is
    // asked for, or when not even the single-symbol slot tables (+ phi at its widest usual bucket) fit
    if (g_opt_rank_layout.load() == RBG_LAYOUT_RUNS) return o;
    if (layout_automatic()) {
        const double pos = (o.force_pos_bytes == 8 || rle.n >= 0xFFFFFFF0ull) ? 8 : 4;
        const double lvl1 = static_cast<double>(sigma) * static_cast<double>((rle.n >> kMaxNarrowShift) + 2) * (sizeof(RankSlot) + sizeof(uint32_t)) +
                            static_cast<double>((rle.n >> 6) + 2) * (4 * pos + 4);
        if (lvl1 > budget) {
            // not even the single-symbol slot tables fit: the run-indexed layout, certainly (where the alphabet allows it)
            if (sigma <= static_cast<unsigned>(kLdsSyms)) {
                *auto_runs = true;
                // An index that large may also be too large for the DEFAULT budget to step by more than a symbol or two (r = 1e9: 36 GB of run lists
                // and phi before any k-mer depth; profiles/r05_pangenome_stream_r1e9_default.json: 5e7 reads/s from the quarter, 1.4e8 from the fast
                // form).  RBG_LAYOUT_AUTO with no budget given then takes up to three quarters of the free HBM -- said on stderr, reported by
                // rbg_info (hbm_budget) and rbg_layout_info (budget_raised); RBG_OPT_HBM_BUDGET_MB decides otherwise.
                const double r = static_cast<double>(rle.heads.size());
                const uint32_t want = static_cast<uint32_t>(std::min(o.kmer_steps, 4));
                // ... and only on a device this load has (almost) to itself: with a tenth of it or more already taken -- the caller's own buffers,
                // another replica, another process -- the quarter stays (a caller that sized its buffers around the quarter rule).
                const size_t whole = std::getenv("RBG_ASSUME_FREE_HBM_MB") ? free_b : total_b;
                const bool device_to_itself = static_cast<double>(free_b) >= 0.9 * static_cast<double>(whole);
                if (opt_mb == 0 && g_opt_rank_layout.load() == RBG_LAYOUT_AUTO && device_to_itself &&
                    planned_depth(r, true, static_cast<uint32_t>(o.kmer_steps), static_cast<double>(free_b), budget, true) < want) {
                    const double raised = 0.75 * static_cast<double>(free_b);
                    if (planned_depth(r, true, static_cast<uint32_t>(o.kmer_steps), static_cast<double>(free_b), raised, true) >
                        planned_depth(r, true, static_cast<uint32_t>(o.kmer_steps), static_cast<double>(free_b), budget, true)) {
                        std::fprintf(stderr, "rbg: device %d: r = %.3g runs: a quarter of the free HBM (%.1f GB) would leave fewer than %u symbols per step; RBG_LAYOUT_AUTO takes up to "
                                             "three quarters (%.1f GB) for this index (RBG_OPT_HBM_BUDGET_MB sets the budget explicitly)\n", device, r, budget / 1e9, want, raised / 1e9);
                        budget = raised;
                        ix->plan_budget = static_cast<uint64_t>(budget);
                        ix->budget_raised = true;
                    }
                }
            }
            return o;
        }
    }
    // the slot layout stages the tables of at most kMaxSlotKmerDepth symbols per gather: more are asked of the run-indexed layout only
    const int slot_steps = std::min(o.kmer_steps, kMaxSlotKmerDepth);
    auto slot_levels_fitting = [&](uint32_t shift) {   // the deepest level whose slot tables (every table at this bucket shift) fit the budget with the levels above it
        const double per_table = static_cast<double>((rle.n >> shift) + 2) * (sizeof(RankSlot) + sizeof(uint32_t));
        double total = major * per_table, tables = major;
        int keep = 1;
        for (int k = 2; k <= slot_steps; ++k) {
            tables *= major;
            total += tables * per_table;
            if (total > budget) break;
            keep = k;
        }
        return keep;
    };
    const int keep = slot_levels_fitting(kMaxWideShift);   // conservative: a level upload() could keep is never excluded
    if (g_opt_rank_layout.load() == RBG_LAYOUT_AUTO && sigma <= static_cast<unsigned>(kLdsSyms) && slot_levels_fitting(kMaxNarrowShift) < slot_steps) {
        // RBG_LAYOUT_AUTO: slot tables only while those of every symbol per step fit the budget at their narrow buckets; rather than give up
        // symbols per step -- or widen the buckets -- the run-indexed layout (all of them, in space proportional to r; deeper steps than the
        // slot layout has: RBG_OPT_KMER_STEPS up to 8).  On the bench index 1.26e9 reads/s from 8.7 GB against 1.18e9 from the 59 GB of four
        // symbols per step (profiles/r04_bench.json space_speed); about 110 bytes per run at its leanest.
        const double runs_least = 110.0 * static_cast<double>(rle.heads.size());
        if (runs_least <= budget) {
            if (std::getenv("RBG_VERBOSE"))
                std::fprintf(stderr, "rbg: device %d: the slot tables of %d symbols per step exceed the %.1f GB replica budget at narrow buckets: the run-indexed layout "
                                     "instead, %d symbols per step (RBG_OPT_RANK_LAYOUT = RBG_LAYOUT_PREFER_SLOTS keeps slot tables with fewer symbols)\n", device, slot_steps,
                             budget / 1e9, o.kmer_steps);
            *auto_runs = true;
            return o;
        }
    }
    if (keep < o.kmer_steps) {
        if (keep < slot_steps) {
            if (std::getenv("RBG_VERBOSE"))
                std::fprintf(stderr, "rbg: device %d: %.1f GB replica budget cannot hold k-mer levels beyond %d at n = %.3g: not composing them\n", device,
                             budget / 1e9, keep, static_cast<double>(rle.n));
            *requested = static_cast<uint64_t>(slot_steps);   // (what the slot layout could have taken of the depth asked for)
        }
        o.kmer_steps = keep;   // (the slot layout: at most kMaxSlotKmerDepth symbols per gather)
    }
    return o;
}

int upload_order_docs(rbg_index *ix);   // (below, beside rbg_set_docs)

int finish(rbg_index *ix, int device, rbg_index **out) {
    ix->device = device;
    if (device != RBG_DEVICE_NONE) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = upload(ix);
        if (!rc) rc = upload_order_docs(ix);   // (a document table loaded with the index: K3's chains in locus order)
        if (rc) { rbg_free(ix); return rc; }
        if (std::getenv("RBG_VERBOSE"))
            std::fprintf(stderr, "rbg: slot tables + upload %.2f s (%.2f GB)\n",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), ix->hbm_bytes / 1e9);
    }
    *out = ix;
    return RBG_OK;
}

bool queryable(const rbg_index *ix) { return ix && ix->device != RBG_DEVICE_NONE; }

bool markers_valid(const uint64_t *s, const uint64_t *e, uint64_t nruns, const uint64_t *off) {
    for (uint64_t j = 0; j < nruns; ++j) {
        if (e[j] < s[j] || off[j] > off[j + 1]) return false;
        if (j && s[j] <= e[j - 1]) return false;  // disjoint, ascending
    }
    return nruns == 0 || off[0] == 0;
}

// common staging for host read batches
struct ReadBatch {
    DevBuf seqs, off;
    int stage(const uint8_t *h_seqs, const uint64_t *h_off, uint64_t N, hipStream_t st) {
        const uint64_t total = N ? h_off[N] : 0;
        int rc;
        if ((rc = seqs.alloc(((total + 15) & ~uint64_t(15)) + 16))) return rc;
        if ((rc = off.alloc((N + 1) * 8))) return rc;
        if (total) HIP_TRY(hipMemcpyAsync(seqs.p, h_seqs, total, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(off.p, h_off, (N + 1) * 8, hipMemcpyHostToDevice, st));
        return RBG_OK;
    }
};

// host-pointer locate paths: order the chains when the batch is big enough for the sort to pay
int make_order(rbg_index *ix, const uint64_t *d_k, uint64_t N, DevBuf &ws, hipStream_t st, const void **order) {
    *order = nullptr;
    if (N < 4096 || N >= 0xFFFFFFFFull) return RBG_OK;
    const size_t bytes = locate_order_ws_bytes(N);
    int rc = ws.alloc(bytes);
    if (rc) return rc;
    if (launch_locate_order(ix->dev, ix->cfg, d_k, N, ws.p, bytes, st)) return RBG_ENODEV;
    *order = ws.p;
    return RBG_OK;
}

int check_offsets(const uint64_t *off, uint64_t N) {
    if (N == 0) return RBG_OK;
    if (!off || off[0] != 0) return RBG_EARG;
    for (uint64_t i = 0; i < N; ++i)
        if (off[i + 1] < off[i]) return RBG_EARG;
    return RBG_OK;
}

// Host memory for a ragged result (released by rbg_free_buffer = free).  The device-to-host copy is the
// first touch of this memory, and for gigabytes of locations the page faults cost more than the PCIe
// transfer (tools/d2h_probe.hip: 3 GB in 0.22 s into fresh malloc memory, 0.13-0.16 s into 2 MB-aligned
// memory marked for transparent huge pages, 0.06 s once touched), so large results ask for huge pages.
// Large results are RECYCLED: rbg_free_buffer keeps blocks of 8 MB and more (up to 6 GB in all) and the next result
// of about that size gets one whose pages are already there -- a batch loop (rb_markers: 1 GB of seed records per 2 M reads;
// rbg_locs_at: 3 GB per 10 M reads) otherwise faults the same pages in again at every call, which costs more than the copy
// (0.098 s of copy-out per 2 M reads in rb_markers, 0.03 s with recycled blocks).  RBG_RESULT_POOL=0 switches it off.
struct ResultPool {
    std::mutex mu;
    std::map<void *, size_t> live;            // blocks handed out by alloc_result (pooled sizes only)
    std::multimap<size_t, void *> idle;
    size_t cached = 0;
    const bool on = !(std::getenv("RBG_RESULT_POOL") && std::getenv("RBG_RESULT_POOL")[0] == '0');
    static constexpr size_t kMax = size_t(6) << 30;
    static ResultPool &get() { static ResultPool p; return p; }
    ~ResultPool() { for (auto &kv : idle) std::free(kv.second); }
};
void *alloc_result(size_t bytes) {
    constexpr size_t kHuge = size_t(2) << 20;
    ResultPool &P = ResultPool::get();
    if (bytes >= 4 * kHuge) {
        const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
        if (P.on) {
            std::lock_guard<std::mutex> g(P.mu);
            auto it = P.idle.lower_bound(rounded);
            if (it != P.idle.end() && it->first <= rounded + rounded / 4) {
                void *p = it->second;
                P.live[p] = it->first;
                P.cached -= it->first;
                P.idle.erase(it);
                return p;
            }
        }
        void *p = std::aligned_alloc(kHuge, rounded);
        if (p) {
            (void)madvise(p, rounded, MADV_HUGEPAGE);
            if (P.on) { std::lock_guard<std::mutex> g(P.mu); P.live[p] = rounded; }
            return p;
        }
    }
    return std::malloc(bytes ? bytes : 8);
}

// pinned staging of big ragged results (ragged_finish): four 64 MB buffers per process, allocated on first use
struct PinnedStage {
    static constexpr size_t kChunk = size_t(64) << 20;
    static constexpr int kBufs = 4;
    std::mutex mu;
    void *buf[kBufs] = {nullptr, nullptr, nullptr, nullptr};   // portable: any device of the process may copy into them
    bool ok = false, tried = false;
    static PinnedStage &get() { static PinnedStage p; return p; }
    bool ensure() {   // (under mu)
        if (tried) return ok;
        tried = true;
        for (int i = 0; i < kBufs; ++i)
            if (rbg_numa::host_malloc_near(&buf[i], kChunk, hipHostMallocPortable, [] { int d = 0; (void)hipGetDevice(&d); return d; }()) != hipSuccess) {
                (void)hipGetLastError();
                return ok = false;
            }
        return ok = true;
    }
};

// Device-to-host copy of a (possibly huge) result into memory that may never have been touched.  Big results leave
// through pinned staging: a copy straight into fresh pageable memory is the first touch of its pages, and for gigabytes
// of locations the page faults (and the driver's own staging) cost more than the transfer (tools/d2h_probe.hip: 3 GB in
// 0.2 s; 0.06 s for the DMA alone).  Chunks of 64 MB are copied into four pinned buffers, two copies ahead, and a team
// of worker threads moves each finished chunk to its place -- which is where the pages get touched, by sixteen threads
// at once and alongside the next chunks' DMA.  Blocks until the data has arrived.
int d2h_result(void *h_dst, const void *d_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return RBG_OK;
    if (bytes >= (size_t(64) << 20)) {
        PinnedStage &ps = PinnedStage::get();
        std::unique_lock<std::mutex> lk(ps.mu, std::try_to_lock);   // (one big result at a time goes this way; a second caller takes the plain copy)
        if (lk.owns_lock() && ps.ensure()) {
            const size_t chunk = PinnedStage::kChunk;
            const size_t nb = (bytes + chunk - 1) / chunk;
            const unsigned T = std::max(1u, std::min(16u, rbg_hostpath::cpu_budget()));
            rbg_hostpath::ThreadTeam team(T);
            char *dst = static_cast<char *>(h_dst);
            const char *src = static_cast<const char *>(d_src);
            hipError_t e = hipSuccess;
            hipEvent_t ev[PinnedStage::kBufs] = {nullptr, nullptr, nullptr, nullptr};   // (per call: events belong to the current device)
            for (hipEvent_t &x : ev)
                if (e == hipSuccess) e = hipEventCreateWithFlags(&x, hipEventDisableTiming);
            auto enqueue = [&](size_t c) {
                const size_t len = std::min(chunk, bytes - c * chunk);
                if (e == hipSuccess) e = hipMemcpyAsync(ps.buf[c % PinnedStage::kBufs], src + c * chunk, len, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipEventRecord(ev[c % PinnedStage::kBufs], st);
            };
            for (size_t c = 0; c < std::min<size_t>(2, nb); ++c) enqueue(c);
            for (size_t c = 0; c < nb && e == hipSuccess; ++c) {
                e = hipEventSynchronize(ev[c % PinnedStage::kBufs]);
                if (e != hipSuccess) break;
                if (c + 2 < nb) enqueue(c + 2);   // its buffer held chunk c - 2, which has been moved out
                const size_t len = std::min(chunk, bytes - c * chunk);
                const char *from = static_cast<const char *>(ps.buf[c % PinnedStage::kBufs]);
                char *to = dst + c * chunk;
                const std::function<void(unsigned)> mv = [&](unsigned t) {
                    const size_t a0 = (len * t / T) & ~size_t(63), z0 = t + 1 == T ? len : (len * (t + 1) / T) & ~size_t(63);
                    if (z0 > a0) std::memcpy(to + a0, from + a0, z0 - a0);
                };
                team.run(mv);
            }
            int rc = RBG_OK;
            if (e != hipSuccess) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); rc = RBG_ENODEV; }
            for (hipEvent_t x : ev)
                if (x) (void)hipEventDestroy(x);
            return rc;
        }
    }
    hipError_t e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e == hipSuccess ? RBG_OK : RBG_ENODEV;
}

// The way in for the big arrays of a load (run lists, samples, phi entries: 5-7 GB each at r = 3e8): worker threads
// copy 64 MB chunks of the pageable source into the pinned buffers while the DMA of the chunks before runs -- the
// driver's own path for pageable memory stages through one thread.  RBG_H2D_STAGED=0: plain hipMemcpy (A/B).
int h2d_big(void *d_dst, const void *h_src, size_t bytes) {
    if (bytes == 0) return RBG_OK;
    static const bool staged = [] { const char *e = std::getenv("RBG_H2D_STAGED"); return !(e && e[0] == '0'); }();
    if (staged && bytes >= (size_t(64) << 20)) {
        PinnedStage &ps = PinnedStage::get();
        std::unique_lock<std::mutex> lk(ps.mu, std::try_to_lock);
        if (lk.owns_lock() && ps.ensure()) {
            const size_t chunk = PinnedStage::kChunk;
            const size_t nb = (bytes + chunk - 1) / chunk;
            const unsigned T = std::max(1u, std::min(16u, rbg_hostpath::cpu_budget()));
            rbg_hostpath::ThreadTeam team(T);
            hipStream_t st = hipStreamPerThread;
            char *dst = static_cast<char *>(d_dst);
            const char *src = static_cast<const char *>(h_src);
            hipError_t e = hipSuccess;
            hipEvent_t ev[PinnedStage::kBufs] = {nullptr, nullptr, nullptr, nullptr};
            for (hipEvent_t &x : ev)
                if (e == hipSuccess) e = hipEventCreateWithFlags(&x, hipEventDisableTiming);
            for (size_t c = 0; c < nb && e == hipSuccess; ++c) {
                const int b = static_cast<int>(c % PinnedStage::kBufs);
                if (c >= static_cast<size_t>(PinnedStage::kBufs)) e = hipEventSynchronize(ev[b]);   // chunk c - kBufs has left this buffer
                if (e != hipSuccess) break;
                const size_t len = std::min(chunk, bytes - c * chunk);
                char *to = static_cast<char *>(ps.buf[b]);
                const char *from = src + c * chunk;
                const std::function<void(unsigned)> mv = [&](unsigned t) {
                    const size_t a0 = (len * t / T) & ~size_t(63), z0 = t + 1 == T ? len : (len * (t + 1) / T) & ~size_t(63);
                    if (z0 > a0) std::memcpy(to + a0, from + a0, z0 - a0);
                };
                team.run(mv);
                e = hipMemcpyAsync(dst + c * chunk, ps.buf[b], len, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipEventRecord(ev[b], st);
            }
            const hipError_t e2 = hipStreamSynchronize(st);
            if (e == hipSuccess) e = e2;
            for (hipEvent_t x : ev)
                if (x) (void)hipEventDestroy(x);
            if (e != hipSuccess) { (void)hipGetLastError(); return RBG_ENODEV; }
            return RBG_OK;
        }
    }
    if (hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return RBG_ENODEV; }
    return RBG_OK;
}

// shared tail of the ragged-output host calls: d_off[N+1] is planned on the device; size, fill, copy back
template <typename FillFn>
int ragged_finish(uint64_t N, DevBuf &d_off, uint64_t *h_off, uint64_t **h_vals, hipStream_t st, FillFn fill) {
    HIP_TRY(hipMemcpyAsync(h_off, d_off.p, (N + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t total = h_off[N];
    *h_vals = static_cast<uint64_t *>(alloc_result(total * 8));
    if (!*h_vals) return RBG_ENOMEM;
    if (total == 0) return RBG_OK;
    DevBuf d_vals;
    int rc = d_vals.alloc(total * 8);
    if (!rc) rc = fill(d_vals.as<uint64_t>());
    if (!rc) rc = d2h_result(*h_vals, d_vals.p, total * 8, st);
    if (rc) { rbg_free_buffer(*h_vals); *h_vals = nullptr; }
    return rc;
}

bool file_readable(const std::string &fname) {
    FILE *f = std::fopen(fname.c_str(), "rb");
    if (!f) return false;
    std::fclose(f);
    return true;
}

// the reference's serialised files -> decoded bundle (load_rowbowt, rowbowt_io.hpp:176-189)
int bundle_from_index_files(const char *prefix, int flags, FlatBundle &b) {
    const std::string pre(prefix);
    int rc = parse_rbwt(pre + ".rbwt", b.rle);  // rowbowt_io.hpp:17,179-182
    if (rc) return rc;
    if (flags & RBG_LOAD_SA) {  // :18,184
        if ((rc = parse_tsa(pre + ".tsa", b.tsa))) return rc;
        b.has_tsa = true;
    }
    if (flags & RBG_LOAD_MA) {  // :19,185
        if ((rc = parse_mab(pre + ".mab", b.ma))) return rc;
        if (!markers_valid(b.ma.start.data(), b.ma.end.data(), b.ma.start.size(), b.ma.off.data())) return RBG_EFORMAT;
        b.has_ma = true;
    }
    if (flags & RBG_LOAD_DL) {  // :20,186
        if ((rc = parse_docs(pre + ".docs", b.dl))) return rc;
        b.has_dl = true;
    }
    return RBG_OK;
}

// rb_build's raw inputs (rb_build.cpp:83-93) -> decoded bundle
int bundle_from_raw_files(const char *bwt_fname, const char *ssa_fname, const char *esa_fname, FlatBundle &b) {
    RawRle &rle = b.rle;
    int rc = read_raw_bwt(bwt_fname, rle);
    if (rc) return rc;
    if (ssa_fname) {
        std::vector<uint64_t> ssa, esa;
        if ((rc = read_raw_samples(ssa_fname, ssa)) || (rc = read_raw_samples(esa_fname, esa))) return rc;
        if (ssa.size() != rle.R || esa.size() != rle.R) return RBG_EFORMAT;  // one sample pair per BWT run
        for (uint64_t i = 0; i < rle.R; ++i)
            if (ssa[i] > rle.n || esa[i] > rle.n) return RBG_EFORMAT;
        tsa_from_samples(rle.n, rle.R, ssa.data(), esa.data(), b.tsa);
        for (uint64_t j = 1; j < rle.R; ++j)
            if (b.tsa.pred_pos[j] == b.tsa.pred_pos[j - 1]) return RBG_EFORMAT;  // run-start samples must be distinct
        b.has_tsa = true;
    }
    return RBG_OK;
}

// RowBowt::build_ftab(k) + FTab::serialize (rowbowt.hpp:726-744, ftab.hpp:29-34): one text line
// "<kmer> <lo> <hi>" for every k-mer over ACGT with a non-empty range, in std::map (lexicographic)
// order, produced chunk by chunk; sink(text) returns false to stop early
template <typename Sink>
int ftab_stream(rbg_index *ix, uint64_t k, Sink sink) {
    const uint64_t total = uint64_t(1) << (2 * k);
    const uint64_t chunk = std::min<uint64_t>(total, uint64_t(1) << 21);
    std::vector<uint8_t> seqs(chunk * k);
    std::vector<uint64_t> off(chunk + 1), lo(chunk), hi(chunk);
    for (uint64_t i = 0; i <= chunk; ++i) off[i] = i * k;
    std::string text;
    for (uint64_t base = 0; base < total; base += chunk) {
        for (uint64_t i = 0; i < chunk; ++i) {
            const uint64_t L = base + i;  // lexicographic rank: first character most significant
            for (uint64_t j = 0; j < k; ++j) seqs[i * k + j] = static_cast<uint8_t>("ACGT"[(L >> (2 * (k - 1 - j))) & 3]);
        }
        const int rc = rbg_find_range(ix, seqs.data(), off.data(), chunk, lo.data(), hi.data());
        if (rc) return rc;
        text.clear();
        for (uint64_t i = 0; i < chunk; ++i) {
            if (lo[i] > hi[i]) continue;  // rowbowt.hpp:737
            text.append(reinterpret_cast<const char *>(&seqs[i * k]), k);
            text += ' ';
            text += std::to_string(lo[i]);
            text += ' ';
            text += std::to_string(hi[i]);
            text += '\n';
        }
        if (!sink(text)) break;
    }
    return RBG_OK;
}

int index_from_bundle(FlatBundle &b, int device, rbg_index **out) {
    rbg_index *ix = new (std::nothrow) rbg_index();
    if (!ix) return RBG_ENOMEM;
    FlattenOptions fo = options_for(device, b.rle, &ix->kmer_steps_requested, &ix->auto_runs, ix);
    fo.defer_kmer = compose_deferred(device);
    int rc = flatten(b.rle, b.has_tsa ? &b.tsa : nullptr, fo, ix->host);
    if (rc) { delete ix; return rc; }
    // (the flat index holds everything the bundle held: 33 bytes per run given back before the upload's own scratch is made)
    { RawRle().heads.swap(b.rle.heads); std::vector<uint64_t>().swap(b.rle.lens); RawTsa empty; std::swap(b.tsa, empty); }
    if (b.has_ma) { ix->H().ma = std::move(b.ma); ix->H().has_ma = true; }
    if (b.has_dl) { ix->H().dl = std::move(b.dl); ix->H().has_dl = true; }
    return finish(ix, device, out);
}

// No exception leaves the C ABI: a corrupt file that makes a reader allocate absurdly, or plain memory
// exhaustion on the host, comes back as an error code (the callers are C, cgo-style bindings, ctypes).
template <typename F>
int guarded(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return RBG_ENOMEM;
    } catch (const std::length_error &) {
        return RBG_ENOMEM;
    } catch (...) {
        return RBG_EFORMAT;
    }
}

}  // namespace

namespace {
struct Reloc {
    std::vector<DevAlloc> from, to;
    const void *operator()(const void *p) const {
        if (!p) return nullptr;
        const char *c = static_cast<const char *>(p);
        for (size_t i = 0; i < from.size(); ++i) {
            const char *b = static_cast<const char *>(from[i].p);
            if (c >= b && c < b + from[i].bytes) return static_cast<const char *>(to[i].p) + (c - b);
        }
        return nullptr;  // not a pointer into the replica
    }
    template <typename T> void fix(T *&p) const { p = static_cast<T *>(const_cast<void *>((*this)(p))); }
};
}  // namespace

extern "C" {

int rbg_abi_version(void) { return RBG_ABI_VERSION; }

const char *rbg_strerror(int code) {
    switch (code) {
        case RBG_OK: return "ok";
        case RBG_EIO: return "file missing or unreadable";
        case RBG_EFORMAT: return "not the sdsl layout written by the reference";
        case RBG_ENODEV: return "no usable gfx950 device / HIP error (there is no CPU path)";
        case RBG_EARG: return "bad argument";
        case RBG_ENOMEM: return "out of memory";
        case RBG_ENOTLOADED: return "required structure (toehold SA / markers / docs) not loaded";
        default: return "unknown error";
    }
}

int rbg_set_default_option(int opt, int64_t value) {
    return guarded([&]() -> int {
    switch (opt) {
        case RBG_OPT_BLOCK_THREADS:
            if (value < 64 || value > 256 || value % 64) return RBG_EARG;  // kernels are built for <= 4 waves per workgroup
            g_opt_block_threads = value; return RBG_OK;
        case RBG_OPT_RANK_BUCKET_SHIFT:
            if (value < -1 || value > 12) return RBG_EARG;  // 8-bit slot offsets up to 8, the wide encoding up to 12
            g_opt_rank_shift = value; return RBG_OK;
        case RBG_OPT_DEEP_BUCKET_SHIFT:
            if (value < -1 || value > 12) return RBG_EARG;
            g_opt_deep_shift = value; return RBG_OK;
        case RBG_OPT_DENSE_OVERFLOW:
            if (value != 0 && value != 1) return RBG_EARG;
            g_opt_dense_overflow = value; return RBG_OK;
        case RBG_OPT_PHI_BUCKET_SHIFT:
            if (value < -1 || value > 8) return RBG_EARG;
            g_opt_phi_shift = value; return RBG_OK;
        case RBG_OPT_POS_BYTES:
            if (value != 0 && value != 4 && value != 8) return RBG_EARG;
            g_opt_pos_bytes = value; return RBG_OK;
        case RBG_OPT_FTAB_K:
            if (value < -1 || value > 16) return RBG_EARG;
            g_opt_ftab_k = value; return RBG_OK;
        case RBG_OPT_HBM_BUDGET_MB:
            if (value < 0) return RBG_EARG;
            g_opt_hbm_budget_mb = value; return RBG_OK;
        case RBG_OPT_KMER_STEPS:
            if (value < 1 || value > kMaxKmerDepth) return RBG_EARG;
            g_opt_kmer_steps = value; return RBG_OK;
        case RBG_OPT_PACKED_READS:
            if (value < 0 || value > 2) return RBG_EARG;
            g_opt_packed_reads = value; return RBG_OK;
        case RBG_OPT_RANK_LAYOUT:
            if (value != RBG_LAYOUT_AUTO && value != RBG_LAYOUT_SLOTS && value != RBG_LAYOUT_RUNS && value != RBG_LAYOUT_PREFER_SLOTS) return RBG_EARG;
            g_opt_rank_layout = value; return RBG_OK;
        case RBG_OPT_RUN_DEPTHS:
            if (value < 0 || value >= (1 << kMaxRunDepth)) return RBG_EARG;
            g_opt_run_depths = value; return RBG_OK;
        case RBG_OPT_RUN_PHI:
            if (value < 0 || value > 2) return RBG_EARG;
            g_opt_run_phi = value; return RBG_OK;
        case RBG_OPT_RUN_REC:
            if (value < 0 || value > 2) return RBG_EARG;
            g_opt_run_rec = value; return RBG_OK;
        case RBG_OPT_RUN_REC_DEPTHS:
            if (value < 0 || value >= (1 << kMaxRunDepth)) return RBG_EARG;
            g_opt_run_rec_depths = value; return RBG_OK;
        default: return RBG_EARG;
    }
    });
}

int rbg_get_default_option(int opt, int64_t *value) {
    return guarded([&]() -> int {
    if (!value) return RBG_EARG;
    switch (opt) {
        case RBG_OPT_BLOCK_THREADS: *value = g_opt_block_threads.load(); return RBG_OK;
        case RBG_OPT_RANK_BUCKET_SHIFT: *value = g_opt_rank_shift.load(); return RBG_OK;
        case RBG_OPT_DEEP_BUCKET_SHIFT: *value = g_opt_deep_shift.load(); return RBG_OK;
        case RBG_OPT_DENSE_OVERFLOW: *value = g_opt_dense_overflow.load(); return RBG_OK;
        case RBG_OPT_PHI_BUCKET_SHIFT: *value = g_opt_phi_shift.load(); return RBG_OK;
        case RBG_OPT_POS_BYTES: *value = g_opt_pos_bytes.load(); return RBG_OK;
        case RBG_OPT_FTAB_K: *value = g_opt_ftab_k.load(); return RBG_OK;
        case RBG_OPT_HBM_BUDGET_MB: *value = g_opt_hbm_budget_mb.load(); return RBG_OK;
        case RBG_OPT_KMER_STEPS: *value = g_opt_kmer_steps.load(); return RBG_OK;
        case RBG_OPT_PACKED_READS: *value = g_opt_packed_reads.load(); return RBG_OK;
        case RBG_OPT_RANK_LAYOUT: *value = g_opt_rank_layout.load(); return RBG_OK;
        case RBG_OPT_RUN_DEPTHS: *value = g_opt_run_depths.load(); return RBG_OK;
        case RBG_OPT_RUN_PHI: *value = g_opt_run_phi.load(); return RBG_OK;
        case RBG_OPT_RUN_REC: *value = g_opt_run_rec.load(); return RBG_OK;
        case RBG_OPT_RUN_REC_DEPTHS: *value = g_opt_run_rec_depths.load(); return RBG_OK;
        default: return RBG_EARG;
    }
    });
}

int rbg_load(const char *prefix, int flags, int device, rbg_index **out) {
    return guarded([&]() -> int {
    if (!prefix || !out) return RBG_EARG;
    *out = nullptr;
    FlatBundle b;
    int rc = bundle_from_index_files(prefix, flags, b);
    // no .rbwt but a native cache next to where it would be: use that (rb_build of this engine writes it);
    // a requested part the cache does not hold is still looked for in its own file (.docs is plain text
    // that rb_build copies, rowbowt_io.hpp:73-80)
    if (rc == RBG_EIO && !file_readable(std::string(prefix) + ".rbwt") && file_readable(std::string(prefix) + ".rbgpu")) {
        const std::string pre(prefix);
        FlatBundle c;
        if ((rc = read_flat(pre + ".rbgpu", c))) return rc;
        if ((flags & RBG_LOAD_SA) && !c.has_tsa) {
            if ((rc = parse_tsa(pre + ".tsa", c.tsa))) return rc;
            if (c.tsa.r != c.rle.R || c.tsa.n != c.rle.n) return RBG_EFORMAT;
            c.has_tsa = true;
        }
        if ((flags & RBG_LOAD_MA) && !c.has_ma) {
            if ((rc = parse_mab(pre + ".mab", c.ma))) return rc;
            if (!markers_valid(c.ma.start.data(), c.ma.end.data(), c.ma.start.size(), c.ma.off.data())) return RBG_EFORMAT;
            c.has_ma = true;
        }
        if ((flags & RBG_LOAD_DL) && !c.has_dl) {
            if ((rc = parse_docs(pre + ".docs", c.dl))) return rc;
            c.has_dl = true;
        }
        c.has_tsa = c.has_tsa && (flags & RBG_LOAD_SA);
        c.has_ma = c.has_ma && (flags & RBG_LOAD_MA);
        c.has_dl = c.has_dl && (flags & RBG_LOAD_DL);
        return index_from_bundle(c, device, out);
    }
    if (rc) return rc;
    return index_from_bundle(b, device, out);
    });
}

int rbg_load_cache(const char *path, int flags, int device, rbg_index **out) {
    return guarded([&]() -> int {
    if (!path || !out) return RBG_EARG;
    *out = nullptr;
    FlatBundle b;
    int rc = read_flat(path, b);
    if (rc) return rc;
    // a part the caller asks for must be in the file, like a missing .tsa/.mab/.docs (rowbowt_io.hpp:166-169)
    if (((flags & RBG_LOAD_SA) && !b.has_tsa) || ((flags & RBG_LOAD_MA) && !b.has_ma) || ((flags & RBG_LOAD_DL) && !b.has_dl))
        return RBG_EIO;
    b.has_tsa = b.has_tsa && (flags & RBG_LOAD_SA);
    b.has_ma = b.has_ma && (flags & RBG_LOAD_MA);
    b.has_dl = b.has_dl && (flags & RBG_LOAD_DL);
    return index_from_bundle(b, device, out);
    });
}

int rbg_convert_index(const char *prefix, int flags, const char *out_path) {
    return guarded([&]() -> int {
    if (!prefix || !out_path) return RBG_EARG;
    FlatBundle b;
    int rc = bundle_from_index_files(prefix, flags, b);
    if (rc) return rc;
    return write_flat(out_path, b);
    });
}

int rbg_convert_raw(const char *bwt_fname, const char *ssa_fname, const char *esa_fname, const char *mab_fname,
                    const char *docs_fname, const char *out_path) {
    return guarded([&]() -> int {
    if (!bwt_fname || !out_path || (!!ssa_fname != !!esa_fname)) return RBG_EARG;
    FlatBundle b;
    int rc = bundle_from_raw_files(bwt_fname, ssa_fname, esa_fname, b);
    if (rc) return rc;
    if (mab_fname) {
        if ((rc = parse_mab(mab_fname, b.ma))) return rc;
        if (!markers_valid(b.ma.start.data(), b.ma.end.data(), b.ma.start.size(), b.ma.off.data())) return RBG_EFORMAT;
        b.has_ma = true;
    }
    if (docs_fname) {
        if ((rc = parse_docs(docs_fname, b.dl))) return rc;
        b.has_dl = true;
    }
    return write_flat(out_path, b);
    });
}

// a run-length BWT (+ both samples of every run) in memory -> the native cache file (what rbg_convert_raw writes for
// the same index from its .bwt/.ssa/.esa files): for builders that never materialise the BWT as text -- n = 5e10 would be
// a 50 GB .bwt -- and for handing one index to several processes of a node (bench.py: rank 0 writes, every rank loads)
static int runs_to_bundle(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y, const uint64_t *esa_y, FlatBundle &b);

int rbg_convert_runs(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y, const uint64_t *esa_y, const char *out_path) {
    return rbg_convert_runs_markers(heads, lens, R, ssa_y, esa_y, nullptr, nullptr, 0, nullptr, nullptr, nullptr, out_path);
}

int rbg_convert_runs_markers(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y, const uint64_t *esa_y,
                             const uint64_t *mk_start, const uint64_t *mk_end, uint64_t mk_nruns, const uint64_t *mk_off, const uint64_t *mk_vals,
                             const char *docs_text, const char *out_path) {
    return guarded([&]() -> int {
    if (!heads || !lens || !out_path || R == 0 || (!!ssa_y != !!esa_y)) return RBG_EARG;
    if (mk_nruns && (!mk_start || !mk_end || !mk_off || !mk_vals)) return RBG_EARG;
    FlatBundle b;
    int rc = runs_to_bundle(heads, lens, R, ssa_y, esa_y, b);
    if (rc) return rc;
    if (mk_nruns) {
        if (!markers_valid(mk_start, mk_end, mk_nruns, mk_off)) return RBG_EARG;
        if (mk_end[mk_nruns - 1] >= b.rle.n) return RBG_EARG;
        b.ma.start.assign(mk_start, mk_start + mk_nruns);
        b.ma.end.assign(mk_end, mk_end + mk_nruns);
        b.ma.off.assign(mk_off, mk_off + mk_nruns + 1);
        b.ma.vals.assign(mk_vals, mk_vals + mk_off[mk_nruns]);
        b.has_ma = true;
    }
    if (docs_text) {   // the text of a .docs file (doclist.hpp:57-73: whitespace-separated name / start pairs)
        std::istringstream ss{std::string(docs_text)};
        std::string name;
        uint64_t pos = 0;
        while (ss >> name >> pos) { b.dl.names.push_back(name); b.dl.starts.push_back(pos); }
        if (b.dl.names.empty()) return RBG_EARG;
        b.dl.sorted = b.dl.starts;
        std::sort(b.dl.sorted.begin(), b.dl.sorted.end());
        b.has_dl = true;
    }
    return write_flat(out_path, b);
    });
}

int rbg_build_from_runs(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y,
                        const uint64_t *esa_y, int device, rbg_index **out) {
    return guarded([&]() -> int {
    if (!heads || !lens || !out || R == 0 || (!!ssa_y != !!esa_y)) return RBG_EARG;
    *out = nullptr;
    FlatBundle b;
    const int rc0 = runs_to_bundle(heads, lens, R, ssa_y, esa_y, b);
    if (rc0) return rc0;
    return index_from_bundle(b, device, out);
    });
}

static int runs_to_bundle(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y, const uint64_t *esa_y, FlatBundle &bundle) {
    {
    RawRle &rle = bundle.rle;
    RawTsa &tsa = bundle.tsa;
    bundle.has_tsa = ssa_y != nullptr;
    rle.R = R;
    rle.B = 2;
    rle.heads.resize(R);
    rle.lens.resize(R);
    // (every loop over the runs is split over the worker threads: 3e8 runs at pangenome scale)
    const unsigned T = load_threads();
    std::vector<uint64_t> part(T + 1, 0);
    std::vector<int> bad(T + 1, 0);
    parallel_for(R, [&](uint64_t b, uint64_t e, unsigned t) {
        uint64_t sum = 0;
        for (uint64_t i = b; i < e; ++i) {
            if (lens[i] == 0 || (i && heads[i] == heads[i - 1])) bad[t] = 1;  // runs are non-empty and maximal
            sum += lens[i];
            rle.heads[i] = heads[i];
            rle.lens[i] = lens[i];
        }
        part[t] = sum;
    });
    uint64_t n = 0;
    for (unsigned t = 0; t <= T; ++t) { if (bad[t]) return RBG_EARG; n += part[t]; }
    rle.n = n;
    if (ssa_y) {
        parallel_for(R, [&](uint64_t b, uint64_t e, unsigned t) {
            for (uint64_t i = b; i < e; ++i)
                if (ssa_y[i] > n || esa_y[i] > n) bad[t] = 1;  // SA values of an n-symbol text
        });
        for (unsigned t = 0; t <= T; ++t) if (bad[t]) return RBG_EARG;
        tsa_from_samples(n, R, ssa_y, esa_y, tsa);
        parallel_for(R, [&](uint64_t b, uint64_t e, unsigned t) {
            for (uint64_t j = std::max<uint64_t>(b, 1); j < e; ++j)
                if (tsa.pred_pos[j] == tsa.pred_pos[j - 1]) bad[t] = 1;  // run-start samples must be distinct
        });
        for (unsigned t = 0; t <= T; ++t) if (bad[t]) return RBG_EARG;
    }
    return RBG_OK;
    }
}

int rbg_build_from_files(const char *bwt_fname, const char *ssa_fname, const char *esa_fname, int device, rbg_index **out) {
    return guarded([&]() -> int {
    if (!bwt_fname || !out || (!!ssa_fname != !!esa_fname)) return RBG_EARG;
    *out = nullptr;
    FlatBundle b;
    int rc = bundle_from_raw_files(bwt_fname, ssa_fname, esa_fname, b);
    if (rc) return rc;
    return index_from_bundle(b, device, out);
    });
}

int rbg_write_ftab(rbg_index *ix, uint64_t k, const char *path) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!path || k == 0 || k > 16) return RBG_EARG;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return RBG_EIO;
    bool io_ok = true;
    int rc = ftab_stream(ix, k, [&](const std::string &t) {
        if (!t.empty() && std::fwrite(t.data(), 1, t.size(), fp) != t.size()) io_ok = false;
        return io_ok;
    });
    if (std::fclose(fp) != 0) io_ok = false;
    if (!rc && !io_ok) rc = RBG_EIO;
    return rc;
    });
}

// FTab::load (ftab.hpp:15-27) keeps k = length of the last line's k-mer.  The file is accepted only if
// it is, byte for byte, the table build_ftab(k) gives for this index: then search_ftab(q) is
// "find_range(q) when q is over ACGT and occurs", which is how the ftab variants are computed here.
int rbg_check_ftab(rbg_index *ix, const char *path, uint64_t *k_out) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!path || !k_out) return RBG_EARG;
    *k_out = 0;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return RBG_EIO;
    char first[64];
    uint64_t k = 0;
    if (std::fgets(first, sizeof first, fp)) while (k < sizeof first && first[k] && first[k] != ' ' && first[k] != '\n') ++k;
    if (k == 0 || k > 16) { std::fclose(fp); return RBG_EFORMAT; }
    std::rewind(fp);
    bool same = true;
    std::vector<char> buf;
    int rc = ftab_stream(ix, k, [&](const std::string &t) {
        buf.resize(t.size());
        if (!t.empty() && (std::fread(buf.data(), 1, t.size(), fp) != t.size() || std::memcmp(buf.data(), t.data(), t.size()) != 0)) same = false;
        return same;
    });
    if (!rc && same && std::fgetc(fp) != EOF) same = false;  // nothing may follow
    std::fclose(fp);
    if (rc) return rc;
    if (!same) return RBG_EFORMAT;
    *k_out = k;
    return RBG_OK;
    });
}

int rbg_set_markers(rbg_index *ix, const uint64_t *run_start, const uint64_t *run_end, uint64_t nruns,
                    const uint64_t *mk_off, const uint64_t *mk_vals) {
    return guarded([&]() -> int {
    if (!ix || !run_start || !run_end || !mk_off || (!mk_vals && mk_off[nruns])) return RBG_EARG;
    if (!markers_valid(run_start, run_end, nruns, mk_off)) return RBG_EARG;
    if (ix->primary) return RBG_EARG;    // attach to the primary, before rbg_replicate
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->H().has_ma) return RBG_EARG;  // immutable once attached
    RawMarkers &m = ix->H().ma;
    m.start.assign(run_start, run_start + nruns);
    m.end.assign(run_end, run_end + nruns);
    m.off.assign(mk_off, mk_off + nruns + 1);
    m.vals.assign(mk_vals, mk_vals + mk_off[nruns]);
    ix->H().has_ma = true;
    if (ix->device != RBG_DEVICE_NONE) {
        DeviceScope scope(ix->device);
        if (scope.rc) return scope.rc;
        return upload_markers(ix);
    }
    return RBG_OK;
    });
}

}  // extern "C"
namespace {
// the document starts on the device for K3's locus order (rbg_dev.h order_docs).  RBG_LOCATE_ORDER=abs keeps the order by absolute position.
int upload_order_docs(rbg_index *ix) {
    DevIndex &d = ix->dev;
    d.order_docs = nullptr;
    d.order_ndocs = d.order_dbits = d.order_lowbits = d.order_obits = 0;
    if (ix->device == RBG_DEVICE_NONE || ix->primary) return RBG_OK;
    // RBG_LOCATE_ORDER = abs: never; locus: whenever there are two documents or more; unset: from 128 documents on -- measured (profiles/
    // r06_locus_order_ab.txt): K3 53.7 -> 50.4 ms per 10 M x 150 bp reads at r = 1.07e9 / 520 haplotypes, equal at 200 haplotypes, and on the bench
    // index (50 haplotypes) the key pass and the wider sort cost more (0.17 ms) than K3 gains (0.02 ms).
    const char *e = std::getenv("RBG_LOCATE_ORDER");
    if (e && std::strcmp(e, "abs") == 0) return RBG_OK;
    const bool forced = e && std::strcmp(e, "locus") == 0;
    const RawDocs &dl = ix->H().dl;
    const uint64_t nd = dl.sorted.size();
    if (!ix->H().has_dl || nd < (forced ? 2u : 128u) || nd > (uint64_t(1) << 20) || dl.sorted[0] != 0) return RBG_OK;   // (one document: the absolute order already is the locus order)
    uint64_t longest = 0;
    for (uint64_t j = 0; j < nd; ++j) longest = std::max(longest, (j + 1 < nd ? dl.sorted[j + 1] : ix->H().n) - dl.sorted[j]);
    uint32_t obits = 1, dbits = 1;
    while (obits < 64 && (longest >> obits)) ++obits;
    while ((nd >> dbits)) ++dbits;
    if (obits + dbits > 60) return RBG_OK;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    const void *p = nullptr;
    int rc = dev_upload(ix, dl.sorted.data(), nd * 8, &p);
    if (rc) return rc;
    d.order_docs = static_cast<const uint64_t *>(p);
    d.order_ndocs = static_cast<uint32_t>(nd);
    d.order_dbits = dbits;
    d.order_obits = obits;
    // positions inside one line of phi slots need no sorting (as in the absolute order: launch_locate_order's begin_bit)
    d.order_lowbits = std::min<uint32_t>(d.phi_shift + 2u, obits > 8 ? obits - 8 : 0u);
    return RBG_OK;
}
}  // namespace
extern "C" {

int rbg_set_docs(rbg_index *ix, const char *names_joined, const uint64_t *starts, uint64_t ndocs) {
    return guarded([&]() -> int {
    if (!ix || !names_joined || !starts || ix->primary) return RBG_EARG;
    std::lock_guard<std::mutex> g(ix->mu);
    RawDocs &d = ix->H().dl;
    d = RawDocs();
    const char *p = names_joined;
    for (uint64_t i = 0; i < ndocs; ++i) {
        d.names.emplace_back(p);
        p += d.names.back().size() + 1;
        d.starts.push_back(starts[i]);
    }
    d.sorted = d.starts;
    std::sort(d.sorted.begin(), d.sorted.end());
    ix->H().has_dl = true;
    {   // rbg_align_text keeps a device copy of the table: made again at its next call (the old arrays stay until rbg_free)
        std::lock_guard<std::mutex> g2(ix->text_mu);
        ix->text_docs = rbg_index::TextDocs();
    }
    return upload_order_docs(ix);   // (K3's chain order by locus: attach the documents BEFORE replicating, like the markers)
    });
}

void rbg_free(rbg_index *ix) {
    if (!ix) return;
    if (ix->device != RBG_DEVICE_NONE) {
        DeviceScope scope(ix->device);
        ix->ws_free.clear();  // pinned + device staging of the host-pointer calls
        for (auto &t : ix->text_out) {
            if (t.pending) (void)hipEventSynchronize(t.done);
            if (t.d_text) DevPool::get().release(t.d_text, t.d_cls, t.d_dev);
            if (t.done) (void)hipEventDestroy(t.done);
            (void)hipHostFree(t.p);
        }
        if (ix->text_copy_stream) (void)hipStreamDestroy(ix->text_copy_stream);
        for (auto &t : ix->text_in) (void)hipHostFree(t.p);
        for (const DevAlloc &a : ix->allocs) (void)hipFree(a.p);
        DevPool::get().trim(ix->device);  // cached scratch blocks of the host-pointer calls
    }
    delete ix;
}

void rbg_free_buffer(void *p) {
    if (!p) return;
    ResultPool &P = ResultPool::get();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const size_t size = it->second;
            P.live.erase(it);
            if (P.on && P.cached + size <= ResultPool::kMax) {
                P.idle.emplace(size, p);
                P.cached += size;
                return;
            }
        }
    }
    std::free(p);
}

// rbg_info for a caller compiled against ANOTHER layout of rbg_info_t (the struct grew with ABI 3 and rbg_info cannot know what its caller
// allocated): fills min(out_bytes, sizeof(rbg_info_t)) bytes -- fields are only ever added at the end from ABI 3 on.
int rbg_info_sized(const rbg_index *ix, rbg_info_t *out, uint64_t out_bytes) {
    if (!ix || !out || out_bytes < 8) return RBG_EARG;
    rbg_info_t v;
    const int rc = rbg_info(ix, &v);
    if (rc) return rc;
    std::memcpy(out, &v, static_cast<size_t>(std::min<uint64_t>(out_bytes, sizeof(v))));
    return RBG_OK;
}

int rbg_info(const rbg_index *ix, rbg_info_t *out) {
    return guarded([&]() -> int {
    if (!ix || !out) return RBG_EARG;
    std::memset(out, 0, sizeof(*out));
    out->n = ix->H().n;
    out->r = ix->H().r;
    out->sigma = ix->H().sigma;
    out->pos_bytes = ix->H().pos_bytes;
    out->device = ix->device;
    out->has_tsa = ix->H().has_tsa;
    out->has_markers = ix->H().has_ma;
    out->has_docs = ix->H().has_dl;
    out->hbm_bytes = ix->hbm_bytes;
    out->marker_runs = ix->H().ma.start.size();
    out->marker_vals = ix->H().ma.vals.size();
    out->rank_bucket_shift = ix->H().sym.empty() ? 0 : ix->H().sym.back().shift;
    out->phi_bucket_shift = ix->H().phi_shift;
    out->rank_slots = ix->rank_slots;
    out->rank_slots_overflow = ix->rank_slots_overflow;
    out->phi_slots = ix->phi_slots;
    out->phi_slots_overflow = ix->phi_slots_overflow;
    out->kmer_steps = ix->H().kmer_levels();
    if (ix->device != RBG_DEVICE_NONE && ix->dev.layout == RBG_LAYOUT_RUNS) out->kmer_steps = ix->dev.run_ksteps;   // depths of the run-indexed search
    out->kmer_symbols = ix->H().kmer(2).empty() ? 0 : ix->H().nmajor;
    out->slot_bytes = ix->device != RBG_DEVICE_NONE && ix->dev.layout == RBG_LAYOUT_SLOTS ? 16 : 0;
    out->ftab_k = ix->dev.ftab_k;
    out->kmer_steps_requested = ix->kmer_steps_requested ? ix->kmer_steps_requested : out->kmer_steps;
    out->hbm_free_at_load = ix->hbm_free_at_load;
    out->hbm_budget = ix->hbm_budget;
    out->rank_layout = ix->runs_layout ? RBG_LAYOUT_RUNS : RBG_LAYOUT_SLOTS;
    out->replicas = ix->device == RBG_DEVICE_NONE ? 0 : 1;
    out->depth_runs[0] = ix->H().r;
    const bool runs_dev = ix->device != RBG_DEVICE_NONE && ix->dev.layout == RBG_LAYOUT_RUNS;
    for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d) {
        if (runs_dev && !(ix->dev.run_depth_mask >> (d - 1) & 1u)) continue;   // depths left without run lists (RBG_OPT_RUN_DEPTHS) report none
        for (const SymTable &t : ix->H().kmer(d)) out->depth_runs[d - 1] += t.nruns;
    }
    out->pair_runs = out->depth_runs[1]; out->triple_runs = out->depth_runs[2]; out->quad_runs = out->depth_runs[3]; out->quint_runs = out->depth_runs[4];
    return RBG_OK;
    });
}

int rbg_layout_info(const rbg_index *ix, rbg_layout_info_t *out, uint64_t out_bytes) {
    return guarded([&]() -> int {
    if (!ix || !out || out_bytes < 8) return RBG_EARG;
    // (ABI 3 changed this struct in the middle -- five-entry arrays became eight -- so a size that is neither this layout's nor a prefix of it at
    //  an 8-byte boundary beyond the ABI-3 head is a caller compiled against ABI 2: refused rather than filled with shifted fields)
    if (out_bytes < offsetof(rbg_layout_info_t, budget_raised) && out_bytes != offsetof(rbg_layout_info_t, rec_bytes) && out_bytes != offsetof(rbg_layout_info_t, entries))
        return RBG_EARG;
    rbg_layout_info_t v;
    std::memset(&v, 0, sizeof(v));
    if (ix->device != RBG_DEVICE_NONE && ix->dev.layout == RBG_LAYOUT_RUNS) {
        const rbg_index::RunsReport &r = ix->runs_report;
        v.run_fmt = r.fmt;
        v.depths_composed = r.depths_composed;
        v.depth_mask_asked = r.depth_mask_asked;
        v.depth_mask_kept = r.depth_mask_kept;
        v.depths_dropped_budget = r.depths_dropped_budget;
        v.rank_directories = r.rank_dirs;
        v.phi_directory = r.phi_dir;
        v.fill_shift = r.fmt == 2 && ix->H().pos_bytes == 8 ? ix->dev.run_fill_shift : 0;
        for (int d = 0; d < kMaxRunDepth; ++d) { v.entries[d] = r.entries[d]; v.fillers[d] = r.fillers[d]; v.dir_bytes[d] = r.dir_bytes[d]; }
        v.phi_entries = r.phi_entries; v.phi_fillers = r.phi_fillers; v.phi_dir_bytes = r.phi_dir_bytes; v.phi_dir_shift = r.phi_dir_shift;
        v.phi_slots = r.phi_slots; v.phi_slot_bytes = r.phi_slot_bytes;
        for (int d = 0; d < kMaxRunDepth; ++d) { v.rec_bytes[d] = r.rec_bytes[d]; v.rec_overflow[d] = r.rec_overflow[d]; }
        v.budget_raised = ix->budget_raised ? 1 : 0;
    }
    std::memcpy(out, &v, static_cast<size_t>(std::min<uint64_t>(out_bytes, sizeof(v))));
    return RBG_OK;
    });
}

int rbg_get_f(const rbg_index *ix, uint64_t f_out[256]) {
    return guarded([&]() -> int {
    if (!ix || !f_out) return RBG_EARG;
    std::memcpy(f_out, ix->H().f, 256 * sizeof(uint64_t));
    return RBG_OK;
    });
}

int rbg_last_run_sample(const rbg_index *ix, uint64_t *out) {
    return guarded([&]() -> int {
    if (!ix || !out) return RBG_EARG;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    *out = ix->H().last_run_sample;
    return RBG_OK;
    });
}

int rbg_host_array(const rbg_index *ix, int which, uint64_t *dst, uint64_t cap, uint64_t *count) {
    return guarded([&]() -> int {
    if (!ix || !count) return RBG_EARG;
    const HostIndex &h = ix->H();
    const std::vector<uint64_t> *v = nullptr;
    std::vector<uint64_t> tmp;
    switch (which) {
        case RBG_ARR_RUN_HEADS: tmp.assign(h.run_heads.begin(), h.run_heads.end()); v = &tmp; break;
        case RBG_ARR_RUN_START: v = &h.run_start; break;
        case RBG_ARR_SAMPLES_LAST: v = &h.samples_last; break;
        case RBG_ARR_PRED_POS: v = &h.pred_pos; break;
        case RBG_ARR_PHI_BASE: v = &h.phi_base; break;
        case RBG_ARR_MARKER_START: v = &h.ma.start; break;
        case RBG_ARR_MARKER_END: v = &h.ma.end; break;
        case RBG_ARR_MARKER_OFF: v = &h.ma.off; break;
        case RBG_ARR_MARKER_VALS: v = &h.ma.vals; break;
        default: return RBG_EARG;
    }
    *count = v->size();
    if (dst) std::memcpy(dst, v->data(), std::min<uint64_t>(cap, v->size()) * 8);
    return RBG_OK;
    });
}

int rbg_resolve_offset(const rbg_index *ix, uint64_t i, const char **name, uint64_t *offset) {
    return guarded([&]() -> int {
    if (!ix || !name || !offset) return RBG_EARG;
    if (!ix->H().has_dl || ix->H().dl.names.empty()) return RBG_ENOTLOADED;
    const RawDocs &d = ix->H().dl;
    // DocList::doc_bounds_rank, doclist.hpp:77-79: rank(min(i+1, size)) over a bit-vector whose
    // size is the LAST start read + 1 (doclist.hpp:66)
    const uint64_t size = d.starts.back() + 1;
    const uint64_t q = i + 1 > size ? size : i + 1;
    const uint64_t rank = std::lower_bound(d.sorted.begin(), d.sorted.end(), q) - d.sorted.begin();
    if (rank == 0) return RBG_EARG;  // reference indexes doc_names_[-1] here
    *offset = i - d.sorted[rank - 1];           // doclist.hpp:48
    *name = d.names[rank - 1].c_str();          // doclist.hpp:49
    return RBG_OK;
    });
}

// The table rbg_resolve_offset answers from, for callers that resolve millions of positions (rb_align -s prints some
// forty per read): a call per position through the ABI was a third of that tool's formatting time.
int rbg_doc_table(rbg_index *ix, uint64_t *ndocs, const uint64_t **sorted_starts, const char *const **names, uint64_t *size) {
    return guarded([&]() -> int {
    if (!ix || !ndocs || !sorted_starts || !names || !size) return RBG_EARG;
    rbg_index *root = ix->primary ? ix->primary : ix;
    if (!root->H().has_dl || root->H().dl.names.empty()) return RBG_ENOTLOADED;
    const RawDocs &d = root->H().dl;
    {
        std::lock_guard<std::mutex> g(root->mu);
        if (root->doc_name_ptrs.size() != d.names.size()) {
            root->doc_name_ptrs.clear();
            for (const std::string &n : d.names) root->doc_name_ptrs.push_back(n.c_str());
        }
    }
    *ndocs = d.names.size();
    *sorted_starts = d.sorted.data();
    *names = root->doc_name_ptrs.data();
    *size = d.starts.back() + 1;
    return RBG_OK;
    });
}

}  // extern "C"
