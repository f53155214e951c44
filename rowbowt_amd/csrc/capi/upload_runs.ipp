// capi/upload_runs.ipp -- the run-indexed layout: its tables on the device (run lists, fillers, folded F, directories / bucket records, phi), and the
// composition of the k-mer depths on the device.  Part of rbg_capi.hip.
namespace {
// ---- the run-indexed layout (rbg_dev.h DevRunTab2; kernels: rbg_runs2_device.hpp) ---------------------------------------
// Inputs: the depth-1 tables of the host index and the k-mer levels composed on the device.
// Everything but the conversion of the depth-1 lists happens in kernels (k_build.hip): fillers (8-byte positions, only
// where a table has a gap of 2^30 rows or more), the low-word pairs, the directories, the phi list, its directory and
// super counts.  Nothing is left out for its size: entry indices are 64-bit, a table may hold up to 2^32 - 16 entries
// (more is an error with a message, not a silent drop), and the phi directory has no size cap.
struct TmpDev {   // device scratch of the load, freed at scope exit
    void *p = nullptr;
    ~TmpDev() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV; }
        return RBG_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    template <typename T> T *as() { return static_cast<T *>(p); }
};
inline size_t scan_tmp_bytes_for(uint64_t N) { return scan_tmp_bytes(N); }

// fillers for a list of m {key, value} u64 pairs at *ent (device; tables closed by sentinels with key n).  When some are
// needed: *ent / *samp are replaced by the expanded arrays (`own` says whether the old ones are tracked allocations of the
// index or plain hipMalloc blocks), *m by the new count, and `at` (indices into the old list) by their new places.
int add_fillers(rbg_index *ix, bool phi, void **ent, void **samp, bool tracked, uint64_t *m, uint64_t n, std::vector<uint64_t> &at, uint64_t *fillers) {
    *fillers = 0;
    const uint32_t fs = ix->dev.run_fill_shift;
    TmpDev tot;
    int rc = tot.alloc(8);
    if (rc) return rc;
    HIP_TRY(hipMemset(tot.p, 0, 8));
    HIP_TRY(static_cast<hipError_t>(launch_fill_count(*ent, *m, n, fs, nullptr, tot.as<unsigned long long>(), nullptr)));
    unsigned long long total = 0;
    HIP_TRY(hipMemcpy(&total, tot.p, 8, hipMemcpyDeviceToHost));
    if (!total) return RBG_OK;
    if (total > (uint64_t(1) << 40) || *m > (uint64_t(1) << 40)) return RBG_ENOMEM;   // (sizes below stay far from 2^64; no index that fits a device comes near)
    TmpDev arr, tmp, idx, out;
    const size_t tb = scan_tmp_bytes_for(*m + 1);
    if ((rc = arr.alloc((*m + 1) * 8)) || (rc = tmp.alloc(tb))) return rc;
    HIP_TRY(hipMemset(tot.p, 0, 8));
    HIP_TRY(static_cast<hipError_t>(launch_fill_count(*ent, *m, n, fs, arr.as<uint64_t>(), tot.as<unsigned long long>(), nullptr)));
    HIP_TRY(static_cast<hipError_t>(launch_scan_u64(arr.as<uint64_t>(), *m + 1, tmp.p, tb, nullptr)));
    const uint64_t m2 = *m + total;
    // the expanded arrays: given back on EVERY error path below (a tracked block through the index's list, a plain one by hipFree),
    // handed to the caller only once everything has succeeded
    struct NewBlock {
        rbg_index *ix; bool tracked; void *p = nullptr;
        NewBlock(rbg_index *i, bool t) : ix(i), tracked(t) {}
        ~NewBlock() { if (!p) return; if (tracked) free_tracked(ix, p); else (void)hipFree(p); }
        int alloc(size_t bytes) {
            if (tracked) {
                // (its own allocation, never a piece of the arena: free_tracked must be able to give it back)
                hipError_t e = hipMalloc(&p, arena_round(bytes));
                if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV; }
                ix->allocs.push_back({p, arena_round(bytes)});
                ix->hbm_bytes += arena_round(bytes);
                return RBG_OK;
            }
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV; }
            return RBG_OK;
        }
        void *release() { void *q = p; p = nullptr; return q; }
    } ent2(ix, tracked), samp2(ix, tracked);
    if ((rc = ent2.alloc((m2 + 2) * 16))) return rc;
    if (*samp && (rc = samp2.alloc(m2 * 8 + 16))) return rc;
    HIP_TRY(static_cast<hipError_t>(launch_fill_expand(phi, *ent, static_cast<const uint64_t *>(*samp), *m, n, fs, arr.as<uint64_t>(), ent2.p, static_cast<uint64_t *>(samp2.p), nullptr)));
    if (!at.empty()) {
        if ((rc = idx.alloc(at.size() * 8)) || (rc = out.alloc(at.size() * 8))) return rc;
        HIP_TRY(hipMemcpy(idx.p, at.data(), at.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(static_cast<hipError_t>(launch_gather_u64(arr.as<uint64_t>(), idx.as<uint64_t>(), at.size(), out.as<uint64_t>(), nullptr)));
        HIP_TRY(hipMemcpy(at.data(), out.p, at.size() * 8, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipDeviceSynchronize());
    if (tracked) { free_tracked(ix, *ent); if (*samp) free_tracked(ix, *samp); }
    else { (void)hipFree(*ent); if (*samp) (void)hipFree(*samp); }
    *ent = ent2.release();
    *samp = samp2.release();
    *m = m2;
    *fillers = total;
    return RBG_OK;
}

template <typename P>
int upload_tables_runs2(rbg_index *ix) {
    constexpr bool W = sizeof(P) == 8;
    HostIndex &h = ix->H();
    rbg_index::RunsReport &rep = ix->runs_report;
    rep.fmt = 2;
    for (SymTable &t : h.sym) {   // (the depth-1 lists compose_on_device left on the device are the slot layout's)
        free_tracked(ix, const_cast<void *>(t.dev_ent));
        free_tracked(ix, const_cast<void *>(t.dev_samp));
        t.dev_ent = t.dev_samp = nullptr;
    }
    const std::vector<SymTable> *depth[kMaxRunDepth];
    depth[0] = &h.sym;
    for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxRunDepth); ++d) depth[d - 1] = &h.kmer(d);
    uint32_t D = 1;
    while (D < static_cast<uint32_t>(kMaxRunDepth) && !depth[D]->empty()) ++D;
    rep.depths_composed = D;
    uint32_t mask = (ix->run_depth_mask ? ix->run_depth_mask : ~0u) & ((1u << D) - 1u);
    mask |= 1u | (1u << (D - 1));   // (the deepest is always kept: the kernels step by it)
    const char *e_dt = std::getenv("RBG_RANK_DIR_RUNS");   // runs per directory bucket at most this on average (default 4)
    const double dir_target = e_dt && std::atof(e_dt) > 0 ? std::atof(e_dt) : 4.0;
    // RBG_RUN_FILL_SHIFT / RBG_PHI_SUPER_SHIFT: test-only overrides so that small indexes meet fillers and several super blocks
    ix->dev.run_fill_shift = static_cast<uint32_t>(env_opt("RBG_RUN_FILL_SHIFT", kRunFillShift, 4, kRunFillShift));
    const uint32_t super_shift = static_cast<uint32_t>(env_opt("RBG_PHI_SUPER_SHIFT", kPhiSuperShift, 1, 24));
    const uint32_t max_shift = W ? ix->dev.run_fill_shift : 31u;   // (a per-lane shift of the low word: rbg_device.hpp pos_bucket)
    // BUCKET RECORDS (RBG_OPT_RUN_REC; rbg_dev.h RunRec2): one aligned 64-byte record per bucket of about three entries instead of
    // the directory -- a rank is one sector.  Automatic: when all kept depths with their records (about 64 / 3 bytes per entry) and
    // the rest of the replica stay within half the budget.  RBG_RUN_REC_PER: entries per bucket on average (default 2.5 inside the
    // bucket; the one before them is held too).
    const char *e_rp = std::getenv("RBG_RUN_REC_PER");
    const double rec_asked = e_rp && std::atof(e_rp) > 0 ? std::atof(e_rp) : 0.0;
    // PER DEPTH, deepest first (a search spends its steps at the deepest depth; the shallower ones take a read's ragged ends): rec_per[d] =
    // entries per bucket on average of depth d's records, 0 = directories.  RBG_OPT_RUN_REC = 2: the depths of RBG_OPT_RUN_REC_DEPTHS
    // (0 = all kept) at RBG_RUN_REC_PER (2.5).  Automatic: each depth in turn gets the narrowest buckets -- 2.5, 4 or 6 entries (a compact
    // record holds eleven) -- with which the replica (phi slots included) stays within the budget and the records stay O(r) (at most one per entry).
    std::vector<double> rec_per(D, 0.0);
    auto records_of = [&](uint32_t d, double per) { return runs_record_count(h, d, per, max_shift); };
    if (g_opt_run_rec.load() == 2) {
        const uint32_t want = g_opt_run_rec_depths.load() ? static_cast<uint32_t>(g_opt_run_rec_depths.load()) : ~0u;
        for (uint32_t d = 0; d < D; ++d)
            if ((mask >> d & 1u) && (want >> d & 1u)) rec_per[d] = rec_asked > 0 ? rec_asked : 2.5;
    } else if (g_opt_run_rec.load() == 0 && ix->hbm_budget) {
        double total = static_cast<double>(W ? runs_replica_bytes<uint64_t>(h, mask) : runs_replica_bytes<uint32_t>(h, mask));
        if (g_opt_run_phi.load() != 1) total += runs_phi_slot_bytes(h);   // phi slots come first (decided after the rank tables, below: the same arithmetic): their room is not the records'
        const double budget = static_cast<double>(ix->hbm_budget);
        const double pers[3] = {2.5, 4.0, 6.0};
        // room at the widest buckets (6 entries) for EVERY kept depth?  Then every depth gets records -- a depth left on directories pays narrowing rounds on
        // its steps -- and a depth takes narrower buckets only with what the shallower ones do not need.  Otherwise: deepest first, while they fit.
        std::vector<double> widest(D, 0.0), entries(D, 0.0);
        double all_widest = 0;
        bool every = true;
        for (uint32_t d = 0; d < D; ++d) {
            if (!(mask >> d & 1u)) continue;
            for (const SymTable &t : *depth[d]) entries[d] += static_cast<double>(t.nruns + 1);
            widest[d] = records_of(d, rec_asked > 0 ? rec_asked : pers[2]) * 64.0;
            all_widest += widest[d];
            every = every && widest[d] <= entries[d] * 64.0;
        }
        every = every && total + all_widest <= budget;
        double shallower_widest = all_widest;
        for (int d = static_cast<int>(D) - 1; d >= 0; --d) {
            if (!(mask >> d & 1u)) continue;
            shallower_widest -= widest[d];
            for (const double per : pers) {
                if (rec_asked > 0 && per != pers[0]) break;
                const double use = rec_asked > 0 ? rec_asked : per;
                const double bytes = records_of(static_cast<uint32_t>(d), use) * 64.0;
                if (bytes <= entries[d] * 64.0 && total + bytes + (every ? shallower_widest : 0.0) <= budget) {
                    rec_per[d] = use;
                    total += bytes;
                    break;
                }
            }
        }
    }
    bool any_recs = false, all_recs = true;
    for (uint32_t d = 0; d < D; ++d)
        if (mask >> d & 1u) { any_recs = any_recs || rec_per[d] > 0; all_recs = all_recs && rec_per[d] > 0; }
    std::vector<DevRunTab2> tabs;
    std::vector<uint64_t> hot;      // rbg_dev.h: dir_off | dir_shift << 56 per table
    int rc;
    for (uint32_t d = 0; d < D; ++d) {
        const std::vector<SymTable> &T = *depth[d];
        ix->dev.run_tab_first[d] = static_cast<uint32_t>(tabs.size());
        ix->dev.run_samp[d] = nullptr;
        ix->dev.run_ent2[d] = nullptr; ix->dev.run_dir2[d] = nullptr;
        if (!(mask >> d & 1u)) {   // no run lists at this depth: nothing steps by it
            release_kmer_level(ix, d + 1);
            for (SymTable &st : kmer_level_tables(h, d + 1)) st.dev_ent = st.dev_samp = nullptr;
            continue;
        }
        uint64_t entries = 0;
        for (const SymTable &t : T) entries += t.nruns + 1;
        // ---- the depth's {start, cum} pairs of P, tables back to back, and its samples (P each) on the device ----
        void *abs_ent = nullptr, *abs_samp = nullptr;
        std::vector<uint64_t> first(T.size() + 1, 0), nr(T.size());
        for (size_t t = 0; t < T.size(); ++t) { first[t + 1] = first[t] + T[t].nruns + 1; nr[t] = T[t].nruns; }
        ComposedLevel *L = (d >= 1 && d - 1 < ix->kmer_levels.size() && ix->kmer_levels[d - 1].ent) ? &ix->kmer_levels[d - 1] : nullptr;
        if (L) {
            if (L->entries != entries || L->first.size() != T.size()) return RBG_EARG;
            for (size_t t = 0; t < T.size(); ++t)
                if (L->first[t] != first[t]) return RBG_EARG;
            abs_ent = L->ent;
            abs_samp = h.has_tsa ? L->samp : nullptr;
            L->ent = L->samp = nullptr;   // (adopted: the index's allocation list keeps them)
        } else {
            HostBuf<RunEnt<P>> ent(entries + 2);
            HostBuf<P> samp(h.has_tsa ? entries + 2 : 0);
            const size_t Wk = std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), T.size()}));
            std::vector<std::thread> workers;
            for (size_t w = 0; w < Wk; ++w)
                workers.emplace_back([&, w] {
                    for (size_t t = w; t < T.size(); t += Wk) {
                        const SymTable &tb = T[t];
                        if (tb.start.size() != tb.nruns + 1) continue;   // (checked below)
                        for (uint64_t k = 0; k <= tb.nruns; ++k) ent[first[t] + k] = RunEnt<P>{static_cast<P>(tb.start[k]), static_cast<P>(tb.cum[k])};
                        if (h.has_tsa) {
                            for (uint64_t k = 0; k < tb.nruns; ++k) samp[first[t] + k] = static_cast<P>(tb.samp[k]);
                            samp[first[t] + tb.nruns] = 0;
                        }
                    }
                });
            for (auto &w : workers) w.join();
            for (const SymTable &tb : T)
                if (tb.start.size() != tb.nruns + 1) return RBG_EARG;   // a table without host arrays and without a device level
            for (uint64_t x = 0; x < 2; ++x) { ent[entries + x] = ent[entries - 1]; if (h.has_tsa) samp[entries + x] = 0; }
            const void *up = nullptr;
            if ((rc = dev_upload(ix, ent.data(), (entries + 2) * sizeof(RunEnt<P>), &up))) return rc;
            abs_ent = const_cast<void *>(up);
            if (h.has_tsa) {
                if ((rc = dev_upload(ix, samp.data(), (entries + 2) * sizeof(P), &up))) return rc;
                abs_samp = const_cast<void *>(up);
            }
        }
        uint64_t E2 = entries, fillers = 0;
        if constexpr (W) {
            std::vector<uint64_t> at;
            for (size_t t = 0; t < T.size(); ++t) { at.push_back(first[t]); at.push_back(first[t] + nr[t]); }
            if ((rc = add_fillers(ix, false, &abs_ent, &abs_samp, true, &E2, h.n, at, &fillers))) return rc;
            if (fillers) {
                for (size_t t = 0; t < T.size(); ++t) { first[t] = at[2 * t]; nr[t] = at[2 * t + 1] - at[2 * t]; }
                first[T.size()] = E2;
            }
        }
        for (size_t t = 0; t < T.size(); ++t)
            if (nr[t] >= 0xFFFFFFF0ull) {
                std::fprintf(stderr, "rbg: a table of k-mer depth %u has %llu entries: the run-indexed layout holds fewer than 2^32 - 16 per table\n", d + 1,
                             static_cast<unsigned long long>(nr[t]));
                return RBG_EARG;
            }
        rep.entries[d] = E2;
        rep.fillers[d] = fillers;
        {   // every cum becomes a ROW of the F column: + the table's F (rbg_dev.h kRunHotShiftBit; k_build.hip k_fold_F)
            TmpDev tf;
            const size_t nt = T.size();
            if ((rc = tf.alloc((2 * nt + 1) * 8))) return rc;
            std::vector<uint64_t> Fv(nt);
            for (size_t t = 0; t < nt; ++t) Fv[t] = T[t].F;
            uint64_t *t_first = tf.as<uint64_t>(), *t_F = t_first + nt + 1;
            HIP_TRY(hipMemcpy(t_first, first.data(), (nt + 1) * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_F, Fv.data(), nt * 8, hipMemcpyHostToDevice));
            HIP_TRY(static_cast<hipError_t>(launch_fold_F(sizeof(P), abs_ent, t_first, t_F, static_cast<uint32_t>(nt), E2 + (W ? 0 : 2), nullptr)));   // (4-byte positions: the two spare entries are final too)
            HIP_TRY(hipDeviceSynchronize());
        }
        // ---- directories: per table the widest bucket that still holds at most about dir_target entries on average ----
        std::vector<uint32_t> dshift(T.size(), 0);
        std::vector<uint64_t> doff(T.size() + 1, 0);
        for (size_t t = 0; t < T.size(); ++t) {
            uint32_t sh = 0;
            const double runs = static_cast<double>(std::max<uint64_t>(1, nr[t]));
            while (sh < max_shift && runs * static_cast<double>(uint64_t(2) << sh) <= dir_target * static_cast<double>(h.n)) ++sh;
            dshift[t] = sh;
            doff[t + 1] = doff[t] + (h.n >> sh) + 2;
        }
        void *dirp = nullptr;
        const size_t dir_ent = W ? sizeof(RunDir64) : 4;
        ix->dev.run_rec2[d] = nullptr;
        const bool use_recs = rec_per[d] > 0;
        const double rec_target = rec_per[d];
        if (d == 0) { ix->dev.run_uni_depth = static_cast<uint32_t>(kMaxRunDepth); ix->dev.run_uni_stride = ix->dev.run_uni_shift = 0; }
        if (use_recs) {
            // the records' buckets: the widest with at most rec_target entries starting inside on average
            auto widest = [&](double runs) {
                uint32_t sh = 0;
                runs = std::max(1.0, runs);
                while (sh < max_shift && runs * static_cast<double>(uint64_t(2) << sh) <= rec_target * static_cast<double>(h.n)) ++sh;
                return sh;
            };
            for (size_t t = 0; t < T.size(); ++t) {
                dshift[t] = widest(static_cast<double>(nr[t]));
                doff[t + 1] = doff[t] + (h.n >> dshift[t]) + 2;
            }
            // UNIFORM geometry (DevIndex::run_uni) for a depth whose hot words a step would read from global memory: one shift -- the widest for the
            // depth's AVERAGE table -- and one record count for every table, so that a step computes its table's hot word.  Taken when it costs no more
            // than a tenth more records than the tables' own shifts, and kept when the buckets that then overflow their record (more than eleven entries:
            // pivots + a scan of the run list) stay rare -- a depth with a few very frequent k-mers keeps per-table shifts.  RBG_RUN_UNIFORM=0 / 1: never /
            // whenever the depth is deep enough (tests, A/B).
            const int uni_env = [] { const char *e = std::getenv("RBG_RUN_UNIFORM"); return e ? std::atoi(e) : -1; }();   // (read per load: tests switch it)
            bool uniform = false;
            std::vector<uint32_t> own_shift = dshift;
            std::vector<uint64_t> own_off = doff;
            bool deepest = true;   // (one uniform depth: the deepest kept one with records -- DevIndex::run_uni_*)
            for (uint32_t d2 = d + 1; d2 < D; ++d2) deepest = deepest && !((mask >> d2 & 1u) && rec_per[d2] > 0);
            if (deepest && d >= static_cast<uint32_t>(kLdsRunDepth) && uni_env != 0 && T.size() > 1) {
                double total_runs = 0;
                for (size_t t = 0; t < T.size(); ++t) total_runs += static_cast<double>(nr[t]);
                const uint32_t su = widest(total_runs / static_cast<double>(T.size()));
                const uint64_t stride = (h.n >> su) + 2;
                if (!(stride >> 27) && su < 32 && tabs.size() + T.size() < (1u << 24) && (uni_env == 1 || static_cast<double>(stride) * static_cast<double>(T.size()) <= 1.1 * static_cast<double>(own_off[T.size()]))) {
                    uniform = true;
                    for (size_t t = 0; t < T.size(); ++t) { dshift[t] = su; doff[t + 1] = doff[t] + stride; }
                }
            }
            void *recp = nullptr;
            unsigned long long novf = 0;
            // the depth's bucket records under the geometry (dshift, doff) as they stand: recp, novf
            auto build_recs = [&]() -> int {
                int rc2;
                if ((rc2 = dev_reserve(ix, doff[T.size()] * sizeof(RunRec2) + 64, &recp))) return rc2;
                TmpDev tmp, ovf;
                const size_t nt = T.size(), bytes = (3 * nt + 1) * 8 + nt * 4;
                if ((rc2 = tmp.alloc(bytes)) || (rc2 = ovf.alloc(8))) return rc2;
                HIP_TRY(hipMemset(ovf.p, 0, 8));
                uint64_t *t_first = tmp.as<uint64_t>(), *t_nr = t_first + nt, *t_doff = t_nr + nt;
                uint32_t *t_sh = reinterpret_cast<uint32_t *>(t_doff + nt + 1);
                HIP_TRY(hipMemcpy(t_first, first.data(), nt * 8, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(t_nr, nr.data(), nt * 8, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(t_doff, doff.data(), (nt + 1) * 8, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(t_sh, dshift.data(), nt * 4, hipMemcpyHostToDevice));
                HIP_TRY(static_cast<hipError_t>(launch_run_recs2(sizeof(P), abs_ent, t_first, t_nr, t_doff, t_sh, static_cast<uint32_t>(nt), doff[nt], recp, ovf.as<unsigned long long>(), nullptr)));
                HIP_TRY(hipMemcpy(&novf, ovf.p, 8, hipMemcpyDeviceToHost));
                return 0;
            };
            if ((rc = build_recs())) return rc;
            // Overflowing records (more than eleven entries: pivots + a scan of the run list, two or three sectors instead of one) come from how a locus's runs
            // cluster, whatever the geometry: on the bench index 4.9 % of the uniform depth-8 records against 4.7 % under the tables' own shifts.  So a uniform
            // depth with MANY of them is measured against the tables' own shifts, not against zero: it stays when it overflows at most a quarter more often.
            if (uniform && uni_env != 1 && static_cast<double>(novf) * 256.0 > static_cast<double>(doff[T.size()])) {
                const std::vector<uint32_t> uni_shift = dshift;
                const std::vector<uint64_t> uni_off = doff;
                const unsigned long long novf_uni = novf;
                free_tracked(ix, recp);
                recp = nullptr;
                dshift = own_shift;
                doff = own_off;
                if ((rc = build_recs())) return rc;
                const unsigned long long novf_own = novf;
                if (static_cast<double>(novf_uni) <= 1.25 * static_cast<double>(novf_own) + static_cast<double>(uni_off[T.size()]) / 256.0) {
                    free_tracked(ix, recp);
                    recp = nullptr;
                    dshift = uni_shift;
                    doff = uni_off;
                    if ((rc = build_recs())) return rc;
                } else {
                    uniform = false;
                    if (std::getenv("RBG_VERBOSE"))
                        std::fprintf(stderr, "rbg:   depth %u: uniform directories would leave %llu records overflowing, the tables' own shifts %llu: the tables keep their own shifts\n",
                                     d + 1, novf_uni, novf_own);
                }
            }
            if (uniform) {
                const uint64_t stride = doff[1] - doff[0];
                if (stride >> 27) return RBG_EARG;   // (load_run_tab's packed constants: a 27-bit stride, a 5-bit shift, a 24-bit first record)
                ix->dev.run_uni_depth = d; ix->dev.run_uni_stride = static_cast<uint32_t>(stride); ix->dev.run_uni_shift = dshift[0];
                if (std::getenv("RBG_VERBOSE"))
                    std::fprintf(stderr, "rbg:   depth %u: uniform directories (shift %u, %llu records per table, %llu of %llu overflowing): hot words computed\n", d + 1, dshift[0],
                                 static_cast<unsigned long long>(stride), novf, static_cast<unsigned long long>(doff[T.size()]));
            }
            ix->dev.run_rec2[d] = static_cast<const RunRec2 *>(recp);
            rep.rec_bytes[d] = doff[T.size()] * sizeof(RunRec2);
            rep.rec_overflow[d] = novf;
        } else {
        if ((rc = dev_reserve(ix, doff[T.size()] * dir_ent + 16, &dirp))) return rc;
        {
            TmpDev tmp;
            const size_t nt = T.size(), bytes = (3 * nt + 1) * 8 + nt * 4;
            if ((rc = tmp.alloc(bytes))) return rc;
            uint64_t *t_first = tmp.as<uint64_t>(), *t_nr = t_first + nt, *t_doff = t_nr + nt;
            uint32_t *t_sh = reinterpret_cast<uint32_t *>(t_doff + nt + 1);
            HIP_TRY(hipMemcpy(t_first, first.data(), nt * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_nr, nr.data(), nt * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_doff, doff.data(), (nt + 1) * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_sh, dshift.data(), nt * 4, hipMemcpyHostToDevice));
            if constexpr (W) HIP_TRY(static_cast<hipError_t>(launch_run_dirs2(abs_ent, t_first, t_nr, t_doff, t_sh, static_cast<uint32_t>(nt), doff[nt], dirp, nullptr)));
            else HIP_TRY(static_cast<hipError_t>(launch_run_dirs(4, abs_ent, t_first, t_nr, t_doff, t_sh, static_cast<uint32_t>(nt), doff[nt], static_cast<uint32_t *>(dirp), nullptr)));
            HIP_TRY(hipDeviceSynchronize());
        }
        rep.dir_bytes[d] = doff[T.size()] * dir_ent;
        }
        ix->dev.run_dir2[d] = dirp;
        // ---- the entries and samples in their final form ----
        if constexpr (W) {
            void *e2 = nullptr, *s6 = nullptr;
            if ((rc = dev_reserve(ix, (E2 + 2) * 8, &e2))) return rc;
            HIP_TRY(static_cast<hipError_t>(launch_pack_pairs32(abs_ent, E2, 2, e2, nullptr)));
            if (abs_samp) {
                if ((rc = dev_reserve(ix, E2 * RunsFmt<P>::samp_bytes + 8, &s6))) return rc;
                HIP_TRY(static_cast<hipError_t>(launch_pack_samp48(static_cast<const uint64_t *>(abs_samp), E2, s6, nullptr)));
            }
            HIP_TRY(hipDeviceSynchronize());
            free_tracked(ix, abs_ent);
            if (abs_samp) free_tracked(ix, abs_samp);
            ix->dev.run_ent2[d] = e2;
            ix->dev.run_samp[d] = s6;
        } else {
            ix->dev.run_ent2[d] = abs_ent;
            ix->dev.run_samp[d] = abs_samp;
        }
        if (std::getenv("RBG_VERBOSE")) {
            size_t f = 0, tt = 0;
            (void)hipMemGetInfo(&f, &tt);
            std::fprintf(stderr, "rbg:   run lists of depth %u in their final form: %llu entries (%llu fillers), directories %.2f GB; HBM in use %.1f GB\n", d + 1,
                         static_cast<unsigned long long>(E2), static_cast<unsigned long long>(fillers), rep.dir_bytes[d] / 1e9, static_cast<double>(tt - f) / 1e9);
        }
        for (size_t t = 0; t < T.size(); ++t) {
            tabs.push_back(DevRunTab2{T[t].F, first[t], doff[t], dshift[t], 0u});
            if (doff[t] >> kRunHotShiftBit) return RBG_EARG;   // (2^56 buckets: no index that fits a device comes near)
            hot.push_back(doff[t] | static_cast<uint64_t>(dshift[t]) << kRunHotShiftBit);
        }
        tabs.push_back(DevRunTab2{0, E2, 0, 0u, 0u});   // closing record
        hot.push_back(0);
    }
    for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d) release_kmer_level(ix, d);   // (levels beyond D, or left over: nothing points at them)
    for (uint32_t d = D; d <= static_cast<uint32_t>(kMaxRunDepth); ++d) ix->dev.run_tab_first[d] = static_cast<uint32_t>(tabs.size());
    if (ix->dev.run_tab_first[std::min<uint32_t>(D, kLdsRunDepth)] > static_cast<uint32_t>(kMaxLdsRunTabs)) return RBG_EARG;
    const void *p = nullptr;
    std::vector<DevSym> syms(h.sym.size());   // (no kernel reads a symbol record on this format: F only, for rbg_get_f-style readers)
    for (size_t t = 0; t < syms.size(); ++t) { syms[t] = DevSym{}; syms[t].F = h.sym[t].F; syms[t].nruns = static_cast<uint32_t>(std::min<uint64_t>(h.sym[t].nruns, 0xFFFFFFFFull)); }
    if ((rc = dev_upload(ix, syms.data(), syms.size() * sizeof(DevSym), &p))) return rc;
    ix->dev.syms = static_cast<const DevSym *>(p);
    if ((rc = dev_upload(ix, tabs.data(), tabs.size() * sizeof(DevRunTab2), &p))) return rc;
    ix->dev.run_tabs2 = static_cast<const DevRunTab2 *>(p);
    if ((rc = dev_upload(ix, hot.data(), hot.size() * 8, &p))) return rc;
    ix->dev.run_hot = static_cast<const uint64_t *>(p);
    ix->dev.run_ntabs = static_cast<uint32_t>(tabs.size());
    ix->dev.run_ksteps = D;
    ix->dev.run_depth_mask = mask;
    ix->run_depth_mask = mask;
    rep.depth_mask_kept = mask;
    rep.rank_dirs = all_recs ? 0 : 1;   // (1: some kept depth answers its ranks through a directory)
    (void)any_recs;
    ix->dev.layout = RBG_LAYOUT_RUNS;
    ix->dev.kmer_steps = 1;
    ix->dev.nmajor = 0;
    if (h.nmajor >= 2) {  // the ftab's word index and the k-mer table index need the major alphabet
        if ((rc = dev_upload(ix, h.major_of, 256, &p))) return rc;
        ix->dev.lut2 = static_cast<const uint8_t *>(p);
        ix->dev.nmajor = h.nmajor;
    }
    // the register tables of the in-kernel read staging (rbg_dev.h stage_*): a shift under which the four major bytes hash to four different
    // three-bit values
    ix->dev.stage_ok = 0;
    if (h.nmajor == 4) {
        for (uint32_t sh = 0; sh <= 5 && !ix->dev.stage_ok; ++sh) {
            uint8_t code[8] = {0, 0, 0, 0, 0, 0, 0, 0}, byte[8];
            bool used[8] = {false, false, false, false, false, false, false, false}, distinct = true;
            for (uint32_t m = 0; m < 4; ++m) {
                const uint32_t t = (h.major_byte[m] >> sh) & 7u;
                if (used[t]) distinct = false;
                used[t] = true;
                code[t] = static_cast<uint8_t>(m);
                byte[t] = h.major_byte[m];
            }
            if (!distinct) continue;
            // an unused place must never equal the byte that hashes to it: a byte whose own hash is another place
            for (uint32_t t = 0; t < 8; ++t)
                if (!used[t]) byte[t] = static_cast<uint8_t>(((t ^ 1u) & 7u) << sh);
            ix->dev.stage_ok = 1;
            ix->dev.stage_shift = sh;
            std::memcpy(ix->dev.stage_code, code, 8);
            std::memcpy(ix->dev.stage_byte, byte, 8);
        }
    }
    ix->dev.phi_slots = nullptr;
    ix->dev.phi_ord = nullptr;
    ix->dev.phi_dir = nullptr;
    ix->dev.phi_super = nullptr;
    ix->dev.phi_super_shift = 0;
    // PHI SLOTS on this layout (RBG_OPT_RUN_PHI = 2; automatic when the whole replica then stays within the budget -- the bucket records of the
    // rank tables, decided before, have left room for them: K3 is the larger kernel at pangenome scale): the slot
    // layout's direct-addressed phi records (rbg_dev.h PhiSlot) with buckets of about n / r rows instead of 32-64 -- so their
    // number is proportional to r, not n -- answer a phi step from ONE sector where the list takes two (directory, entries); at
    // pangenome scale K3 is bound by exactly that sector count.  Cost: about 54 bytes per run at 8-byte positions against 16.
    bool phi_by_slots = false;
    uint32_t slot_shift = 0;
    if (h.has_tsa) {
        const double rows_per_sample = static_cast<double>(h.n) / static_cast<double>(std::max<uint64_t>(1, h.r));
        while (slot_shift < 8 && static_cast<double>(uint64_t(2) << slot_shift) <= rows_per_sample) ++slot_shift;   // the widest bucket with at most one sampled position on average
        if (slot_shift < h.phi_shift) slot_shift = h.phi_shift;
        const bool packed = sizeof(P) == 8 && (h.n >> kPhiPackedPosBits) == 0 && slot_shift <= kPhiPackedMaxShift;
        const size_t slot_b = packed ? sizeof(PhiSlotPacked) : sizeof(PhiSlot<P>);
        const size_t need = ((h.n >> slot_shift) + 2) * (slot_b + 4) + (h.r + 1) * sizeof(PhiEnt<P>);
        const int64_t mode = g_opt_run_phi.load();
        // automatic: only while the slots are O(r) -- at most two buckets per sampled position (the bucket shift stops at 8: an index with
        // n / r far beyond 256 would get n / 256 of them) -- and the whole replica stays within the budget
        phi_by_slots = mode == 2 || (mode == 0 && ix->hbm_budget && ((h.n >> slot_shift) + 2) <= 2 * h.r && ix->hbm_bytes + need <= ix->hbm_budget);
        if (phi_by_slots) {
            VStage vs("phi slots of the run-indexed layout");
            HostBuf<PhiEnt<P>> pe(h.r + 1);
            parallel_for(h.r, [&](uint64_t a, uint64_t b, unsigned) {
                for (uint64_t j = a; j < b; ++j) { pe[j].pos = static_cast<P>(h.pred_pos[j]); pe[j].base = static_cast<P>(h.phi_base[j]); }
            });
            pe[h.r].pos = static_cast<P>(h.n); pe[h.r].base = 0;
            if ((rc = dev_upload(ix, pe.data(), (h.r + 1) * sizeof(PhiEnt<P>), &ix->dev.phi_ent))) return rc;
            const uint64_t nb = (h.n >> slot_shift) + 2;
            void *slots = nullptr, *ord = nullptr;
            if ((rc = dev_reserve(ix, nb * slot_b, &slots)) || (rc = dev_reserve(ix, nb * sizeof(uint32_t), &ord))) return rc;
            TmpDev ovf;
            if ((rc = ovf.alloc(8))) return rc;
            HIP_TRY(hipMemset(ovf.p, 0, 8));
            ix->dev.phi_packed = packed ? 1 : 0;
            ix->dev.phi_shift = slot_shift;
            if (launch_build_phi_slots(sizeof(P), packed, ix->dev.phi_ent, h.r, h.n, slot_shift, slots, static_cast<uint32_t *>(ord), ovf.as<unsigned long long>(), nullptr))
                return RBG_ENODEV;
            unsigned long long novf = 0;
            HIP_TRY(hipMemcpy(&novf, ovf.p, 8, hipMemcpyDeviceToHost));
            ix->phi_slots = nb;
            ix->phi_slots_overflow = novf;
            ix->dev.phi_slots = slots;
            ix->dev.phi_ord = static_cast<const uint32_t *>(ord);
            ix->dev.phi_m = h.r;
            ix->dev.phi_last_pos = h.pred_pos[h.r - 1];
            ix->dev.phi_last_base = h.phi_base[h.r - 1];
            rep.phi_entries = h.r; rep.phi_dir = 0; rep.phi_dir_shift = slot_shift; rep.phi_slots = nb; rep.phi_slot_bytes = nb * (slot_b + 4);
        }
    }
    if (h.has_tsa && !phi_by_slots) {
        // sampled positions per directory bucket: between per and 2 * per on average (RBG_PHI_DIR_PER, default 1: the scan's
        // first four requests then cover the bucket and its predecessor nineteen times in twenty)
        const char *e_pp = std::getenv("RBG_PHI_DIR_PER");
        const double per = e_pp && std::atof(e_pp) > 0 ? std::atof(e_pp) : 1.0;
        uint32_t ds = 2;
        while (ds < max_shift && ds < 30 && (static_cast<double>(h.r) * static_cast<double>(uint64_t(1) << ds)) / static_cast<double>(h.n) < per) ++ds;
        const uint64_t nd = (h.n >> ds) + 2;
        void *dirp = nullptr;
        if ((rc = dev_reserve(ix, nd * 4 + 16, &dirp))) return rc;
        uint64_t m2 = h.r, fillers = 0;
        if constexpr (W) {
                    HostBuf<uint64_t> pe((h.r + 1) * 2);
            parallel_for(h.r, [&](uint64_t a, uint64_t b, unsigned) {
                for (uint64_t j = a; j < b; ++j) { pe[2 * j] = h.pred_pos[j]; pe[2 * j + 1] = h.phi_base[j]; }
            });
            pe[2 * h.r] = h.n; pe[2 * h.r + 1] = 0;   // sentinel: never below a query
            void *abs = nullptr, *none = nullptr;
            HIP_TRY(hipMalloc(&abs, (h.r + 1 + 2) * 16));
            if ((rc = h2d_big(abs, pe.data(), (h.r + 1) * 16))) { (void)hipFree(abs); return rc; }
            uint64_t m_all = h.r + 1;
            std::vector<uint64_t> at;
            rc = add_fillers(ix, true, &abs, &none, false, &m_all, h.n, at, &fillers);
            if (rc) { (void)hipFree(abs); return rc; }
            m2 = m_all - 1;
            void *e12 = nullptr, *sup = nullptr;
            const uint64_t nsup = (nd >> super_shift) + 2;
            rc = dev_reserve(ix, (m2 + 1 + 3) * sizeof(PhiEnt12), &e12);
            if (!rc) rc = dev_reserve(ix, nsup * 8, &sup);
            hipError_t e = hipSuccess;
            if (!rc) e = static_cast<hipError_t>(launch_pack_phi12(abs, m2 + 1, 3, e12, nullptr));
            if (!rc && e == hipSuccess) e = static_cast<hipError_t>(launch_phi_dir(8, abs, m2, ds, nd, static_cast<uint32_t *>(dirp), super_shift, static_cast<uint64_t *>(sup), nullptr));
            if (!rc && e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(abs);
            if (rc) return rc;
            HIP_TRY(e);
            ix->dev.phi_ent = e12;
            ix->dev.phi_super = static_cast<const uint64_t *>(sup);
            ix->dev.phi_super_shift = super_shift;
        } else {
            typedef PhiFmt<P> Fmt;
            HostBuf<unsigned char> pe((h.r + 1 + Fmt::spare) * Fmt::ent_bytes);
            parallel_for(h.r, [&](uint64_t a, uint64_t b, unsigned) {
                for (uint64_t j = a; j < b; ++j) Fmt::put_ent(pe.data(), j, h.pred_pos[j], h.phi_base[j]);
            });
            for (size_t x = 0; x <= Fmt::spare; ++x) Fmt::put_ent(pe.data(), h.r + x, h.n, 0);
            if ((rc = dev_upload(ix, pe.data(), pe.size(), &ix->dev.phi_ent))) return rc;
            HIP_TRY(static_cast<hipError_t>(launch_phi_dir(4, ix->dev.phi_ent, h.r, ds, nd, static_cast<uint32_t *>(dirp), 0, nullptr, nullptr)));
            HIP_TRY(hipDeviceSynchronize());
        }
        ix->dev.phi_dir = static_cast<const uint32_t *>(dirp);
        ix->dev.phi_dir_shift = ds;
        ix->dev.phi_m = m2;
        ix->dev.phi_last_pos = h.pred_pos[h.r - 1];
        ix->dev.phi_last_base = h.phi_base[h.r - 1];
        rep.phi_entries = m2; rep.phi_fillers = fillers; rep.phi_dir_bytes = nd * 4; rep.phi_dir_shift = ds; rep.phi_dir = 1;
    }
    HIP_TRY(hipDeviceSynchronize());
    return RBG_OK;
}

int upload_markers(rbg_index *ix) {
    const RawMarkers &m = ix->H().ma;
    const void *p = nullptr;
    int rc;
    if ((rc = dev_upload(ix, m.start.data(), m.start.size() * 8, &p))) return rc;
    ix->dev.mk_start = static_cast<const uint64_t *>(p);
    if ((rc = dev_upload(ix, m.end.data(), m.end.size() * 8, &p))) return rc;
    ix->dev.mk_end = static_cast<const uint64_t *>(p);
    if ((rc = dev_upload(ix, m.off.data(), m.off.size() * 8, &p))) return rc;
    ix->dev.mk_off = static_cast<const uint64_t *>(p);
    if ((rc = dev_upload(ix, m.vals.data(), m.vals.size() * 8, &p))) return rc;
    ix->dev.mk_vals = static_cast<const uint64_t *>(p);
    ix->dev.mk_nruns = m.start.size();
    ix->dev.mk_bucket = nullptr;
    ix->dev.mk_shift = 0;
    ix->dev.mk_rec = nullptr;
    const uint64_t nruns = m.start.size(), n = ix->H().n;
    if (nruns && nruns < 0xFFFFFFFFull) {
        // about two buckets per run: at_range's two predecessor searches (2 x log2(nruns) dependent
        // loads) become one table read and a scan over the runs of one bucket
        uint32_t shift = 0;
        while (shift < 20 && (n >> shift) > 2 * nruns) ++shift;
        const uint64_t nb = (n >> shift) + 2;
        std::vector<uint32_t> bucket(nb);
        uint64_t j = 0;
        for (uint64_t b = 0; b < nb; ++b) {
            const uint64_t first_row = b << shift;
            while (j < nruns && m.end[j] < first_row) ++j;
            bucket[b] = static_cast<uint32_t>(j);
        }
        if ((rc = dev_upload(ix, bucket.data(), nb * 4, &p))) return rc;
        ix->dev.mk_bucket = static_cast<const uint32_t *>(p);
        ix->dev.mk_shift = shift;
        // the bucket records (rbg_dev.h MkRec): 32 bytes per bucket, i.e. about 64 per run.  RBG_MK_REC=0: the arrays only (A/B, tests)
        // (32 bytes per bucket = about 64 per marker run: only while that is a small part of the device -- at most an eighth of the free HBM and 16 GB; a marker array
        //  of 1e9 runs keeps the 4-byte directory)
        const char *e = std::getenv("RBG_MK_REC");
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const bool rec_fits = nb * sizeof(MkRec) <= std::min<size_t>(free_b / 8, size_t(16) << 30);
        if (shift <= 16 && !(e && e[0] == '0') && (m.vals.size() >> 40) == 0 && rec_fits) {
            std::vector<MkRec> recs(nb);
            for (uint64_t b = 0; b < nb; ++b) {
                MkRec &R = recs[b];
                std::memset(&R, 0, sizeof(R));
                const uint64_t a = bucket[b], first_row = b << shift, end_row = first_row + (uint64_t(1) << shift);
                R.a = static_cast<uint32_t>(a);
                const uint64_t off_a = a < nruns ? m.off[a] : m.vals.size();
                R.off_lo = static_cast<uint32_t>(off_a);
                R.off_hi = static_cast<uint8_t>(off_a >> 32);
                uint32_t k = 0;
                bool over = false;
                for (uint64_t j2 = a; j2 < nruns && m.start[j2] < end_row; ++j2) {
                    const uint64_t c = m.off[j2 + 1] - m.off[j2];
                    if (k == kMkRecRuns || c > 0xFFFF) { over = true; break; }
                    R.s_off[k] = static_cast<uint16_t>(m.start[j2] > first_row ? m.start[j2] - first_row : 0);
                    R.e_off[k] = static_cast<uint16_t>(std::min<uint64_t>(m.end[j2] - first_row, 0xFFFF));   // (end >= first_row: j2 >= a)
                    R.cnt[k] = static_cast<uint16_t>(c);
                    ++k;
                }
                R.nin = over ? static_cast<uint8_t>(kMkRecOverflow) : static_cast<uint8_t>(k);
            }
            if ((rc = dev_upload(ix, recs.data(), nb * sizeof(MkRec), &p))) return rc;
            ix->dev.mk_rec = static_cast<const MkRec *>(p);
        }
    }
    return RBG_OK;
}

FlattenOptions current_options();

// give back the device arrays of the k-mer level `depth` (2..5) -- a level the budget rule drops, or one the run-indexed
// layout has copied out
void release_kmer_level(rbg_index *ix, uint32_t depth) {
    if (depth < 2 || depth - 2 >= ix->kmer_levels.size()) return;
    ComposedLevel &L = ix->kmer_levels[depth - 2];
    for (void *p : {L.ent, L.samp}) {
        if (!p) continue;
        for (size_t i = 0; i < ix->allocs.size(); ++i)
            if (ix->allocs[i].p == p) { ix->hbm_bytes -= ix->allocs[i].bytes; ix->allocs.erase(ix->allocs.begin() + static_cast<std::ptrdiff_t>(i)); break; }
        (void)hipFree(p);
    }
    L = ComposedLevel();
}
std::vector<SymTable> &kmer_level_tables(HostIndex &h, uint32_t depth) { return h.kmer(depth); }
uint32_t depth_of_level(const HostIndex &h, const std::vector<SymTable> *lvl) { return static_cast<uint32_t>(lvl - h.kmer_lv) + 2u; }
void drop_kmer_level(rbg_index *ix, std::vector<SymTable> &lvl) {
    release_kmer_level(ix, depth_of_level(ix->H(), &lvl));
    std::vector<SymTable>().swap(lvl);
}

// Depths 2 .. kmer_deferred composed on the device (k_compose.hip) from the depth-1 tables of the k-mer alphabet and the
// BWT's own runs; the host tables get their metadata (runs, total, F, bucket shift) and pointers into the level arrays.
// Without the memory for it (or with RBG_HOST_COMPOSE=1 at flatten time) the host composes as before.
template <typename P> int compose_on_device_k(rbg_index *ix, uint32_t K);

// Depths 2 .. kmer_deferred on the device; when neither the device (transient HBM: about 100 bytes per piece of the deepest
// intermediate depth) nor the host (24 bytes per run and depth, refused when the container's memory would not hold it) can
// compose that many symbols per step, one symbol less is tried -- said on stderr, and rbg_info reports the depth asked for beside
// the depth kept.  (Round 4: an r = 1e9 index gets 3 symbols per step this way where 5 would need more than the device has.)
template <typename P>
int compose_on_device(rbg_index *ix) {
    HostIndex &h = ix->H();
    const uint32_t M = h.nmajor, K0 = h.kmer_deferred;
    h.kmer_deferred = 0;
    if (M < 1 || K0 < 2) return RBG_OK;
    if (ix->kmer_steps_requested == 0) ix->kmer_steps_requested = K0;
    // How deep is worth composing is decided BEFORE composing (planned_depth): a depth takes minutes and hundreds of GB of transient HBM at r = 1e9,
    // and one the budget rule of upload() then drops -- or whose composition fails after the shallower ones were made -- was composed for nothing.
    // The fallback below still catches an estimate that was too kind.
    uint32_t K_plan = K0;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            free_b = assumed_free_hbm(free_b);
            const int64_t opt_mb = g_opt_hbm_budget_mb.load();
            const double budget = ix->plan_budget ? static_cast<double>(ix->plan_budget) : static_cast<double>(opt_mb > 0 ? static_cast<size_t>(opt_mb) << 20 : default_budget(free_b));
            const bool runs_certain = g_opt_rank_layout.load() == RBG_LAYOUT_RUNS || ix->auto_runs;
            K_plan = planned_depth(static_cast<double>(h.r), h.has_tsa, K0, static_cast<double>(free_b), budget, runs_certain);
            if (K_plan < K0)
                std::fprintf(stderr, "rbg: r = %.3g runs, %.1f GB free, %.1f GB replica budget: composing %u symbol(s) per step, not the %u asked for (estimated: depth %u would "
                                     "hold about %.3g runs; RBG_OPT_HBM_BUDGET_MB / RBG_OPT_RUN_DEPTHS change what fits)\n", static_cast<double>(h.r), free_b / 1e9, budget / 1e9,
                             K_plan, K0, K0, est_depth_runs(static_cast<double>(h.r), K0));
        }
    }
    if (K_plan < 2) return RBG_OK;   // single-symbol steps: nothing to compose
    for (uint32_t K = K_plan; K >= 2; --K) {
        // (a pass that failed partway -- the host fallback included -- must leave nothing of a deeper level behind: levels() and
        //  level_has_data() count what they find)
        for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d) { release_kmer_level(ix, d); std::vector<SymTable>().swap(kmer_level_tables(h, d)); }
        ix->kmer_levels.clear();
        ix->runs_forced = false;
        const int rc = compose_on_device_k<P>(ix, K);
        if (rc != RBG_ENOMEM) return rc;
        std::fprintf(stderr, "rbg: %u symbols per step cannot be composed in the memory there is: trying %u\n", K, K - 1);
        (void)hipGetLastError();
    }
    return RBG_OK;   // single-symbol steps: nothing to compose
}

template <typename P>
int compose_on_device_k(rbg_index *ix, const uint32_t K) {
    HostIndex &h = ix->H();
    const uint32_t M = h.nmajor;
    const FlattenOptions opt = current_options();
    const auto t0 = std::chrono::steady_clock::now();
    struct Hold {
        std::vector<void *> p;
        ~Hold() { for (void *q : p) if (q) (void)hipFree(q); }
        int put(const void *src, size_t bytes, void **out) {
            void *d = nullptr;
            hipError_t e = hipMalloc(&d, bytes ? bytes : 16);
            if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV; }
            p.push_back(d);
            if (bytes && h2d_big(d, src, bytes) != RBG_OK) return RBG_ENODEV;
            *out = d;
            return RBG_OK;
        }
    } hold;
    int rc = RBG_OK;
    ComposeTable major[4];
    for (uint32_t m = 0; m < M && !rc; ++m) {
        const SymTable &t = h.sym[h.major_slot[m]];
        PreparedSym<P> ps;
        prepare_sym<P>(t, h.has_tsa, ps);
        void *de = nullptr, *dsp = nullptr;
        rc = hold.put(ps.ent.data(), ps.ent.size() * sizeof(RunEnt<P>), &de);
        if (!rc && h.has_tsa) rc = hold.put(ps.samp.data(), ps.samp.size() * sizeof(P), &dsp);
        major[m] = ComposeTable{de, dsp, t.nruns, t.total, t.F};
    }
    void *g_start = nullptr, *g_id = nullptr, *g_samp = nullptr;
    if (!rc) {   // depth 1: the BWT runs themselves, id = major index of the head, sample = samples_last_ (SA - 1)
        HostBuf<P> gs(h.r + 1), sp(h.has_tsa ? h.r : 0);
        HostBuf<uint32_t> gi(h.r);
        gs[h.r] = static_cast<P>(h.run_start[h.r]);
        parallel_for(h.r, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t g = b; g < e; ++g) {
                gs[g] = static_cast<P>(h.run_start[g]);
                const uint8_t m = h.major_of[h.run_heads[g]];
                gi[g] = m == 0xFF ? 0xFFFFFFFFu : m;
                if (h.has_tsa) sp[g] = static_cast<P>(h.samples_last[g]);
            }
        });
        rc = hold.put(gs.data(), gs.size() * sizeof(P), &g_start);
        if (!rc) rc = hold.put(gi.data(), gi.size() * 4, &g_id);
        if (!rc && h.has_tsa) rc = hold.put(sp.data(), sp.size() * sizeof(P), &g_samp);
    }
    if (std::getenv("RBG_VERBOSE"))
        std::fprintf(stderr, "rbg:   compose: depth-1 tables and runs converted and copied in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    std::vector<ComposedLevel> levels;
    // When the run-indexed layout is certain (asked for, or not even the single-symbol slot tables fit the budget: the test
    // options_for makes) the depths its depth set leaves out give their arrays back as soon as the next depth is made.
    uint32_t keep_mask = 0;
    {
        bool runs_certain = g_opt_rank_layout.load() == RBG_LAYOUT_RUNS || ix->auto_runs;
        if (!runs_certain && layout_automatic()) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const int64_t opt_mb = g_opt_hbm_budget_mb.load();
                const double budget = static_cast<double>(opt_mb > 0 ? static_cast<size_t>(opt_mb) << 20 : default_budget(free_b));
                const double lvl1 = static_cast<double>(h.sigma) * static_cast<double>((h.n >> kMaxNarrowShift) + 2) * (sizeof(RankSlot) + sizeof(uint32_t)) +
                                    static_cast<double>((h.n >> 6) + 2) * (4.0 * sizeof(P) + 4);
                runs_certain = lvl1 > budget;
            }
        }
        if (runs_certain && h.sigma <= static_cast<uint32_t>(kLdsSyms)) {
            keep_mask = g_opt_run_depths.load() ? static_cast<uint32_t>(g_opt_run_depths.load()) | 1u : default_depth_mask(K);
            keep_mask |= 1u << (K - 1);
        }
    }
    // (with the run-indexed layout certain, the composition also frees its inputs as soon as they have been read: nothing
    //  after it needs the depth-1 lists in this form -- upload_tables_runs2 builds depth 1 from the host tables)
    bool released[2] = {false, false};
    if (!rc) rc = compose_levels_device(sizeof(P), h.n, M, major, g_start, static_cast<const uint32_t *>(g_id), g_samp, h.r, K, h.has_tsa, levels, nullptr, keep_mask,
                                        keep_mask ? released : nullptr);
    if (released[0])
        for (void *&held : hold.p)
            if (held == g_start || held == g_id || held == g_samp) held = nullptr;
    if (released[1])
        for (uint32_t m = 0; m < M; ++m) {
            for (void *&held : hold.p)
                if (held == major[m].ent || held == major[m].samp) held = nullptr;
            major[m].ent = major[m].samp = nullptr;
        }
    if (rc == RBG_EARG) {   // 2^32 pieces in one depth (r beyond about 1.7e9 at five symbols): the device sweeps index pieces with 32 bits, the host composition does not
        std::fprintf(stderr, "rbg: a k-mer depth has 2^32 pieces or more: the device composition indexes them with 32 bits\n");
        rc = RBG_ENOMEM;
    }
    if (rc == RBG_ENOMEM || rc == RBG_ENODEV) {   // not enough HBM for the sweeps' temporaries: the host composes instead
        for (ComposedLevel &L : levels) { if (L.ent) (void)hipFree(L.ent); if (L.samp) (void)hipFree(L.samp); }
        (void)hipGetLastError();
        // the host composition holds every depth as three 8-byte vectors per run: 24 bytes x (about 1.6 + 2.1 + 2.6 + 3.2) runs of the
        // BWT at pangenome scale -- it must not be what exhausts the machine (a container's memory limit kills the process, and on a
        // shared box more than that)
        const double need_host = 24.0 * 3.3 * static_cast<double>(K - 1) * static_cast<double>(h.r);
        const double have_host = host_memory_available();
        if (need_host > 0.8 * have_host) {
            std::fprintf(stderr, "rbg: composing the k-mer tables on the device failed (%s), and the host composition would need about %.0f GB of the %.0f GB "
                                 "this process may still use: not attempted (fewer symbols per step -- RBG_OPT_KMER_STEPS -- need less of both)\n",
                         rbg_strerror(rc), need_host / 1e9, have_host / 1e9);
            return RBG_ENOMEM;
        }
        std::fprintf(stderr, "rbg: composing the k-mer tables on the device failed (%s): composing on the host\n", rbg_strerror(rc));
        return compose_kmer_tables_host(h, static_cast<int>(K), opt);
    }
    if (rc) {
        for (ComposedLevel &L : levels) { if (L.ent) (void)hipFree(L.ent); if (L.samp) (void)hipFree(L.samp); }
        return rc;
    }
    ix->kmer_levels = std::move(levels);
    ix->runs_forced = keep_mask != 0;
    // the depth-1 run lists of the k-mer alphabet are on the device in the very form the slot tables are built from
    // (commit_sym): they stay, instead of being converted and copied a second time (5 + 2.5 GB at r = 3e8)
    for (uint32_t m = 0; m < M && !released[1]; ++m) {
        SymTable &t = h.sym[h.major_slot[m]];
        for (void *q : {const_cast<void *>(major[m].ent), const_cast<void *>(major[m].samp)}) {
            if (!q) continue;
            for (void *&held : hold.p)
                if (held == q) held = nullptr;
            const size_t bytes = q == major[m].ent ? (t.nruns + 1) * sizeof(RunEnt<P>) : std::max<size_t>(16, t.nruns * sizeof(P));
            ix->allocs.push_back({q, bytes});
            ix->hbm_bytes += bytes;
        }
        t.dev_ent = major[m].ent;
        t.dev_samp = major[m].samp;
    }
    for (uint32_t d = 2; d <= K; ++d) {
        ComposedLevel &L = ix->kmer_levels[d - 2];
        if (L.ent) {   // (a depth outside the run-indexed layout's depth set has given its arrays back already: metadata only)
            ix->allocs.push_back({L.ent, (L.entries + 2) * sizeof(RunEnt<P>)});
            ix->hbm_bytes += (L.entries + 2) * sizeof(RunEnt<P>);
        }
        if (L.samp) { ix->allocs.push_back({L.samp, (L.entries + 2) * sizeof(P)}); ix->hbm_bytes += (L.entries + 2) * sizeof(P); }
        std::vector<SymTable> &tabs = kmer_level_tables(h, d);
        tabs.assign(L.nruns.size(), SymTable());
        for (size_t t = 0; t < tabs.size(); ++t) {
            SymTable &st = tabs[t];
            st.byte = h.major_byte[t % M];
            st.nruns = L.nruns[t];
            st.total = L.total[t];
            st.F = L.F[t];
            st.shift = kmer_table_shift(h.n, st.nruns, d, opt);
            if (st.shift > 12 || (st.shift > 8 && (h.n >> 40))) return RBG_EARG;  // wide buckets carry 40-bit ranks (rbg_dev.h)
            if (st.nruns >= 0xFFFFFFF0ull) return RBG_EARG;
            st.dev_ent = L.ent ? static_cast<const char *>(L.ent) + L.first[t] * sizeof(RunEnt<P>) : nullptr;
            st.dev_samp = L.samp ? static_cast<const char *>(L.samp) + L.first[t] * sizeof(P) : nullptr;
        }
    }
    if (std::getenv("RBG_VERBOSE"))
        std::fprintf(stderr, "rbg: k-mer tables composed on the device %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return RBG_OK;
}

// does k-mer depth d (2..5) still have its run lists -- on the host, or composed on the device and not given back?
bool level_has_data(const rbg_index *ix, uint32_t d) {
    const std::vector<SymTable> &T = ix->H().kmer(d);
    if (T.empty()) return false;
    if (d - 2 < ix->kmer_levels.size() && ix->kmer_levels[d - 2].ent) return true;
    for (const SymTable &t : T)
        if (t.start.size() == t.nruns + 1) return true;
    return false;
}

bool compose_deferred(int device);
inline int levels_of(const HostIndex &h) { return static_cast<int>(h.kmer_levels()); }

}  // namespace
