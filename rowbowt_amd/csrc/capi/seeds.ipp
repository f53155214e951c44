// capi/seeds.ipp -- markers, marker seeds and greedy seeding (rb_markers' path), device and host entry points.  Part of rbg_capi.hip.
extern "C" {
int rbg_markers_at(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, uint64_t N, uint64_t *mk_off, uint64_t **mk) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_ma) return RBG_ENOTLOADED;
    if (!mk_off || !mk || (N && (!lo || !hi))) return RBG_EARG;
    *mk = nullptr;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    DevBuf dlo, dhi, doff, dtmp;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    int rc;
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = doff.alloc((N + 1) * 8)) || (rc = dtmp.alloc(tmp_bytes)))
        return rc;
    if (N) {
        HIP_TRY(hipMemcpyAsync(dlo.p, lo, N * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dhi.p, hi, N * 8, hipMemcpyHostToDevice, st));
    }
    if (launch_markers_plan(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, doff.as<uint64_t>(), dtmp.p, tmp_bytes, st))
        return RBG_ENODEV;
    return ragged_finish(N, doff, mk_off, mk, st, [&](uint64_t *d_vals) {
        return launch_markers_fill(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, doff.as<uint64_t>(), d_vals, st)
                   ? RBG_ENODEV : RBG_OK;
    });
    });
}

int rbg_find_range_w_markers(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize,
                             uint64_t max_range, uint64_t *lo, uint64_t *hi, uint64_t *mk_off, uint64_t **mk) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_ma) return RBG_ENOTLOADED;  // reference: "warning: no marker array found!", default LFData
    if (!mk_off || !mk || wsize == 0 || (N && (!lo || !hi || !off))) return RBG_EARG;
    *mk = nullptr;
    int rc = check_offsets(off, N);
    if (rc) return rc;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    ReadBatch rb;
    if ((rc = rb.stage(seqs, off, N, st))) return rc;
    DevBuf dlo, dhi, doff, dtmp;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = doff.alloc((N + 1) * 8)) || (rc = dtmp.alloc(tmp_bytes)))
        return rc;
    if (launch_find_range_markers_plan(ix->dev, ix->cfg, rb.seqs.as<uint8_t>(), rb.off.as<uint64_t>(), N, wsize, max_range,
                                       dlo.as<uint64_t>(), dhi.as<uint64_t>(), doff.as<uint64_t>(), dtmp.p, tmp_bytes, st))
        return RBG_ENODEV;
    if (N) {
        HIP_TRY(hipMemcpyAsync(lo, dlo.p, N * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hi, dhi.p, N * 8, hipMemcpyDeviceToHost, st));
    }
    return ragged_finish(N, doff, mk_off, mk, st, [&](uint64_t *d_vals) {
        return launch_find_range_markers_fill(ix->dev, ix->cfg, rb.seqs.as<uint8_t>(), rb.off.as<uint64_t>(), N, wsize,
                                              max_range, doff.as<uint64_t>(), d_vals, st) ? RBG_ENODEV : RBG_OK;
    });
    });
}

// ---- marker seeds (next-row f4): get_markers_greedy_seeding, rowbowt.hpp:406-482 ---------------------

int rbg_marker_seeds_plan_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                              uint64_t max_range, uint64_t ftab_k, uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes,
                              void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!d_seed_off || !d_mk_off || (N && (!d_seqs || !d_off))) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    if (tmp_bytes < scan_tmp_bytes(N) || (N && !d_tmp)) return RBG_EARG;
    return launch_marker_seeds_plan(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, ftab_k, d_seed_off, d_mk_off, d_tmp, tmp_bytes,
                                    stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_marker_seeds_fill_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                              uint64_t max_range, uint64_t ftab_k, const uint64_t *d_seed_off, const uint64_t *d_mk_off,
                              rbg_marker_seed_t *d_seeds, uint64_t *d_mk, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N && (!d_seqs || !d_off || !d_seed_off || !d_mk_off || !d_seeds)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    return launch_marker_seeds_fill(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, ftab_k, d_seed_off, d_mk_off,
                                    reinterpret_cast<uint64_t *>(d_seeds), d_mk, stream) ? RBG_ENODEV : RBG_OK;
    });
}

// The same two phases with a LOG between them (rbg_dev.h SeedLog): the plan leaves every sequence's seed records and the
// places of its markers in d_log, the fill copies from there and walks only the sequences that exceeded their quota.
size_t rbg_marker_seeds_log_bytes(const rbg_index *ix, uint64_t N, uint32_t seeds_per_read) {
    if (!ix) return 0;
    if (seeds_per_read == 0) seeds_per_read = kSeedLogSeedsDefault;
    if (seeds_per_read < 2) seeds_per_read = 2;
    if (seeds_per_read > 255) seeds_per_read = 255;
    return seed_log_bytes(N, ix->H().pos_bytes, seeds_per_read);
}

int rbg_marker_seeds_plan_log_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                                  uint64_t max_range, uint64_t ftab_k, uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes,
                                  void *d_log, size_t log_bytes, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!d_seed_off || !d_mk_off || (N && (!d_seqs || !d_off))) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    if (tmp_bytes < scan_tmp_bytes(N) || (N && !d_tmp)) return RBG_EARG;
    if (N && !make_seed_log(d_log, log_bytes, N, ix->H().pos_bytes).base) return RBG_EARG;   // unaligned, or no room for two seeds per sequence
    return launch_marker_seeds_plan(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, ftab_k, d_seed_off, d_mk_off, d_tmp, tmp_bytes,
                                    stream, d_log, log_bytes) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_marker_seeds_fill_log_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                                  uint64_t max_range, uint64_t ftab_k, const uint64_t *d_seed_off, const uint64_t *d_mk_off,
                                  rbg_marker_seed_t *d_seeds, uint64_t *d_mk, void *d_log, size_t log_bytes, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N && (!d_seqs || !d_off || !d_seed_off || !d_mk_off || !d_seeds)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15 || reinterpret_cast<uintptr_t>(d_seeds) & 15) return RBG_EARG;
    if (N && !make_seed_log(d_log, log_bytes, N, ix->H().pos_bytes).base) return RBG_EARG;
    return launch_marker_seeds_fill(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, ftab_k, d_seed_off, d_mk_off,
                                    reinterpret_cast<uint64_t *>(d_seeds), d_mk, stream, d_log, log_bytes) ? RBG_ENODEV : RBG_OK;
    });
}

static int marker_seeds_host(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize, uint64_t max_range,
                             uint64_t ftab_k, uint64_t *seed_off, rbg_marker_seed_t **seeds, uint64_t **mk);

struct SeedsReq : CombineReq {
    const uint8_t *seq = nullptr;
    uint64_t len = 0, wsize = 0, max_range = 0, ftab_k = 0;
    uint64_t nseeds = 0;
    rbg_marker_seed_t *seeds = nullptr;
    uint64_t *mk = nullptr;
};

// one read through the combiner (get_markers_greedy_seeding(query, wsize, max_range, fn) from a thread pool):
// requests with the same parameters share a launch; each gets its own slice, its marker offsets starting at 0
static int marker_seeds_one(rbg_index *ix, const uint8_t *seq, uint64_t len, uint64_t wsize, uint64_t max_range, uint64_t ftab_k,
                            uint64_t *seed_off, rbg_marker_seed_t **seeds, uint64_t **mk) {
    SeedsReq mine;
    mine.seq = seq; mine.len = len; mine.wsize = wsize; mine.max_range = max_range; mine.ftab_k = ftab_k;
    const int rc = combine_submit(ix, ix->comb_seeds, mine,
        [](const SeedsReq &a, const SeedsReq &b) { return a.wsize == b.wsize && a.max_range == b.max_range && a.ftab_k == b.ftab_k; },
        [&](std::vector<SeedsReq *> &batch) {
            const uint64_t K = batch.size();
            std::vector<uint64_t> off(K + 1, 0), soff(K + 1, 0);
            for (uint64_t i = 0; i < K; ++i) off[i + 1] = off[i] + batch[i]->len;
            std::vector<uint8_t> flat(off[K] + 1);
            for (uint64_t i = 0; i < K; ++i)
                if (batch[i]->len) std::memcpy(flat.data() + off[i], batch[i]->seq, batch[i]->len);
            rbg_marker_seed_t *all = nullptr;
            uint64_t *allmk = nullptr;
            int rc2 = marker_seeds_host(ix, flat.data(), off.data(), K, mine.wsize, mine.max_range, mine.ftab_k, soff.data(), &all, &allmk);
            if (!rc2 && K == 1) {   // nothing to split
                batch[0]->nseeds = soff[1];
                batch[0]->seeds = all;
                batch[0]->mk = allmk;
                all = nullptr;
                allmk = nullptr;
            } else if (!rc2) {
                for (uint64_t i = 0; i < K && !rc2; ++i) {
                    const uint64_t s0 = soff[i], s1 = soff[i + 1];
                    const uint64_t m0 = s1 > s0 ? all[s0].mk_begin : 0, m1 = s1 > s0 ? all[s1 - 1].mk_end : 0;
                    auto *hs = static_cast<rbg_marker_seed_t *>(std::malloc(std::max<size_t>(1, (s1 - s0) * sizeof(rbg_marker_seed_t))));
                    auto *hm = static_cast<uint64_t *>(std::malloc(std::max<size_t>(1, (m1 - m0) * 8)));
                    if (!hs || !hm) { std::free(hs); std::free(hm); rc2 = RBG_ENOMEM; break; }   // (plain malloc blocks)
                    for (uint64_t j = s0; j < s1; ++j) {
                        hs[j - s0] = all[j];
                        hs[j - s0].mk_begin -= m0;
                        hs[j - s0].mk_end -= m0;
                    }
                    if (m1 > m0) std::memcpy(hm, allmk + m0, (m1 - m0) * 8);
                    batch[i]->nseeds = s1 - s0;
                    batch[i]->seeds = hs;
                    batch[i]->mk = hm;
                }
            }
            rbg_free_buffer(all);
            rbg_free_buffer(allmk);
            if (rc2)
                for (SeedsReq *r : batch) { rbg_free_buffer(r->seeds); rbg_free_buffer(r->mk); r->seeds = nullptr; r->mk = nullptr; }
            for (SeedsReq *r : batch) r->rc = rc2;
        });
    if (rc) return rc;
    seed_off[0] = 0;
    seed_off[1] = mine.nseeds;
    *seeds = mine.seeds;
    *mk = mine.mk;
    return RBG_OK;
}

int rbg_get_markers_greedy_seeding(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize,
                                   uint64_t max_range, uint64_t ftab_k, uint64_t *seed_off, rbg_marker_seed_t **seeds, uint64_t **mk) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!seed_off || !seeds || !mk || (N && !off)) return RBG_EARG;
    *seeds = nullptr;
    *mk = nullptr;
    int rc = check_offsets(off, N);
    if (rc) return rc;
    if (N == 1 && combine_enabled()) return marker_seeds_one(ix, seqs, off[1], wsize, max_range, ftab_k, seed_off, seeds, mk);
    return marker_seeds_host(ix, seqs, off, N, wsize, max_range, ftab_k, seed_off, seeds, mk);
    });
}

static int marker_seeds_host(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize, uint64_t max_range,
                             uint64_t ftab_k, uint64_t *seed_off, rbg_marker_seed_t **seeds, uint64_t **mk) {
    {
    int rc;
    *seeds = nullptr;
    *mk = nullptr;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    ReadBatch rb;
    if ((rc = rb.stage(seqs, off, N, st))) return rc;
    DevBuf dsoff, dmoff, dtmp, dseeds, dmk, dlog;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    if ((rc = dsoff.alloc((N + 1) * 8)) || (rc = dmoff.alloc((N + 1) * 8)) || (rc = dtmp.alloc(tmp_bytes))) return rc;
    // the log between the two phases (one walk instead of two); without the memory for it the fill pass walks again
    size_t log_bytes = seed_log_bytes(N, ix->H().pos_bytes, kSeedLogSeedsDefault);
    if (dlog.alloc(log_bytes)) log_bytes = 0;
    if (launch_marker_seeds_plan(ix->dev, ix->cfg, rb.seqs.as<uint8_t>(), rb.off.as<uint64_t>(), N, wsize, max_range, ftab_k,
                                 dsoff.as<uint64_t>(), dmoff.as<uint64_t>(), dtmp.p, tmp_bytes, st, log_bytes ? dlog.p : nullptr, log_bytes))
        return RBG_ENODEV;
    uint64_t total_mk = 0;
    HIP_TRY(hipMemcpyAsync(seed_off, dsoff.p, (N + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&total_mk, dmoff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t total_seeds = seed_off[N];
    auto *h_seeds = static_cast<rbg_marker_seed_t *>(alloc_result(total_seeds * sizeof(rbg_marker_seed_t)));
    auto *h_mk = static_cast<uint64_t *>(alloc_result(total_mk * 8));
    if (!h_seeds || !h_mk) { rbg_free_buffer(h_seeds); rbg_free_buffer(h_mk); return RBG_ENOMEM; }
    rc = RBG_OK;
    if (total_seeds) {
        if (!(rc = dseeds.alloc(total_seeds * sizeof(rbg_marker_seed_t))) && !(rc = dmk.alloc(total_mk ? total_mk * 8 : 8))) {
            if (launch_marker_seeds_fill(ix->dev, ix->cfg, rb.seqs.as<uint8_t>(), rb.off.as<uint64_t>(), N, wsize, max_range, ftab_k,
                                         dsoff.as<uint64_t>(), dmoff.as<uint64_t>(), dseeds.as<uint64_t>(), dmk.as<uint64_t>(), st,
                                         log_bytes ? dlog.p : nullptr, log_bytes))
                rc = RBG_ENODEV;
            if (!rc) rc = d2h_result(h_seeds, dseeds.p, total_seeds * sizeof(rbg_marker_seed_t), st);
            if (!rc && total_mk) rc = d2h_result(h_mk, dmk.p, total_mk * 8, st);
        }
    }
    if (rc) { rbg_free_buffer(h_seeds); rbg_free_buffer(h_mk); return rc; }
    *seeds = h_seeds;
    *mk = h_mk;
    return RBG_OK;
    }
}

// ---- greedy seeding (next-row f4) -----------------------------------------------------------------

int rbg_greedy_longest_seed_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t min_length,
                                uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_qstart, uint64_t *d_qend, uint64_t *d_ssamp,
                                void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_seqs || !d_off || !d_lo || !d_hi || !d_qstart || !d_qend || !d_ssamp)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    return launch_greedy_seed(ix->dev, ix->cfg, d_seqs, d_off, N, min_length, d_lo, d_hi, d_qstart, d_qend, d_ssamp, stream)
               ? RBG_ENODEV : RBG_OK;
    });
}

// The instrumented instantiations of the seeding kernels (run-indexed layout; include/rbg.h RBG_SEED_STATS): the same walks and outputs, plus what
// they touched.  bench.py prices the kernels' rooflines from these sums.
int rbg_greedy_longest_seed_stats_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t min_length,
                                      uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_qstart, uint64_t *d_qend, uint64_t *d_ssamp, uint64_t *d_stats,
                                      void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!d_stats || (N && (!d_seqs || !d_off || !d_lo || !d_hi || !d_qstart || !d_qend || !d_ssamp))) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    if (ix->dev.layout != RBG_LAYOUT_RUNS) return RBG_EARG;
    return launch_greedy_seed(ix->dev, ix->cfg, d_seqs, d_off, N, min_length, d_lo, d_hi, d_qstart, d_qend, d_ssamp, stream,
                              reinterpret_cast<unsigned long long *>(d_stats)) ? RBG_ENODEV : RBG_OK;
    });
}

// plan (count walk + scans) and fill (second walk) of the marker seeds in one call, both instrumented, sums added to d_stats
int rbg_marker_seeds_stats_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize, uint64_t max_range,
                               uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes, rbg_marker_seed_t *d_seeds, uint64_t *d_mk,
                               uint64_t *d_stats, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!d_stats || !d_seed_off || !d_mk_off || (N && (!d_seqs || !d_off || !d_seeds))) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    if (tmp_bytes < scan_tmp_bytes(N) || (N && !d_tmp)) return RBG_EARG;
    if (ix->dev.layout != RBG_LAYOUT_RUNS) return RBG_EARG;
    unsigned long long *st = reinterpret_cast<unsigned long long *>(d_stats);
    if (launch_marker_seeds_plan(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, 0, d_seed_off, d_mk_off, d_tmp, tmp_bytes, stream, nullptr, 0, st)) return RBG_ENODEV;
    return launch_marker_seeds_fill(ix->dev, ix->cfg, d_seqs, d_off, N, wsize, max_range, 0, d_seed_off, d_mk_off, reinterpret_cast<uint64_t *>(d_seeds), d_mk,
                                    stream, nullptr, 0, st) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_locate_fill_offset_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                               uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const uint64_t *d_sub,
                               const void *d_order, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_lo || !d_hi || !d_k || !d_loc_off || !d_locs)) return RBG_EARG;
    return launch_locate_fill(ix->dev, ix->cfg, d_lo, d_hi, d_k, N, max_hits, d_loc_off, d_locs, d_sub, d_order, stream) ? RBG_ENODEV : RBG_OK;
    });
}

static int greedy_host(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t min_length,
                       uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, bool locate, uint64_t max_hits,
                       uint64_t *loc_off, uint64_t **locs) {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && !off) return RBG_EARG;
    int rc = check_offsets(off, N);
    if (rc) return rc;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    ReadBatch rb;
    if ((rc = rb.stage(seqs, off, N, st))) return rc;
    DevBuf d[5], doff, dtmp;
    for (auto &b : d)
        if ((rc = b.alloc(N * 8))) return rc;
    if (launch_greedy_seed(ix->dev, ix->cfg, rb.seqs.as<uint8_t>(), rb.off.as<uint64_t>(), N, min_length, d[0].as<uint64_t>(),
                           d[1].as<uint64_t>(), d[2].as<uint64_t>(), d[3].as<uint64_t>(), d[4].as<uint64_t>(), st))
        return RBG_ENODEV;
    uint64_t *outs[5] = {lo, hi, qs, qe, ss};
    for (int a = 0; a < 5; ++a)
        if (outs[a] && N) HIP_TRY(hipMemcpyAsync(outs[a], d[a].p, N * 8, hipMemcpyDeviceToHost, st));
    if (!locate) {
        HIP_TRY(hipStreamSynchronize(st));
        return RBG_OK;
    }
    const size_t tmp_bytes = scan_tmp_bytes(N);
    if ((rc = doff.alloc((N + 1) * 8)) || (rc = dtmp.alloc(tmp_bytes))) return rc;
    if (launch_locate_plan(ix->dev, ix->cfg, d[0].as<uint64_t>(), d[1].as<uint64_t>(), N, max_hits, doff.as<uint64_t>(), dtmp.p, tmp_bytes, st))
        return RBG_ENODEV;
    DevBuf dord;
    const void *order = nullptr;
    if ((rc = make_order(ix, d[4].as<uint64_t>(), N, dord, st, &order))) return rc;
    return ragged_finish(N, doff, loc_off, locs, st, [&](uint64_t *d_vals) {
        return launch_locate_fill(ix->dev, ix->cfg, d[0].as<uint64_t>(), d[1].as<uint64_t>(), d[4].as<uint64_t>(), N, max_hits,
                                  doff.as<uint64_t>(), d_vals, d[2].as<uint64_t>(), order, st) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_greedy_longest_seed(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t min_length,
                            uint64_t *lo, uint64_t *hi, uint64_t *qstart, uint64_t *qend, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (N && (!lo || !hi || !qstart || !qend || !ssamp)) return RBG_EARG;
    return greedy_host(ix, seqs, off, N, min_length, lo, hi, qstart, qend, ssamp, false, 0, nullptr, nullptr);
    });
}

int rbg_find_locs_greedy_seeding(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t min_length,
                                 uint64_t max_hits, uint64_t *loc_off, uint64_t **locs) {
    return guarded([&]() -> int {
    if (!loc_off || !locs) return RBG_EARG;
    *locs = nullptr;
    return greedy_host(ix, seqs, off, N, min_length, nullptr, nullptr, nullptr, nullptr, nullptr, true, max_hits, loc_off, locs);
    });
}

}  // extern "C"
