// capi/device_api.ipp -- the device-resident entry points (*_dev: HBM in, HBM out, the caller's stream).  Part of rbg_capi.hip.
extern "C" {
// ---- device-resident entry points ------------------------------------------------------------------

int rbg_find_range_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t *d_lo,
                       uint64_t *d_hi, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N && (!d_seqs || !d_off || !d_lo || !d_hi)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;  // reads are fetched as aligned 16-byte chunks
    return launch_find_range(ix->dev, ix->cfg, d_seqs, d_off, N, d_lo, d_hi, nullptr, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_find_range_w_toehold_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                                 uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_ssamp, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_seqs || !d_off || !d_lo || !d_hi || !d_ssamp)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    return launch_find_range(ix->dev, ix->cfg, d_seqs, d_off, N, d_lo, d_hi, d_ssamp, stream) ? RBG_ENODEV : RBG_OK;
    });
}

static_assert(RBG_SEARCH_STATS == kStatSearchN && RBG_LOCATE_STATS == kStatLocateN, "rbg.h mirrors rbg_dev.h");
static_assert(RBG_SS_CHUNKS == kStChunks && RBG_SS_SYMBOLS == kStSymbols && RBG_LS_LOCS == kLsLocs, "rbg.h mirrors rbg_dev.h");

int rbg_find_range_stats_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t *d_lo,
                             uint64_t *d_hi, uint64_t *d_ssamp, uint64_t *d_stats, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (d_ssamp && !ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!d_stats || (N && (!d_seqs || !d_off || !d_lo || !d_hi))) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15) return RBG_EARG;
    return launch_find_range_stats(ix->dev, ix->cfg, d_seqs, d_off, N, d_lo, d_hi, d_ssamp, reinterpret_cast<unsigned long long *>(d_stats), stream)
               ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_locate_fill_stats_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                              uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const void *d_order, uint64_t *d_stats,
                              void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!d_stats || (N && (!d_lo || !d_hi || !d_k || !d_loc_off || !d_locs))) return RBG_EARG;
    return launch_locate_fill(ix->dev, ix->cfg, d_lo, d_hi, d_k, N, max_hits, d_loc_off, d_locs, nullptr, d_order, stream,
                              reinterpret_cast<unsigned long long *>(d_stats)) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_sample_reads_dev(const uint8_t *d_text, uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first_read,
                         uint64_t N, uint32_t sub_ppm, uint8_t *d_seqs, uint64_t *d_off, uint64_t *d_start, void *stream) {
    return guarded([&]() -> int {
    if (!d_text || !d_seqs || !d_off) return RBG_EARG;
    if (m == 0 || m > L || H == 0 || L > unit || sub_ppm > 1000000u) return RBG_EARG;
    return launch_sample_reads(d_text, unit, H, L, m, seed, first_read, N, sub_ppm, d_seqs, d_off, d_start, stream) ? RBG_ENODEV : RBG_OK;
    });
}

// ---- packed reads (device API) ------------------------------------------------------------------
size_t rbg_pack_ws_bytes(uint64_t N, uint64_t total_bytes) { return pack_ws_bytes(N, total_bytes); }

int rbg_sample_reads_pangenome_dev(const uint8_t *d_base, const uint64_t *d_sites, const uint8_t *d_alt, const uint8_t *d_G, uint64_t S, const uint32_t *d_site_dir,
                                   uint32_t site_dir_shift, uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first_read, uint64_t N,
                                   uint32_t sub_ppm, uint8_t *d_seqs, uint64_t *d_off, uint64_t *d_start, void *stream) {
    return guarded([&]() -> int {
    if (!d_base || !d_seqs || !d_off || (S && (!d_sites || !d_alt || !d_G))) return RBG_EARG;
    if (m == 0 || m > L || H == 0 || L > unit || sub_ppm > 1000000u || (d_site_dir && (site_dir_shift > 40 || S >= 0xFFFFFFFFull))) return RBG_EARG;
    return launch_sample_reads_pg(d_base, d_sites, d_alt, d_G, S, d_site_dir, site_dir_shift, unit, H, L, m, seed, first_read, N, sub_ppm, d_seqs, d_off, d_start, stream)
               ? RBG_ENODEV : RBG_OK;
    });
}

static int packed_args_ok(const rbg_index *ix, const void *d_ws, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                          uint64_t total_bytes) {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N && (!d_ws || !d_seqs || !d_off)) return RBG_EARG;
    if (reinterpret_cast<uintptr_t>(d_seqs) & 15 || reinterpret_cast<uintptr_t>(d_ws) & 15) return RBG_EARG;
    if (total_bytes / 64 + N + 1 >= (uint64_t(1) << 32)) return RBG_EARG;  // chunk indices are 32-bit
    return RBG_OK;
}

int rbg_pack_reads_dev(rbg_index *ix, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t total_bytes,
                       void *d_ws, size_t ws_bytes, void *stream) {
    return guarded([&]() -> int {
    int rc = packed_args_ok(ix, d_ws, d_seqs, d_off, N, total_bytes);
    if (rc) return rc;
    if (ws_bytes < pack_ws_bytes(N, total_bytes)) return RBG_EARG;
    return launch_pack_reads(ix->dev, ix->cfg, d_seqs, d_off, N, total_bytes, d_ws, ws_bytes, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_find_range_packed_dev(rbg_index *ix, const void *d_ws, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                              uint64_t total_bytes, uint64_t *d_lo, uint64_t *d_hi, void *stream) {
    return guarded([&]() -> int {
    int rc = packed_args_ok(ix, d_ws, d_seqs, d_off, N, total_bytes);
    if (rc) return rc;
    if (N && (!d_lo || !d_hi)) return RBG_EARG;
    return launch_find_range_packed(ix->dev, ix->cfg, d_ws, d_seqs, d_off, N, total_bytes, d_lo, d_hi, nullptr, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_find_range_w_toehold_packed_dev(rbg_index *ix, const void *d_ws, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                                        uint64_t total_bytes, uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_ssamp, void *stream) {
    return guarded([&]() -> int {
    int rc = packed_args_ok(ix, d_ws, d_seqs, d_off, N, total_bytes);
    if (rc) return rc;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_lo || !d_hi || !d_ssamp)) return RBG_EARG;
    return launch_find_range_packed(ix->dev, ix->cfg, d_ws, d_seqs, d_off, N, total_bytes, d_lo, d_hi, d_ssamp, stream) ? RBG_ENODEV : RBG_OK;
    });
}

size_t rbg_locate_plan_tmp_bytes(uint64_t N) { return scan_tmp_bytes(N); }

int rbg_locate_plan_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N, uint64_t max_hits,
                        uint64_t *d_loc_off, void *d_tmp, size_t tmp_bytes, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!d_loc_off || (N && (!d_lo || !d_hi || !d_tmp))) return RBG_EARG;
    return launch_locate_plan(ix->dev, ix->cfg, d_lo, d_hi, N, max_hits, d_loc_off, d_tmp, tmp_bytes, stream) ? RBG_ENODEV : RBG_OK;
    });
}

size_t rbg_locate_order_ws_bytes(uint64_t N) { return locate_order_ws_bytes(N); }

int rbg_locate_order_dev(rbg_index *ix, const uint64_t *d_k, uint64_t N, void *d_ws, size_t ws_bytes, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_k || !d_ws)) return RBG_EARG;
    if (N >= 0xFFFFFFFFull || ws_bytes < locate_order_ws_bytes(N) || (reinterpret_cast<uintptr_t>(d_ws) & 255)) return RBG_EARG;
    return launch_locate_order(ix->dev, ix->cfg, d_k, N, d_ws, ws_bytes, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_locate_fill_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                        uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const void *d_order, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!d_lo || !d_hi || !d_k || !d_loc_off || !d_locs)) return RBG_EARG;
    return launch_locate_fill(ix->dev, ix->cfg, d_lo, d_hi, d_k, N, max_hits, d_loc_off, d_locs, nullptr, d_order, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_locate_fill_dev32(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                          uint64_t max_hits, const uint64_t *d_loc_off, uint32_t *d_locs32, const void *d_order, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (ix->H().pos_bytes != 4) return RBG_EARG;   // text positions beyond 32 bits: rbg_locate_fill_dev
    if (N && (!d_lo || !d_hi || !d_k || !d_loc_off || !d_locs32)) return RBG_EARG;
    return launch_locate_fill(ix->dev, ix->cfg, d_lo, d_hi, d_k, N, max_hits, d_loc_off, nullptr, nullptr, d_order, stream, nullptr, d_locs32) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_markers_plan_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N, uint64_t *d_mk_off,
                         void *d_tmp, size_t tmp_bytes, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_ma) return RBG_ENOTLOADED;
    if (!d_mk_off || (N && (!d_lo || !d_hi || !d_tmp))) return RBG_EARG;
    return launch_markers_plan(ix->dev, ix->cfg, d_lo, d_hi, N, d_mk_off, d_tmp, tmp_bytes, stream) ? RBG_ENODEV : RBG_OK;
    });
}

int rbg_markers_fill_dev(rbg_index *ix, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N,
                         const uint64_t *d_mk_off, uint64_t *d_mk, void *stream) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_ma) return RBG_ENOTLOADED;
    if (N && (!d_lo || !d_hi || !d_mk_off || !d_mk)) return RBG_EARG;
    return launch_markers_fill(ix->dev, ix->cfg, d_lo, d_hi, N, d_mk_off, d_mk, stream) ? RBG_ENODEV : RBG_OK;
    });
}

}  // extern "C"
