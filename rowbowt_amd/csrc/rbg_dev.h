// rbg_dev.h -- device-visible description of the HBM-resident index replica (shared by the
// C-ABI translation unit and the kernels).  Layout rationale: DESIGN.md "Data layout in HBM".
#pragma once

#include <cstddef>
#include <cstdint>

namespace rbg {

// One run of one symbol.  P = uint32_t when n < 2^32-1, else uint64_t.
template <typename P>
struct RunEnt {
    P start;  // BWT position where the run begins
    P cum;    // occurrences of the symbol before the run
};

// One phi predecessor record (ToeholdSA::phi, toehold_sa.hpp:56-72), pre-joined:
// base = samples_last_[pred_to_run_[j] - 1].
template <typename P>
struct PhiEnt {
    P pos;
    P base;
};

// First level of every predecessor search, direct-addressed by (position >> shift): one record
// per bucket of 2^shift BWT positions that already CONTAINS the answer in the common case, so a
// rank is ONE aligned 8-word load instead of "bucket word -> entries" (two dependent gathers).
//   prev      = last run of the symbol starting strictly before the bucket (start = kSent if none)
//   e0, e1    = first two runs starting inside the bucket (start = kSent if absent)
//   cum of an absent entry = cum of the run that would come next, so len(p) = next(p).cum - p.cum
//   a         = ordinal (within the symbol's run list) of the first run starting inside the bucket
//   e1.start == kOvf: more than two runs start in the bucket -> search ent[a .. next slot's a)
template <typename P>
struct alignas(8 * sizeof(P)) RankSlot {
    P pstart, pcum;
    P s0, c0;
    P s1, c1;
    P next_cum;
    P a;
};
// same idea for phi's predecessor structure over text positions (payload = phi base)
template <typename P>
struct alignas(8 * sizeof(P)) PhiSlot {
    P ppos, pbase;
    P p0, b0;
    P p1, b1;
    P a;
    P pad;
};
template <typename P> constexpr P kSent = static_cast<P>(~static_cast<P>(0));
template <typename P> constexpr P kOvf = static_cast<P>(~static_cast<P>(0) - 1);

struct DevSym {
    const void *ent;    // RunEnt<P>[nruns + 1] (sentinel: start = n, cum = total): overflow buckets, samples
    const void *samp;   // P[nruns]: samples_last_ of each run (nullptr without toehold SA)
    const void *slots;  // RankSlot<P>[(n >> shift) + 2]
    uint64_t nruns;
    uint64_t F;      // RowBowt::f_[byte]
    uint64_t total;  // occurrences of the symbol
    uint32_t shift;
    uint32_t pad;
};

constexpr int kLdsSyms = 8;  // symbol tables staged in LDS per workgroup; rarer slots read from HBM

struct DevIndex {
    uint64_t n, r;
    uint64_t last_run_sample;
    const DevSym *syms;  // sigma entries, device memory
    uint32_t sigma;
    uint32_t pos_bytes;
    // phi
    const void *phi_ent;    // PhiEnt<P>[r]
    const void *phi_slots;  // PhiSlot<P>[(n >> phi_shift) + 2]
    uint32_t phi_shift;
    uint32_t has_tsa;
    // markers
    const uint64_t *mk_start, *mk_end, *mk_off, *mk_vals;
    uint64_t mk_nruns;
    // {reads, matched, sum occ, sum locs}
    unsigned long long *counters;
    const uint8_t *lut;  // 256 bytes, device memory
};

struct LaunchCfg {
    int block_threads = 256;
    int max_blocks = 0;  // 0: derive from the device
};

// launchers (rbg_kernels.hip).  All asynchronous on `stream`; return hipError_t as int.
int launch_find_range(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, void *stream);
size_t scan_tmp_bytes(uint64_t N);
int launch_locate_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                       uint64_t max_hits, uint64_t *loc_off, void *tmp, size_t tmp_bytes, void *stream);
int launch_locate_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi,
                       const uint64_t *k, uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs,
                       void *stream);
int launch_markers_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                        uint64_t *mk_off, void *tmp, size_t tmp_bytes, void *stream);
int launch_markers_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                        const uint64_t *mk_off, uint64_t *mk, void *stream);
int launch_find_range_markers_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off,
                                   uint64_t N, uint64_t wsize, uint64_t max_range, uint64_t *lo, uint64_t *hi,
                                   uint64_t *mk_off, void *tmp, size_t tmp_bytes, void *stream);
int launch_find_range_markers_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off,
                                   uint64_t N, uint64_t wsize, uint64_t max_range, const uint64_t *mk_off, uint64_t *mk,
                                   void *stream);
int launch_count_from_ranges(const uint64_t *lo, const uint64_t *hi, uint64_t N, uint64_t *count, void *stream);

}  // namespace rbg
