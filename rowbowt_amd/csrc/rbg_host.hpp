// rbg_host.hpp -- host side of the engine: readers for the reference's on-disk files and the
// flattener that produces the HBM layout.  Pure C++17, no HIP, no sdsl.
//
// What it replaces in the reference (paths relative to the reference tree):
//   rle_string::load            include/rle_string.hpp:265-275
//   sparse_sd_vector::load      include/sparse_sd_vector.hpp:194-200   (sdsl::sd_vector<> bytes)
//   huff_string::load           include/huff_string.hpp:61-63          (sdsl::wt_huff<> bytes)
//   ToeholdSA::load / build_phi include/toehold_sa.hpp:85-91, :105-131
//   DocList::load               include/doclist.hpp:57-73
//   MarkerArray::load           pfbwt-f marker_array.hpp (un-vendored; layout per SURVEY 8b-format)
//   RowBowt::build_f            include/rowbowt.hpp:770-778
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include <functional>

namespace rbg {

// fn(begin, end) over [0, n) cut into contiguous chunks, one per worker thread (as many as the process may use: the
// container's CPU quota counts, rbg_thread_team.hpp cpu_budget); runs inline when n is small.  The load path's loops
// over the runs (3e8 of them at pangenome scale) go through this.
void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)> &fn, uint64_t min_per_thread = uint64_t(1) << 16);
unsigned load_threads();

// ---- decoded contents of the reference's files -------------------------------------------------
struct RawRle {          // ri::rle_string, as a plain run-length BWT
    uint64_t n = 0, R = 0, B = 0;
    std::vector<uint8_t> heads;   // R
    std::vector<uint64_t> lens;   // R
};
struct RawTsa {          // ToeholdSA members, toehold_sa.hpp:157-161
    uint64_t r = 0, n = 0;
    std::vector<uint64_t> pred_pos;      // set bits of pred_, ascending
    std::vector<uint64_t> samples_last;  // BWT-run order
    std::vector<uint64_t> pred_to_run;   // text order
};
struct RawMarkers {
    std::vector<uint64_t> start, end;  // inclusive SA-index runs, ascending
    std::vector<uint64_t> off;         // nruns+1 offsets into vals
    std::vector<uint64_t> vals;        // MarkerT values
    int32_t wsize = 0;
};
struct RawDocs {
    std::vector<std::string> names;    // file order
    std::vector<uint64_t> starts;      // file order
    std::vector<uint64_t> sorted;      // ascending (the bit-vector view)
};

int parse_rbwt(const std::string &fname, RawRle &out);
int parse_tsa(const std::string &fname, RawTsa &out);
int parse_mab(const std::string &fname, RawMarkers &out);
int parse_docs(const std::string &fname, RawDocs &out);
// raw build inputs of rb_build (rb_build.cpp:83-93): <pre>.bwt as rle_string(std::string fname, B) reads it
// (rle_string.hpp:44-97: `ifs >> c` skips whitespace bytes, byte 0 becomes 1), <pre>.ssa / <pre>.esa as
// (x, y) u64 pairs of which y is used (toehold_sa.hpp:133-155)
int read_raw_bwt(const std::string &fname, RawRle &out);
int read_raw_samples(const std::string &fname, std::vector<uint64_t> &y_out);
// raw .ssa/.esa y values -> RawTsa (toehold_sa.hpp:105-155)
void tsa_from_samples(uint64_t n, uint64_t r, const uint64_t *ssa_y, const uint64_t *esa_y, RawTsa &out);

// ---- native cache file (next-row f1): everything the reference's .rbwt/.tsa/.mab/.docs hold, as flat
// little-endian arrays ("<prefix>.rbgpu").  Loading it is one mapping of the file: no wavelet-tree or
// sd_vector decoding (rle_string::load, rle_string.hpp:265-275) and no re-encoding of a raw BWT
// (rle_string.hpp:44-97).  Layout (all u64 unless noted, sections padded to 8 bytes):
//   "RBGPUIX2" | flags (bit0 tsa, bit1 markers, bit2 docs) | n | R | B | len_width (4|8) | pos_width (4|8)
//   | ma_nruns | ma_nvals | ma_wsize | docs_bytes
//   | heads u8[R] | lens len_width[R]
//   | tsa:  pred_pos pos_width[R] | samples_last pos_width[R] | pred_to_run pos_width[R]
//   | ma:   start pos_width[nruns] | end pos_width[nruns] | off u64[nruns+1] | vals u64[nvals]
//   | docs: the .docs text (doclist.hpp:57-73) | checksum
// Checksum: FlatSum (rbg_host.cpp) over the FlatSums of the preceding words taken in chunks of 2^20 words, so that the
// reader's threads verify a 9 GB file in 0.14 s.  "RBGPUIX1" (one FlatSum chain over all words; written until round 3)
// differs in nothing else and is still read.
struct FlatBundle {
    RawRle rle;
    bool has_tsa = false, has_ma = false, has_dl = false;
    RawTsa tsa;
    RawMarkers ma;
    RawDocs dl;
};
int write_flat(const std::string &fname, const FlatBundle &b);
int read_flat(const std::string &fname, FlatBundle &b);  // RBG_EIO / RBG_EFORMAT on a missing, torn or inconsistent file

// ---- the flat layout, host copy ------------------------------------------------------------------
// One table per distinct BWT symbol ("slot").  Entry k describes the k-th run of that symbol:
//   start[k] = BWT position where the run begins (ascending),
//   cum[k]   = number of this symbol in BWT[0, start[k])   (cum[nruns] = total),
//   samp[k]  = ToeholdSA::samples_last_ of that run (only with a toehold SA).
// The direct-addressed first level (one slot per 2^shift BWT positions, rbg_dev.h RankSlot) is
// generated from start/cum at upload time; `shift` is chosen here.
struct SymTable {
    uint8_t byte = 0;
    uint32_t shift = 0;
    uint64_t nruns = 0, total = 0, F = 0;
    std::vector<uint64_t> start, cum;  // nruns + 1 (sentinel: start = n, cum = total)
    std::vector<uint64_t> samp;        // nruns or empty
    // a k-mer table composed on the device (k_compose.hip): the arrays above stay empty, the run list lives in HBM
    // as {start, cum} pairs of the position width (+ sentinel) and its samples beside it
    const void *dev_ent = nullptr, *dev_samp = nullptr;
};

// symbols one search step may consume: the slot layout stages the tables of up to 5 symbols per gather in LDS, the run-indexed
// layout reads the records of deeper tables from a global array (rbg_dev.h kMaxRunDepth is this constant)
constexpr int kMaxKmerDepth = 8;
constexpr int kMaxSlotKmerDepth = 5;

struct HostIndex {
    uint64_t n = 0, r = 0;
    uint32_t sigma = 0;
    uint32_t pos_bytes = 8;
    uint8_t lut[256];        // byte -> slot, 0xFF = symbol absent from the BWT
    uint64_t f[257];         // f[c] = # symbols < c  (RowBowt::f_, plus f[256] = n)
    std::vector<SymTable> sym;
    std::vector<uint8_t> run_heads;    // R
    std::vector<uint64_t> run_start;   // R + 1
    // toehold SA
    // multi-symbol steps: for the <= 4 most frequent non-terminator symbols ("major"),
    // kmer(2)[m1 * nmajor + m0] / kmer(3)[(m2 * nmajor + m1) * nmajor + m0] are the tables of the rows
    // whose preceding text characters are x1 x0 / x2 x1 x0 (x0 = bwt[p] adjacent to the suffix).
    // F is the first row of the SA interval of that k-mer; samp[] holds SA - k at each run end.
    uint32_t nmajor = 0;
    uint8_t major_byte[4] = {0, 0, 0, 0};
    uint8_t major_of[256];             // byte -> 0..nmajor-1, 0xFF otherwise
    std::vector<SymTable> kmer_lv[kMaxKmerDepth - 1];   // [d - 2]: the nmajor^d tables of depth d = 2 .. kMaxKmerDepth, or empty
    std::vector<SymTable> &kmer(uint32_t d) { return kmer_lv[d - 2]; }
    const std::vector<SymTable> &kmer(uint32_t d) const { return kmer_lv[d - 2]; }
    // the deepest depth that has tables (1: single-symbol steps only)
    uint32_t kmer_levels() const {
        uint32_t k = 1;
        for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxKmerDepth); ++d)
            if (!kmer_lv[d - 2].empty()) k = d;
        return k;
    }
    void clear_kmer() { for (auto &v : kmer_lv) std::vector<SymTable>().swap(v); }
    // > 0: flatten() only chose the k-mer alphabet and left the composition of depths 2 .. kmer_deferred to the device
    // (FlattenOptions::defer_kmer; rbg_capi.hip upload() runs k_compose.hip before it sizes the replica)
    uint32_t kmer_deferred = 0;
    std::vector<uint32_t> major_slot;  // symbol slot of each major symbol, ascending
    bool has_tsa = false;
    uint64_t last_run_sample = 0;      // toehold_sa.hpp:97-99
    std::vector<uint64_t> samples_last, pred_pos, phi_base;
    uint32_t phi_shift = 0;
    bool has_ma = false;
    RawMarkers ma;
    bool has_dl = false;
    RawDocs dl;
};

struct FlattenOptions {
    int rank_bucket_shift = -1;  // <0: automatic (about one run per two buckets, at most 8); 9..12 = wide buckets (rbg_dev.h)
    int deep_bucket_shift = -1;  // >= 0: bucket shift of the 4-mer and deeper levels (their runs are sparse)
    int phi_bucket_shift = -1;
    int force_pos_bytes = 0;     // 0: 4 when n fits, else 8
    int kmer_steps = 5;          // symbols consumed per step: 1 (reference shape) .. kMaxKmerDepth
    bool defer_kmer = false;     // choose the k-mer alphabet but leave the tables of depth >= 2 to the device (HostIndex::kmer_deferred)
};

int flatten(const RawRle &rle, const RawTsa *tsa, const FlattenOptions &opt, HostIndex &out);
// depths 2 .. kmer_steps on the host from a flattened index (the reference statement of the composition; what
// flatten() runs itself unless defer_kmer is set)
int compose_kmer_tables_host(HostIndex &ix, int kmer_steps, const FlattenOptions &opt);
// bucket shift of a table with `nruns` runs under the options (what compose() gives the tables it makes)
uint32_t kmer_table_shift(uint64_t n, uint64_t nruns, uint32_t depth, const FlattenOptions &opt);

}  // namespace rbg
