// rbg_dev.h -- device-visible description of the HBM-resident index replica (shared by the
// C-ABI translation unit and the kernels).  Layout rationale: DESIGN.md "Data layout in HBM".
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <vector>

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

// THE SORTED LISTS OF THE RUN-INDEXED LAYOUT (RBG_LAYOUT_RUNS below: run lists per k-mer depth, the phi list) are
// {key, value} pairs as above at both position widths; the SAMPLES beside the run lists are 6 bytes each at 8-byte
// positions (positions stay below 2^48: rbg_host.cpp flatten()) -- they are read once per materialised re-sample, so
// their alignment costs nothing.  12-byte ENTRIES were tried (round 3: 18 instead of 24 bytes per run and depth, the
// n = 5e10 replica 59 GB instead of 77, bit-exact) and not kept: the probes' unaligned 12- and 16-byte requests made K1/K2
// 12 % slower on the bench index and 49 % slower at n = 5e10 at the same number of L2 misses and VALU instructions
// (profiles/r03_ent48_ab.txt, tools/ent48_probe.hip; commit "experiment: 12-byte entries ...").
struct Samp48 {
    uint16_t w[3];   // little end first
};
// The PHI LIST {sampled position, base} keeps the 12-byte form at 8-byte positions: its probes always read the lane's
// four consecutive entries (three 16-byte requests, no clamping to a candidate count), and K3's ordered walk takes 41.4
// -> 36.8 ms per batch at n = 5e10 with it (7.2 -> 6.7 on the bench index).
struct Ent48 {
    uint32_t key_lo, val_lo;
    uint16_t key_hi, val_hi;
};
static_assert(sizeof(Ent48) == 12, "three words per entry");
template <typename P> struct PhiFmt;
template <> struct PhiFmt<uint32_t> {
    static constexpr size_t ent_bytes = 8, spare = 1;
    static void put_ent(void *base, uint64_t i, uint64_t key, uint64_t val) { static_cast<PhiEnt<uint32_t> *>(base)[i] = {static_cast<uint32_t>(key), static_cast<uint32_t>(val)}; }
};
template <> struct PhiFmt<uint64_t> {
    static constexpr size_t ent_bytes = sizeof(Ent48), spare = 3;   // (the four-entry loads may touch three entries past the sentinel)
    static void put_ent(void *base, uint64_t i, uint64_t key, uint64_t val) {
        static_cast<Ent48 *>(base)[i] = {static_cast<uint32_t>(key), static_cast<uint32_t>(val), static_cast<uint16_t>(key >> 32), static_cast<uint16_t>(val >> 32)};
    }
};
template <typename P> struct RunsFmt;
template <> struct RunsFmt<uint32_t> {
    static constexpr size_t ent_bytes = 8, samp_bytes = 4, spare = 1;   // spare entries after the last sentinel (the two-entry loads may touch one)
    static void put_ent(void *base, uint64_t i, uint64_t key, uint64_t val) { static_cast<RunEnt<uint32_t> *>(base)[i] = {static_cast<uint32_t>(key), static_cast<uint32_t>(val)}; }
    static void put_samp(void *base, uint64_t i, uint64_t v) { static_cast<uint32_t *>(base)[i] = static_cast<uint32_t>(v); }
};
template <> struct RunsFmt<uint64_t> {
    static constexpr size_t ent_bytes = 16, samp_bytes = sizeof(Samp48), spare = 1;
    static void put_ent(void *base, uint64_t i, uint64_t key, uint64_t val) { static_cast<RunEnt<uint64_t> *>(base)[i] = {key, val}; }
    static void put_samp(void *base, uint64_t i, uint64_t v) {
        static_cast<Samp48 *>(base)[i] = {{static_cast<uint16_t>(v), static_cast<uint16_t>(v >> 16), static_cast<uint16_t>(v >> 32)}};
    }
};

// First level of every rank: one record per bucket of 2^shift BWT positions (shift <= 8),
// direct-addressed by (position >> shift), that ANSWERS rank(i, c) for every i in the bucket in
// the common case.  Four 32-bit words = 16 bytes whatever the position width (n < 2^48), so one
// rank is one load request: tools/gather_roof.hip shows the chip is bound by the number of gather
// requests, not by their bytes (51 G/s at 16 B, 30 G/s at 32 B).
//   r0  = low 32 bits of rank(B0, c), B0 = bucket begin; bits 32-47 live in w3
//   w1  = ext (bits 0-8: how many positions from B0 are covered by a run of c that began before
//         B0) | cnt (bits 9-11: runs of c that START inside the bucket, 0..4; 7 = more than 4)
//         | prev_is_c (bit 12: position B0-1 holds c) | run0 (bits 16-31)
//   w2  = run1 | run2 << 16,   w3 = run3 | rank(B0,c)[47:32] << 16
//   run = off (8 bits, start - B0) | (len - 1) << 8, len clipped to the bucket; absent = 0xFFFF
//   rank(i) = r0 + min(o, ext) + sum_t clamp(o - off_t, 0, len_t),  o = i - B0
// cnt == 7 (an "overflow" bucket): w2 = offset (16-byte units) of the bucket's dense table in
// DevIndex::dense -- two bytes per row o: rank(B0 + o, c) - r0, and 255 if row B0 + o - 1 holds c, else the
// number of runs of c that start in [B0, B0 + o) (<= 128) -- so rank, "inside" and the predecessor run's
// ordinal are one more 2-byte load.  Reads
// sampled from the text land in these buckets 40x more often than their share of the table (2 % of all
// ranks on the bench, 72 % of the wave-steps have at least one such lane: DESIGN.md 4).  Without a dense
// pool (or for wide buckets) the run list ent[ord[b] .. ord[b+1]) is searched.
//
// Wide buckets (8 < shift <= 12, meant for the deep k-mer levels whose runs are sparse: 0.01 run starts per
// 256 rows at depth 5) keep the 16 bytes but spend them differently, for rank values below 2^40:
//   r0  = low 32 bits of rank(B0, c)
//   w1  = rank(B0,c)[39:32] (bits 0-7) | ext (bits 8-20, 0..4096) | cnt (bits 21-23: 0..2; 7 = more than 2)
//         | prev_is_c (bit 24)
//   w2  = run0, w3 = run1;   run = off (12 bits) | (len - 1) << 12, absent = 0xFFFFFF
struct alignas(16) RankSlot {
    uint32_t r0, w1, w2, w3;
};
constexpr int kSlotRuns = 4;
constexpr int kSlotRunsWide = 2;
constexpr uint32_t kMaxNarrowShift = 8, kMaxWideShift = 12;
// phi's first level over text positions: phi(i) = (D + i) mod n where D = base - pos (mod n) of
// the last sampled position strictly before i.  dprev is that D at the bucket begin; d0/d1 belong
// to the first two sampled positions inside the bucket.
//   meta = off0 (bits 0-7) | off1 (8-15) | cnt (16-17: 0..2; 3 = more than 2); absent off = 0xFF
template <typename P>
struct alignas(4 * sizeof(P)) PhiSlot {
    P dprev, d0, d1, meta;
};
// 8-byte positions, n < 2^38 and buckets of at most 64 positions: the same slot in 16 bytes instead of 32 (three
// 38-bit D values, two 6-bit offsets, the 2-bit count) -- one phi step then moves half the bytes and a cache line holds
// twice the slots (every index the slot layout can hold on 288 GB has n < 2^38).  cnt: 0..2, 3 = more than 2.
//   w0 = dprev | (d0 & (2^26 - 1)) << 38        w1 = d0 >> 26 | d1 << 12 | off0 << 50 | off1 << 56 | cnt << 62
struct alignas(16) PhiSlotPacked {
    uint64_t w0, w1;
};
constexpr uint32_t kPhiPackedPosBits = 38, kPhiPackedMaxShift = 6;
constexpr uint32_t kSlotOvf = 7;
constexpr uint32_t kPhiOvf = 3;
constexpr uint32_t kMaxSlotShift = 8;

struct DevSym {        // 48 bytes: 344 of them (symbols + 2/3/4-mers) are staged in LDS per workgroup
    const void *ent;    // RunEnt<P>[nruns + 1] (sentinel: start = n, cum = total): overflow buckets
    const void *samp;   // P[nruns]: samples_last_ of each run (nullptr without toehold SA)
    const void *slots;  // RankSlot[(n >> shift) + 2]; nullptr in the run-indexed layout (the run list is searched)
    const uint32_t *ord;  // (n >> shift) + 2: # runs of the symbol starting before each bucket; nullptr likewise
    uint64_t F;      // RowBowt::f_[byte] (k-mer: first row of its SA interval)
    uint32_t shift;
    uint32_t nruns;
};

// ---- run-indexed layout (RBG_LAYOUT_RUNS): space proportional to r, nothing proportional to n ------------------
// rank(i, c) = predecessor search over the symbol's sorted run starts (the reference's own structure is O(r) too:
// rle_string::rank, rle_string.hpp:131-161 over sparse_sd_vector, :110-163), phi(i) = predecessor search over the
// sampled text positions (toehold_sa.hpp:56-72).  The arrays are described below (DevRunTab2, RunRec2); the
// wave-cooperative form of rounds 2-3 (16-ary sampled index, quads of lanes probing 16-entry blocks, 128-byte bucket
// records) lost its A/Bs to the one-lane form and was retired in round 5 (profiles/r04_fmt_ab.txt, DESIGN_HISTORY.md).
constexpr int kLdsSyms = 8;
constexpr int kMaxMajor = 4;  // symbol tables staged in LDS per workgroup; rarer slots read from HBM

// Run-indexed layout, K1/K2: every k-mer depth d = 1..run_ksteps (a depth-d table lists the rows whose d
// preceding text characters spell the k-mer, as runs) keeps the run lists of all its tables back to back in table order
// -- each table's entries ascend and end with its sentinel {n, total} -- and one record per table (DevRunTab2).
// Each table has a coarse DIRECTORY over its run starts -- dir[b] = # runs of the table starting below b << dir_shift,
// the shift chosen per table so that a bucket holds about four runs (about a byte per run: still O(r)) -- or BUCKET
// RECORDS (RunRec2) that fuse the directory entry with the entries it names.
// ---- the arrays ("format 2" in profiles/ and DESIGN_HISTORY.md; the only format since round 5) ----------------------
// Per kept k-mer depth the run lists of all its tables back to back, a directory per table, a sample per entry, laid
// out for ONE LANE answering its own ranks (rbg_runs2_device.hpp): the directory names the
// few entries that can hold the answer, the lane fetches them with independent 16-byte requests and scans them in
// registers; no cross-lane traffic.  At 4-byte positions the entries are {start, cum} pairs of u32; at 8-byte positions:
//   entries  {start mod 2^32, cum mod 2^32}: 8 bytes instead of 16.  The directory bucket (shift <= 30) bounds the
//            answer to the runs starting inside the bucket and the one before them, and FILLER entries (a continuation
//            of the run, or an empty run, every 2^30 rows of a gap) keep every entry within 2^30 rows of the next one
//            of its table: all candidates of a bucket B lie in (B - 2^30, B + 2^30), so keys compare correctly as
//            32-bit distances from the anchor B - 2^31, `position - start` and run lengths are exact in 32 bits.
//            This is the hi/lo split Elias-Fano makes in the reference (sparse_sd_vector.hpp:110-163): the bucket
//            number carries the high part.
//   directory entry {count, hi}: count = # entries of the table that start below the bucket (fillers included), hi =
//            (cum of entry count - 1) >> 31: the rank at any position of the bucket lies in [hi << 31, (hi << 31) + 2^32),
//            so its low 32 bits (computed from the entries' low words) determine it: ONE 64-bit add per rank.
//   entry indices are 64-bit (table's first entry, 64-bit; + count, 32-bit): no depth is left out for having 2^32 entries.
struct DevRunTab2 {
    uint64_t F;          // first row of the k-mer's SA interval
    uint64_t first;      // index of the table's first entry in its depth's arrays
    uint64_t dir_off;    // the table's directory starts at entry dir_off of its depth's directory array
    uint32_t dir_shift, pad;
};
static_assert(sizeof(DevRunTab2) == 32, "two 16-byte reads per record");
// What a search step needs of its table is where its directory / bucket records start and how wide a bucket is: ONE 8-byte word,
//   hot = dir_off | dir_shift << 56,
// staged in LDS for the depths up to kLdsRunDepth, read from a global array (512 KB at depth 8: L2-resident) for the deeper ones.  The
// rest of DevRunTab2 is COLD: `first` is read when a step scans the run list (overflowing buckets, directories) or a toehold re-sample
// is materialised, F never -- every cum of the run lists, records and directories is stored with the table's F already added
// (k_build.hip k_fold_F): a rank comes out of the arithmetic as the ROW it maps to, F + rank, which is all LF wants of it
// (rowbowt.hpp:86), and a position below a table's first run finds that table's F in the first entry's cum.
// (Round 5, first form: the whole 32-byte record per step.  At depth 8 that was 2 MB of records read at random beside the index: 10
//  L2 requests and 4.4 L2 misses per read of 23, profiles/r05_k2_pmc_tabs32.txt.)
constexpr uint32_t kRunHotShiftBit = 56;
struct RunDir64 { uint32_t count, hi; };   // directory entry at 8-byte positions (4-byte positions: the count alone)
// BUCKET RECORDS (RBG_OPT_RUN_REC; DevIndex::run_rec2): the directory entry and the entries it names, fused into ONE aligned 64-byte
// record per bucket of a table, direct-addressed by (position >> shift) -- a rank is then one sector instead of a directory sector plus
// the 1.6 sectors an unaligned stretch of the run list takes, and at every scale K1/K2 on this layout are bound by exactly that sector
// count.  About 64 / 2.5 bytes per entry: the price of the speed, off when the budget is short.  (profiles/r04_rec_per_sweep.txt: 2.5
// entries per bucket is the fastest on the bench index; at n = 5e10 five per bucket run as fast from 31 GB less -- RBG_RUN_REC_PER.)
// COMPACT form (meta bit 5; nearly every bucket): the last entry that starts before the bucket (or the table's first) in full --
// {cum (cum_end's place), start, length} -- and up to kRec2CompactIn entries that start inside it as {offset : sh, length : 32 - sh} bits,
// sh = the table's bucket shift: dense tables (narrow buckets) get long length fields, the sparse tables of the deep k-mer depths
// (buckets of 2^22 rows at depth 8 of the bench index) wide offsets.  Unused places hold {all ones, 0}.  The runs are disjoint and
// ascending, so  rank = cum of the first + sum over ALL places of min(max(position - start, 0), length)  -- branch-free.
// OVERFLOW form (meta bit 4): more candidates than that, or a run too long for its field: e0 / cum_end = their first / their number, and
// twelve PIVOTS where the entries would be -- the starts of candidates stride, 2 x stride, ... with stride = ceil(number / 13) -- so
// that the lane narrows thirteen-fold from the record itself and reads the run list once.
// (Round 4 also had a form of six full {start, cum} pairs for short buckets with long runs; the variable split made it rare, and its
//  compare-and-select scan was a third of the record code every wave executed: such buckets take the overflow form now.)
constexpr uint32_t kRec2Pivots = 12;
constexpr uint32_t kRec2Overflow = 16u;  // meta bit 4
constexpr uint32_t kRec2Compact = 32u;   // meta bit 5
constexpr uint32_t kRec2CompactIn = 10;
struct alignas(64) RunRec2 {
    uint32_t e0;        // index (relative to the table's first entry) of the first entry held -- or of the first candidate of an overflowing bucket
    uint32_t hi;        // 8-byte positions: (cum of the entry before the bucket) >> 31 (RunDir64::hi); 0 otherwise
    uint32_t meta;      // bits 0-3: entries held (0..11); bit 4: overflow (none held); bit 5: compact
    uint32_t cum_end;   // compact: cum of the first entry held; overflow: the number of candidates from e0 on
    uint32_t ent[kRec2Pivots];   // compact: {start, length} of the first entry, then ten packed ones; overflow: twelve pivot starts
};
static_assert(sizeof(RunRec2) == 64, "one sector per bucket");
constexpr uint32_t kRunFillShift = 30;     // fillers every 2^30 rows; directory shifts stay <= 30 at 8-byte positions (DevIndex::run_fill_shift;
                                           // RBG_RUN_FILL_SHIFT lowers it so that tests meet fillers on small indexes)
// the phi list of format 2 at 8-byte positions: {sampled position mod 2^32, base} with fillers likewise (a filler at
// position X after the sample (p, b) carries base (b + X - p) mod n: phi(i) = base + (i - pos) is unchanged)
struct PhiEnt12 { uint32_t pos_lo, base_lo, base_hi; };
static_assert(sizeof(PhiEnt12) == 12, "three words per entry");
constexpr uint32_t kPhiSuperShift = 16;    // phi_super[j] = # entries below bucket j << 16 (64-bit); phi_dir holds the low 32 bits of the counts

// Symbols a search step may consume (RBG_OPT_KMER_STEPS; the reference's own multi-symbol shortcut is its ftab, rowbowt.hpp:121-131,
// :726-758).  The table records {F, first, dir_off, dir_shift} of depths up to kLdsRunDepth are staged in LDS by every workgroup
// (8 + 16 + 64 + 256 + 1024 of them at most: 44 KB); a deeper depth has 4^6 .. 4^8 tables -- 2 MB of records at depth 8 -- and its
// record is read from the global array (two 16-byte loads of one line that stays in L2 / MALL: not an HBM sector).
constexpr int kMaxRunDepth = 8;
constexpr int kLdsRunDepth = 5;
constexpr int kMaxLdsRunTabs = kLdsSyms + 16 + 64 + 256 + 1024 + kLdsRunDepth;  // records staged in LDS by k_find_range_runs

// One 32-byte record per bucket of the marker directory (round 6): at_range(lo, hi) -- MarkerArray::at_range as rowbowt.hpp:272-290 / :437-441 call it -- from the
// records of the buckets of lo and hi, ONE or two sectors, instead of a directory entry, the run ends, the run starts and the value offsets (4.9 sectors per query,
// a third of the marker seeds' misses: profiles/r06_pmc_markers.txt).  `a` = the first run whose end is >= the bucket's first row (what mk_bucket holds), its value
// offset, and EVERY run from `a` on that starts before the bucket's end, as {start, end} relative to the bucket's first row (start clamped to 0 from below, end to
// 0xFFFF from above) and its number of values.  Runs are disjoint and ascending, so for lo in this bucket the first run with end >= lo is a + #{listed: end < lo},
// and for hi in this bucket one past the last run with start <= hi is a + #{listed: start <= hi}; the value offsets follow from off_a and the listed counts.
// nin == kMkRecOverflow: more than kMkRecRuns such runs, or a run with more than 65535 values: the arrays answer (from `a`, as before).
constexpr uint32_t kMkRecRuns = 3, kMkRecOverflow = 0xFF;
struct MkRec {
    uint32_t a;
    uint32_t off_lo;
    uint8_t off_hi, nin;
    uint16_t s_off[kMkRecRuns], e_off[kMkRecRuns], cnt[kMkRecRuns];
    uint16_t pad[2];
};
static_assert(sizeof(MkRec) == 32, "two marker records per 64-byte sector");

struct DevIndex {
    uint64_t n, r;
    uint64_t last_run_sample;
    const DevSym *syms;  // sigma entries, device memory
    uint32_t sigma;
    uint32_t pos_bytes;
    // phi
    const void *phi_ent;    // PhiEnt<P>[r]
    const void *phi_slots;  // PhiSlot<P>[(n >> phi_shift) + 2]
    const uint32_t *phi_ord;  // (n >> phi_shift) + 2: # sampled positions before each bucket
    uint32_t phi_shift;
    uint32_t has_tsa;
    // markers
    const uint64_t *mk_start, *mk_end, *mk_off, *mk_vals;
    uint64_t mk_nruns;
    // direct-addressed entry into the marker runs: mk_bucket[b] = index of the first run whose end is
    // >= b << mk_shift (mk_nruns if none); (n >> mk_shift) + 2 entries; nullptr = binary search only
    const uint32_t *mk_bucket;
    uint32_t mk_shift;
    const MkRec *mk_rec;     // the same buckets as 32-byte records (nullptr: RBG_MK_REC=0, or buckets wider than 2^16 rows)
    // {reads, matched, sum occ, sum locs}
    unsigned long long *counters;
    const uint8_t *lut;  // 256 bytes, device memory
    // two-symbol steps: pairs[m1 * nmajor + m2] describes the pair symbol (c1,c2) with the same
    // record shape as a single symbol (F = first row of the SA interval of "c1c2", samp = SA-2 at
    // pair-run ends); lut2 maps a byte to its major index 0..nmajor-1 or 0xFF.  nmajor == 0: off.
    const DevSym *pairs;    // nmajor^2
    const DevSym *triples;  // nmajor^3 (kmer_steps >= 3)
    const DevSym *quads;    // nmajor^4 (kmer_steps >= 4)
    const DevSym *quints;   // nmajor^5 (kmer_steps == 5)
    const uint8_t *lut2;
    uint32_t nmajor;
    uint32_t kmer_steps;    // 1 .. 5
    // ftab (reference: RowBowt::search_ftab, rowbowt.hpp:745-758; result-neutral by construction,
    // :124-125): for every word of ftab_k major symbols the state after searching it -- {lo, hi,
    // toehold, 0} (4 x u32 at 4-byte positions, else 4 x u64) -- indexed by the word read as a base-nmajor number, most significant
    // digit = leftmost symbol.  Built on the GPU at load time with k_find_range itself.  0 = none.
    const void *ftab;       // 16-byte entries at 4-byte positions, 32-byte entries at 8-byte positions
    uint32_t ftab_k;
    uint32_t phi_packed;    // 1: phi_slots holds PhiSlotPacked (8-byte positions only)
    const uint8_t *dense;   // dense tables of the overflow buckets of every narrow rank table (RankSlot); nullptr = none
    // run-indexed layout (layout == 2)
    uint32_t layout;        // 1 = slot tables (RBG_LAYOUT_SLOTS), 2 = run-indexed (RBG_LAYOUT_RUNS)
    uint32_t run_ksteps;    // symbols a step of k_find_range_runs may consume (1..kMaxRunDepth); kmer_steps stays 1 for the per-lane kernels
    uint32_t run_ntabs;     // records in run_tabs2
    const void *run_samp[kMaxRunDepth];   // per depth: one sample per entry (run-end sample; SA - d for a depth-d run: 4 bytes, 6 at 8-byte positions), nullptr without toehold SA
    uint32_t run_tab_first[kMaxRunDepth + 1];   // depth d's records start at run_tab_first[d - 1]
    // run-indexed phi: a coarse directory over the sampled positions -- phi_dir[b] = # sampled positions below b << phi_dir_shift
    // ((n >> shift) + 2 entries; the shift keeps two to four sampled positions per bucket) -- so that a phi step is one
    // 8-byte gather (two neighbouring entries) and ONE row probe of the run list instead of a descent through the sampled
    // levels; a bucket with more than 8 sampled positions is narrowed by pivot probes first.  nullptr: phi slots answer phi.
    const uint32_t *phi_dir;
    uint32_t phi_dir_shift;
    uint32_t run_depth_mask;   // bit d - 1: the k-mer depth d has run lists (bit 0 always; RBG_OPT_RUN_DEPTHS / the budget rule may leave depths out)
    uint32_t phi_super_shift;               // 0 = no super counts (4-byte positions: r < 2^32)
    uint32_t run_fill_shift;                // 8-byte positions: entries of a table (and of the phi list) lie less than 2^this rows apart
    const void *run_ent2[kMaxRunDepth];     // uint2 {start, cum} (low words at 8-byte positions) + 2 spare entries
    const void *run_dir2[kMaxRunDepth];     // uint32_t counts (4-byte positions) or RunDir64 (8-byte positions)
    const DevRunTab2 *run_tabs2;            // run_ntabs records, depth d's from run_tab_first[d - 1] (cold: `first` for scans and re-samples)
    const uint64_t *run_hot;                // run_ntabs words dir_off | dir_shift << 56: what a step reads of its table
    // UNIFORM geometry of a depth beyond kLdsRunDepth (round 6, second half): every table of the depth has the same bucket shift and the same
    // number of bucket records, so a step COMPUTES its table's hot word -- (table's ordinal in the depth) x stride | shift << 56 -- instead of reading it
    // from the 512 KB global array: an L2-served scattered request costs a tenth of a miss (profiles/r06_gather_width.txt), and a search made one
    // per step.  ONE depth can be uniform (the deepest kept one: where a search spends its steps), so that its three constants sit in scalar registers;
    // run_uni_depth = kMaxRunDepth: none (every table has its own shift; the load decides: capi/upload_runs.ipp).
    uint32_t run_uni_depth;                 // depth index d (k-mer depth d + 1) of the uniform depth
    uint32_t run_uni_stride;                // bucket records per table there
    uint32_t run_uni_shift;                 // the bucket shift of all its tables
    const RunRec2 *run_rec2[kMaxRunDepth];  // per depth: the tables' bucket records back to back (DevRunTab2::dir_off / dir_shift then address them); nullptr = directories
    const uint64_t *phi_super;              // 8-byte positions: full counts every 2^phi_super_shift buckets
    uint64_t phi_m;                         // entries of the phi list (fillers included); entry phi_m is the sentinel
    uint64_t phi_last_pos, phi_last_base;   // the last sampled position and its base (circular predecessor, toehold_sa.hpp:59,65)
    // In-kernel 2-bit staging of the reads (k_find_range_runs STAGE): with four major symbols whose bytes differ in the three bits
    // from `stage_shift` up (ACGT and acgt: shift 0), a byte's code and the byte it must be come out of two 8-byte register tables by v_perm_b32 --
    // four symbols per instruction, no per-symbol LDS lookup.  stage_ok == 0: the alphabet does not allow it; the kernel reads bytes as before.
    // Chain order of K3 by LOCUS (launch_locate_order): with a document table attached (DocList, doclist.hpp:46-79: one document per
    // haplotype sequence in a pangenome index) the chains are sorted by {offset inside the document (coarse), document, offset (fine)} instead of
    // the absolute text position: reads of one locus whose toeholds lie in the same haplotype become neighbours whatever that haplotype is, and
    // chains of one locus visit the haplotypes in the same order -- neighbouring lanes share phi sectors for the whole walk.  Result-neutral.
    const uint64_t *order_docs;   // sorted document starts (order_ndocs of them), nullptr = order by absolute position
    uint32_t order_ndocs, order_dbits, order_lowbits, order_obits;   // key = (offset >> lowbits) << (dbits + lowbits) | doc << lowbits | offset & (2^lowbits - 1)
    uint32_t stage_ok, stage_shift;
    uint32_t stage_code[2];    // byte t of {[0], [1]}: the major index of the symbol whose (byte >> shift) & 7 == t (0 where there is none)
    uint32_t stage_byte[2];    // ... and that symbol's byte (a byte no symbol of the alphabet has where there is none)
};

// What the instrumented instantiations count (sums over the launch; include/rbg.h rbg_search_stats_t mirrors it).
// bench.py turns them into the bytes of the algorithm AS RUN: ftab entry + slots x 16 + dense x 2 + read chunks x 16
// + offsets + outputs + 2 gathers per materialised re-sample (K1/K2); phi slots + sorted keys + offsets + stores (K3).
enum SearchStat {
    kStSteps = 0,   // LF gathers issued (one per single-symbol or k-mer step)
    kStSlots,       // RankSlot loads (1 per step, 2 when lo and hi+1 fall in different buckets)
    kStDense,       // 2-byte loads from dense overflow tables
    kStSearch,      // ranks answered by searching the run list (overflow bucket without a dense table)
    kStFtab,        // ftab entries fetched
    kStResample,    // toehold re-samples materialised at the end of a read (ord + samp gathers)
    kStChunks,      // aligned 16-byte chunks of read bytes fetched
    kStSymbols,     // read symbols consumed (reference LF iterations covered)
    kStatSearchN
};
enum LocateStat {
    kLsPhiSteps = 0,  // phi evaluations (PhiSlot loads)
    kLsPhiOvf,        // of them: overflow buckets searched in the run list
    kLsChains,        // reads with at least one location
    kLsLocs,          // locations stored
    kStatLocateN
};

// ---- marker-seed log (one walk instead of two) ------------------------------------------------------------------------
// get_markers_greedy_seeding's output is ragged twice over (records per read, markers per record), so it takes a count
// pass, two scans and a fill pass -- and round 2's fill pass walked every read again (24 of 43 ms per 10 M reads on both
// strands).  With a log the count pass leaves behind, per sequence, what the fill pass needs: its seed records
// {lo, hi, qs, qe, marker sub-range} and, for every window query that found markers, where they sit in mk_vals.  The fill
// pass then only copies.  Fixed quota per sequence (the caller's scratch decides it); a sequence that needs more is
// flagged and listed, and only those are walked again.
template <typename P>
struct SeedLogRec {
    P lo, hi;
    uint32_t qs, qe, mb, me;   // q.first, seed_ei; markers [mb, me) of this read's markers
};
struct SeedLogWin {
    uint32_t src, cnt;         // mk_vals[src, src + cnt)
};
struct SeedLog {
    unsigned char *base;       // nullptr: no log.  Sequence i: base + i * stride = {u32 ns, u32 nw, SeedLogRec[qs], SeedLogWin[qw]}
    uint64_t stride;
    uint32_t qs, qw;
    uint32_t *nsel;            // nsel[0] = # sequences over quota, their indices from nsel + 4 on
    unsigned long long *stats = nullptr;   // != nullptr (run-indexed layout, no log): the INSTRUMENTED instantiation adds what it touched (kSeedStatN sums)
};
// What the instrumented seeding kernels of the run-indexed layout count (rbg_marker_seeds_stats_dev / rbg_greedy_longest_seed_stats_dev): the
// eight sums of SearchStat with that layout's meanings (rbg_runs2_device.hpp), then the marker side.
enum SeedStat {
    kSdMarkerQueries = kStatSearchN,   // window queries that went to the marker runs (range <= max_range)
    kSdMarkerDir,                      // bucket records read for them (32 bytes each; directory entries of 4 bytes with RBG_MK_REC=0)
    kSdMarkerProbes,                   // mk_end / mk_start entries read (8 bytes each): behind an overflowing record only
    kSdMarkerOff,                      // mk_off entries read (8 bytes each): likewise
    kSdMarkerVals,                     // marker values copied (8 read + 8 written)
    kSdSeedRecs,                       // seed records written (48 bytes each)
    kSdSequences,                      // sequences walked
    kSeedStatN
};
constexpr uint32_t kSeedLogOverflow = 0xFFFFFFFFu;
constexpr uint32_t kSeedLogWindows = 6, kSeedLogSeedsDefault = 12;
inline size_t seed_log_rec_bytes(uint32_t pos_bytes) { return pos_bytes == 4 ? sizeof(SeedLogRec<uint32_t>) : sizeof(SeedLogRec<uint64_t>); }
inline size_t seed_log_stride(uint32_t pos_bytes, uint32_t seeds_per_read) {
    return 8 + static_cast<size_t>(seeds_per_read) * seed_log_rec_bytes(pos_bytes) + kSeedLogWindows * sizeof(SeedLogWin);
}
inline size_t seed_log_bytes(uint64_t N, uint32_t pos_bytes, uint32_t seeds_per_read) {
    return static_cast<size_t>(N) * seed_log_stride(pos_bytes, seeds_per_read) + 64 + (N + 4) * 4;
}
// the log a scratch area of `bytes` can hold for N sequences (base == nullptr: too small for two seeds each)
inline SeedLog make_seed_log(void *scratch, size_t bytes, uint64_t N, uint32_t pos_bytes) {
    SeedLog lg{nullptr, 0, 0, kSeedLogWindows, nullptr, nullptr};
    if (!scratch || N == 0 || (reinterpret_cast<uintptr_t>(scratch) & 15)) return lg;
    const size_t tail = 64 + (N + 4) * 4, fixed = 8 + kSeedLogWindows * sizeof(SeedLogWin), rec = seed_log_rec_bytes(pos_bytes);
    if (bytes < tail + N * (fixed + 2 * rec)) return lg;
    size_t q = ((bytes - tail) / N - fixed) / rec;
    if (q > 255) q = 255;
    lg.qs = static_cast<uint32_t>(q);
    lg.stride = fixed + q * rec;
    lg.base = static_cast<unsigned char *>(scratch);
    lg.nsel = reinterpret_cast<uint32_t *>(lg.base + ((N * lg.stride + 15) & ~size_t(15)));
    return lg;
}

// ---- k-mer tables composed on the device at load time (k_compose.hip) ---------------------------------------------------
struct ComposeTable {       // one major symbol's own (depth-1) table, resident on the device
    const void *ent;        // RunEnt<P>[nruns + 1]
    const void *samp;       // P[nruns] or nullptr
    uint64_t nruns, total, F;
};
struct ComposedLevel {      // one k-mer depth: its tables back to back in table order, each closed by its sentinel {n, total}
    void *ent = nullptr;    // RunEnt<P>[entries + 2] (hipMalloc; the caller owns it)
    void *samp = nullptr;   // P[entries + 2] or nullptr
    uint64_t entries = 0;   // runs of all tables + one sentinel per table
    std::vector<uint64_t> nruns, total, F, first;   // per table; first = index of its first entry
};
// depths 2 .. kmax from the depth-1 tables of the M major symbols and the depth-1 segmentation (g_start[g_n + 1] with
// sentinel n, g_id[g_n] = major index or 0xFFFFFFFF, g_samp[g_n] = samples_last or nullptr); returns an RBG_* code
int compose_levels_device(uint32_t pos_bytes, uint64_t n, uint32_t M, const ComposeTable *major, const void *g_start, const uint32_t *g_id,
                          const void *g_samp, uint64_t g_n, uint32_t kmax, bool with_samples, std::vector<ComposedLevel> &out, void *stream,
                          uint32_t keep_mask = 0 /*bit d - 1: depth d's arrays are kept; others are freed once the next depth is made (their metadata stays); 0 = keep all*/,
                          bool *inputs_released = nullptr /*[2], non-null: the inputs are handed over (hipMalloc blocks) and freed as soon as they have been read --
                                                            [0] = the depth-1 segmentation was freed, [1] = the major symbols' tables were*/);

// bytes of host memory this process may still take: the smaller of the machine's MemAvailable and what its cgroup (v2) has left
inline double host_memory_available() {
    double avail = 1e18;
    if (FILE *f = std::fopen("/proc/meminfo", "r")) {
        char line[256];
        while (std::fgets(line, sizeof line, f)) {
            unsigned long long kb = 0;
            if (std::sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = static_cast<double>(kb) * 1024.0; break; }
        }
        std::fclose(f);
    }
    unsigned long long mx = 0, cur = 0;
    bool have_mx = false, have_cur = false;
    if (FILE *f = std::fopen("/sys/fs/cgroup/memory.max", "r")) { have_mx = std::fscanf(f, "%llu", &mx) == 1; std::fclose(f); }
    if (FILE *f = std::fopen("/sys/fs/cgroup/memory.current", "r")) { have_cur = std::fscanf(f, "%llu", &cur) == 1; std::fclose(f); }
    if (have_mx && have_cur && mx > cur) avail = std::min(avail, static_cast<double>(mx - cur));
    else if (have_mx && have_cur) avail = 0;
    return avail;
}


struct LaunchCfg {
    int block_threads = 256;
    int max_blocks = 0;  // 0: derive from the device
};

// launchers (k_search.hip, k_locate.hip, k_markers.hip, k_build.hip).  All asynchronous on `stream`; return hipError_t as int.
int launch_find_range(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, void *stream);
// run-indexed layout (k_runs.hip)
int launch_find_range_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                           uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, void *stream,
                           unsigned long long *stats = nullptr /*kStatSearchN: the instrumented instantiation*/);
int launch_find_range_runs_sel(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                               uint64_t *hi, uint64_t *ssamp /*nullable*/, const uint32_t *sel, const uint32_t *nsel, void *stream);
int launch_find_range_runs_packed(const DevIndex &ix, const LaunchCfg &cfg, const uint2 *meta, const uint4 *chunks, uint64_t N,
                                  uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, void *stream);
int launch_locate_fill_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint64_t *k,
                            uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs, const uint64_t *sub,
                            const void *order, const uint64_t *skeys, void *stream, unsigned long long *stats = nullptr /*kStatRunLocateN*/,
                            uint32_t *locs32 = nullptr);
// the kernels beside the rb_align path on the run-indexed layout (k_runs_seeds.hip): k-mer steps through the depths' run lists
int launch_lf_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym, uint64_t N,
                   uint64_t *lo_out, uint64_t *hi_out, void *stream);
int launch_find_range_markers_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                                   uint64_t wsize, uint64_t max_range, uint64_t *lo, uint64_t *hi, uint64_t *cnt, const uint64_t *mk_off,
                                   uint64_t *mk, bool fill, void *stream);
int launch_greedy_seed_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                            uint64_t min_length, uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, void *stream,
                            unsigned long long *stats = nullptr /*kSeedStatN: the instrumented instantiation*/);
// lg.base != nullptr: the count pass (fill == false) writes the log; the fill pass walks only the sequences listed behind lg.nsel
int launch_marker_seeds_runs(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t wsize,
                             uint64_t max_range, uint64_t *seed_cnt, uint64_t *mk_cnt, const uint64_t *seed_off, const uint64_t *mk_off,
                             uint64_t *seeds, uint64_t *mk, bool fill, void *stream, const SeedLog &lg, uint64_t ftab_k = 0 /*format 2: rb_markers --ftab*/);
int launch_find_range_stats(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                            uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, unsigned long long *stats /*kStatSearchN*/,
                            void *stream);
size_t scan_tmp_bytes(uint64_t N);
int launch_locate_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                       uint64_t max_hits, uint64_t *loc_off, void *tmp, size_t tmp_bytes, void *stream);
int launch_locate_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi,
                       const uint64_t *k, uint64_t N, uint64_t max_hits, const uint64_t *loc_off, uint64_t *locs,
                       const uint64_t *sub /*nullable: per-read value subtracted from every location*/,
                       const void *order /*nullable: workspace filled by launch_locate_order*/, void *stream,
                       unsigned long long *stats = nullptr /*kStatLocateN: launches the instrumented instantiation*/,
                       uint32_t *locs32 = nullptr /*4-byte positions only: store the locations as uint32_t here instead of `locs`*/);
size_t locate_order_ws_bytes(uint64_t N);
int launch_locate_order(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *k, uint64_t N, void *ws, size_t ws_bytes,
                        void *stream);
// first-level slot tables from the uploaded run lists (pos_bytes = 4 or 8 selects RunEnt<P> / PhiEnt<P>)
// dense_cursor (nullable): running total of dense-table space handed to overflow buckets, in 16-byte units
int launch_build_rank_slots(uint32_t pos_bytes, const void *ent, uint64_t nruns, uint64_t n, uint32_t shift, void *slots,
                            uint32_t *ord, unsigned long long *overflow, unsigned long long *dense_cursor, void *stream);
// second pass, once the pool of dense_cursor * 16 bytes exists: fills the dense tables of one rank table
int launch_fill_dense(uint32_t pos_bytes, const void *ent, uint64_t n, uint32_t shift, const void *slots, const uint32_t *ord,
                      uint8_t *dense, void *stream);
// the text of `rb_align -s` on the device (k_text.hip; rbg_align_text): phase 1 = element lengths and offsets in the workspace,
// phase 2 = the bytes.  E = N + loc_off[N] elements (a head per read, then its locations), + N with a markers line per read
size_t text_ws_bytes(uint64_t E);
int launch_copy16(const void *pinned_src, void *dst, uint64_t bytes, void *stream);   // host-mapped memory -> device by a kernel (not the copy engine)
int launch_text_plan(const uint64_t *lo, const uint64_t *hi, const uint64_t *loc_off, const uint64_t *locs, uint64_t N, uint64_t E, const char *names,
                     const uint32_t *name_off, const uint64_t *doc_start, const char *doc_names, const uint32_t *doc_name_off, uint64_t ndocs,
                     uint64_t text_size, bool with_locs, const uint64_t *mk_off, const uint64_t *mk, void *ws, size_t ws_bytes, unsigned int *d_bad, void *stream);
void text_total_ptrs(void *ws, uint64_t E, const uint64_t **last_at, const uint32_t **last_len);
int launch_text_fill(const uint64_t *lo, const uint64_t *hi, const uint64_t *loc_off, const uint64_t *locs, uint64_t N, uint64_t E, const char *names,
                     const uint32_t *name_off, const uint64_t *doc_start, const char *doc_names, const uint32_t *doc_name_off, uint64_t ndocs,
                     uint64_t text_size, bool with_locs, const uint64_t *mk_off, const uint64_t *mk, void *ws, uint64_t total, char *text, void *stream);
// run-indexed layout from run lists already on the device (k_build.hip): the tables' directories, 8-byte samples packed to 6
int launch_run_dirs(uint32_t pos_bytes, const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *doff, const uint32_t *dshift,
                    uint32_t T, uint64_t total, uint32_t *dir, void *stream);
int launch_pack_samp48(const uint64_t *in, uint64_t n, void *out, void *stream);
// format 2 of the run-indexed layout (k_build.hip): {key, value} u64 pairs on the device -> fillers, low-word pairs, directories
int launch_fill_count(const void *ent, uint64_t m, uint64_t n, uint32_t fill_shift, uint64_t *arr /*m + 1, nullable*/, unsigned long long *total, void *stream);
int launch_scan_u64(uint64_t *vals, uint64_t N, void *tmp, size_t tmp_bytes, void *stream);   // inclusive, in place (tmp: scan_tmp_bytes(N))
int launch_fill_expand(bool phi, const void *ent, const uint64_t *samp, uint64_t m, uint64_t n, uint32_t fill_shift, const uint64_t *pos, void *ent_out, uint64_t *samp_out, void *stream);
int launch_gather_u64(const uint64_t *src, const uint64_t *idx, uint64_t count, uint64_t *out, void *stream);
int launch_pack_pairs32(const void *ent, uint64_t m, uint64_t spare, void *out, void *stream);
int launch_pack_phi12(const void *ent, uint64_t m, uint64_t spare, void *out, void *stream);
int launch_run_dirs2(const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *doff, const uint32_t *dshift, uint32_t T, uint64_t total,
                     void *dir, void *stream);
// adds F[t] to the cum of every entry of table t (entries [first[t], first[t + 1]) of `ent`, RunEnt<P>; entries from first[T] on take the last table's)
int launch_fold_F(uint32_t pos_bytes, void *ent, const uint64_t *first, const uint64_t *F, uint32_t T, uint64_t total, void *stream);
int launch_run_recs2(uint32_t pos_bytes, const void *ent, const uint64_t *first, const uint64_t *nruns, const uint64_t *roff, const uint32_t *rshift, uint32_t T, uint64_t total,
                     void *recs, unsigned long long *overflow, void *stream);
int launch_phi_dir(uint32_t pos_bytes, const void *ent, uint64_t m, uint32_t shift, uint64_t nb, uint32_t *dir, uint32_t ss, uint64_t *super, void *stream);
int launch_build_phi_slots(uint32_t pos_bytes, bool packed, const void *ent, uint64_t r, uint64_t n, uint32_t shift, void *slots, uint32_t *ord,
                           unsigned long long *overflow, void *stream);
// packed reads (2 bits per symbol): pack the byte batch once, then search the packed form
size_t pack_ws_bytes(uint64_t N, uint64_t total_bytes);
int launch_pack_reads(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                      uint64_t total_bytes, void *ws, size_t ws_bytes, void *stream);
int launch_find_range_packed(const DevIndex &ix, const LaunchCfg &cfg, const void *ws, const uint8_t *seqs, const uint64_t *off,
                             uint64_t N, uint64_t total_bytes, uint64_t *lo, uint64_t *hi, uint64_t *ssamp, void *stream);
int launch_find_range_packed_only(const DevIndex &ix, const LaunchCfg &cfg, const uint2 *meta, const uint4 *chunks, uint64_t N,
                                  uint64_t *lo, uint64_t *hi, uint64_t *ssamp /*nullable*/, void *stream);
// log (nullable scratch): when it can hold a log (make_seed_log) the plan writes one and the fill copies from it
int launch_marker_seeds_plan(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range, uint64_t ftab_k, uint64_t *seed_off, uint64_t *mk_off, void *tmp,
                             size_t tmp_bytes, void *stream, void *log = nullptr, size_t log_bytes = 0, unsigned long long *stats = nullptr /*kSeedStatN; run-indexed layout, no log*/);
int launch_marker_seeds_fill(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range, uint64_t ftab_k, const uint64_t *seed_off, const uint64_t *mk_off,
                             uint64_t *seeds, uint64_t *mk, void *stream, void *log = nullptr, size_t log_bytes = 0, unsigned long long *stats = nullptr);
int launch_greedy_seed(const DevIndex &ix, const LaunchCfg &cfg, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                       uint64_t min_length, uint64_t *lo, uint64_t *hi, uint64_t *qs, uint64_t *qe, uint64_t *ss, void *stream, unsigned long long *stats = nullptr);
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
int launch_lf(const DevIndex &ix, const LaunchCfg &cfg, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym,
              uint64_t N, uint64_t *lo_out, uint64_t *hi_out, void *stream);
// fills ix.ftab-shaped table `tab` (nmajor^k entries x 4 u64) by searching every k-symbol word
int launch_build_ftab(const DevIndex &ix, const LaunchCfg &cfg, uint32_t k, void *tab, void *stream);
size_t ftab_build_scratch_bytes(uint64_t words, uint32_t k);  // device scratch the build takes besides the table
int launch_sample_reads(const uint8_t *text, uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first,
                        uint64_t N, uint32_t sub_ppm, uint8_t *seqs, uint64_t *off, uint64_t *start_out /*nullable*/, void *stream);
int launch_sample_reads_pg(const uint8_t *base, const uint64_t *sites, const uint8_t *alt, const uint8_t *G, uint64_t S, const uint32_t *site_dir, uint32_t site_dir_shift,
                           uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed, uint64_t first, uint64_t N, uint32_t sub_ppm, uint8_t *seqs, uint64_t *off,
                           uint64_t *start_out, void *stream);
int launch_count_from_ranges(const uint64_t *lo, const uint64_t *hi, uint64_t N, uint64_t *count, void *stream);

}  // namespace rbg
