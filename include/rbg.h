/*
 * rbg.h -- C-ABI of the MI355X-native backward-search engine for rowbowt's rb_align hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Everything the
 * reference computes on this path through rbwt::RowBowt<ri::rle_string_sd> is reachable here,
 * batched over N reads.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference tree).  The C++17 header shim that keeps the reference's own
 * signatures on top of this ABI is rowbowt_amd/include/rowbowt_gpu.hpp; the binding a reference
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - Ranges are inclusive [lo,hi]; the empty range is exactly {1,0} (rowbowt.hpp:77,85).
 *  - All positions are uint64_t.  Reads are raw bytes, matched as-is (no upper-casing, no N
 *    handling: rb_align.cpp:121 passes seq->seq.s straight to find_range).
 *  - A batch of reads is `seqs` (concatenated bytes) + `off[N+1]` (byte offsets, off[0]=0).
 *  - Return value: RBG_OK (0) or a negative RBG_E* code; rbg_strerror() names it.  The reference
 *    itself prints to stderr and exit(1)s (rowbowt_io.hpp:166-169); the shim maps codes to that.
 *  - There is NO CPU compute path.  Every query entry point runs HIP kernels on a gfx950 device
 *    and fails with RBG_ENODEV when none is usable.
 *  - *_dev entry points take DEVICE pointers and a hipStream_t (as void*), launch asynchronously
 *    and neither allocate nor synchronise (graph-capturable).  The plain entry points take HOST
 *    pointers, stage through HBM and block until the results are in the caller's buffers.
 *  - Thread safety: an rbg_index is immutable after load; concurrent query calls on one index
 *    are allowed (the reference calls const methods concurrently, rb_markers.cpp:321-326).
 */
#ifndef RBG_H
#define RBG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 3 (round 5): RBG_OPT_KMER_STEPS goes up to 8 (default 8: the run-indexed layout steps by up to eight symbols; the slot layout by
 * up to five); rbg_info_t ends with depth_runs[8]; rbg_layout_info_t holds eight depths and lost the fields of the retired first format;
 * the options RBG_OPT_TREE_TOP_KB (12), RBG_OPT_SLOT_BYTES (13) and RBG_OPT_RUN_FMT (15) are gone (RBG_EARG) together with the kernels they
 * selected.  ABI 2 -> 3 also records what round 4 changed without a bump: the default HBM budget is a
 * quarter of the free HBM (was three quarters), RBG_LAYOUT_AUTO builds the run-indexed layout when the slot tables of every symbol per step
 * do not fit it, rbg_layout_info / rbg_sample_reads_pangenome_dev / RBG_LAYOUT_PREFER_SLOTS / options 16-17 exist.  A binding should check
 * rbg_abi_version() before it binds the symbols of a newer ABI (rowbowt_amd/capi.py does). */
#define RBG_ABI_VERSION 3

typedef struct rbg_index rbg_index;

/* rbwt::LoadRbwtFlag, rowbowt_io.hpp:146-152 (same values) */
enum { RBG_LOAD_NONE = 0, RBG_LOAD_SA = 1, RBG_LOAD_MA = 2, RBG_LOAD_DL = 4, RBG_LOAD_FT = 8 };

enum {
    RBG_OK = 0,
    RBG_EIO = -1,      /* file missing / unreadable (reference: "bad file", exit(1)) */
    RBG_EFORMAT = -2,  /* not the sdsl layout the reference's files use */
    RBG_ENODEV = -3,   /* no usable gfx950 device / HIP error */
    RBG_EARG = -4,     /* bad argument */
    RBG_ENOMEM = -5,
    RBG_ENOTLOADED = -6 /* structure (toehold SA / markers / docs) was not loaded */
};

/* device = HIP device ordinal to hold the index replica, or RBG_DEVICE_NONE to parse and flatten
 * on the host only (no queries possible; used to inspect the layout without a GPU). */
#define RBG_DEVICE_NONE (-1)

int rbg_abi_version(void);
const char *rbg_strerror(int code);

/* ---- loading ------------------------------------------------------------------------------ */

/* rbwt::load_rowbowt<ri::rle_string_sd>(prefix, flag), rowbowt_io.hpp:176-189:
 * reads <prefix>.rbwt always, .tsa if SA, .mab if MA, .docs if DL (suffixes :17-21) in the
 * sdsl-serialised layout the reference writes (rle_string.hpp:248-275, toehold_sa.hpp:74-91),
 * flattens to the HBM layout and uploads one replica to `device`.
 * RBG_LOAD_FT is accepted and ignored (rb_align never sets it, rb_align.cpp:149-157). */
int rbg_load(const char *prefix, int flags, int device, rbg_index **out);

/* Construction from raw inputs, replacing rle_string(std::string fname, B) rle_string.hpp:44-97
 * and ToeholdSA(n, r, ssa, esa) toehold_sa.hpp:27-35 at run granularity:
 * heads[R]/lens[R] = run-length BWT (byte 0 must already be stored as 1, rle_string.hpp:59,62);
 * ssa_y/esa_y = the second u64 of each (x,y) pair of <pre>.ssa / <pre>.esa, one per BWT run
 * (toehold_sa.hpp:133-155), or both NULL for no toehold SA. */
int rbg_build_from_runs(const uint8_t *heads, const uint64_t *lens, uint64_t R,
                        const uint64_t *ssa_y, const uint64_t *esa_y, int device, rbg_index **out);

/* Next-row f1: the same construction straight from rb_build's raw input files (rb_build.cpp:83-93):
 * <pre>.bwt read as rle_string(fname) reads it (formatted extraction skips whitespace bytes,
 * rle_string.hpp:58-62; byte 0 -> 1), <pre>.ssa / <pre>.esa as (x,y) u64 pairs (both or neither). */
int rbg_build_from_files(const char *bwt_fname, const char *ssa_fname, const char *esa_fname, int device,
                         rbg_index **out);

/* ---- next-row f1: native cache file and the rb_build outputs ----------------------------------
 * "<prefix>.rbgpu" holds what the reference's .rbwt/.tsa/.mab/.docs hold, as flat little-endian
 * arrays with a checksum (layout: rowbowt_amd/csrc/rbg_host.hpp; the file is mapped and checked by all loader threads:
 * 1.2 s for the 9 GB of an n = 5e10 index); loading it needs no sdsl decoding.
 * rbg_load() falls back to "<prefix>.rbgpu" when "<prefix>.rbwt" does not exist.
 * rbg_convert_index: the reference's serialised files (rb_build's output, rowbowt_io.hpp:49-89) -> cache.
 * rbg_convert_raw:   rb_build's raw inputs <pre>.bwt [+ .ssa/.esa] (rb_build.cpp:83-93; rle_string.hpp:44-97,
 *                    toehold_sa.hpp:27-35,133-155), plus optionally an already serialised .mab and a
 *                    .docs file -> cache.  (The raw .ma input is read by pfbwt-f's MarkerArray
 *                    constructor, which is not vendored: not supported.)
 * Both run on the host only.  rbg_load_cache: flags select the stored parts; asking for a part the
 * file lacks is RBG_EIO, like a missing .tsa/.mab/.docs. */
int rbg_convert_index(const char *prefix, int flags, const char *out_path);
int rbg_convert_raw(const char *bwt_fname, const char *ssa_fname, const char *esa_fname, const char *mab_fname,
                    const char *docs_fname, const char *out_path);
int rbg_load_cache(const char *path, int flags, int device, rbg_index **out);
/* the cache file from a run-length BWT in memory (the arguments of rbg_build_from_runs): for builders that never write
 * the BWT as text, and for handing one index to the ranks of a node (rank 0 writes, every rank rbg_load_cache's) */
int rbg_convert_runs(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y /* nullable, with esa_y */,
                     const uint64_t *esa_y, const char *out_path);
/* the same with a marker array (the arguments of rbg_set_markers; mk_nruns = 0: none) and the text of a .docs file
 * (nullable) stored beside it: one file that rb_align -s -m / rb_markers can run from */
int rbg_convert_runs_markers(const uint8_t *heads, const uint64_t *lens, uint64_t R, const uint64_t *ssa_y, const uint64_t *esa_y,
                             const uint64_t *mk_start, const uint64_t *mk_end, uint64_t mk_nruns, const uint64_t *mk_off,
                             const uint64_t *mk_vals, const char *docs_text /* nullable */, const char *out_path);
/* RowBowt::build_ftab(k) + FTab::serialize (rowbowt.hpp:726-744, ftab.hpp:29-34): the reference's
 * text .ftab ("<kmer> <lo> <hi>" per line, lexicographic), computed with the search kernel. 1 <= k <= 16. */
int rbg_write_ftab(rbg_index *, uint64_t k, const char *path);
/* What loading a .ftab amounts to here (LoadRbwtFlag::FT, rowbowt_io.hpp:187; FTab::load, ftab.hpp:15-27):
 * the file must be, byte for byte, the table build_ftab(k) makes for this index (RBG_EFORMAT if not);
 * *k_out = its k-mer size, the `ftab_k` argument of the marker-seed calls below. */
int rbg_check_ftab(rbg_index *, const char *path, uint64_t *k_out);
/* MarkerArray contents (pfbwt-f marker_array.hpp, loaded at rowbowt_io.hpp:185):
 * inclusive SA-index runs + mk_off[nruns+1] offsets into mk_vals. */
int rbg_set_markers(rbg_index *, const uint64_t *run_start, const uint64_t *run_end, uint64_t nruns,
                    const uint64_t *mk_off, const uint64_t *mk_vals);
/* DocList contents, doclist.hpp:57-73: names '\0'-joined, starts[ndocs]. */
int rbg_set_docs(rbg_index *, const char *names_joined, const uint64_t *starts, uint64_t ndocs);

void rbg_free(rbg_index *);

/* ---- introspection ------------------------------------------------------------------------ */

typedef struct rbg_info_t {
    uint64_t n;             /* rle_string::size() */
    uint64_t r;             /* rle_string::number_of_runs() */
    uint32_t sigma;         /* distinct BWT symbols */
    uint32_t pos_bytes;     /* 4 or 8: width of positions in the HBM layout */
    int32_t device;         /* HIP ordinal or RBG_DEVICE_NONE */
    uint32_t has_tsa, has_markers, has_docs;
    uint64_t hbm_bytes;     /* bytes of the device replica */
    uint64_t marker_runs, marker_vals;
    uint32_t rank_bucket_shift, phi_bucket_shift;
    uint32_t slot_bytes;    /* bytes per rank slot of the slot layout: 16; 0 on the run-indexed layout or without a device */
    /* first-level slot tables: totals and how many buckets overflow their 2 inline entries */
    uint64_t rank_slots, rank_slots_overflow, phi_slots, phi_slots_overflow;
    /* multi-symbol LF steps: symbols consumed per gather (1..5), size of the major alphabet that has
     * k-mer tables (0 = none), total runs of the 2-mer and 3-mer tables */
    uint64_t kmer_steps, kmer_symbols, pair_runs, triple_runs, quad_runs;
    /* ftab (RowBowt::build_ftab / search_ftab, rowbowt.hpp:726-758): word length of the device table, 0 = none */
    uint64_t ftab_k;
    uint64_t quint_runs;    /* total runs of the 5-mer tables (kmer_steps == 5) */
    /* ABI 2: how the space/speed point was chosen at load (the same call is up to 4x slower with fewer levels,
     * DESIGN.md 2b): the depth asked for (RBG_OPT_KMER_STEPS), the free HBM seen at load, the budget the replica had
     * to fit (a quarter of it, or RBG_OPT_HBM_BUDGET_MB); kmer_steps above is what was kept.  Also printed on
     * stderr at load when a level is dropped (always with RBG_VERBOSE). */
    uint64_t kmer_steps_requested, hbm_free_at_load, hbm_budget;
    uint64_t rank_layout;   /* RBG_LAYOUT_SLOTS or RBG_LAYOUT_RUNS (what RBG_OPT_RANK_LAYOUT / the budget rule chose) */
    uint64_t replicas;      /* per handle: 1 when this handle holds a device replica, 0 for a host-only index
                             * (RBG_DEVICE_NONE); further replicas are handles of their own (rbg_replicate[_many]) */
    uint64_t depth_runs[8]; /* ABI 3: [d - 1] = total runs of the tables of k-mer depth d ([0] = r; 0 for a depth that has no tables,
                             * or no run lists on the run-indexed layout); pair_runs .. quint_runs above are [1] .. [4] */
} rbg_info_t;
enum { RBG_LAYOUT_AUTO = 0, RBG_LAYOUT_SLOTS = 1, RBG_LAYOUT_RUNS = 2, RBG_LAYOUT_PREFER_SLOTS = 3 };
int rbg_info(const rbg_index *, rbg_info_t *out);   /* writes sizeof(rbg_info_t) bytes of THIS header: a client built against another ABI uses ... */
int rbg_info_sized(const rbg_index *, rbg_info_t *out, uint64_t out_bytes);   /* ... this: min(out_bytes, sizeof) bytes; fields are only added at the end from ABI 3 on */

/* What the load decided about the run-indexed layout (RBG_LAYOUT_RUNS), so that no table is left out silently: the
 * reference's structures have no size limits (rle_string.hpp:131-161, toehold_sa.hpp:56-72 are plain uint64_t), and where
 * this layout has one -- or the HBM budget bites -- the decision is here and on stderr.  All zero on the slot layout.
 * out_bytes = sizeof(rbg_layout_info_t) of the caller (fields beyond it are not written: the struct may grow at its end). */
typedef struct rbg_layout_info_t {
    uint32_t run_fmt;                 /* 2 (the per-lane search; the wave-cooperative format 1 of rounds 2-3 was retired in round 5) */
    uint32_t depths_composed;         /* k-mer depths the load composed run lists for (1..8) */
    uint32_t depth_mask_asked;        /* RBG_OPT_RUN_DEPTHS as given (0 = the default rule) */
    uint32_t depth_mask_kept;         /* bit d - 1: depth d has run lists in HBM */
    uint32_t depths_dropped_budget;   /* bit d - 1: depth d was left out because the replica exceeded the HBM budget */
    uint32_t rank_directories;        /* 1: some depth's ranks go through per-table directories (a depth with bucket records has rec_bytes > 0) */
    uint32_t phi_directory;           /* 1: phi goes through its directory */
    uint32_t fill_shift;              /* 8-byte positions: entries of a table lie less than 2^fill_shift rows apart */
    uint64_t entries[8];              /* per depth: entries of its run lists (sentinels and fillers included) */
    uint64_t fillers[8];              /* per depth: filler entries among them (8-byte positions; 0 unless a table has a gap >= 2^fill_shift) */
    uint64_t dir_bytes[8];            /* per depth: bytes of its tables' directories */
    uint64_t phi_entries, phi_fillers, phi_dir_bytes, phi_dir_shift;
    uint64_t phi_slots, phi_slot_bytes; /* phi SLOTS (RBG_OPT_RUN_PHI): their number (buckets of 2^phi_dir_shift text positions) and the
                                         * bytes of slots + ordinals; 0 = phi goes through the list of sampled positions and its directory */
    uint64_t rec_bytes[8];            /* per depth: bytes of its tables' bucket records (RBG_OPT_RUN_REC; 0 = directories) */
    uint64_t rec_overflow[8];         /* per depth: buckets with more entries than a record holds (their ranks go through the run list) */
    uint64_t budget_raised;           /* 1: RBG_LAYOUT_AUTO, no budget given, took three quarters of the free HBM instead of a quarter -- an index of so many runs
                                       * that the quarter would have left it one or two symbols per step (rbg_info().hbm_budget is the budget applied) */
} rbg_layout_info_t;
int rbg_layout_info(const rbg_index *, rbg_layout_info_t *out, uint64_t out_bytes);

/* RowBowt::get_f(), rowbowt.hpp:719 / build_f :770-778: 256 entries. */
int rbg_get_f(const rbg_index *, uint64_t f_out[256]);
/* ToeholdSA::get_last_run_sample(), toehold_sa.hpp:97-99 */
int rbg_last_run_sample(const rbg_index *, uint64_t *out);

/* Host copies of the flattened layout, for layout tests (no compute).  Each call returns the
 * element count in *count and, if dst != NULL, copies min(*count, cap) uint64 values.
 * which: */
enum {
    RBG_ARR_RUN_HEADS = 0,   /* R values (head byte of each BWT run) */
    RBG_ARR_RUN_START = 1,   /* R+1 values (BWT position where each run starts; last = n) */
    RBG_ARR_SAMPLES_LAST = 2,/* r values: ToeholdSA::samples_last_ (toehold_sa.hpp:158) */
    RBG_ARR_PRED_POS = 3,    /* r values: set bits of ToeholdSA::pred_ (:157) */
    RBG_ARR_PHI_BASE = 4,    /* r values: samples_last_[pred_to_run_[j]-1] (:70), 0 where run 0 */
    RBG_ARR_MARKER_START = 5, RBG_ARR_MARKER_END = 6, RBG_ARR_MARKER_OFF = 7, RBG_ARR_MARKER_VALS = 8
};
int rbg_host_array(const rbg_index *, int which, uint64_t *dst, uint64_t cap, uint64_t *count);

/* ---- queries, host buffers (drop-in) ------------------------------------------------------ */

/* RowBowt::LF(range_t, uint8_t c), rowbowt.hpp:74-88: one backward step for N (range, symbol)
 * triples.  An input range outside [0,n) x [0,n) or an absent symbol yields {1,0}. */
int rbg_lf(rbg_index *, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym, uint64_t N,
           uint64_t *lo_out, uint64_t *hi_out);
/* RowBowt::find_range(const std::string&), rowbowt.hpp:121-131, for N reads. */
int rbg_find_range(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                   uint64_t *lo, uint64_t *hi);
/* RowBowt::count, rowbowt.hpp:266-269. */
int rbg_count(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *count);
/* find_range (ssamp == NULL) / find_range_w_toehold for reads that lie scattered in a larger host buffer: read i =
 * base[begin[i], begin[i] + len[i]).  What a FASTA/FASTQ parser that leaves the sequence bytes in its input buffer
 * hands over (the reference copies every read into a std::string first, rb_align.cpp:97): no gather on the host. */
int rbg_find_range_spans(rbg_index *, const uint8_t *base, const uint64_t *begin, const uint32_t *len, uint64_t N,
                         uint64_t *lo, uint64_t *hi, uint64_t *ssamp /* nullable */);

/* RowBowt::find_range_w_toehold, rowbowt.hpp:169-184 (LFData.rn / .ssamp); a failed read gets
 * {1,0}, ssamp=0 (LFData::clear :153-159).  RBG_ENOTLOADED without a toehold SA. */
int rbg_find_range_w_toehold(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t *lo, uint64_t *hi, uint64_t *ssamp);
/* RowBowt::locs_at(range, k, max_hits, locs), rowbowt.hpp:613-615 -> ToeholdSA::locate_range
 * toehold_sa.hpp:37-49, for N (range, toehold) triples.  loc_off[N+1] is written (exclusive scan
 * of min(occ,max_hits)); *locs is malloc()ed by the library with loc_off[N] entries and must be
 * released with rbg_free_buffer.  Per read the order is SA[hi], SA[hi-1], ... (phi chain). */
int rbg_locs_at(rbg_index *, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N,
                uint64_t max_hits, uint64_t *loc_off, uint64_t **locs);
/* RowBowt::markers_at(range_t, vec), rowbowt.hpp:282-285 -> MarkerArray::at_range. */
int rbg_markers_at(rbg_index *, const uint64_t *lo, const uint64_t *hi, uint64_t N,
                   uint64_t *mk_off, uint64_t **mk);
/* RowBowt::find_range_w_markers(query, wsize, max_range), rowbowt.hpp:292-339: final range plus
 * the windowed marker list (window results PREPENDED, :320,:333). */
int rbg_find_range_w_markers(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                             uint64_t wsize, uint64_t max_range,
                             uint64_t *lo, uint64_t *hi, uint64_t *mk_off, uint64_t **mk);
/* Next-row f4 (marker seeds).  RowBowt::get_markers_greedy_seeding(query, wsize, max_range, fn),
 * rowbowt.hpp:406-482, as rb_markers runs it (rb_markers.cpp:411-413): one record per call of the
 * callback fn(range, (q.first, q.second), mbuf), in call order.  ftab_k = 0: no ftab loaded
 * (rb_markers' default, rb_markers.cpp:25); ftab_k = K > 0: with the ftab of k-mer size K loaded
 * (rb_markers --ftab): the first K bases and every restart after a failed seed go through
 * search_ftab (rowbowt.hpp:430-433, :454-464), computed as find_range of the ACGT-only k-mer (see
 * rbg_check_ftab).  A read shorter than K makes the reference throw (substr, :431); here its
 * first lookup counts as a miss.  The reference requires K - 1 <= wsize (:423-426).  The
 * markers are NOT sorted or deduplicated (the caller's out_fn does that, rb_markers.cpp:374-380).
 * Works without a marker array too (every mbuf empty), like the reference (:273, :283). */
typedef struct rbg_marker_seed {
    uint64_t lo, hi;     /* range passed to fn (never empty in this variant) */
    uint64_t qstart;     /* q.first */
    uint64_t qend;       /* q.second + 1 (exclusive end; q.second itself wraps for an empty seed) */
    uint64_t mk_begin;   /* this seed's mbuf = mk[mk_begin, mk_end) */
    uint64_t mk_end;
} rbg_marker_seed_t;
/* seed_off[N+1] written; *seeds (seed_off[N] records) and *mk are malloc()ed by the library:
 * release both with rbg_free_buffer. */
int rbg_get_markers_greedy_seeding(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                                   uint64_t wsize, uint64_t max_range, uint64_t ftab_k,
                                   uint64_t *seed_off, rbg_marker_seed_t **seeds, uint64_t **mk);
/* Next-row f4 (greedy seeding).  RowBowt::get_seeds_greedy_w_sample(query, min_length),
 * rowbowt.hpp:222-256, reduced by the choice locate_from_longest_seed makes (rowbowt.hpp:669-677):
 * per read the FIRST seed of strictly greatest length, as LFData {rn, qstart, qend, ssamp}; a read
 * with no seed of at least min_length gets rn={1,0}, qstart=qend=ssamp=0. */
int rbg_greedy_longest_seed(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t min_length,
                            uint64_t *lo, uint64_t *hi, uint64_t *qstart, uint64_t *qend, uint64_t *ssamp);
/* RowBowt::find_locs_greedy_seeding(s, min_length, max_hits), rowbowt.hpp:633-657 (== get_seeds +
 * locate_from_longest_seed :664-685): locations of the longest seed, each minus the seed's qstart. */
int rbg_find_locs_greedy_seeding(rbg_index *, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t min_length,
                                 uint64_t max_hits, uint64_t *loc_off, uint64_t **locs);
/* gives back a result array one of the calls above allocated (*locs, *mk, *seeds).  Always through this call, never free():
 * large blocks are kept and handed to the next result of about that size -- their pages are already there, which is most of
 * what a 3 GB result costs (RBG_RESULT_POOL=0 in the environment: plain malloc / free). */
void rbg_free_buffer(void *);

/* RowBowt::resolve_offset, rowbowt.hpp:623-625 -> DocList::doc_and_offset_at doclist.hpp:46-50.
 * *name points into the index (valid until rbg_free). */
int rbg_resolve_offset(const rbg_index *, uint64_t i, const char **name, uint64_t *offset);
/* The table rbg_resolve_offset answers from, for callers that resolve positions by the million: position i belongs to
 * document names[k - 1] at offset i - sorted_starts[k - 1], k = # sorted_starts < min(i + 1, size) (doclist.hpp:46-50,
 * :77-79; k == 0 has no answer: rbg_resolve_offset returns RBG_EARG there).  Pointers stay valid until rbg_free. */
int rbg_doc_table(rbg_index *, uint64_t *ndocs, const uint64_t **sorted_starts, const char *const **names, uint64_t *size);

/* `rb_align -s` as text, made on the device (rb_report, rb_align.cpp:118-139): for every read
 *     <name> (<lo>,<hi>), count=<hi - lo + 1>\n\tlocs: <pos>/<doc>:<pos - doc start> ... \n
 * with locs_at (rowbowt.hpp:613-621, max_hits as in rbg_locs_at) and resolve_offset (:623-625, doclist.hpp:46-79) done
 * where the ranges are: the locations never cross PCIe, the decimals are written by kernels, the host gets the finished
 * bytes.  lo / hi / k: host arrays of N (from rbg_find_range_w_toehold / rbg_find_range_spans); the names are N spans of
 * name_base (name_begin[i], name_len[i]).  *text points into a pinned buffer the handle owns: valid until
 * rbg_release_text(ix, *text) (several may be out at once: a writer thread can still hold one while the next batch is
 * made).  The call returns while the text is still being copied out (on the handle's own copy stream, under the caller's
 * next calls): rbg_wait_text(ix, *text) before the first byte is read.  RBG_ENOTLOADED without the toehold SA or the document list; RBG_EARG when a location lies before every
 * document (rbg_resolve_offset's error).  k == NULL: the report without -s, one line per read (no toehold SA or
 * document list needed). */
enum { RBG_TEXT_MARKERS = 1 };   /* flags: also "\tmarkers: <pos>/<allele> ...\n" per read (markers_at, rb_align.cpp:134-142; needs the marker array) */
int rbg_align_text(rbg_index *, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N, uint64_t max_hits, uint32_t flags,
                   const char *name_base, const uint64_t *name_begin, const uint32_t *name_len, const char **text, uint64_t *text_len);
int rbg_wait_text(rbg_index *, const char *text);
int rbg_release_text(rbg_index *, const char *text);
/* make `count` pinned text buffers of `bytes` now (pinning a few hundred MB takes tenths of a second: a tool does it before its clock starts) */
int rbg_reserve_text(rbg_index *, uint64_t bytes, int count);
/* ---- queries, device-resident buffers (HBM in, HBM out; asynchronous on `stream`) ---------- */
/* d_seqs: the reads back to back as in the host calls, in device memory, 16-BYTE ALIGNED (RBG_EARG otherwise), and the allocation must reach the next
 * multiple of 16 bytes at or past its last read's end: the kernels fetch reads as aligned 16-byte chunks (the bytes beyond a read's end are never used).
 * d_off[N + 1]: byte offsets.  Same answers as the host calls (find_range rowbowt.hpp:121-131, find_range_w_toehold :169-184); nothing is allocated, nothing
 * synchronised: graph-capturable. */

int rbg_find_range_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                       uint64_t *d_lo, uint64_t *d_hi, void *stream);
int rbg_find_range_w_toehold_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                                 uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_ssamp, void *stream);
/* Packed reads: the same searches over a 2-bit form of the batch.  One lane per read fetching its
 * own bytes is 7 uncoalesced 16-byte requests per 100 bp; rbg_pack_reads_dev reads the bytes once,
 * coalesced, and writes every read as 2-bit codes in consumption order (64 symbols per 16 bytes)
 * into the caller's workspace; the *_packed_dev searches then take ceil(len/64) requests per read and
 * give the same results as rbg_find_range_dev / rbg_find_range_w_toehold_dev.  Reads with a symbol
 * outside the index's 4 most frequent symbols are searched from the bytes (pass the same d_seqs /
 * d_off).  total_bytes: any upper bound on d_off[N], the same in all three calls;
 * total_bytes/64 + N must stay below 2^32.  Measured on the bench batch (10 M x 100 bp): pack 0.8 ms,
 * searches 0.45-0.55 ms faster each -- worth it when a resident batch is searched more than once. */
size_t rbg_pack_ws_bytes(uint64_t N, uint64_t total_bytes);
int rbg_pack_reads_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t total_bytes,
                       void *d_ws, size_t ws_bytes, void *stream);
int rbg_find_range_packed_dev(rbg_index *, const void *d_ws, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                              uint64_t total_bytes, uint64_t *d_lo, uint64_t *d_hi, void *stream);
int rbg_find_range_w_toehold_packed_dev(rbg_index *, const void *d_ws, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N,
                                        uint64_t total_bytes, uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_ssamp, void *stream);
/* locate, two-phase because the output is ragged:
 *  1. rbg_locate_plan_dev writes d_loc_off[N+1]; d_tmp/tmp_bytes is scratch (query the size with
 *     rbg_locate_plan_tmp_bytes).  Read d_loc_off[N] to size d_locs.
 *  2. (optional) rbg_locate_order_dev; 3. rbg_locate_fill_dev walks the phi chains into d_locs. */
size_t rbg_locate_plan_tmp_bytes(uint64_t N);
int rbg_locate_plan_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N, uint64_t max_hits,
                        uint64_t *d_loc_off, void *d_tmp, size_t tmp_bytes, void *stream);
/* Optional, any time before fill: order the phi chains by toehold text position.  Chains of reads
 * from nearby loci visit the haplotypes in the same order, so neighbouring lanes then touch
 * neighbouring slots for the whole walk (k_locate_fill 9 -> 2.3 ms per 10M reads on the bench
 * index; the sort costs 0.36 ms).  Results are unchanged.  d_ws: 256-byte aligned, N < 2^32-1. */
size_t rbg_locate_order_ws_bytes(uint64_t N);
int rbg_locate_order_dev(rbg_index *, const uint64_t *d_k, uint64_t N, void *d_ws, size_t ws_bytes, void *stream);
/* d_order: the workspace prepared by rbg_locate_order_dev for the same d_k, or NULL (input order).
 * With d_order the walk takes each read's toehold from the workspace (it travelled with the sort) and
 * its count from d_loc_off (rbg_locate_plan_dev with the same max_hits): d_lo/d_hi/d_k are then not read.
 * d_locs needs 8-byte alignment only: the walk stores whole 64-byte windows of the array wherever it starts. */
int rbg_locate_fill_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                        uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const void *d_order, void *stream);
/* locate_fill with a per-read value subtracted from every location (d_sub nullable): the
 * `locs[i] - best_range.qstart` of locate_from_longest_seed, rowbowt.hpp:681-683 */
/* The same walk storing each location as 32 bits (d_loc_off still counts locations): for device pipelines on an
 * index with 4-byte positions (rbg_info().pos_bytes == 4, i.e. n < 2^32 - 16; RBG_EARG otherwise) -- half the write
 * requests of the step that dominates count+locate.  The values are the low 32 bits of rbg_locate_fill_dev's, so a
 * toehold that wrapped below zero (a match at text position 0 of a range wider than one row) reads 0xFFFFFFFF.
 * The host API and the reference's signature (vector<uint64_t>, toehold_sa.hpp:37-49) keep 64 bits. */
int rbg_locate_fill_dev32(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                          uint64_t max_hits, const uint64_t *d_loc_off, uint32_t *d_locs32, const void *d_order, void *stream);
int rbg_locate_fill_offset_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                               uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const uint64_t *d_sub,
                               const void *d_order, void *stream);
int rbg_greedy_longest_seed_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t min_length,
                                uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_qstart, uint64_t *d_qend, uint64_t *d_ssamp,
                                void *stream);
/* marker seeds, two-phase: plan writes the exclusive scans d_seed_off[N+1] (records per read) and
 * d_mk_off[N+1] (markers per read); the caller sizes d_seeds / d_mk from their last entries. */
int rbg_marker_seeds_plan_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                              uint64_t max_range, uint64_t ftab_k, uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes,
                              void *stream);
int rbg_marker_seeds_fill_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                              uint64_t max_range, uint64_t ftab_k, const uint64_t *d_seed_off, const uint64_t *d_mk_off,
                              rbg_marker_seed_t *d_seeds, uint64_t *d_mk, void *stream);
/* The same two phases with a LOG between them, so that the reads are walked once: the plan also writes, per sequence, its
 * seed records and where its markers sit in the marker array into d_log (rbg_marker_seeds_log_bytes(ix, N, q) bytes for a
 * quota of q seeds per sequence, 0 = the default 12; 16-byte aligned; any larger area raises the quota), the fill copies
 * from there and walks only the sequences that exceeded their quota (listed at the log's end).  Same outputs as the pair
 * above; d_seeds 16-byte aligned; d_log must be left untouched between the two calls.  On the bench index the pair
 * takes 43 ms per 10 M reads (both strands) walking twice and about half with the log. */
size_t rbg_marker_seeds_log_bytes(const rbg_index *, uint64_t N, uint32_t seeds_per_read);
int rbg_marker_seeds_plan_log_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                                  uint64_t max_range, uint64_t ftab_k, uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp,
                                  size_t tmp_bytes, void *d_log, size_t log_bytes, void *stream);
int rbg_marker_seeds_fill_log_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize,
                                  uint64_t max_range, uint64_t ftab_k, const uint64_t *d_seed_off, const uint64_t *d_mk_off,
                                  rbg_marker_seed_t *d_seeds, uint64_t *d_mk, void *d_log, size_t log_bytes, void *stream);
/* markers, same two-phase shape */
int rbg_markers_plan_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N,
                         uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes, void *stream);
int rbg_markers_fill_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, uint64_t N,
                         const uint64_t *d_mk_off, uint64_t *d_mk, void *stream);

/* ---- global counters (the values the 8-GPU run reduces over RCCL) -------------------------- */
/* {reads processed, reads matched (non-empty range), sum of occ, sum of located positions}
 * accumulated on the device by every query since load / the last reset. */
int rbg_counters(rbg_index *, uint64_t out[4]);
int rbg_counters_reset(rbg_index *);
/* One-read calls (N = 1) of rbg_find_range / rbg_count / rbg_find_range_w_toehold / rbg_get_markers_greedy_seeding
 * made concurrently by several host threads are combined into one batched launch per round: what lets a caller
 * written against the reference's one-query-at-a-time methods from a thread pool (its only parallel dispatcher:
 * rb_markers.cpp:318-535) get batched launches unmodified.  out = {launches, requests served by them} since load;
 * requests / launches is the mean batch size.  (Environment RBG_HOST_COMBINE=0 switches the combining off.) */
int rbg_combine_stats(rbg_index *, uint64_t out[2]);

/* ---- several GPUs (index replicated, read stream sharded, no data-path collective; the reference is
 * single-device, its only parallel dispatcher is rb_markers' thread pool, rb_markers.cpp:318-535) --------------- */
/* A further replica of `primary`'s device index on `device`: built once, copied peer to peer (hipMemcpyPeer over
 * xGMI), its pointer-bearing records re-pointed.  The handle works with every query entry point above (host or
 * *_dev), shares the host-side index with the primary and must be freed before it; markers / docs are attached
 * to the primary BEFORE replicating. */
int rbg_replicate(rbg_index *primary, int device, rbg_index **replica_out);
/* G replicas at once: every target's peer copies are enqueued (one stream per target) before any is waited for, so the
 * transfers overlap -- each MI355X has its own xGMI link to the primary, the fan-out takes about one copy's time.
 * All or nothing: on failure no replica is left behind.  devices[] may repeat (tests put several on one device). */
int rbg_replicate_many(rbg_index *primary, const int *devices, int G, rbg_index **replicas_out /* [G] */);
/* What the fan-out cost one replica (measurement plumbing of the multi-GPU start, SURVEY 8e's "one-time distribution by parallel
 * peer copies"; the reference has no counterpart): out = {milliseconds its copies took on its own stream (HIP events), bytes
 * copied, peer access to the primary's device (1 = direct, 0 = staged through the host by the runtime, -1 = same device)}. */
int rbg_replicate_stats(const rbg_index *replica, double out[3]);
/* contiguous block [begin, end) of `rank` out of `world` (sizes differ by at most one; concatenating the ranks'
 * outputs restores the input order): read i of N goes to rank i * world / N */
int rbg_shard_bounds(uint64_t n_items, int rank, int world, uint64_t *begin, uint64_t *end);
/* find_range (ssamp == NULL) / find_range_w_toehold over G replicas: shard g of the batch runs on replicas[g],
 * the G shards concurrently; outputs in input order. */
int rbg_find_range_sharded(rbg_index *const *replicas, int G, const uint8_t *seqs, const uint64_t *off, uint64_t N,
                           uint64_t *lo, uint64_t *hi, uint64_t *ssamp /* nullable */);
/* The run's only collective: one RCCL all-reduce (sum) of the four counters of rbg_counters.  `nccl_comm` is the
 * caller's ncclComm_t for this replica's device (one rank per GPU), `stream` a HIP stream or NULL. */
int rbg_counters_allreduce(rbg_index *, void *nccl_comm, void *stream, uint64_t out[4]);
/* the same for G replicas on G distinct devices held by one process: one grouped all-reduce over a communicator
 * clique that is made on the first call for that set of devices (ncclCommInitAll) and kept for the later ones;
 * rbg_comm_cache_clear() destroys the kept cliques.  RCCL itself is opened on the first call that needs it
 * (dlopen): librbg.so does not link it, and RBG_ENODEV is the answer where it is absent. */
int rbg_counters_allreduce_local(rbg_index *const *replicas, int G, uint64_t out[4]);
int rbg_comm_cache_clear(void);

/* ---- measurement: what a launch touched (SURVEY 8d "report mean executed steps"; the reference's only
 * instrumentation is the stderr timer line rb_align.cpp:192) ------------------------------------------ */
/* The *_stats_dev calls run an INSTRUMENTED instantiation of the same kernel on the same arguments (same
 * outputs, same counters) and add what it touched to a device array of RBG_SEARCH_STATS / RBG_LOCATE_STATS
 * 64-bit sums, which the caller zeroes first.  bench.py derives the bytes of the algorithm as run from them;
 * the timed launches are the plain ones. */
/* On the run-indexed layout (RBG_LAYOUT_RUNS) the same sums count that layout's accesses: RBG_SS_SLOTS = directory
 * gathers (8 bytes each), RBG_SS_DENSE = run-list entries the probes needed (2 positions' width each; at most 16 per
 * probe), RBG_SS_SEARCH = narrowing rounds of crowded buckets (16 pivot keys each), RBG_SS_RESAMPLE = one sample gather
 * each; RBG_LS_PHI_STEPS = phi evaluations (one 8-byte directory gather + one probe each), RBG_LS_PHI_SEARCH = sampled
 * positions those probes and their narrowing rounds needed. */
enum { RBG_SS_STEPS = 0,      /* LF gathers issued (single-symbol or k-mer steps) */
       RBG_SS_SLOTS,          /* 16-byte rank slots loaded (1 per step, 2 when lo and hi+1 fall in different buckets) */
       RBG_SS_DENSE,          /* 2-byte loads from dense overflow tables */
       RBG_SS_SEARCH,         /* ranks answered by a run-list search (overflow bucket without a dense table) */
       RBG_SS_FTAB,           /* ftab entries fetched */
       RBG_SS_RESAMPLE,       /* toehold re-samples materialised (2 gathers each) */
       RBG_SS_CHUNKS,         /* aligned 16-byte chunks of read bytes fetched */
       RBG_SS_SYMBOLS,        /* read symbols consumed = reference LF iterations covered */
       RBG_SEARCH_STATS };
enum { RBG_LS_PHI_STEPS = 0,  /* phi evaluations (one phi slot each) */
       RBG_LS_PHI_SEARCH,     /* of them: answered by a run-list search */
       RBG_LS_CHAINS,         /* reads with at least one location */
       RBG_LS_LOCS,           /* locations stored */
       RBG_LOCATE_STATS };
/* The seeding kernels' instrumented instantiations (run-indexed layout only; RBG_EARG elsewhere): RBG_SEED_STATS sums -- the eight of
 * RBG_SEARCH_STATS with that layout's meanings, then the marker side.  rbg_marker_seeds_stats_dev = rbg_marker_seeds_plan_dev +
 * rbg_marker_seeds_fill_dev (the two-walk pair, no log, no --ftab) with both walks instrumented: same offsets, records and markers. */
enum { RBG_SD_MARKER_QUERIES = 8, /* window queries that went to the marker runs (rowbowt.hpp:437-441 with range <= max_range) */
       RBG_SD_MARKER_DIR,         /* bucket records read for them (32 bytes each: one per end of the range, one when both ends share a bucket; with RBG_MK_REC=0
                                     directory entries, 4 bytes each) */
       RBG_SD_MARKER_PROBES,      /* run starts / ends read from the arrays (8 bytes each): only behind an overflowing bucket's record */
       RBG_SD_MARKER_OFF,         /* value offsets read from the array (8 bytes each): likewise */
       RBG_SD_MARKER_VALS,        /* marker values copied (8 read + 8 written each) */
       RBG_SD_SEED_RECS,          /* seed records written (48 bytes each) */
       RBG_SD_SEQUENCES,          /* sequences walked */
       RBG_SEED_STATS };
int rbg_greedy_longest_seed_stats_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t min_length,
                                      uint64_t *d_lo, uint64_t *d_hi, uint64_t *d_qstart, uint64_t *d_qend, uint64_t *d_ssamp,
                                      uint64_t *d_stats /* RBG_SEED_STATS */, void *stream);
int rbg_marker_seeds_stats_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t wsize, uint64_t max_range,
                               uint64_t *d_seed_off, uint64_t *d_mk_off, void *d_tmp, size_t tmp_bytes, rbg_marker_seed_t *d_seeds,
                               uint64_t *d_mk, uint64_t *d_stats /* RBG_SEED_STATS */, void *stream);
int rbg_find_range_stats_dev(rbg_index *, const uint8_t *d_seqs, const uint64_t *d_off, uint64_t N, uint64_t *d_lo,
                             uint64_t *d_hi, uint64_t *d_ssamp /* NULL = count-only kernel */,
                             uint64_t *d_stats /* RBG_SEARCH_STATS */, void *stream);
int rbg_locate_fill_stats_dev(rbg_index *, const uint64_t *d_lo, const uint64_t *d_hi, const uint64_t *d_k, uint64_t N,
                              uint64_t max_hits, const uint64_t *d_loc_off, uint64_t *d_locs, const void *d_order,
                              uint64_t *d_stats /* RBG_LOCATE_STATS */, void *stream);

/* Synthetic reads generated on the device (measurement plumbing for BASELINE.json configs[3], "1B synthetic 150 bp
 * reads": the host cannot feed them, and the reference has no generator -- its reads come from a FASTQ,
 * rb_align.cpp:169-178).  Counter-based: read g = first_read + i is a pure function of (seed, g): a haplotype
 * g_h < H, an offset in [0, L - m] inside it (text position g_h * unit + offset of a text laid out as H units of
 * `unit` symbols whose first L are the haplotype), and with probability sub_ppm / 1e6 one substituted base.
 * Writes N reads of m bytes at stride m into d_seqs (16-byte aligned), d_off[N+1], and (optional) each read's
 * text position.  Needs no index. */
int rbg_sample_reads_dev(const uint8_t *d_text, uint64_t unit, uint64_t H, uint64_t L, uint64_t m, uint64_t seed,
                         uint64_t first_read, uint64_t N, uint32_t sub_ppm, uint8_t *d_seqs, uint64_t *d_off,
                         uint64_t *d_start /* nullable */, void *stream);
/* The same reads (byte for byte, for the same seed) from the STRUCTURE of such a text instead of the text: a pangenome of
 * n = 3e11 symbols does not fit the HBM it would be sampled from.  Symbol (h, p) is d_base[p] (L bytes), or d_alt[j] when p
 * is variant site j (d_sites: S ascending offsets) and haplotype h carries the alternative allele (d_G[j * H + h] != 0).
 * d_site_dir (nullable): d_site_dir[b] = # sites below b << site_dir_shift, (L >> site_dir_shift) + 2 entries (S < 2^32): the first
 * site of a read is then found in its bucket instead of by a search over all sites. */
int rbg_sample_reads_pangenome_dev(const uint8_t *d_base, const uint64_t *d_sites, const uint8_t *d_alt, const uint8_t *d_G, uint64_t S,
                                   const uint32_t *d_site_dir, uint32_t site_dir_shift, uint64_t unit, uint64_t H, uint64_t L, uint64_t m,
                                   uint64_t seed, uint64_t first_read, uint64_t N, uint32_t sub_ppm, uint8_t *d_seqs, uint64_t *d_off,
                                   uint64_t *d_start /* nullable */, void *stream);

/* ---- tuning (never changes results) -------------------------------------------------------- */
/* Process-wide defaults read when an index is built/loaded: BLOCK_THREADS (64, 128, 192 or 256; the search
 * kernels use 1024-thread workgroups instead while the 5-mer level is resident),
 * RANK/PHI_BUCKET_SHIFT (-1 = automatic, else 0..8; RANK also 9..12 = wide buckets), DEEP_BUCKET_SHIFT (-1 = like
 * RANK, else the bucket shift of the 4-mer and deeper tables only, whose runs are sparse), POS_BYTES (0 = automatic, 4 or 8 to force a width),
 * KMER_STEPS (1..8, default 8: symbols the backward search consumes per step; the slot layout takes at most 5 of them per gather; 2.. build the k-mer
 * tables of DESIGN.md 2b -- each level is four times the tables of the one before, 218 GB in all for a
 * 2-Gbase index; 1 keeps the reference's one-symbol steps only; the deepest levels are dropped
 * automatically when the replica would not fit), HBM_BUDGET_MB (0 = A QUARTER of the free HBM -- three quarters until round 3; a drop-in library leaves the device to its caller unless told otherwise:
 * upper bound for the replica, deciding how many k-mer levels are kept.  ONE EXCEPTION, reported by rbg_layout_info().budget_raised and on stderr: under
 * RBG_LAYOUT_AUTO with no budget given, an index of so many runs that the quarter would leave it fewer than four symbols per step (r about 1e9) takes up
 * to three quarters -- and only on a device the load has to itself (at least nine tenths of it free: with the caller's own buffers, another replica or
 * another process already there the quarter stays); any explicit HBM_BUDGET_MB switches the exception off), FTAB_K (-1 = automatic (the longest word of at most 12 symbols with 4^k <= n/16),
 * 0 = no ftab, else the word length of the ftab built on the GPU at load time: the state after the
 * last FTAB_K symbols of a read is one gather; result-neutral like the reference's ftab,
 * rowbowt.hpp:124-125,726-758).
 * DENSE_OVERFLOW (1 = default): buckets with more run starts than a slot holds get a two-bytes-per-row table
 * (512 bytes per such bucket, about 1.2 % on top of the replica) so that their rank is one more load
 * instead of a search of the run list; 0 = search the run list.
 * PACKED_READS applies to the host-pointer search calls, at call time: how the reads cross PCIe -- 0 = as bytes,
 * 1 (default) = as 2-bit codes packed by CPU threads for batches of >= 4096 reads (a quarter of the bytes; reads
 * holding symbols outside the index's four k-mer symbols are searched from their bytes afterwards), 2 = always
 * as 2-bit codes.  (Packing a batch that is already in HBM is rbg_pack_reads_dev.)
 * RANK_LAYOUT: RBG_LAYOUT_AUTO (default): slot tables while those of every requested symbol per step (KMER_STEPS), at their
 * narrow buckets, fit the HBM budget; otherwise the run-indexed layout rather than slot tables with wider buckets or fewer
 * symbols per step (on the bench index at the default budget 1.35-1.47e9 count+locate reads/s from 12 GB at eight symbols per step,
 * against 1.26e9 from the 221 GB of five symbols per gather and 1.18e9 from the 59 GB of four: profiles/r05_bench.json) -- unless that does
 * not fit either (about 110 bytes per run).  RBG_LAYOUT_PREFER_SLOTS: slot tables with as many symbols per step as fit, the
 * run-indexed layout only when not even the single-symbol level does (the rule of rounds 2-3; the seeding kernels are a quarter faster
 * on slot tables).  rbg_info::rank_layout reports the outcome.
 * RBG_LAYOUT_SLOTS, RBG_LAYOUT_RUNS = the run-indexed layout: the run lists of the k-mer depths RUN_DEPTHS names (by default the
 * deepest KMER_STEPS asks for and the budget holds, half of it, a quarter of it ... and 1) with a directory or bucket records per
 * table, space proportional to r and nothing proportional to n; rank and phi are predecessor searches over the few entries of
 * one bucket by the lane that owns the query (rle_string::rank rle_string.hpp:131-161 / ToeholdSA::phi toehold_sa.hpp:56-72 keep
 * their O(r) shape). */
/* Environment switches (read at load or per call; none of them changes an answer -- they exist for A/B measurements
 * and for the tests that pin both sides):  RBG_PHI_PACKED=0 keeps 32-byte phi slots at 8-byte positions;
 * RBG_RANK_DIR_RUNS=<x> sets the runs per rank-directory bucket of the run-indexed layout (default 4), RBG_RUN_REC_PER=<x> the entries
 * per bucket record (default 2.5), RBG_PHI_DIR_PER=<x> the sampled positions per phi-directory bucket (default 1..2);
 * RBG_RUN_FILL_SHIFT / RBG_PHI_SUPER_SHIFT lower the filler distance / super-count spacing of 8-byte positions so that tests
 * meet both on small indexes;  RBG_HOST_THREADS, RBG_HOST_CHUNK_READS, RBG_HOST_DIRECT_OUT=0, RBG_HOST_COMBINE=0,
 * RBG_HOST_TRACE=1|2 tune / trace the host-pointer pipeline;  RBG_LAYOUT=auto|slots|runs|prefer-slots, RBG_RUN_DEPTHS,
 * RBG_KMER_STEPS, RBG_HBM_BUDGET_MB, RBG_FTAB_K, RBG_RUN_PHI, RBG_RUN_REC give the options of the same names their initial values (for
 * the command-line tools, which keep the reference's flags; rbg_set_default_option overrides them);  RBG_H2D_STAGED=0 uploads the big
 * arrays of a load by plain hipMemcpy;  RBG_VERBOSE=1 prints what the budget rule did and the seconds of every stage of a load. */
enum { RBG_OPT_BLOCK_THREADS = 1, RBG_OPT_RANK_BUCKET_SHIFT = 2, RBG_OPT_PHI_BUCKET_SHIFT = 3, RBG_OPT_POS_BYTES = 4,
       RBG_OPT_KMER_STEPS = 5, RBG_OPT_HBM_BUDGET_MB = 6, RBG_OPT_FTAB_K = 7, RBG_OPT_PACKED_READS = 8,
       RBG_OPT_DEEP_BUCKET_SHIFT = 9, RBG_OPT_DENSE_OVERFLOW = 10, RBG_OPT_RANK_LAYOUT = 11,
       /* 12, 13 and 15 were RBG_OPT_TREE_TOP_KB, RBG_OPT_SLOT_BYTES and RBG_OPT_RUN_FMT (retired with ABI 3: RBG_EARG) */
       RBG_OPT_RUN_DEPTHS = 14 /* run-indexed layout: bit d - 1 set = keep run lists for the k-mer depth d (bit 0 is implied; depths
                                  above the highest bit are not built).  A search step consumes the longest stretch a kept
                                  depth covers, so any set gives the same answers; the lists grow with the depth (DESIGN.md
                                  2c).  0 = default: the deepest depth RBG_OPT_KMER_STEPS asks for, half of it, a quarter of
                                  it ... and 1 (1, 2, 4, 8 of eight: whole reads go by eight symbols a step, a remainder takes
                                  one step per set bit); 0xFF keeps all eight; over budget the depths between the first and
                                  the deepest go before the deepest does.  rbg_info(): kmer_steps is the deepest depth kept,
                                  depth_runs[d - 1] is 0 for the depths left out */,
       RBG_OPT_RUN_PHI = 16 /* run-indexed layout -- how phi (toehold_sa.hpp:56-72) is answered: 1 = from the list of sampled positions
                                  through its directory (12-16 bytes per run: two dependent sectors per step); 2 = from direct-addressed phi SLOTS
                                  (the slot layout's PhiSlot records) whose buckets are about n / r rows wide, so that their number is proportional
                                  to r (about 54 bytes per run at 8-byte positions: one sector per step -- at pangenome scale K3 is bound by that
                                  count); 0 (default) = slots when the whole replica then stays within the HBM budget.  RBG_RUN_PHI gives the
                                  initial value; rbg_layout_info().phi_slots says what was built. */,
       RBG_OPT_RUN_REC = 17 /* run-indexed layout -- BUCKET RECORDS: 2 = every bucket of a table (about three entries wide) gets one aligned
                                  64-byte record holding its entries and the one before them (up to eleven in the compact form, six otherwise; a bucket with
                                  more holds twelve pivots into the run list instead: rbg_dev.h RunRec2), direct-addressed: a rank is ONE sector instead of a
                                  directory sector plus an unaligned stretch of the run list (K1/K2 on this layout are bound by that count); about
                                  21-26 bytes per entry on top of the run lists, which stay for crowded buckets and the samples; 1 = directories only;
                                  0 (default) = decided PER DEPTH, deepest first (where a search spends its steps): a depth gets records -- at 2.5, else 4,
                                  else 6 entries per bucket -- while the replica with them (and with the phi slots, which come first) stays within the HBM budget.  RBG_RUN_REC gives the initial
                                  value, RBG_RUN_REC_PER the entries per bucket; rbg_layout_info().rec_bytes says what was built, per depth. */,
       RBG_OPT_RUN_REC_DEPTHS = 18 /* with RBG_OPT_RUN_REC = 2: bit d - 1 = the k-mer depth d gets bucket records, the other kept depths keep their
                                  directories (0, the default: every kept depth).  An index of r = 1e9 runs has room for the records of its deepest depth
                                  but not of all (profiles/r05_pangenome_stream_r1e9.json).  RBG_RUN_REC_DEPTHS gives the initial value. */ };
int rbg_set_default_option(int opt, int64_t value);
/* the value a later load would use (so that a caller can change a knob for one load and put it back) */
int rbg_get_default_option(int opt, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* RBG_H */
