// capi/core.ipp -- the handle (rbg_index), the tuning options, device / host scratch pools, the arena, the slot layout's upload helpers.
// Part of rbg_capi.hip (one translation unit: the parts share the pools and options of this anonymous namespace).
using namespace rbg;
static_assert(kMaxRunDepth == kMaxKmerDepth && kLdsRunDepth == kMaxSlotKmerDepth, "rbg_dev.h and rbg_host.hpp name the same depths");

#include "../rbg_hostpath.hpp"

struct DevAlloc {
    void *p;
    size_t bytes;
};
// a device array of records that hold device pointers (DevSym): what a peer copy has to re-point
struct PtrTable {
    const void *d_ptr;
    size_t count, stride;
    std::vector<size_t> ptr_offsets;
};

struct rbg_index {
    HostIndex host;
    rbg_index *primary = nullptr;  // set in a replica handle (rbg_replicate): the host-side index lives in the primary
    HostIndex &H() { return primary ? primary->host : host; }
    const HostIndex &H() const { return primary ? primary->host : host; }
    int device = RBG_DEVICE_NONE;
    DevIndex dev{};
    LaunchCfg cfg;
    std::vector<DevAlloc> allocs;  // every device allocation of the replica
    std::vector<PtrTable> ptr_tables;
    uint64_t hbm_bytes = 0;
    void *arena = nullptr;       // one allocation holding every table of the replica
    size_t arena_bytes = 0, arena_used = 0;
    uint64_t rank_slots = 0, rank_slots_overflow = 0, phi_slots = 0, phi_slots_overflow = 0;
    uint64_t kmer_steps_requested = 0, hbm_free_at_load = 0, hbm_budget = 0;  // how the space/speed point was chosen (rbg_info)
    bool runs_layout = false;
    double replicate_ms = 0.0;     // replica handle: duration of its peer copies on its own stream (HIP events; rbg_replicate_stats)
    int replicate_peer = -1;       // replica handle: 1 = peer access to the source's device was enabled, 0 = the runtime stages the copy, -1 = same device / primary
    uint64_t plan_free = 0, plan_budget = 0;   // free HBM and replica budget as options_for() saw them BEFORE anything of this load was on the device (0: not taken)
    bool budget_raised = false;                // RBG_LAYOUT_AUTO raised the default budget from a quarter to three quarters of the free HBM (an index too large for the quarter)
    bool auto_runs = false;        // RBG_LAYOUT_AUTO chose the run-indexed layout because the slot tables of every requested symbol per step exceed the budget
    bool runs_forced = false;      // the composition already gave back the depths the run-indexed layout leaves out: no way back to slot tables
    uint32_t run_depth_mask = 0;   // run-indexed layout: the k-mer depths that have run lists (bit d - 1)
    // what the load decided about the run-indexed layout (rbg_layout_info): nothing is left out without a line here
    struct RunsReport {
        uint32_t fmt = 0, depth_mask_asked = 0, depth_mask_kept = 0, depths_composed = 0;
        uint64_t entries[kMaxRunDepth] = {}, fillers[kMaxRunDepth] = {}, dir_bytes[kMaxRunDepth] = {};
        uint64_t phi_entries = 0, phi_fillers = 0, phi_dir_bytes = 0, phi_dir_shift = 0;
        uint32_t rank_dirs = 0, phi_dir = 0;       // 1: present
        uint32_t depths_dropped_budget = 0;        // mask of depths the HBM budget left out
        uint64_t phi_slots = 0, phi_slot_bytes = 0;   // format 2 with phi slots (RBG_OPT_RUN_PHI): their number and bytes (slots + ordinals)
        uint64_t rec_bytes[kMaxRunDepth] = {}, rec_overflow[kMaxRunDepth] = {};   // bucket records (RBG_OPT_RUN_REC)
    } runs_report;
    // one-read host calls from concurrent threads are combined into one launch ("group commit", see Combiner below)
    struct Combiner {
        std::mutex mu;
        std::condition_variable cv;
        bool leader = false;
        std::vector<void *> pending;
    } comb_range, comb_seeds;
    std::atomic<uint64_t> comb_launches{0}, comb_requests{0};
    std::mutex ws_mu;            // host-call workspaces (rbg_hostpath.hpp): one per concurrent caller, kept for reuse
    std::vector<std::unique_ptr<rbg_hostpath::Workspace>> ws_free;
    std::vector<ComposedLevel> kmer_levels;  // k-mer depths composed on the device (k_compose.hip): [0] = depth 2; arrays listed in `allocs`
    std::vector<DevSym> dense_todo;  // load time only: rank tables whose overflow buckets still need their dense tables
    std::vector<const char *> doc_name_ptrs;  // rbg_doc_table's view of the document names
    // rbg_align_text: the document table on this handle's device (made at the first call) and the pinned buffers its texts are
    // copied out into (handed to the caller until rbg_release_text)
    struct TextDocs { const uint64_t *start = nullptr; const char *names = nullptr; const uint32_t *name_off = nullptr; uint64_t n = 0, size = 0; } text_docs;
    // (a text is copied out on the handle's own copy stream while the caller goes on: `done` is recorded behind the copy, the
    //  device-side text block goes back to the scratch pool once it has been waited for)
    struct TextOut { char *p = nullptr; size_t cap = 0; bool busy = false, pending = false; hipEvent_t done = nullptr; void *d_text = nullptr; size_t d_cls = 0; int d_dev = 0; };
    std::vector<TextOut> text_out;
    hipStream_t text_copy_stream = nullptr;
    struct TextIn { char *p = nullptr; size_t cap = 0; bool busy = false; };   // pinned staging of a call's inputs (ranges, names)
    std::vector<TextIn> text_in;
    std::mutex text_mu;
    std::mutex mu;               // guards marker/doc attachment only; queries are lock-free
};

namespace {

// Initial values of the load-time knobs a command-line user may need (rb_align / rb_markers / rb_build keep the reference's
// flags, so these come by environment): RBG_LAYOUT = auto | slots | runs, RBG_RUN_DEPTHS = mask, RBG_KMER_STEPS = 1..5,
// RBG_HBM_BUDGET_MB, RBG_FTAB_K = -1..16.  rbg_set_default_option overrides them; a value out of range is reported and ignored.
int64_t env_opt(const char *name, int64_t dflt, int64_t lo, int64_t hi) {
    const char *e = std::getenv(name);
    if (!e || !*e) return dflt;
    if (std::strcmp(name, "RBG_LAYOUT") == 0) {
        if (std::strcmp(e, "auto") == 0) return RBG_LAYOUT_AUTO;
        if (std::strcmp(e, "slots") == 0) return RBG_LAYOUT_SLOTS;
        if (std::strcmp(e, "runs") == 0) return RBG_LAYOUT_RUNS;
        if (std::strcmp(e, "prefer-slots") == 0) return RBG_LAYOUT_PREFER_SLOTS;
    }
    char *end = nullptr;
    const long long v = std::strtoll(e, &end, 0);
    if (end == e || *end || v < lo || v > hi) {
        std::fprintf(stderr, "rbg: %s=%s ignored (expected %lld..%lld)\n", name, e, static_cast<long long>(lo), static_cast<long long>(hi));
        return dflt;
    }
    return v;
}
std::atomic<int64_t> g_opt_block_threads{256};
std::atomic<int64_t> g_opt_rank_shift{-1};
std::atomic<int64_t> g_opt_phi_shift{-1};
std::atomic<int64_t> g_opt_pos_bytes{0};
std::atomic<int64_t> g_opt_kmer_steps{env_opt("RBG_KMER_STEPS", kMaxKmerDepth, 1, kMaxKmerDepth)};   // (the slot layout stages at most kMaxSlotKmerDepth = 5)
std::atomic<int64_t> g_opt_hbm_budget_mb{env_opt("RBG_HBM_BUDGET_MB", 0, 0, int64_t(1) << 40)};
std::atomic<int64_t> g_opt_ftab_k{env_opt("RBG_FTAB_K", -1, -1, 16)};
std::atomic<int64_t> g_opt_deep_shift{-1};
std::atomic<int64_t> g_opt_dense_overflow{1};
std::atomic<int64_t> g_opt_rank_layout{env_opt("RBG_LAYOUT", RBG_LAYOUT_AUTO, RBG_LAYOUT_AUTO, RBG_LAYOUT_PREFER_SLOTS)};    // RBG_LAYOUT_AUTO / _SLOTS / _RUNS / _PREFER_SLOTS
// the two automatic settings (include/rbg.h): both take the run-indexed layout when not even the single-symbol slot tables fit the budget;
// RBG_LAYOUT_AUTO also when the slot tables would have to give up symbols per step for it (rbg_index::auto_runs, decided by options_for)
// The k-mer depths that get run lists when RBG_OPT_RUN_DEPTHS names none: the deepest K, then K / 2, K / 4, ... and 1 (of eight: 1, 2, 4, 8).
// A search step consumes the longest stretch a kept depth covers, so whole reads go by K symbols a step and the remainder of a
// read (or of a seed) takes one step per set bit; every depth kept costs its run lists.
inline uint32_t default_depth_mask(uint32_t K) {
    uint32_t mask = 1u;
    for (uint32_t d = K; d >= 1; d /= 2) mask |= 1u << (d - 1);
    return mask;
}
inline bool layout_automatic() { const int64_t v = g_opt_rank_layout.load(); return v == RBG_LAYOUT_AUTO || v == RBG_LAYOUT_PREFER_SLOTS; }
std::atomic<int64_t> g_opt_run_depths{env_opt("RBG_RUN_DEPTHS", 0, 0, (1 << kMaxRunDepth) - 1)};    // run-indexed layout: bit d - 1 = keep the k-mer depth d (0 = default_depth_mask: the deepest, half of it, a quarter ..., 1)
std::atomic<int64_t> g_opt_run_phi{env_opt("RBG_RUN_PHI", 0, 0, 2)};   // run-indexed layout, format 2: 0 = automatic, 1 = phi over the sampled-position list (12-16 bytes per run), 2 = phi SLOTS of about n/r rows (about 54 bytes per run at 8-byte positions; one sector per step instead of two)
std::atomic<int64_t> g_opt_run_rec_depths{env_opt("RBG_RUN_REC_DEPTHS", 0, 0, (1 << kMaxRunDepth) - 1)};   // with RBG_OPT_RUN_REC = 2: the depths (bit d - 1) that get bucket records; 0 = every kept depth
std::atomic<int64_t> g_opt_run_rec{env_opt("RBG_RUN_REC", 0, 0, 2)};   // run-indexed layout, format 2: bucket records (rbg_dev.h RunRec2) -- 0 = automatic (when the replica with them stays within half the budget), 1 = off, 2 = on
std::atomic<int64_t> g_opt_packed_reads{1};  // host-pointer calls: 0 bytes over PCIe, 1 (default) 2-bit codes for batches >= 4096, 2 always

// The replica's share of the free HBM when no budget is given (RBG_OPT_HBM_BUDGET_MB): A QUARTER.  Until round 3 a default
// rbg_load took three quarters -- the bench index then got its 5-symbol slot level (218 GB) for the last 10-15 % of K1/K2's
// speed and left its caller 80 GB of a 288 GB device.  A drop-in library should leave the device to its caller unless told
// otherwise: with a quarter the same load keeps the 4-symbol level (58 GB), and the budget option is one call away.
inline size_t default_budget(size_t free_b) { return free_b / 4; }
// How many symbols per step of the run-indexed layout are worth composing, estimated BEFORE composing: a depth adds at most about 0.62 r runs to
// the one before it (measured 0.55-0.69 r per depth at r = 1.07e9, n / r = 282; 0.33 r on the bench index), pieces are indexed with 32 bits, the
// sweeps hold about 70 bytes per piece of the depth being made (profiles/r04_pangenome_stream_r1e9_k5.log), and -- with_budget -- the least the budget
// rule of upload() keeps of a depth K (the single symbols, K itself, phi) must fit the budget at 18 bytes per entry.
// RBG_ASSUME_FREE_HBM_MB (tests only): the free HBM the PLANNING of a load assumes (budget, composition depth), capped to this many MiB -- so that
// the rules an index of r = 1e9 runs meets on a 288 GB device (budget raised, depth planned before composing) are exercised by a test-sized index.
// Allocation itself is not limited by it.
inline size_t assumed_free_hbm(size_t free_b) {
    const char *e = std::getenv("RBG_ASSUME_FREE_HBM_MB");
    if (!e || std::atoll(e) <= 0) return free_b;
    return std::min<size_t>(free_b, static_cast<size_t>(std::atoll(e)) << 20);
}
inline double est_depth_runs(double r, uint32_t d) { return r * (1.0 + 0.62 * static_cast<double>(d - 1)); }
inline uint32_t planned_depth(double r, bool samples, uint32_t K0, double free_b, double budget, bool with_budget) {
    const double per_entry = 8.0 + (samples ? 6.0 : 0.0) + 4.0;
    uint32_t K = K0;
    while (K > 1) {
        bool ok = est_depth_runs(r, K) < 0.9 * 4294967296.0 && 70.0 * est_depth_runs(r, K) <= 0.95 * free_b;
        if (ok && with_budget) ok = (samples ? 16.0 * r : 0.0) + (est_depth_runs(r, 1) + est_depth_runs(r, K)) * per_entry <= budget;
        if (ok) break;
        --K;
    }
    return K;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "rbg: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV;                           \
        }                                                                                         \
    } while (0)

// RBG_VERBOSE: seconds a stage of a load took (device work is synchronised first when `sync`)
struct VStage {
    const char *what;
    bool on, sync;
    std::chrono::steady_clock::time_point t0;
    explicit VStage(const char *w, bool sync_ = true) : what(w), on(std::getenv("RBG_VERBOSE") != nullptr), sync(sync_), t0(std::chrono::steady_clock::now()) {}
    ~VStage() {
        if (!on) return;
        if (sync) (void)hipDeviceSynchronize();
        std::fprintf(stderr, "rbg:   %s %.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

// RAII: make `device` current for the scope of one API call
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    int rc = RBG_OK;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { rc = RBG_ENODEV; return; }
        if (prev != device) {
            if (hipSetDevice(device) != hipSuccess) { rc = RBG_ENODEV; return; }
            changed = true;
        }
    }
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
};

// device scratch freed at scope exit
// Scratch device memory of the host-pointer calls.  hipMalloc / hipFree per call cost more than a one-read query itself
// (and hipFree synchronises the whole device, which serialises concurrent callers), so freed blocks are kept per device
// and size class and handed out again: in steady state a call allocates nothing.  Blocks beyond 512 MiB and whatever
// would take the cache past 2 GiB go back to the driver at once; rbg_free() of an index trims its device's cache.
class DevPool {
   public:
    static DevPool &get() { static DevPool p; return p; }
    static size_t size_class(size_t bytes) {
        if (bytes < 4096) return 4096;
        if (bytes <= (size_t(64) << 20)) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
        // large blocks: eighths of the power of two below (steps of at most 12.5 %): the ragged results of successive batches
        // (locations, text) differ by a few per cent and must find each other's blocks (with 2 MB classes every batch missed)
        size_t p2 = size_t(64) << 20;
        while ((p2 << 1) <= bytes) p2 <<= 1;
        const size_t step = p2 >> 3;
        return (bytes + step - 1) / step * step;
    }
    int alloc(size_t bytes, void **out, size_t *cls_out) {
        const size_t cls = size_class(bytes);
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(mu_);
            // the smallest cached block of this class or one up to a quarter larger
            for (auto it = free_.lower_bound({dev, cls}); it != free_.end() && it->first.first == dev && it->first.second <= cls + cls / 4; ++it)
                if (!it->second.empty()) {
                    *out = it->second.back();
                    it->second.pop_back();
                    cached_ -= it->first.second;
                    *cls_out = it->first.second;
                    return RBG_OK;
                }
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, cls);
        if (e == hipErrorOutOfMemory) {   // give the cache back and try once more
            (void)hipGetLastError();
            trim(dev);
            e = hipMalloc(&p, cls);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV; }
        *out = p;
        *cls_out = cls;
        return RBG_OK;
    }
    void release(void *p, size_t cls, int dev) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (cls <= kMaxBlock && cached_ + cls <= kMaxCached) {
                free_[{dev, cls}].push_back(p);
                cached_ += cls;
                return;
            }
        }
        (void)hipFree(p);
    }
    void trim(int dev) {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &kv : free_)
                if (kv.first.first == dev) {
                    cached_ -= kv.first.second * kv.second.size();
                    drop.insert(drop.end(), kv.second.begin(), kv.second.end());
                    kv.second.clear();
                }
        }
        for (void *p : drop) (void)hipFree(p);
    }

   private:
    static constexpr size_t kMaxBlock = size_t(1) << 30, kMaxCached = size_t(4) << 30;
    std::mutex mu_;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_;
    size_t cached_ = 0;
};

struct DevBuf {
    void *p = nullptr;
    size_t cls = 0;
    int dev = 0;
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        (void)hipGetDevice(&dev);
        return DevPool::get().alloc(bytes, &p, &cls);
    }
    ~DevBuf() {
        if (!p) return;
        // every user works on hipStreamPerThread and has synchronised by the time its buffers go out of scope, except on
        // an error path: make sure nothing still runs on the block before another caller may get it
        (void)hipStreamSynchronize(hipStreamPerThread);
        DevPool::get().release(p, cls, dev);
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T> T *as() { return static_cast<T *>(p); }
};

// Gigabytes of host scratch that worker threads fill (run lists converted to the device's width, phi entries): NOT
// value-initialised -- a std::vector's zero fill is one thread touching every page first (0.5 s per 2.5 GB at r = 3e8,
// three to five such arrays per load); here the first touch is the parallel fill itself, on huge pages where it can be.
template <typename T>
struct HostBuf {
    T *p = nullptr;
    size_t n = 0;
    HostBuf() = default;
    explicit HostBuf(size_t count) { resize(count); }
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    HostBuf(HostBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    HostBuf &operator=(HostBuf &&o) noexcept { if (this != &o) { std::free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~HostBuf() { std::free(p); }
    void resize(size_t count) {   // (contents are not kept)
        std::free(p);
        p = nullptr; n = 0;
        if (!count) return;
        constexpr size_t kHuge = size_t(2) << 20;
        const size_t bytes = count * sizeof(T);
        if (bytes >= 4 * kHuge) {
            p = static_cast<T *>(std::aligned_alloc(kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)));
            if (p) (void)madvise(p, bytes, MADV_HUGEPAGE);
        } else {
            p = static_cast<T *>(std::malloc(bytes));
        }
        if (!p) throw std::bad_alloc();
        n = count;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// host-to-device copy of a (possibly huge) pageable array: through pinned staging when it is big (defined beside d2h_result)
int h2d_big(void *d_dst, const void *h_src, size_t bytes);

// The replica lives in ONE device allocation (the arena) that the tables are carved out of: a
// thousand separate hipMallocs leave the tables scattered over physical memory, and the gather
// rate of these kernels is sensitive to that.  Anything that does not fit the
// pre-computed arena (markers attached later) gets its own allocation.
constexpr size_t kArenaAlign = 64 * 1024;
inline size_t arena_round(size_t bytes) { return ((bytes ? bytes : 1) + kArenaAlign - 1) & ~(kArenaAlign - 1); }

// space for `bytes` in the arena (or its own allocation when the arena is full / absent)
int dev_reserve(rbg_index *ix, size_t bytes, void **dst) {
    void *p = nullptr;
    const size_t alloc = arena_round(bytes);
    if (ix->arena && ix->arena_used + alloc <= ix->arena_bytes) {
        p = static_cast<char *>(ix->arena) + ix->arena_used;
        ix->arena_used += alloc;
    } else {
        HIP_TRY(hipMalloc(&p, alloc));
        ix->allocs.push_back({p, alloc});
        ix->hbm_bytes += alloc;
    }
    *dst = p;
    return RBG_OK;
}

int dev_upload(rbg_index *ix, const void *src, size_t bytes, const void **dst) {
    void *p = nullptr;
    int rc = dev_reserve(ix, bytes, &p);
    if (rc) return rc;
    if (bytes && (rc = h2d_big(p, src, bytes))) return rc;
    *dst = p;
    return RBG_OK;
}

// bytes of one table in the replica; in_arena: what the arena has to hold of it (a table composed on the device keeps
// its run list and samples in the level's own allocation)
template <typename P>
size_t table_bytes(const SymTable &t, bool with_samples, uint64_t n, bool in_arena = false) {
    const uint64_t nb = (n >> t.shift) + 2;
    const size_t lists = (in_arena && t.dev_ent) ? 0 : arena_round((t.nruns + 1) * sizeof(RunEnt<P>)) + (with_samples ? arena_round(t.nruns * sizeof(P)) : 0);
    return lists + arena_round(nb * sizeof(RankSlot)) + arena_round(nb * sizeof(uint32_t));
}

// 8-byte positions with n < 2^38 and phi buckets of at most 64 positions: 16-byte packed phi slots (rbg_dev.h)
template <typename P>
bool phi_slots_packed(const HostIndex &h) {
    const char *e = std::getenv("RBG_PHI_PACKED");   // "0": keep the 32-byte slots (A/B measurements, tests)
    if (e && e[0] == '0') return false;
    return sizeof(P) == 8 && (h.n >> kPhiPackedPosBits) == 0 && h.phi_shift <= kPhiPackedMaxShift;
}
template <typename P>
size_t phi_slot_bytes(const HostIndex &h) { return phi_slots_packed<P>(h) ? sizeof(PhiSlotPacked) : sizeof(PhiSlot<P>); }

template <typename P>
size_t replica_bytes(const HostIndex &h, bool in_arena = false) {
    size_t total = 0;
    for (const SymTable &t : h.sym) total += table_bytes<P>(t, h.has_tsa, h.n, in_arena);
    total += arena_round(h.sym.size() * sizeof(DevSym)) + 3 * arena_round(256);
    for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxSlotKmerDepth); ++d) {
        for (const SymTable &t : h.kmer(d)) total += table_bytes<P>(t, h.has_tsa, h.n, in_arena);
        total += arena_round(h.kmer(d).size() * sizeof(DevSym));
    }
    if (h.has_tsa) {
        const uint64_t nb = (h.n >> h.phi_shift) + 2;
        total += arena_round(h.r * sizeof(PhiEnt<P>)) + arena_round(nb * phi_slot_bytes<P>(h)) + arena_round(nb * sizeof(uint32_t));
    }
    if (h.has_ma)
        total += arena_round(h.ma.start.size() * 8) + arena_round(h.ma.end.size() * 8) + arena_round(h.ma.off.size() * 8) +
                 arena_round(h.ma.vals.size() * 8);
    return total + 16 * kArenaAlign;
}

// one symbol (or k-mer) table -> its device record, in two halves so that the host-side array
// building of many tables can run on worker threads while the uploads stay on the calling thread
template <typename P>
struct PreparedSym {
    HostBuf<RunEnt<P>> ent;
    HostBuf<P> samp;
};

template <typename P>
void prepare_sym(const SymTable &t, bool with_samples, PreparedSym<P> &p) {
    if (t.dev_ent) return;   // composed on the device: the run list is there already
    p.ent.resize(t.nruns + 1);
    if (with_samples) p.samp.resize(t.nruns);
    parallel_for(t.nruns + 1, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t k = b; k < e; ++k) {
            p.ent[k].start = static_cast<P>(t.start[k]);
            p.ent[k].cum = static_cast<P>(t.cum[k]);
            if (with_samples && k < t.nruns) p.samp[k] = static_cast<P>(t.samp[k]);
        }
    }, uint64_t(1) << 18);
}

// upload the run list (+ samples); the RankSlot / ord tables are generated from it on the device
template <typename P>
int commit_sym(rbg_index *ix, const SymTable &t, bool with_samples, PreparedSym<P> &p, DevSym &d, unsigned long long *d_overflow) {
    int rc = RBG_OK;
    d.samp = nullptr;
    if (t.dev_ent) {
        d.ent = t.dev_ent;
        if (with_samples) d.samp = t.dev_samp;
    } else {
        if ((rc = dev_upload(ix, p.ent.data(), p.ent.size() * sizeof(RunEnt<P>), &d.ent))) return rc;
        if (with_samples && (rc = dev_upload(ix, p.samp.data(), p.samp.size() * sizeof(P), &d.samp))) return rc;
    }
    const uint64_t nb = (ix->H().n >> t.shift) + 2;
    void *slots = nullptr, *ord = nullptr;
    if ((rc = dev_reserve(ix, nb * sizeof(RankSlot), &slots)) || (rc = dev_reserve(ix, nb * sizeof(uint32_t), &ord))) return rc;
    const bool dense = g_opt_dense_overflow.load() != 0;
    if (launch_build_rank_slots(sizeof(P), d.ent, t.nruns, ix->H().n, t.shift, slots, static_cast<uint32_t *>(ord), d_overflow,
                                dense ? d_overflow + 2 : nullptr, nullptr))
        return RBG_ENODEV;
    d.slots = slots;
    d.ord = static_cast<const uint32_t *>(ord);
    ix->rank_slots += nb;
    d.F = t.F;
    d.shift = t.shift;
    d.nruns = static_cast<uint32_t>(t.nruns);
    if (dense) ix->dense_todo.push_back(d);
    return RBG_OK;
}

template <typename P>
int upload_many(rbg_index *ix, const std::vector<SymTable> &tabs, bool with_samples, std::vector<DevSym> &recs,
                unsigned long long *d_overflow) {
    recs.resize(tabs.size());
    const size_t T = std::max<size_t>(1, std::min<size_t>(16, std::thread::hardware_concurrency()));
    for (size_t b = 0; b < tabs.size(); b += T) {
        const size_t e = std::min(tabs.size(), b + T);
        std::vector<PreparedSym<P>> prep(e - b);
        std::vector<std::thread> workers;
        for (size_t i = b + 1; i < e; ++i)
            workers.emplace_back([&, i] { prepare_sym<P>(tabs[i], with_samples, prep[i - b]); });
        prepare_sym<P>(tabs[b], with_samples, prep[0]);
        for (auto &w : workers) w.join();
        for (size_t i = b; i < e; ++i) {
            int rc = commit_sym<P>(ix, tabs[i], with_samples, prep[i - b], recs[i], d_overflow);
            if (rc) return rc;
            prep[i - b] = PreparedSym<P>();  // release before the next batch
        }
    }
    return RBG_OK;
}

template <typename P>
int upload_tables(rbg_index *ix) {
    HostIndex &h = ix->H();
    DevBuf d_ovf;  // [0] rank slots, [1] phi slots that overflow their inline entries; [2] dense-table space handed out (16-byte units)
    int rc = d_ovf.alloc(32);
    if (rc) return rc;
    HIP_TRY(hipMemset(d_ovf.p, 0, 32));
    ix->dense_todo.clear();
    ix->dev.dense = nullptr;
    unsigned long long *ovf = d_ovf.as<unsigned long long>();
    std::vector<DevSym> syms;
    {
        VStage vs("depth-1 tables: run lists up, slot tables built");
        if ((rc = upload_many<P>(ix, h.sym, h.has_tsa, syms, ovf))) return rc;
    }
    const void *p = nullptr;
    rc = dev_upload(ix, syms.data(), syms.size() * sizeof(DevSym), &p);
    if (rc) return rc;
    ix->dev.syms = static_cast<const DevSym *>(p);
    ix->ptr_tables.push_back({p, syms.size(), sizeof(DevSym), {offsetof(DevSym, ent), offsetof(DevSym, samp), offsetof(DevSym, slots), offsetof(DevSym, ord)}});
    ix->dev.nmajor = 0;
    ix->dev.kmer_steps = 1;
    if (!h.kmer(2).empty()) {
        auto upload_set = [&](const std::vector<SymTable> &tabs, const DevSym **dst) -> int {
            VStage vs("one k-mer level: slot tables built");
            std::vector<DevSym> recs;
            int r2 = upload_many<P>(ix, tabs, h.has_tsa, recs, ovf);
            if (r2) return r2;
            const void *pp = nullptr;
            r2 = dev_upload(ix, recs.data(), recs.size() * sizeof(DevSym), &pp);
            if (r2) return r2;
            ix->ptr_tables.push_back({pp, recs.size(), sizeof(DevSym), {offsetof(DevSym, ent), offsetof(DevSym, samp), offsetof(DevSym, slots), offsetof(DevSym, ord)}});
            *dst = static_cast<const DevSym *>(pp);
            return RBG_OK;
        };
        const DevSym **slot_tabs[kMaxSlotKmerDepth - 1] = {&ix->dev.pairs, &ix->dev.triples, &ix->dev.quads, &ix->dev.quints};
        for (uint32_t d = 2; d <= static_cast<uint32_t>(kMaxSlotKmerDepth) && !h.kmer(d).empty(); ++d) {
            if ((rc = upload_set(h.kmer(d), slot_tabs[d - 2]))) return rc;
            ix->dev.kmer_steps = d;
        }
        rc = dev_upload(ix, h.major_of, 256, &p);
        if (rc) return rc;
        ix->dev.lut2 = static_cast<const uint8_t *>(p);
        ix->dev.nmajor = h.nmajor;
    }
    if (h.has_tsa) {
        VStage vs("phi: entries up, slots built");
        {
            HostBuf<PhiEnt<P>> pe(h.r);
            parallel_for(h.r, [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t j = b; j < e; ++j) {
                    pe[j].pos = static_cast<P>(h.pred_pos[j]);
                    pe[j].base = static_cast<P>(h.phi_base[j]);
                }
            });
            rc = dev_upload(ix, pe.data(), pe.size() * sizeof(PhiEnt<P>), &ix->dev.phi_ent);
            if (rc) return rc;
        }
        const uint64_t nb = (h.n >> h.phi_shift) + 2;
        void *slots = nullptr, *ord = nullptr;
        const bool packed = phi_slots_packed<P>(h);
        if ((rc = dev_reserve(ix, nb * phi_slot_bytes<P>(h), &slots)) || (rc = dev_reserve(ix, nb * sizeof(uint32_t), &ord))) return rc;
        ix->dev.phi_packed = packed ? 1 : 0;
        if (launch_build_phi_slots(sizeof(P), packed, ix->dev.phi_ent, h.r, h.n, h.phi_shift, slots, static_cast<uint32_t *>(ord), ovf + 1, nullptr))
            return RBG_ENODEV;
        ix->phi_slots = nb;
        ix->dev.phi_slots = slots;
        ix->dev.phi_ord = static_cast<const uint32_t *>(ord);
    }
    unsigned long long counts[3] = {0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());  // every table is generated before the first query (and before d_ovf goes away)
    HIP_TRY(hipMemcpy(counts, d_ovf.p, 24, hipMemcpyDeviceToHost));
    ix->rank_slots_overflow = counts[0];
    ix->phi_slots_overflow = counts[1];
    // second pass over the rank tables: now that the number of overflow buckets is known, give each its
    // dense table (rbg_dev.h).  The slots hold 32-bit offsets in 16-byte units: a pool beyond 64 GB (never seen:
    // 2.7 GB for the bench index) leaves the run-list search in place, as does an allocation failure.
    if (counts[2] > 0 && counts[2] < (1ull << 32)) {
        VStage vs("dense tables of the overflow buckets");
        void *pool = nullptr;
        const size_t bytes = static_cast<size_t>(counts[2]) * 16 + 64;
        if (hipMalloc(&pool, bytes) == hipSuccess) {
            ix->allocs.push_back({pool, bytes});
            ix->hbm_bytes += bytes;
            for (const DevSym &d : ix->dense_todo)
                if (launch_fill_dense(sizeof(P), d.ent, h.n, d.shift, d.slots, d.ord, static_cast<uint8_t *>(pool), nullptr)) return RBG_ENODEV;
            HIP_TRY(hipDeviceSynchronize());
            ix->dev.dense = static_cast<const uint8_t *>(pool);
        } else {
            (void)hipGetLastError();
        }
    }
    ix->dense_todo.clear();
    ix->dense_todo.shrink_to_fit();
    return RBG_OK;
}

// give back one allocation the index tracks
void free_tracked(rbg_index *ix, void *p) {
    if (!p) return;
    for (size_t i = 0; i < ix->allocs.size(); ++i)
        if (ix->allocs[i].p == p) { ix->hbm_bytes -= ix->allocs[i].bytes; ix->allocs.erase(ix->allocs.begin() + static_cast<std::ptrdiff_t>(i)); break; }
    (void)hipFree(p);
}

// bytes of the run-indexed replica with the k-mer depths of `mask` (bit d - 1) among those h holds (run lists, samples,
// 1/15 of sampled keys, phi)
template <typename P>
size_t runs_replica_bytes(const HostIndex &h, uint32_t mask = ~0u) {
    size_t total = 0;
    // 8-byte entries at either width, directory entries of 4 / 8 bytes per (at most) half a run
    const size_t ent_bytes = 8, dir_per_entry = sizeof(P) == 8 ? 4 : 2;
    for (uint32_t di = 0; di < static_cast<uint32_t>(kMaxKmerDepth); ++di) {
        const std::vector<SymTable> *lv = di == 0 ? &h.sym : &h.kmer(di + 1);
        if (!((mask | 1u) >> di & 1u)) continue;
        size_t entries = 0;
        for (const SymTable &t : *lv) entries += t.nruns + 1;
        total += entries * (ent_bytes + (h.has_tsa ? RunsFmt<P>::samp_bytes : 0)) + entries * dir_per_entry + lv->size() * 8 + 8 * kArenaAlign;
    }
    if (h.has_tsa) total += (h.r + 1) * PhiFmt<P>::ent_bytes + h.r * 4;   // (+ the phi directory: at most r entries)
    return total + 16 * kArenaAlign;
}

// bucket records of one k-mer depth at `per` entries per bucket on average: their number (a sparse table's bucket shift stops at max_shift)
inline double runs_record_count(const HostIndex &h, uint32_t depth_index, double per, uint32_t max_shift) {
    const std::vector<SymTable> &lv = depth_index == 0 ? h.sym : h.kmer(depth_index + 1);
    double nrec = 0;
    for (const SymTable &t : lv) {
        uint32_t sh = 0;
        const double runs = static_cast<double>(std::max<uint64_t>(1, t.nruns));
        while (sh < max_shift && runs * static_cast<double>(uint64_t(2) << sh) <= per * static_cast<double>(h.n)) ++sh;
        nrec += static_cast<double>((h.n >> sh) + 2);
    }
    return nrec;
}
// phi slots of about n / r rows on the run-indexed layout: their bytes, or 0 where the automatic rule would not build them (more than 2 r buckets)
inline double runs_phi_slot_bytes(const HostIndex &h) {
    if (!h.has_tsa) return 0;
    uint32_t ss = 0;
    while (ss < 8 && static_cast<double>(uint64_t(2) << ss) <= static_cast<double>(h.n) / static_cast<double>(std::max<uint64_t>(1, h.r))) ++ss;
    if (ss < h.phi_shift) ss = h.phi_shift;
    const double nb = static_cast<double>((h.n >> ss) + 2);
    return nb <= 2.0 * static_cast<double>(h.r) ? nb * (h.pos_bytes == 8 ? 36.0 : 20.0) : 0.0;
}

void release_kmer_level(rbg_index *ix, uint32_t depth);
std::vector<SymTable> &kmer_level_tables(HostIndex &h, uint32_t depth);

}  // namespace
