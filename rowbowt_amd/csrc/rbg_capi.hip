// rbg_capi.hip -- the C-ABI of include/rbg.h: owns the host copy of the flat index and its
// HBM replica, stages host batches, launches the kernels of k_search.hip / k_locate.hip / k_markers.hip / k_build.hip.
// There is deliberately no CPU compute path in this library.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>

#include "../../include/rbg.h"
#include "rbg_dev.h"
#include "rbg_host.hpp"

using namespace rbg;
static_assert(kMaxRunDepth == kMaxKmerDepth && kLdsRunDepth == kMaxSlotKmerDepth, "rbg_dev.h and rbg_host.hpp name the same depths");

#include "rbg_hostpath.hpp"

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

void release_kmer_level(rbg_index *ix, uint32_t depth);
std::vector<SymTable> &kmer_level_tables(HostIndex &h, uint32_t depth);

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
    auto records_of = [&](uint32_t d, double per) {   // records of depth d at `per` entries per bucket (a sparse table's shift stops at max_shift)
        double nrec = 0;
        for (const SymTable &t : *depth[d]) {
            uint32_t sh = 0;
            const double runs = static_cast<double>(std::max<uint64_t>(1, t.nruns));
            while (sh < max_shift && runs * static_cast<double>(uint64_t(2) << sh) <= per * static_cast<double>(h.n)) ++sh;
            nrec += static_cast<double>((h.n >> sh) + 2);
        }
        return nrec;
    };
    if (g_opt_run_rec.load() == 2) {
        const uint32_t want = g_opt_run_rec_depths.load() ? static_cast<uint32_t>(g_opt_run_rec_depths.load()) : ~0u;
        for (uint32_t d = 0; d < D; ++d)
            if ((mask >> d & 1u) && (want >> d & 1u)) rec_per[d] = rec_asked > 0 ? rec_asked : 2.5;
    } else if (g_opt_run_rec.load() == 0 && ix->hbm_budget) {
        double total = static_cast<double>(W ? runs_replica_bytes<uint64_t>(h, mask) : runs_replica_bytes<uint32_t>(h, mask));
        if (h.has_tsa && g_opt_run_phi.load() != 1) {   // phi slots come first (decided after the rank tables, below: the same arithmetic): their room is not the records'
            uint32_t ss = 0;
            while (ss < 8 && static_cast<double>(uint64_t(2) << ss) <= static_cast<double>(h.n) / static_cast<double>(std::max<uint64_t>(1, h.r))) ++ss;
            if (ss < h.phi_shift) ss = h.phi_shift;
            const double nb = static_cast<double>((h.n >> ss) + 2);
            if (nb <= 2.0 * static_cast<double>(h.r)) total += nb * (W ? 36.0 : 20.0);
        }
        for (int d = static_cast<int>(D) - 1; d >= 0; --d) {
            if (!(mask >> d & 1u)) continue;
            double entries_d = 0;
            for (const SymTable &t : *depth[d]) entries_d += static_cast<double>(t.nruns + 1);
            const double pers[3] = {2.5, 4.0, 6.0};
            for (const double per : pers) {
                if (rec_asked > 0 && per != pers[0]) break;
                const double nrec = records_of(static_cast<uint32_t>(d), rec_asked > 0 ? rec_asked : per);
                if (nrec <= entries_d && total + nrec * 64.0 <= static_cast<double>(ix->hbm_budget)) {
                    rec_per[d] = rec_asked > 0 ? rec_asked : per;
                    total += nrec * 64.0;
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
        if (use_recs) {
            // the records' buckets: the widest with at most rec_target entries starting inside on average
            for (size_t t = 0; t < T.size(); ++t) {
                uint32_t sh = 0;
                const double runs = static_cast<double>(std::max<uint64_t>(1, nr[t]));
                while (sh < max_shift && runs * static_cast<double>(uint64_t(2) << sh) <= rec_target * static_cast<double>(h.n)) ++sh;
                dshift[t] = sh;
                doff[t + 1] = doff[t] + (h.n >> sh) + 2;
            }
            void *recp = nullptr;
            if ((rc = dev_reserve(ix, doff[T.size()] * sizeof(RunRec2) + 64, &recp))) return rc;
            TmpDev tmp, ovf;
            const size_t nt = T.size(), bytes = (3 * nt + 1) * 8 + nt * 4;
            if ((rc = tmp.alloc(bytes)) || (rc = ovf.alloc(8))) return rc;
            HIP_TRY(hipMemset(ovf.p, 0, 8));
            uint64_t *t_first = tmp.as<uint64_t>(), *t_nr = t_first + nt, *t_doff = t_nr + nt;
            uint32_t *t_sh = reinterpret_cast<uint32_t *>(t_doff + nt + 1);
            HIP_TRY(hipMemcpy(t_first, first.data(), nt * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_nr, nr.data(), nt * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_doff, doff.data(), (nt + 1) * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t_sh, dshift.data(), nt * 4, hipMemcpyHostToDevice));
            HIP_TRY(static_cast<hipError_t>(launch_run_recs2(sizeof(P), abs_ent, t_first, t_nr, t_doff, t_sh, static_cast<uint32_t>(nt), doff[nt], recp, ovf.as<unsigned long long>(), nullptr)));
            unsigned long long novf = 0;
            HIP_TRY(hipMemcpy(&novf, ovf.p, 8, hipMemcpyDeviceToHost));
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
    if (std::getenv("RBG_VERBOSE") || static_cast<uint64_t>(levels()) != ix->kmer_steps_requested)
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
    if (device == RBG_DEVICE_NONE || o.kmer_steps < 2) return o;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return o;
    DeviceScope scope(device);
    if (scope.rc) return o;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return o;
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
                if (opt_mb == 0 && g_opt_rank_layout.load() == RBG_LAYOUT_AUTO &&
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

int finish(rbg_index *ix, int device, rbg_index **out) {
    ix->device = device;
    if (device != RBG_DEVICE_NONE) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = upload(ix);
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
    return RBG_OK;
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

// ---- host-buffer entry points ----------------------------------------------------------------------

// Reads of a host batch: either the C-ABI's packed layout (read i = seqs[off[i], off[i+1])) or spans of a larger
// buffer (read i = base[begin[i], begin[i] + len[i]): what a parser that leaves the bytes in its input buffer has).
struct HostReads {
    const uint8_t *base = nullptr;
    const uint64_t *off = nullptr;     // packed layout (N + 1), or
    const uint64_t *begin = nullptr;   // spans
    const uint32_t *len = nullptr;
    const uint8_t *ptr(uint64_t i) const { return base + (off ? off[i] : begin[i]); }
    uint64_t length(uint64_t i) const { return off ? off[i + 1] - off[i] : len[i]; }
};

// a workspace of the index for the duration of one call
struct WsLease {
    rbg_index *ix;
    std::unique_ptr<rbg_hostpath::Workspace> ws;
    explicit WsLease(rbg_index *ix_) : ix(ix_) {
        std::lock_guard<std::mutex> g(ix->ws_mu);
        if (!ix->ws_free.empty()) { ws = std::move(ix->ws_free.back()); ix->ws_free.pop_back(); }
        if (!ws) { ws.reset(new rbg_hostpath::Workspace()); ws->device = ix->device; }
    }
    ~WsLease() {
        std::lock_guard<std::mutex> g(ix->ws_mu);
        ix->ws_free.push_back(std::move(ws));
    }
};

constexpr uint64_t kHostChunkReads = uint64_t(1) << 20;    // reads per in-flight chunk ...
constexpr uint64_t kHostChunkBytes = uint64_t(384) << 20;  // ... and symbols per chunk (long reads)

static int find_range_host_core(rbg_index *ix, const HostReads &R, uint64_t N, uint64_t *lo, uint64_t *hi, uint64_t *ssamp,
                                uint64_t *count, bool allow_pack);

// ---- micro-batching of one-read calls ---------------------------------------------------------------------------------
// A caller written against the reference asks one read at a time (RowBowt::find_range(query), rowbowt.hpp:121-131);
// when several of its threads do so concurrently (the reference's only parallel dispatcher is rb_markers' thread pool,
// rb_markers.cpp:318-535) their calls are combined: whoever arrives while no launch is being prepared becomes the
// leader, takes everything that has queued up, runs ONE batched call for it and hands the answers back; whoever arrives
// in the meantime queues for the next round.  No timer and no added latency: a lone caller's request is a batch of
// one, and the batch size follows the concurrency by itself.  RBG_HOST_COMBINE=0 switches it off (A/B).
struct CombineReq {
    bool done = false;
    int rc = RBG_OK;
};
// exec(batch) answers every request of the batch (sets rc); match(a, b): may b ride in a's batch?
extern "C++" {
template <typename Req, typename Match, typename Exec>
int combine_submit(rbg_index *ix, rbg_index::Combiner &C, Req &mine, Match match, Exec exec) {
    std::unique_lock<std::mutex> lk(C.mu);
    C.pending.push_back(&mine);
    while (!mine.done) {
        if (C.leader) { C.cv.wait(lk); continue; }
        C.leader = true;
        std::vector<Req *> batch;
        std::vector<void *> rest;
        for (void *p : C.pending) {
            Req *r = static_cast<Req *>(p);
            if (r == &mine || match(mine, *r)) batch.push_back(r); else rest.push_back(p);
        }
        C.pending.swap(rest);
        lk.unlock();
        int rc_all = RBG_OK;
        try {
            exec(batch);
        } catch (const std::bad_alloc &) {
            rc_all = RBG_ENOMEM;
        } catch (...) {
            rc_all = RBG_EFORMAT;
        }
        ix->comb_launches.fetch_add(1, std::memory_order_relaxed);
        ix->comb_requests.fetch_add(batch.size(), std::memory_order_relaxed);
        lk.lock();
        for (Req *r : batch) { if (rc_all) r->rc = rc_all; r->done = true; }
        C.leader = false;
        C.cv.notify_all();
    }
    return mine.rc;
}
}  // extern "C++"
inline bool combine_enabled() {
    static const bool on = [] { const char *e = std::getenv("RBG_HOST_COMBINE"); return !(e && e[0] == '0'); }();
    return on;
}

struct RangeReq : CombineReq {
    const uint8_t *seq = nullptr;
    uint64_t len = 0;
    bool want_ss = false;
    uint64_t lo = 1, hi = 0, ss = 0;
};

// one read through the combiner: find_range / count / find_range_w_toehold with N = 1
static int find_range_one(rbg_index *ix, const uint8_t *seq, uint64_t len, uint64_t *lo, uint64_t *hi, uint64_t *ssamp, uint64_t *count) {
    RangeReq mine;
    mine.seq = seq;
    mine.len = len;
    mine.want_ss = ssamp != nullptr;
    const int rc = combine_submit(ix, ix->comb_range, mine, [](const RangeReq &a, const RangeReq &b) { return a.len <= 0xFFFFFFFFull && b.len <= 0xFFFFFFFFull; },
        [&](std::vector<RangeReq *> &batch) {
            const uint64_t K = batch.size();
            bool any_ss = false;
            const uint8_t *base = nullptr;
            for (RangeReq *r : batch) {
                any_ss = any_ss || r->want_ss;
                if (r->len && (!base || r->seq < base)) base = r->seq;
            }
            std::vector<uint64_t> begin(K), blo(K), bhi(K), bss(any_ss ? K : 0);
            std::vector<uint32_t> blen(K);
            HostReads R;
            int rc2;
            if (K == 1 && batch[0]->len > 0xFFFFFFFFull) {   // (a read beyond 4 GB: the packed layout takes any length)
                const uint64_t off[2] = {0, batch[0]->len};
                R.base = batch[0]->seq;
                R.off = off;
                rc2 = find_range_host_core(ix, R, 1, blo.data(), bhi.data(), any_ss ? bss.data() : nullptr, nullptr, true);
            } else {
                static const uint8_t kNone = 0;
                if (!base) base = &kNone;
                for (uint64_t i = 0; i < K; ++i) {
                    begin[i] = batch[i]->len ? static_cast<uint64_t>(batch[i]->seq - base) : 0;
                    blen[i] = static_cast<uint32_t>(batch[i]->len);
                }
                R.base = base;
                R.begin = begin.data();
                R.len = blen.data();
                rc2 = find_range_host_core(ix, R, K, blo.data(), bhi.data(), any_ss ? bss.data() : nullptr, nullptr, true);
            }
            for (uint64_t i = 0; i < K; ++i) {
                batch[i]->rc = rc2;
                batch[i]->lo = blo[i];
                batch[i]->hi = bhi[i];
                if (batch[i]->want_ss) batch[i]->ss = bss[i];
            }
        });
    if (rc) return rc;
    if (lo) { *lo = mine.lo; *hi = mine.hi; }
    if (ssamp) *ssamp = mine.ss;
    if (count) *count = mine.hi >= mine.lo ? mine.hi - mine.lo + 1 : 0;
    return RBG_OK;
}

static int find_range_host(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo, uint64_t *hi,
                           uint64_t *ssamp, uint64_t *count) {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!off || (!seqs && off[N])) return RBG_EARG;
    if (N == 1 && off[0] == 0 && combine_enabled()) return find_range_one(ix, seqs, off[1], lo, hi, ssamp, count);
    if (off[0] != 0) return RBG_EARG;   // (the rest of check_offsets() is done by the staging passes, chunk by chunk, before any byte is read)
    HostReads R;
    R.base = seqs;
    R.off = off;
    return find_range_host_core(ix, R, N, lo, hi, ssamp, count, true);
}

// The pipeline of rbg_hostpath.hpp.  Outputs: lo/hi (both or neither), ssamp (toehold search), count.
static int find_range_host_core(rbg_index *ix, const HostReads &R, uint64_t N, uint64_t *lo, uint64_t *hi, uint64_t *ssamp,
                                uint64_t *count, bool allow_pack) {
    using rbg_hostpath::Slot;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    WsLease lease(ix);
    rbg_hostpath::Workspace &W = *lease.ws;
    const int64_t pk = g_opt_packed_reads.load();
    const HostIndex &h = ix->H();
    // 2-bit transfer: needs the packed search kernel's alphabet (four k-mer symbols) and the slot-table layout
    const bool pack = allow_pack && ix->dev.nmajor == 4 && (pk == 2 || (pk == 1 && N >= 4096));
    const bool acgt = h.major_byte[0] == 'A' && h.major_byte[1] == 'C' && h.major_byte[2] == 'G' && h.major_byte[3] == 'T';
    if (!W.team) {
        // a quarter of the hardware's CPUs, at most 64 and at most what the container's CPU quota lets run at once
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned nt = std::min({64u, std::max(1u, hw / 4), rbg_hostpath::cpu_budget()});
        if (const char *e = std::getenv("RBG_HOST_THREADS")) nt = static_cast<unsigned>(std::max(1, std::min(256, std::atoi(e))));
        W.team.reset(new rbg_hostpath::ThreadTeam(nt));
        W.bad.resize(W.team->size());
    }
    rbg_hostpath::ThreadTeam &team = *W.team;
    const unsigned T = team.size();
    // small batches (the shim's one-read calls among them) stay on the calling thread: waking the team costs more
    auto par = [&](uint64_t work_items, const std::function<void(unsigned)> &fn) {
        if (work_items < 16384) { for (unsigned t = 0; t < T; ++t) fn(t); }
        else team.run(fn);
    };
    for (auto &v : W.bad) v.clear();
    const int nout = (lo ? 2 : 0) + (ssamp ? 1 : 0) + (count ? 1 : 0);
    const bool need_lohi_dev = true;  // the kernels always write lo/hi
    (void)need_lohi_dev;

    int rc = RBG_OK;
    const bool trace = std::getenv("RBG_HOST_TRACE") != nullptr;   // per-call breakdown on stderr
    double t_pack = 0, t_wait = 0, t_out = 0, t_enq = 0;
    // RBG_HOST_TRACE=2: also the device-side timeline of every chunk (timing events around copy in / search / copy out)
    const bool timeline = trace && std::atoi(std::getenv("RBG_HOST_TRACE")) >= 2;
    const char *e_direct = std::getenv("RBG_HOST_DIRECT_OUT");
    const bool direct_out = !(e_direct && e_direct[0] == '0');
    struct ChunkEvents { hipEvent_t e[4]; double host_ms; };
    std::vector<ChunkEvents> tl;
    auto mark = [&](int which, hipStream_t st) {
        if (!timeline) return;
        if (which == 0) { tl.emplace_back(); for (hipEvent_t &e : tl.back().e) (void)hipEventCreate(&e); }
        (void)hipEventRecord(tl.back().e[which], st);
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_call = now();
    auto drain = [&](Slot &s) -> int {  // wait for the chunk in flight in `s`, hand its results to the caller
        if (!s.busy) return RBG_OK;
        const auto tw0 = now();
        if (hipEventSynchronize(s.done) != hipSuccess) return RBG_ENODEV;
        const auto tw1 = now();
        t_wait += secs(tw0, tw1);
        const uint64_t *src = static_cast<const uint64_t *>(s.h_out);
        uint64_t *dsts[4];
        int ncol = 0;
        if (lo) { dsts[ncol++] = lo + s.begin; dsts[ncol++] = hi + s.begin; }
        if (ssamp) dsts[ncol++] = ssamp + s.begin;
        if (count) dsts[ncol++] = count + s.begin;
        const uint64_t cnt = s.cnt;
        par(cnt, [&](unsigned t) {  // pinned -> the caller's (pageable) arrays, every member a slice of every column
            const uint64_t i0 = cnt * t / T, i1 = cnt * (t + 1) / T;
            for (int c = 0; c < ncol; ++c) std::memcpy(dsts[c] + i0, src + static_cast<uint64_t>(c) * cnt + i0, (i1 - i0) * 8);
        });
        t_out += secs(tw1, now());
        s.busy = false;
        return RBG_OK;
    };

    uint64_t chunk_reads = kHostChunkReads;
    if (const char *e = std::getenv("RBG_HOST_CHUNK_READS")) chunk_reads = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));   // (tests: many chunks from a small batch)
    std::vector<uint64_t> part(T + 1), psym(T);
    std::atomic<bool> bad_offsets{false};   // an offset smaller than its predecessor (check_offsets), found by a sizing pass
    uint64_t b = 0;
    unsigned seq = 0;   // chunks enqueued so far
    while (b < N && !rc) {
        // the chunk [b, e): bounded in reads and in symbols (no pass over the reads just to find the bound: the packed
        // layout has the symbol count in its offsets, spans are sampled and measured by the staging pass itself)
        uint64_t e = std::min<uint64_t>(N, b + chunk_reads), sym = 0;
        if (R.off) {
            while (e > b + 1 && R.off[e] - R.off[b] > kHostChunkBytes) e = b + (e - b) / 2;
        } else {
            uint64_t mx = 0;
            for (uint64_t i = b; i < e; i += 1 + (e - b) / 64) mx = std::max<uint64_t>(mx, R.len[i]);
            while (e > b + 1 && (e - b) * std::max<uint64_t>(mx, 64) > 4 * kHostChunkBytes) e = b + (e - b) / 2;   // long reads: fewer per chunk
        }
        // sizing pass: every member measures its slice (symbols, or 16-byte chunks of the 2-bit form) and checks that the
        // offsets ascend (check_offsets): nothing of the caller's read bytes is touched, and nothing sized, before that
        const auto ts0 = now();
        par(e - b, [&](unsigned t) {
            const uint64_t i0 = b + (e - b) * t / T, i1 = b + (e - b) * (t + 1) / T;
            uint64_t c = 0, sy = 0;
            bool bad = false;
            if (R.off) {
                for (uint64_t i = i0; i < i1; ++i) { const uint64_t m = R.off[i + 1] - R.off[i]; bad |= R.off[i + 1] < R.off[i]; sy += m; c += (m + 63) >> 6; }
            } else {
                for (uint64_t i = i0; i < i1; ++i) { const uint64_t m = R.len[i]; sy += m; c += (m + 63) >> 6; }
            }
            if (bad) bad_offsets = true;
            part[t + 1] = pack ? c : sy;
            psym[t] = sy;
        });
        t_pack += secs(ts0, now());
        if (bad_offsets) { rc = RBG_EARG; break; }
        for (uint64_t v : psym) sym += v;
        part[0] = 0;
        for (unsigned t = 0; t < T; ++t) part[t + 1] += part[t];
        if (pack && part[T] >= (uint64_t(1) << 32)) { rc = RBG_EARG; break; }   // chunk indices are 32-bit
        const uint64_t cnt = e - b;
        Slot &s = W.slot[seq % rbg_hostpath::kSlots];
        ++seq;
        if ((rc = drain(s))) break;   // (waits only when every buffer is in flight: normally drained below)
        // device columns: lo, hi, [ssamp], [count]
        const uint64_t dev_cols = 2 + (ssamp ? 1 : 0) + (count ? 1 : 0);
        size_t in_bytes;
        if (pack) in_bytes = cnt * 8 + 16 + (sym / 64 + cnt + 1) * 16;
        else in_bytes = (cnt + 1) * 8 + 16 + sym + 32;
        const int er = W.ensure(s, in_bytes, dev_cols * cnt * 8);
        if (er) { rc = er == 2 ? RBG_ENOMEM : RBG_ENODEV; break; }
        const auto tp0 = now();
        char *hin = static_cast<char *>(s.h_in);
        char *din = static_cast<char *>(s.d_in);
        uint64_t *dout = static_cast<uint64_t *>(s.d_out);
        uint64_t *d_lo = dout, *d_hi = dout + cnt, *d_ss = ssamp ? dout + 2 * cnt : nullptr;
        uint64_t *d_cnt = count ? dout + (ssamp ? 3 : 2) * cnt : nullptr;
        // Results leave without a copy engine: the kernels store the columns the caller wants straight into the pinned
        // buffer (device-visible host memory; 8 bytes per lane, whole lines per wave, posted writes over PCIe).  A
        // device-to-host copy enqueued behind the search of chunk c holds up the copy IN of chunk c + 1 on this
        // platform until that search has finished (one engine serves both directions, in order: measured with
        // RBG_HOST_TRACE=2), which serialised copy in / search / copy out of successive chunks.  RBG_HOST_DIRECT_OUT=0
        // keeps the copies (A/B measurements).
        if (direct_out) {
            uint64_t *hcol = static_cast<uint64_t *>(s.h_out);
            if (lo) { d_lo = hcol; d_hi = hcol + cnt; hcol += 2 * cnt; }
            if (ssamp) { d_ss = hcol; hcol += cnt; }
            if (count) { d_cnt = hcol; hcol += cnt; }
        }
        size_t used = 0;
        if (pack) {
            uint2 *meta = reinterpret_cast<uint2 *>(hin);
            const size_t chunks_at = (cnt * 8 + 15) & ~size_t(15);
            uint32_t *chunks = reinterpret_cast<uint32_t *>(hin + chunks_at);
            // every member packs from its own prefix of 16-byte chunks (sizing pass above)
            par(cnt, [&](unsigned t) {
                const uint64_t i0 = b + cnt * t / T, i1 = b + cnt * (t + 1) / T;
                uint64_t c = part[t];
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint64_t m = R.length(i);
                    uint32_t *dst = chunks + c * 4;
                    const bool ok = m < 0x80000000ull &&
                                    (acgt ? rbg_hostpath::pack_read_acgt(R.ptr(i), m, dst) : rbg_hostpath::pack_read_lut(R.ptr(i), m, h.major_of, dst));
                    meta[i - b] = make_uint2(static_cast<uint32_t>(c), ok ? static_cast<uint32_t>(m) : 0x80000000u);
                    if (!ok) W.bad[t].push_back(i);
                    c += (m + 63) >> 6;
                }
            });
            used = chunks_at + part[T] * 16;
            t_pack += secs(tp0, now());
            mark(0, s.st);
            if (timeline) tl.back().host_ms = secs(t_call, now()) * 1e3;
            if (hipMemcpyAsync(din, hin, used, hipMemcpyHostToDevice, s.st) != hipSuccess) rc = RBG_ENODEV;
            mark(1, s.st);
            if (!rc && launch_find_range_packed_only(ix->dev, ix->cfg, reinterpret_cast<const uint2 *>(din), reinterpret_cast<const uint4 *>(din + chunks_at), cnt,
                                                     d_lo, d_hi, d_ss, s.st))
                rc = RBG_ENODEV;
            mark(2, s.st);
        } else {
            uint64_t *off2 = reinterpret_cast<uint64_t *>(hin);
            const size_t bytes_at = ((cnt + 1) * 8 + 15) & ~size_t(15);
            char *bytes = hin + bytes_at;
            par(cnt, [&](unsigned t) {
                const uint64_t i0 = b + cnt * t / T, i1 = b + cnt * (t + 1) / T;
                uint64_t c = part[t];
                if (R.off && i1 > i0) {  // contiguous in the source: one copy per slice
                    std::memcpy(bytes + c, R.ptr(i0), R.off[i1] - R.off[i0]);
                    for (uint64_t i = i0; i < i1; ++i) off2[i - b] = c + (R.off[i] - R.off[i0]);
                } else {
                    for (uint64_t i = i0; i < i1; ++i) {
                        const uint64_t m = R.length(i);
                        std::memcpy(bytes + c, R.ptr(i), m);
                        off2[i - b] = c;
                        c += m;
                    }
                }
            });
            off2[cnt] = part[T];
            used = bytes_at + part[T];
            t_pack += secs(tp0, now());
            mark(0, s.st);
            if (timeline) tl.back().host_ms = secs(t_call, now()) * 1e3;
            if (hipMemcpyAsync(din, hin, used, hipMemcpyHostToDevice, s.st) != hipSuccess) rc = RBG_ENODEV;
            mark(1, s.st);
            if (!rc && launch_find_range(ix->dev, ix->cfg, reinterpret_cast<const uint8_t *>(din + bytes_at), reinterpret_cast<const uint64_t *>(din), cnt, d_lo,
                                         d_hi, d_ss, s.st))
                rc = RBG_ENODEV;
            mark(2, s.st);
        }
        if (rc) break;
        if (count && launch_count_from_ranges(d_lo, d_hi, cnt, d_cnt, s.st)) { rc = RBG_ENODEV; break; }
        // results through pinned memory: [lo | hi] (when asked for) | ssamp | count, the columns the caller wants
        {
            char *hout = static_cast<char *>(s.h_out);
            size_t at = 0;
            hipError_t e2 = hipSuccess;
            if (!direct_out) {
                if (lo) { e2 = hipMemcpyAsync(hout, d_lo, 2 * cnt * 8, hipMemcpyDeviceToHost, s.st); at += 2 * cnt * 8; }
                if (e2 == hipSuccess && ssamp) { e2 = hipMemcpyAsync(hout + at, d_ss, cnt * 8, hipMemcpyDeviceToHost, s.st); at += cnt * 8; }
                if (e2 == hipSuccess && count) { e2 = hipMemcpyAsync(hout + at, d_cnt, cnt * 8, hipMemcpyDeviceToHost, s.st); at += cnt * 8; }
            }
            mark(3, s.st);
            if (e2 == hipSuccess) e2 = hipEventRecord(s.done, s.st);
            if (e2 != hipSuccess) { rc = RBG_ENODEV; break; }
        }
        s.begin = b;
        s.cnt = cnt;
        s.busy = true;
        b = e;
        t_enq = secs(t_call, now()) - t_pack - t_wait - t_out;
        // while the GPU works: hand the chunks that have finished to the caller, oldest first, without waiting
        for (unsigned j = 1; j < rbg_hostpath::kSlots && !rc; ++j) {
            Slot &o = W.slot[(seq - 1 + j) % rbg_hostpath::kSlots];
            if (!o.busy) continue;
            if (hipEventQuery(o.done) != hipSuccess) break;
            rc = drain(o);
        }
    }
    (void)nout;
    for (unsigned j = 0; j < rbg_hostpath::kSlots; ++j) {   // what is still in flight, oldest first
        Slot &s = W.slot[(seq + j) % rbg_hostpath::kSlots];
        const int r2 = rc ? RBG_OK : drain(s);
        if (!rc) rc = r2;
        // on an error EVERY stream that exists is drained, marked busy or not: a chunk whose copy or search was enqueued
        // before a later step of the same chunk failed is in flight without the mark, and the workspace (its pinned
        // buffers, which a direct-out kernel writes) goes back to the pool when this call returns
        if (rc && s.st) { (void)hipStreamSynchronize(s.st); s.busy = false; }
    }
    if (rc) {
        for (ChunkEvents &c : tl) for (hipEvent_t &e : c.e) (void)hipEventDestroy(e);
        return rc;
    }
    if (trace)
        std::fprintf(stderr, "rbg host call: %llu reads, %s, %u threads: %.2f ms = stage %.2f + enqueue/other %.2f + wait for the GPU %.2f + copy out %.2f\n",
                     static_cast<unsigned long long>(N), pack ? "2-bit" : "bytes", T, secs(t_call, now()) * 1e3, t_pack * 1e3, t_enq * 1e3, t_wait * 1e3,
                     t_out * 1e3);
    if (timeline && !tl.empty()) {
        for (size_t c = 0; c < tl.size(); ++c) {
            float t[4] = {0, 0, 0, 0};
            for (int j = 0; j < 4; ++j) (void)hipEventElapsedTime(&t[j], tl[0].e[0], tl[c].e[j]);
            std::fprintf(stderr, "  chunk %2zu: enqueued at %7.2f ms (host clock); device clock from the first copy: copy in %7.2f..%7.2f, search ..%7.2f, copy out ..%7.2f\n",
                         c, tl[c].host_ms, t[0], t[1], t[2], t[3]);
        }
        for (ChunkEvents &c : tl) for (hipEvent_t &e : c.e) (void)hipEventDestroy(e);
    }
    // reads the 2-bit form cannot express (any symbol outside the k-mer alphabet): searched from their bytes
    std::vector<uint64_t> bad;
    for (auto &v : W.bad) bad.insert(bad.end(), v.begin(), v.end());
    if (!bad.empty()) {
        std::sort(bad.begin(), bad.end());
        std::vector<uint64_t> bb(bad.size());
        std::vector<uint32_t> bl(bad.size());
        bool fits = true;
        for (size_t j = 0; j < bad.size(); ++j) {
            bb[j] = static_cast<uint64_t>(R.ptr(bad[j]) - R.base);
            const uint64_t m = R.length(bad[j]);
            if (m > 0xFFFFFFFFull) fits = false;
            bl[j] = static_cast<uint32_t>(m);
        }
        std::vector<uint64_t> t_lo(bad.size()), t_hi(bad.size()), t_ss(ssamp ? bad.size() : 0), t_cnt(count ? bad.size() : 0);
        HostReads Rb;
        Rb.base = R.base;
        std::vector<uint64_t> off3;
        std::string flat;
        if (fits) {
            Rb.begin = bb.data();
            Rb.len = bl.data();
        } else {  // a read beyond 4 GB: gather into the packed layout
            off3.assign(1, 0);
            for (uint64_t i : bad) { flat.append(reinterpret_cast<const char *>(R.ptr(i)), R.length(i)); off3.push_back(flat.size()); }
            Rb.base = reinterpret_cast<const uint8_t *>(flat.data());
            Rb.off = off3.data();
        }
        // (the lease is still held: the recursive call takes another workspace)
        rc = find_range_host_core(ix, Rb, bad.size(), t_lo.data(), t_hi.data(), ssamp ? t_ss.data() : nullptr, count ? t_cnt.data() : nullptr, false);
        if (rc) return rc;
        for (size_t j = 0; j < bad.size(); ++j) {
            if (lo) { lo[bad[j]] = t_lo[j]; hi[bad[j]] = t_hi[j]; }
            if (ssamp) ssamp[bad[j]] = t_ss[j];
            if (count) count[bad[j]] = t_cnt[j];
        }
    }
    return RBG_OK;
}

int rbg_lf(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, const uint8_t *sym, uint64_t N, uint64_t *lo_out,
           uint64_t *hi_out) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!lo || !hi || !sym || !lo_out || !hi_out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    DevBuf dlo, dhi, dsym, dlo2, dhi2;
    int rc;
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = dsym.alloc(N)) || (rc = dlo2.alloc(N * 8)) || (rc = dhi2.alloc(N * 8)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dlo.p, lo, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dhi.p, hi, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dsym.p, sym, N, hipMemcpyHostToDevice, st));
    if (launch_lf(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), dsym.as<uint8_t>(), N, dlo2.as<uint64_t>(), dhi2.as<uint64_t>(), st))
        return RBG_ENODEV;
    HIP_TRY(hipMemcpyAsync(lo_out, dlo2.p, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hi_out, dhi2.p, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RBG_OK;
    });
}

int rbg_find_range(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo, uint64_t *hi) {
    return guarded([&]() -> int {
    if (N && (!lo || !hi)) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, lo, hi, nullptr, nullptr);
    });
}

int rbg_count(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *count) {
    return guarded([&]() -> int {
    if (N && !count) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, nullptr, nullptr, nullptr, count);
    });
}

int rbg_find_range_spans(rbg_index *ix, const uint8_t *base, const uint64_t *begin, const uint32_t *len, uint64_t N, uint64_t *lo,
                         uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (ssamp && !ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N == 0) return RBG_OK;
    if (!base || !begin || !len || !lo || !hi) return RBG_EARG;
    HostReads R;
    R.base = base;
    R.begin = begin;
    R.len = len;
    return find_range_host_core(ix, R, N, lo, hi, ssamp, nullptr, true);
    });
}

int rbg_find_range_w_toehold(rbg_index *ix, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                             uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (ix && !ix->H().has_tsa) return RBG_ENOTLOADED;
    if (N && (!lo || !hi || !ssamp)) return RBG_EARG;
    return find_range_host(ix, seqs, off, N, lo, hi, ssamp, nullptr);
    });
}

int rbg_locs_at(rbg_index *ix, const uint64_t *lo, const uint64_t *hi, const uint64_t *k, uint64_t N,
                uint64_t max_hits, uint64_t *loc_off, uint64_t **locs) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!ix->H().has_tsa) return RBG_ENOTLOADED;
    if (!loc_off || !locs || (N && (!lo || !hi || !k))) return RBG_EARG;
    *locs = nullptr;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = hipStreamPerThread;
    DevBuf dlo, dhi, dk, doff, dtmp;
    const size_t tmp_bytes = scan_tmp_bytes(N);
    int rc;
    if ((rc = dlo.alloc(N * 8)) || (rc = dhi.alloc(N * 8)) || (rc = dk.alloc(N * 8)) || (rc = doff.alloc((N + 1) * 8)) ||
        (rc = dtmp.alloc(tmp_bytes)))
        return rc;
    if (N) {
        HIP_TRY(hipMemcpyAsync(dlo.p, lo, N * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dhi.p, hi, N * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dk.p, k, N * 8, hipMemcpyHostToDevice, st));
    }
    if (launch_locate_plan(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), N, max_hits, doff.as<uint64_t>(), dtmp.p, tmp_bytes, st))
        return RBG_ENODEV;
    DevBuf dord;
    const void *order = nullptr;
    if ((rc = make_order(ix, dk.as<uint64_t>(), N, dord, st, &order))) return rc;
    return ragged_finish(N, doff, loc_off, locs, st, [&](uint64_t *d_vals) {
        return launch_locate_fill(ix->dev, ix->cfg, dlo.as<uint64_t>(), dhi.as<uint64_t>(), dk.as<uint64_t>(), N, max_hits,
                                  doff.as<uint64_t>(), d_vals, nullptr, order, st) ? RBG_ENODEV : RBG_OK;
    });
    });
}

}  // extern "C"
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

// ---- more than one GPU in one process (index replicated, reads sharded, no data-path collective) ----
// The replica is built ONCE (load / build on the primary's device) and copied to the other devices peer to peer
// (xGMI); records that hold device pointers (DevSym arrays, DevIndex) are re-pointed into the copy.


}  // extern "C"

namespace {

// one target of a fan-out: the new handle, its stream (on the target device) and the relocation map of its copy
struct ReplicaJob {
    rbg_index *r = nullptr;
    hipStream_t st = nullptr;
    Reloc reloc;
};

// allocate on `device` and ENQUEUE the peer copies of every allocation of `src` on the job's own stream: nothing here
// waits, so the copies of several targets run side by side (each target pulls over its own xGMI link to the source)
int replicate_begin(rbg_index *src, int device, ReplicaJob &job) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RBG_ENODEV;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RBG_ENODEV;
    rbg_index *r = new (std::nothrow) rbg_index();
    if (!r) return RBG_ENOMEM;
    job.r = r;
    r->primary = src;
    r->device = device;
    r->cfg = src->cfg;
    r->cfg.max_blocks = prop.multiProcessorCount * 32;
    r->runs_layout = src->runs_layout;
    r->run_depth_mask = src->run_depth_mask;
    r->runs_report = src->runs_report;
    r->rank_slots = src->rank_slots; r->rank_slots_overflow = src->rank_slots_overflow;
    r->phi_slots = src->phi_slots; r->phi_slots_overflow = src->phi_slots_overflow;
    r->kmer_steps_requested = src->kmer_steps_requested;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return RBG_ENODEV;
    r->hbm_free_at_load = free_b;
    r->hbm_budget = src->hbm_budget;
    if (device != src->device) {
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, device, src->device);
        if (can) (void)hipDeviceEnablePeerAccess(src->device, 0);  // already enabled is fine
        (void)hipGetLastError();
    }
    HIP_TRY(hipStreamCreateWithFlags(&job.st, hipStreamNonBlocking));
    for (const DevAlloc &a : src->allocs) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, a.bytes);
        if (e == hipSuccess) {
            r->allocs.push_back({p, a.bytes});
            r->hbm_bytes += a.bytes;
            e = hipMemcpyPeerAsync(p, device, a.p, src->device, a.bytes, job.st);
        }
        if (e != hipSuccess) {
            std::fprintf(stderr, "rbg: replicating %.1f GB to device %d failed: %s\n", src->hbm_bytes / 1e9, device, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? RBG_ENOMEM : RBG_ENODEV;
        }
        job.reloc.from.push_back(a);
        job.reloc.to.push_back({p, a.bytes});
    }
    return RBG_OK;
}

// wait for the job's copies, then re-point the records that hold device pointers
int replicate_finish(rbg_index *src, ReplicaJob &job) {
    rbg_index *r = job.r;
    const Reloc &reloc = job.reloc;
    DeviceScope scope(r->device);
    if (scope.rc) return scope.rc;
    HIP_TRY(hipStreamSynchronize(job.st));
    for (const PtrTable &t : src->ptr_tables) {
        std::vector<char> buf(t.count * t.stride);
        void *dst = const_cast<void *>(reloc(t.d_ptr));
        if (!dst) return RBG_ENODEV;
        if (hipMemcpy(buf.data(), dst, buf.size(), hipMemcpyDeviceToHost) != hipSuccess) return RBG_ENODEV;
        for (size_t i = 0; i < t.count; ++i)
            for (size_t o : t.ptr_offsets) {
                const void *old;
                std::memcpy(&old, buf.data() + i * t.stride + o, sizeof(old));
                const void *nw = reloc(old);
                std::memcpy(buf.data() + i * t.stride + o, &nw, sizeof(nw));
            }
        if (hipMemcpy(dst, buf.data(), buf.size(), hipMemcpyHostToDevice) != hipSuccess) return RBG_ENODEV;
        r->ptr_tables.push_back({dst, t.count, t.stride, t.ptr_offsets});
    }
    DevIndex d = src->dev;
    reloc.fix(d.syms); reloc.fix(d.phi_ent); reloc.fix(d.phi_slots); reloc.fix(d.phi_ord);
    reloc.fix(d.mk_start); reloc.fix(d.mk_end); reloc.fix(d.mk_off); reloc.fix(d.mk_vals); reloc.fix(d.mk_bucket);
    reloc.fix(d.counters); reloc.fix(d.lut); reloc.fix(d.pairs); reloc.fix(d.triples); reloc.fix(d.quads); reloc.fix(d.quints);
    reloc.fix(d.lut2); reloc.fix(d.ftab); reloc.fix(d.dense);
    reloc.fix(d.phi_dir);
    for (int t = 0; t < kMaxRunDepth; ++t) reloc.fix(d.run_samp[t]);
    reloc.fix(d.run_tabs2); reloc.fix(d.run_hot); reloc.fix(d.phi_super);
    for (int t = 0; t < kMaxRunDepth; ++t) { reloc.fix(d.run_ent2[t]); reloc.fix(d.run_dir2[t]); reloc.fix(d.run_rec2[t]); }
    r->dev = d;
    if (hipMemset(d.counters, 0, 4 * sizeof(uint64_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RBG_ENODEV;
    return RBG_OK;
}

}  // namespace

extern "C" {

// Replicas on `devices[0..G)` from ONE finished index: every target's copies are enqueued before any is waited for, so
// the G transfers overlap -- on MI355X's fully connected xGMI each target has its own link to the source, and the
// fan-out takes about the time of one copy instead of G (a chain 0 -> 1 -> ... would use each link once as well but
// serialise on the first hop; with a direct link per pair the star is the better shape).
int rbg_replicate_many(rbg_index *src, const int *devices, int G, rbg_index **out) {
    return guarded([&]() -> int {
    if (!src || !out || !devices || G <= 0) return RBG_EARG;
    for (int g = 0; g < G; ++g) out[g] = nullptr;
    if (!queryable(src)) return RBG_ENODEV;
    if (src->primary) return RBG_EARG;  // replicate the primary, not a replica
    {
        DeviceScope s0(src->device);
        if (s0.rc) return s0.rc;
        HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<ReplicaJob> jobs(G);
    int rc = RBG_OK;
    for (int g = 0; g < G && !rc; ++g) rc = replicate_begin(src, devices[g], jobs[g]);
    for (int g = 0; g < G; ++g) {
        if (!jobs[g].r) continue;
        if (!rc && jobs[g].st) rc = replicate_finish(src, jobs[g]);
        else if (jobs[g].st) (void)hipStreamSynchronize(jobs[g].st);   // never free memory a copy is still writing
        if (jobs[g].st) {
            DeviceScope scope(jobs[g].r->device);
            (void)hipStreamDestroy(jobs[g].st);
        }
    }
    if (rc) {
        for (int g = 0; g < G; ++g)
            if (jobs[g].r) rbg_free(jobs[g].r);
        return rc;
    }
    for (int g = 0; g < G; ++g) out[g] = jobs[g].r;
    return RBG_OK;
    });
}

int rbg_replicate(rbg_index *src, int device, rbg_index **out) {
    if (!out) return RBG_EARG;
    return rbg_replicate_many(src, &device, 1, out);
}

int rbg_shard_bounds(uint64_t n_items, int rank, int world, uint64_t *begin, uint64_t *end) {
    if (world <= 0 || rank < 0 || rank >= world || !begin || !end) return RBG_EARG;
    *begin = static_cast<uint64_t>((static_cast<unsigned __int128>(n_items) * static_cast<unsigned>(rank)) / static_cast<unsigned>(world));
    *end = static_cast<uint64_t>((static_cast<unsigned __int128>(n_items) * (static_cast<unsigned>(rank) + 1u)) / static_cast<unsigned>(world));
    return RBG_OK;
}

int rbg_find_range_sharded(rbg_index *const *replicas, int G, const uint8_t *seqs, const uint64_t *off, uint64_t N, uint64_t *lo,
                           uint64_t *hi, uint64_t *ssamp) {
    return guarded([&]() -> int {
    if (!replicas || G <= 0) return RBG_EARG;
    for (int g = 0; g < G; ++g)
        if (!queryable(replicas[g])) return RBG_ENODEV;
    if (N == 0) return RBG_OK;
    if (!off || !lo || !hi || (!seqs && off[N])) return RBG_EARG;
    int rc0 = check_offsets(off, N);
    if (rc0) return rc0;
    std::vector<int> rcs(G, RBG_OK);
    auto work = [&](int g) {
        uint64_t b, e;
        (void)rbg_shard_bounds(N, g, G, &b, &e);
        if (e == b) return;
        std::vector<uint64_t> o(e - b + 1);  // the shard's offsets, re-based
        for (uint64_t i = b; i <= e; ++i) o[i - b] = off[i] - off[b];
        rcs[g] = find_range_host(replicas[g], seqs + off[b], o.data(), e - b, lo + b, hi + b, ssamp ? ssamp + b : nullptr, nullptr);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(work, g);
    work(0);
    for (auto &t : th) t.join();
    for (int rc : rcs)
        if (rc) return rc;
    return RBG_OK;
    });
}

// ---- counters --------------------------------------------------------------------------------------

int rbg_counters(rbg_index *ix, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, ix->dev.counters, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RBG_OK;
    });
}

int rbg_combine_stats(rbg_index *ix, uint64_t out[2]) {
    if (!ix || !out) return RBG_EARG;
    rbg_index *root = ix;
    out[0] = root->comb_launches.load(std::memory_order_relaxed);
    out[1] = root->comb_requests.load(std::memory_order_relaxed);
    return RBG_OK;
}

int rbg_counters_reset(rbg_index *ix) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ix->dev.counters, 0, 4 * sizeof(uint64_t)));
    return RBG_OK;
    });
}

}  // extern "C"

namespace {

// RCCL is needed by two optional calls only (the counters' all-reduce), so the library does not link it: it is opened
// on first use, and a process that never reduces counters loads librbg.so on a machine without RCCL.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
    static Rccl &get() {
        static Rccl r = [] {
            Rccl x;
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.h) break;
            }
            if (!x.h) {
                std::fprintf(stderr, "rbg: RCCL not found (%s): the counters' all-reduce is unavailable\n", dlerror());
                return x;
            }
            x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.h, "ncclAllReduce"));
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.h, "ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.h, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.h, "ncclGroupEnd"));
            x.ok = x.AllReduce && x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd;
            return x;
        }();
        return r;
    }
};

// One communicator clique per set of devices, made on first use and kept for the life of the process: a reduce per
// batch must not pay ncclCommInitAll (hundreds of milliseconds on 8 GPUs) every time.  rbg_comm_cache_clear() drops them.
struct CliqueCache {
    std::mutex mu;
    std::map<std::vector<int>, std::vector<ncclComm_t>> cliques;
    static CliqueCache &get() {
        static CliqueCache *c = new CliqueCache();   // never destroyed: communicators must not be torn down at exit time
        return *c;
    }
};

}  // namespace

extern "C" {

int rbg_comm_cache_clear(void) {
    return guarded([&]() -> int {
    CliqueCache &cc = CliqueCache::get();
    std::lock_guard<std::mutex> lk(cc.mu);
    if (!cc.cliques.empty() && Rccl::get().ok)
        for (auto &kv : cc.cliques)
            for (ncclComm_t c : kv.second) (void)Rccl::get().CommDestroy(c);
    cc.cliques.clear();
    return RBG_OK;
    });
}

// One RCCL all-reduce (sum) of the four 64-bit counters over the communicator's ranks: the run's only collective
// (the reference has none).  `nccl_comm` is the caller's ncclComm_t for this replica's device.
int rbg_counters_allreduce(rbg_index *ix, void *nccl_comm, void *stream, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!queryable(ix)) return RBG_ENODEV;
    if (!nccl_comm || !out) return RBG_EARG;
    DeviceScope scope(ix->device);
    if (scope.rc) return scope.rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DevBuf sum;
    int rc = sum.alloc(4 * sizeof(uint64_t));
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());  // every query launched so far has added its counts
    if (!Rccl::get().ok) return RBG_ENODEV;
    if (Rccl::get().AllReduce(ix->dev.counters, sum.p, 4, ncclUint64, ncclSum, static_cast<ncclComm_t>(nccl_comm), st) != ncclSuccess) return RBG_ENODEV;
    HIP_TRY(hipMemcpyAsync(out, sum.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RBG_OK;
    });
}

// The same for G replicas held by ONE process: a communicator clique over their devices (ncclCommInitAll once per
// device set, kept: CliqueCache), one grouped all-reduce, every replica ends with the same sums.  Needs G distinct devices.
int rbg_counters_allreduce_local(rbg_index *const *replicas, int G, uint64_t out[4]) {
    return guarded([&]() -> int {
    if (!replicas || G <= 0 || !out) return RBG_EARG;
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) {
        if (!queryable(replicas[g])) return RBG_ENODEV;
        devs[g] = replicas[g]->device;
        for (int h = 0; h < g; ++h)
            if (devs[h] == devs[g]) return RBG_EARG;
    }
    Rccl &nc = Rccl::get();
    if (!nc.ok) return RBG_ENODEV;
    CliqueCache &cc = CliqueCache::get();
    std::lock_guard<std::mutex> lk(cc.mu);   // one reduce at a time per process: the clique is shared
    auto it = cc.cliques.find(devs);
    if (it == cc.cliques.end()) {
        std::vector<ncclComm_t> fresh(G);
        if (nc.CommInitAll(fresh.data(), G, devs.data()) != ncclSuccess) return RBG_ENODEV;
        it = cc.cliques.emplace(devs, std::move(fresh)).first;
    }
    const std::vector<ncclComm_t> &comms = it->second;
    std::vector<void *> sums(G, nullptr);
    int rc = RBG_OK;
    for (int g = 0; g < G && !rc; ++g) {
        DeviceScope scope(devs[g]);
        if (hipDeviceSynchronize() != hipSuccess || hipMalloc(&sums[g], 32) != hipSuccess) rc = RBG_ENODEV;
    }
    if (!rc) {
        (void)nc.GroupStart();
        for (int g = 0; g < G; ++g) {
            DeviceScope scope(devs[g]);
            if (nc.AllReduce(replicas[g]->dev.counters, sums[g], 4, ncclUint64, ncclSum, comms[g], nullptr) != ncclSuccess) rc = RBG_ENODEV;
        }
        if (nc.GroupEnd() != ncclSuccess) rc = RBG_ENODEV;
    }
    for (int g = 0; g < G; ++g) {
        DeviceScope scope(devs[g]);
        if (!rc && hipDeviceSynchronize() != hipSuccess) rc = RBG_ENODEV;
        if (!rc && g == 0 && hipMemcpy(out, sums[0], 32, hipMemcpyDeviceToHost) != hipSuccess) rc = RBG_ENODEV;
        if (sums[g]) (void)hipFree(sums[g]);
    }
    return rc;
    });
}

}  // extern "C"
