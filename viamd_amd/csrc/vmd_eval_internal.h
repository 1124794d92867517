#pragma once
// viamd_amd/csrc/vmd_eval_internal.h - the evaluator's shared declarations: C++ host side of the drop-in boundary (include/vmd_eval.h).
//
// Mirrors the md_script_eval_* lifecycle VIAMD drives (/root/reference/src/main.cpp:951-1039): create -> clear_data -> frame_range from
// pool threads -> property_data / frame_mask polled by the GUI thread.  All arithmetic happens in the HIP kernels of vmd_kernels.hip;
// the host code only batches frames, owns the device buffers and keeps the md_script_property_data_t views up to date.  There is no
// CPU compute path.  Round 6: the one translation unit this used to be (4 300 lines) is eight files by concern -
//     vmd_eval_runtime.cpp  errors, options, profiling, resource pool        vmd_eval_ir.cpp     property descriptors
//     vmd_eval_core.cpp     the eval object, host views, accessors           vmd_eval_stage.cpp  static uploads, trajectory staging
//     vmd_eval_batch.cpp    grids, cell builds, batches, process_range       vmd_eval_calls.cpp  queue, read-ahead, deferred settle
//     vmd_eval_traj.cpp     trajectory kinds, checkpoint / mapping caches    vmd_eval_post.cpp   histogram post-processing
// - and this header holds every struct they share, in the order the single file declared them, with each function's prototype where
// its definition used to stand.  Nothing here is part of the ABI.
#include <hip/hip_runtime.h>

#include <float.h>

#include <algorithm>
#include <sys/mman.h>

#include <atomic>
#include <climits>
#include <chrono>
#include <unistd.h>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

#include "vmd_eval.h"
#include "vmd_hip.h"

// ------------------------------------------------------------------------------------------------ errors / options
extern thread_local std::string g_last_error;

// md_log_register analogue (VIAMD installs a logger that turns messages into toasts, src/main.cpp:384-420): failures go to
// the registered callback, or to stderr when there is none.  The callback may be invoked from any thread that calls the API.
extern std::mutex g_log_mtx;

extern vmd_log_fn g_log_fn;

extern void* g_log_user;

void vmd_log(int level, const char* msg);

bool vmd_fail(const char* fmt, ...);

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return vmd_fail("%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)
#define KRN_OK(expr)                                                                              \
    do {                                                                                          \
        int e_ = (expr);                                                                          \
        if (e_ != 0) return vmd_fail("%s failed: %s", #expr, hipGetErrorString((hipError_t)e_));  \
    } while (0)

struct Options {
    std::atomic<int> rdf_variant{0};     // 0 queue, 1 inline
    std::atomic<int> batch_frames{0};    // 0 = auto
    std::atomic<int> force_brute{0};
    std::atomic<int> load_threads{0};    // host threads decoding one staged batch through load_frame; 0 = auto (see load_threads())
    // pencils of cross-section rmax/split (walk reach = split).  z: explicit (A/B).  y: 0 = by density - selections of >= 0.08 atoms / A^3
    // in the lanes of every pass of a group (all heavy atoms of a liquid; SURVEY 8d's C3-dense) walk half-width pencils in y: a 64-atom
    // chunk of such a selection is only ~4 A long, so the x windows are dominated by the 2 r_max of padding and thinner pencils pay
    // (c3d 2 028 -> 2 131 frames/s, profiles/r05d_pencil_split_by_density.txt; at c3's 0.033 / A^3 the same split costs 10 %, at c5's mix
    // 6 %); 1 / 2 / .. = fixed
    std::atomic<int> pencil_split_y{0}, pencil_split_z{1};
    // fine x cell = rmax / nxf_divisor (8 / 12 / 16 / 24 / 32 measured: 16 is +0.8 % on c3, profiles/r02l_ab_fine_cells.txt)
    std::atomic<int> nxf_divisor{16};
    std::atomic<int> cells_aos{1};       // sort through 16-byte records + repack
    // frames offered raw (load_raw) are decompressed on the device (0 = on the host threads): 1 = one thread per
    std::atomic<int> xtc_device_decode{3};
                                             // frame (k_xtc_decode), 2 = index pass + one thread per chunk (k_xtc_index / k_xtc_chunks),
                                             // 3 = one wave per frame (k_xtc_wave)
    std::atomic<int> xtc_chunk{256};         // atoms per chunk of variant 2
    // process-wide cache of device blocks freed by evals (MB; pinned host blocks: a quarter of it); 0 = off
    std::atomic<int> pool_mb{16384};
    // combining queue: how long the leader waits for the other pool threads of the previous round to come back with their next ranges (0 =
    // take what is there)
    std::atomic<int> gather_us{150};
    // combining queue: the host views are refreshed when no call is waiting (and every lazy_views_ms at the latest), not after every batch
    std::atomic<int> lazy_views{1};
    std::atomic<int> lazy_views_ms{20};
    // the next batch is queued before the host waits for the current one (evals without block partials); measured r03ad: no gain (the
    // per-batch host gap is ~0.06 ms; the next decode then lands on the cell build), off
    std::atomic<int> defer_sync{0};
    // filtered evaluation: consecutive frame blocks share ONE batch (one cell build, one synchronisation; a pair launch per block)
    std::atomic<int> block_superbatch{1};
    // ... and the blocks' pair launches alternate between two streams, so that the tail of one runs under the head of the next
    std::atomic<int> block_two_streams{1};
    // file-backed device decode: small first and last batches (pipeline fill / drain); r03o: no gain, off
    std::atomic<int> xtc_ramp{0};
    std::atomic<int> xtc_decode_ahead{1};    // batches the device decoder runs ahead of the kernels (1 or 2); r03n: 2 changes nothing
    // TRR / DCD: frames DMA'd out of the mapped file, swapped / scaled / transposed by k_raw_f32 (0: host threads)
    std::atomic<int> raw_f32_device{1};
    std::atomic<int> xtc_cold_streams{1};    // first pass out of a mapped file: up to four batches walked side by side on their own streams
    // variant 3: DMA the compressed frames straight out of the mapped file (raw_mapped_view), no host copy
    std::atomic<int> xtc_mapped{1};
    std::atomic<int> xtc_map_limit_mb{0};    // pinned bytes of mapped files, all trajectories together (0 = half of the physical memory)
    // variant 3: the first decode also leaves a 16-bit record per group; later decodes place every group from them, no walk.  Measured
    // (r03t2, c2): +2 % from a file, +3 % compressed-resident - the walk was a fifth of a re-decode, the per-group arithmetic is the rest.
    // 1 = for file-backed trajectories (records in the process-wide store), 2 = also for vmd_rawtraj_* objects, whose point is a small
    // footprint (atoms x 2 bytes per frame on top of ~5 bytes per atom of bit stream)
    std::atomic<int> xtc_records{1};
    std::atomic<int> xtc_record_mb{2048};    // ... as long as frames x atoms x 2 bytes of a trajectory stay below this
    std::atomic<int> xtc_checkpoints{1};     // variant 3: the first decode of a frame leaves checkpoints, later ones decode it in sections
    // oracle/SPEC.md's DECISION: tags as switches - 0 = the documented default, 1 = the alternative; read when an eval is created
    std::atomic<int> spec_rdf_closed{0};          // D-RDF-OPEN: r_min <= d <= r_max instead of the open interval
    std::atomic<int> spec_sdf_include_self{0};    // D-SDF-EXCL: targets that are atoms of structure k are scattered like any other
    std::atomic<int> spec_sdf_density{0};         // D-SDF-NORM: values = counts / (frames evaluated x voxel volume) instead of raw counts
    std::atomic<int> spec_dist_geometric_com{0};  // D-DIST-COM: distance(a, b) between geometric centres, not centres of mass
    // D-WRAP: positions enter rdf() as they are, minimum image by rounding - evaluated by k_rdf_brute (all pairs: a
    std::atomic<int> spec_rdf_raw{0};
                                                  // setting for matching an mdlib that does it this way, not a fast path)
    // D-RDF-NORM: 0 = cell volume when fully periodic, else the cutoff sphere; 1 = always the cutoff sphere;
    std::atomic<int> spec_rdf_norm{0};
                                                  // 2 = per reference atom (the weights do not carry N_ref)
    std::atomic<int> sdf_direct_view{1};          // k_counts_to_float writes the volume's float view into its pinned host pages itself
    std::atomic<int> stage_frames{128};      // frames per staged batch of a host / file trajectory (batch_frames <= 0)
    // pair-kernel grid while batches are decompressed on the device: 6 blocks per CU leave every
    std::atomic<int> rdf_blocks_decode{1536};
                                                // SIMD a wave slot and 80 VGPRs, so k_xtc_wave of batch k + 1 (wave priority 3) runs under
                                                // the pair kernel of batch k; costs the pair kernel ~4 % (0 = leave the grid alone)
    std::atomic<int> sdf_dense{0};       // dense-target SDF scatter (stream whole frames, select by tag): measured slower, off
    // SDF target lists that are arithmetic progressions are generated on the device (0 = always load the index list)
    std::atomic<int> sdf_arith{1};
    // selections of at most this many atoms are sorted by one block per frame (k_cells_fused), never through pencil buckets (0: buckets for
    // everyone)
    std::atomic<int> cells_small{8192};
    // co-evaluated RDFs of one range share pair passes through disjoint atom classes (0 = one pass per property)
    std::atomic<int> rdf_classes{1};
    // read-ahead under VIAMD's call pattern (many pool threads, ranges of a frame or a few): the first small call that finds
    // company evaluates a whole REGION of frame blocks ahead into block partials, later calls for those frames only mark them requested
    std::atomic<int> readahead{1};           // 0 = every call is evaluated when it arrives (the combining queue of round 3)
    std::atomic<int> readahead_frames{128};  // frames of the first region of an evaluation (rounded to whole blocks) ...
    std::atomic<int> readahead_growth{4};    // ... every further region is this many times larger (up to one kernel batch)
    std::atomic<int> readahead_small{64};    // calls of at most this many frames take part; larger ranges are evaluated directly
    // frames per block partial (0 = by script: 256 without pair passes, 16 - 128 by selection size with)
    std::atomic<int> readahead_block{0};
    // a call that leaves alone waits this long for another call before it settles the eval (commit + views)
    std::atomic<int> readahead_linger_us{60};
    // the FIRST call of an evaluation waits this long for a second caller before it decides it is alone
    std::atomic<int> readahead_company_us{80};
    // test hook: the block partials' allocation "fails" (the eval must fall back to the combining queue)
    std::atomic<int> readahead_fail_alloc{0};
    // Opt-in: small calls are served by read-ahead even when they come from ONE thread (a host that walks a range frame by frame), and the
    // settle a pool's last leaver performs is DEFERRED to a helper thread that runs once the eval has been quiet for
    // readahead_lone_settle_us. The price is the contract: results then trail the last call by that long (a polling reader like VIAMD's GUI
    // does not notice; vmd_eval_wait_settled / finalize / reduce / the exporters wait for them), and system + trajectory must stay valid
    // until then.
    std::atomic<int> readahead_lone{0};
    std::atomic<int> readahead_lone_settle_us{300};
    // the cell build computes the atom index of a periodic selection instead of reading its index list (round 6; A/B, read at creation)
    std::atomic<int> cells_sel_pattern{1};
    // bucket capacities of the two-level cell build are measured on 4 frames each from the beginning, middle and end of a batch (3,
    // round 6) or from its beginning and end only (2)
    std::atomic<int> cells_cap_sample{3};
    // ... and never below the selection's mean population per pencil x 1.15 (0: measured populations only)
    std::atomic<int> cells_cap_floor{1};
};

extern Options g_opt;

size_t load_threads();


// Where the evaluator is (process-wide, last writer wins): a static string set at every stage of a batch.  Costs one relaxed store; a
// crash handler (tests/native/stress_eval.cpp installs one for SIGABRT / SIGSEGV) can print it when the process dies inside the HIP
// runtime without a message - round 2 saw one such abort and could not say where.
extern std::atomic<const char*> g_stage;

#define VMD_STAGE(text) g_stage.store(text, std::memory_order_relaxed)

// ------------------------------------------------------------------------------------------------ profiling (hipEvents)
struct ProfEntry { double ms = 0.0; uint64_t launches = 0; };

extern std::mutex g_prof_mtx;

extern std::map<std::string, ProfEntry> g_prof;

extern std::atomic<bool> g_prof_on;

// host wall time of a scope, booked under `name` next to the device event times (profiling only: where the eval thread waits)
struct HostTimer {
    const char* name; std::chrono::steady_clock::time_point t0; bool on;
    explicit HostTimer(const char* n) : name(n), on(g_prof_on.load()) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~HostTimer() {
        if (!on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> l(g_prof_mtx);
        g_prof[name].ms += ms; g_prof[name].launches += 1;
    }
};

// ------------------------------------------------------------------------------------------------ resource cache
// VIAMD creates a fresh md_script_eval_t for every script edit and frees the old one (src/main.cpp:966-972, 960).  An eval owns ~60
// device buffers, a dozen pinned blocks, nine streams and twenty events; created and destroyed through the runtime that is
// 1.5 - 3 ms + 3.5 - 6 ms per life cycle (+ 0.7 ms of first-touch allocations inside the first frame_range) against 2.6 ms for
// the whole 10 000-frame SDF evaluation and 8.2 ms for the 100k-atom RDF (profiles/r03ai).  Blocks, streams and events an eval
// gives up are therefore kept, process-wide and per device, and handed to the next eval.
//   * a block is only reused for a request of (nearly) its size: at most 25 % + 1 MB of slack;
//   * a device block that may still be in use by queued work is given back behind a device synchronisation - what hipFree did
//     implicitly; vmd_eval_free synchronises the eval's streams once and releases everything inside a PoolIdle scope instead;
//   * option pool_mb bounds the cached device bytes (pinned: a quarter of it); beyond it, and with pool_mb = 0, blocks go back to
//     the runtime.  An allocation the runtime refuses is retried once after the cache has been emptied.
struct ResourcePool {
    std::mutex mtx;
    std::multimap<size_t, void*> blocks[65];                       // [device 0..63, 64 = pinned host]: bytes -> free block
    std::unordered_map<void*, std::pair<size_t, int>> owner;      // every block that came through the pool: bytes, kind
    size_t pooled[2] = {0, 0};                                     // cached bytes: device, pinned
    std::vector<hipStream_t> streams[64][2];                       // [device][0 = default priority, 1 = highest]
    std::vector<hipEvent_t> events[64][2];                         // [device][0 = with timing, 1 = hipEventDisableTiming]
};

ResourcePool& pool();

extern thread_local int t_pool_idle;

struct PoolIdle { PoolIdle() { ++t_pool_idle; } ~PoolIdle() { --t_pool_idle; } };

static const int kPinned = 64;

int pool_device();

hipError_t pool_raw_alloc(int kind, void** p, size_t bytes);

void pool_raw_free(int kind, void* p);

hipError_t pool_take(int kind, void** out, size_t bytes);

void pool_give(void* p);

hipStream_t pool_stream(bool high_priority);

void pool_stream_give(hipStream_t s, bool high_priority);

hipEvent_t pool_event(bool timing);

void pool_event_give(hipEvent_t e, bool timing);

struct ProfPending { const char* name; hipEvent_t a, b; };

struct Profiler {
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        return pool_event(true);
    }
    void begin(const char* name, hipStream_t s) {
        if (!g_prof_on) return;
        ProfPending p{name, get(), get()};
        if (!p.a || !p.b) return;
        hipEventRecord(p.a, s);
        pending.push_back(p);
    }
    void end(hipStream_t s) {
        if (!g_prof_on || pending.empty()) return;
        hipEventRecord(pending.back().b, s);
    }
    void resolve() {   // call after the stream is synchronised
        if (pending.empty()) return;
        std::lock_guard<std::mutex> l(g_prof_mtx);
        std::vector<ProfPending> later;
        for (auto& p : pending) {
            float ms = 0.0f;
            const hipError_t rc = hipEventElapsedTime(&ms, p.a, p.b);
            // queued on another stream, still running
            if (rc == hipErrorNotReady) { (void)hipGetLastError(); later.push_back(p); continue; }
            if (rc == hipSuccess) { g_prof[p.name].ms += ms; g_prof[p.name].launches += 1; }
            pool.push_back(p.a); pool.push_back(p.b);
        }
        pending.swap(later);
    }
    ~Profiler() { for (auto e : pool) pool_event_give(e, true); for (auto& p : pending) { (void)hipEventSynchronize(p.b);
            pool_event_give(p.a, true); pool_event_give(p.b, true); } }
};

// ------------------------------------------------------------------------------------------------ device buffer helper
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) pool_give(p);
        p = nullptr; cap = 0;
        hipError_t e = pool_take(-1, (void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) return vmd_fail("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        cap = n;
        return true;
    }
    bool upload(const T* src, size_t n, hipStream_t s) {
        if (!ensure(n)) return false;
        if (n) HIP_OK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
        return true;
    }
    void release() { if (p) pool_give(p); p = nullptr; cap = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Host array behind a md_script_property_data_t view.  The views of a volume (8 + 17 MB) are read and written by the copy engine
// after every range: those live in their own pinned allocation (hipHostMalloc) - registering the pages of a heap block
// (hipHostRegister) shares pages with neighbouring blocks, fails when two evals sit next to each other, and leaks the
// registration when only one of two succeeds.
template <typename T>
struct HostBuf {
    T* p = nullptr;
    size_t n = 0;
    bool pinned = false;
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { release(); }
    void release() { if (p) { if (pinned) pool_give(p); else delete[] p; } p = nullptr; n = 0; pinned = false; }
    void assign(size_t count, T v, bool want_pinned = false) {
        release();
        if (count == 0) return;
        if (want_pinned && pool_take(kPinned, (void**)&p, count * sizeof(T)) == hipSuccess) pinned = true;
        else { (void)hipGetLastError(); p = new T[count]; }
        n = count;
        std::fill(p, p + n, v);
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* begin() { return p; }
    T* end() { return p + n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// ------------------------------------------------------------------------------------------------ IR
enum PropKind { PROP_RDF = 0, PROP_SDF = 1, PROP_DIST = 2 };

struct Property {
    std::string name;
    PropKind kind;
    vmd_property_flags_t flags;
    std::vector<int32_t> a, b;      // RDF: ref/target; SDF: structures (K*m)/target; DIST: a/b
    float rmin = 0.0f, rmax = 0.0f; // RDF range; SDF: rmax = cutoff (half extent)
    size_t K = 0, m = 0;
    int dist_kind = 0;
    std::vector<int32_t> aoff, boff;   // DIST: context offsets into a / b (population), size P + 1
};

struct vmd_script_ir_t {
    std::vector<Property> props;
    std::vector<const char*> names;
    mutable std::atomic<uint64_t> fingerprint{0};     // 0 = not computed: hashing a 1M-atom script's index lists takes ~1 ms, and
                                                      // vmd_eval_frame_range compares fingerprints on every call
    void rebuild_names() { names.clear(); for (auto& p : props) names.push_back(p.name.c_str()); fingerprint = 0; }
};

uint64_t fnv1a(uint64_t h, const void* data, size_t n);

bool ir_name_ok(vmd_script_ir_t* ir, const char* name);

bool idx_ok(const int32_t* idx, size_t n, const char* what);

// ------------------------------------------------------------------------------------------------ eval
// a distinct atom selection that needs a cell-sorted copy per frame batch (shared between RDF properties)
struct Selection {
    std::vector<int32_t> idx;
    DevBuf<int32_t> d_idx;
    // the periodic form of idx - atom(t) = first + (t / m) * period + off[t % m] - found once at creation (intern_selection); m = 0: none.
    // The cell-build kernels then compute the index instead of reading the list (vmd_hip_set_cells_sel_pattern)
    int pat_m = 0, pat_first = 0, pat_period = 0, pat_off[4] = {0, 0, 0, 0};
    DevBuf<uint32_t> cell_count, rank, cell_start;
    DevBuf<float> sorted, aos;
    int nsel_pad = 0;
    bool built = false;     // for the current batch ...
    vmd_grid_t built_grid;  // ... on this grid
    // two-level build: bucket capacity per pencil (records), measured on a few frames and kept for the eval's lifetime
    std::vector<uint32_t> pen_off;      // [npen + 1] exclusive prefix; empty = not measured
    int pen_ny = 0, pen_nz = 0;         // the pencil layout the capacities belong to
    int cap_max = 0, total_cap = 0;
    float cap_margin = 1.25f;
    int overflows = 0;                  // times a bucket overflowed; after 3 the selection stays on the single-level builds
    bool used_pencil = false;           // the current batch was built through the buckets
    uint32_t overflow_bit = 1u;         // this selection's bit in the device overflow flag (1 << (index % 32))
    DevBuf<uint32_t> d_pen_off, pen_count, pen_start;
    DevBuf<float> bucket;
    // capacities measured for other pencil layouts: two RDF groups with different cutoffs on one selection alternate between two
    // grids in every batch, and re-measuring costs two launches, a readback and a synchronisation each time
    struct Caps { int ny, nz, cap_max, total_cap; std::vector<uint32_t> pen_off; };
    std::vector<Caps> caps_cache;
};

// One launch of the pair kernel and the histograms it feeds.  Co-evaluated RDF properties of the same range are decomposed
// into disjoint atom classes (by which reference / target sets an atom belongs to): every class pair is evaluated once and added
// to each property that contains it - `goo = rdf(O, O)` and `ghv = rdf(heavy, heavy)` share the O-O pass, which is nearly all
// of ghv (BASELINE config 5).  mult: ordered-pair multiplicity of the pass in the property (same-class pass: the kernel already
// counts both orders; cross pass (c, d): 1 for (c in ref, d in target), +1 for (d in ref, c in target)).
struct PairPass {
    int sel_a = -1, sel_b = -1;
    bool same = false;
    std::vector<std::pair<int, uint64_t>> targets;      // (index into eval->props, mult)
};

struct RdfGroup {
    float rmin = 0.0f, rmax = 0.0f;
    std::vector<int> props;                             // indices into eval->props
    std::vector<PairPass> passes;
    bool classes = false;                               // passes come from the class decomposition
};

float* zero_volume_view(size_t nfloats);

struct PropState {
    Property prop;                      // private copy of the descriptor
    vmd_script_property_data_t data;
    vmd_script_aggregate_t aggregate;
    HostBuf<float> values;              // what data.values points at
    // DIST: temporal rows of frames evaluated ahead, copied into `values` when their block is committed (read-ahead)
    std::vector<float> ahead_values;
    std::vector<float> weights, agg_mean, agg_var, agg_ext;
    HostBuf<uint64_t> counts;           // host mirror of d_counts
    std::vector<double> weights64;
    size_t ncounts = 0;                 // bins or voxels
    size_t dim1 = 0;                    // temporal population
    size_t dist_P = 1, dist_per = 1;    // DIST: contexts x values per context
    DevBuf<uint64_t> d_counts;
    DevBuf<uint64_t> d_blocks;          // [nblocks][ncounts]: per-frame-block partial accumulators (filtered evaluation)
    std::vector<double> block_weights64; // [nblocks][ncounts], distributions only
    DevBuf<float> d_values;             // volume float view (device)
    DevBuf<float> d_max;
    int sel_a = -1, sel_b = -1;         // RDF: indices into eval->sels
    bool same_set = false;
    // SDF
    DevBuf<int32_t> d_structs, d_tgt;
    DevBuf<int8_t> d_owner;
    bool have_owner = false;
    bool unowned = false;               // no target atom belongs to any structure
    int tgt_first = 0, tgt_stride = 0;  // > 0: the target list is the arithmetic progression first + t * stride
    DevBuf<uint8_t> d_tag;              // dense-target path: one tag per atom
    bool have_tag = false;
    size_t tag_len = 0;
    DevBuf<float> d_mass;
    DevBuf<double> d_ref_pose;
    DevBuf<int32_t> d_tree_order, d_tree_parent;     // bond trees of the K structures ([K][m] local indices), when the system carries bonds
    DevBuf<double> d_tree_pos;                       // scratch of the tree walk, [B*K][m][3]
    bool have_tree = false;
    DevBuf<float> d_R32, d_c32, d_group;
    bool ref_pose_ready = false;
    // DIST
    DevBuf<int32_t> d_a, d_b, d_aoff, d_boff;
    DevBuf<float> d_ma, d_mb, d_out;
    bool uploaded = false;
    bool pinned = false;
    bool dirty = false;                 // device accumulators changed since the last host refresh
    bool counts_stale = false;          // volume: host u64 mirror older than the device accumulators
};

// Flags several evals share (checkpoint tables of one compressed trajectory, evaluated by "Eval Full" and "Eval Filt" side by side,
// src/main.cpp:982-1039): "this frame's checkpoints are valid", a frame's signature, "the group records lie".  A flag is raised after the
// stream that wrote the table has been synchronised and is looked at before a launch that reads the table: release / acquire, so that
// the hand-over is defined (and ThreadSanitizer-clean: tests/native/concurrent_evals.cpp).  Two evals that decode the same frame at the
// same time write the same bytes into the table.
template <class T> static inline T flag_get(const T* p) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_ACQUIRE); return v; }

template <class T> static inline void flag_set(T* p, T v) { __atomic_store(p, &v, __ATOMIC_RELEASE); }

// the frame mask is read by the host's GUI thread (vmd_eval_frame_mask_bits, one byte per frame) while pool threads complete frames: bytes go
// through relaxed atomic accesses - plain moves on x86-64 -, so the hand-over is defined behaviour (found by ThreadSanitizer once the
// reference's own polling loop ran against the shim, round 6)
static inline void mask_set(std::vector<uint8_t>& m, size_t f, uint8_t v = 1) { __atomic_store_n(&m[f], v, __ATOMIC_RELAXED); }
static inline uint8_t mask_get(const std::vector<uint8_t>& m, size_t f) { return __atomic_load_n(&m[f], __ATOMIC_RELAXED); }

// Decoder checkpoints of file-backed trajectories (k_xtc_wave), kept per TRAJECTORY for the whole process: VIAMD creates
// a fresh md_script_eval_t for every script edit (src/main.cpp:966-972), so a cache inside the eval would never be hit by the
// re-evaluations it exists for.  Keyed by the trajectory's instance pointer; a frame's checkpoints are only used while the frame's
// signature (stream length, decoder parameters, its first bytes) is the one they were written for - a different file behind a
// recycled pointer can therefore not be entered at a stale bit position.  vmd_ckcache_drop(inst) forgets a trajectory (the native
// readers call it when they close).
struct CkCache {
    size_t frames = 0, atoms = 0;
    int device = -1;
    DevBuf<vmd_xtc_ck_t> ck;
    DevBuf<uint32_t> nck;
    std::vector<uint8_t> have;
    std::vector<uint64_t> sig;
    // group records (vmd_hip.h: vmd_hip_xtc_decode_wave_rec): rec_stride entries per frame, 0 = none (option off, over the budget)
    DevBuf<uint16_t> rec;
    DevBuf<uint32_t> nrec;
    size_t rec_stride = 0;
    bool rec_failed = false;         // a decode from records was rejected: this trajectory goes back to walking its sections
};

size_t record_stride_for(size_t frames, size_t atoms, int level = 1);

extern std::mutex g_ck_mtx;

// keyed by (trajectory instance, device): two devices decoding the same file keep a table each instead of replacing each other's on
// every batch; stages with a decode in flight hold their own reference (Stage::ck_hold), so an eviction never frees what they point into
typedef std::pair<const void*, int> CkKey;

extern std::map<CkKey, std::shared_ptr<CkCache>> g_ck_store;

std::shared_ptr<CkCache> ckcache_for(const void* inst_, size_t frames, size_t atoms, int device);

// ---- checkpoint sidecar.  A first pass over an XTC file walks every bit stream from its first bit (29k c2 frames/s against 84k once
// the decoder checkpoints exist) - and the checkpoints die with the process.  mdlib keeps a frame-offset cache file next to a
// trajectory for the same reason; this is the same idea for the decoder state: 1 KB per frame (64 checkpoints of 16 bytes).  A loaded
// table is only ever a hint: a frame's checkpoints are used while the frame's signature (stream length, decoder parameters, first
// and last bytes) is the one stored with them, and the sectioned decode verifies every section's end state against the next
// checkpoint - a sidecar of another file, or a damaged one, costs a first pass, never a wrong coordinate.
// Group records are not stored (2 bytes per group: 13 - 40 % of the XTC file itself); a trajectory whose checkpoints came from a
// sidecar decodes in sections from them (r03t2: 80.7k against 82.3k frames/s with records).
struct CkFileHeader { char magic[8]; uint32_t version, ck_max; uint64_t frames, atoms; };

static const char kCkMagic[8] = {'V', 'M', 'D', 'X', 'T', 'C', 'C', 'K'};

// Mapped trajectory files (vmd_trajectory_i::raw_mapped_view), pinned for the copy engine in windows of 1 GiB on first use
// (hipHostRegister on the mapping: 5 ms per 512 MB once, then DMA at the rate of hipHostMalloc memory - profiles/r03c_hostio.txt).
// Process-wide, keyed by the mapping's base; the reader that owns the mapping calls vmd_mapreg_drop before it unmaps.  A window
// that cannot be pinned (limit reached, the driver refuses) stays unpinned: its batches take the load_raw copy instead.
struct MapReg {
    size_t bytes = 0;
    std::vector<uint8_t> state;              // per window: 0 = not tried, 1 = pinned, 2 = refused
};

extern std::mutex g_map_mtx;

extern std::map<const unsigned char*, MapReg> g_map_store;

extern size_t g_map_pinned;

static const size_t kMapWindow = (size_t)1 << 30;

void mapreg_release(const unsigned char* base, MapReg& m);

bool mapreg_pin(const unsigned char* base, size_t bytes, size_t lo, size_t hi);

uint64_t frame_signature(const vmd_xtc_frame_t& fi, const unsigned char* bytes);

// one pending vmd_eval_frame_range call (lives on the caller's stack)
struct RangeRequest {
    uint32_t beg = 0, end = 0;
    const vmd_system_t* sys = nullptr;
    vmd_trajectory_i* traj = nullptr;
    bool done = false, ok = true;
    std::string error;
};

// Which trajectory an interface stands for: its instance pointer AND its frame source.  Hosts whose callbacks close over their state
// (ctypes, lambdas, file-static readers) pass inst = NULL for every trajectory; their callbacks differ.
struct TrajId {
    const void* inst = nullptr;
    const void* fn = nullptr;
    bool operator==(const TrajId& o) const { return inst == o.inst && fn == o.fn; }
    bool operator!=(const TrajId& o) const { return !(*this == o); }
};

TrajId traj_id(const vmd_trajectory_i* t);

struct vmd_script_eval_t {
    uint64_t ir_fingerprint = 0;
    size_t num_frames = 0;
    std::vector<uint8_t> frame_mask;
    std::atomic<size_t> frames_done{0};
    std::atomic<bool> interrupt{false};
    std::mutex mtx;                                   // serialises device work (frame ranges, finalize, clear, vis payloads)
    std::mutex queue_mtx;                             // combining queue of concurrent frame_range calls
    std::condition_variable queue_cv;
    std::vector<RangeRequest*> queue;
    bool leader_active = false;
    size_t last_round = 0;                            // requests the leader served in its previous round: how many callers to expect back
    long last_round_us = 0;                           // ... and how long that round took to evaluate
    std::chrono::steady_clock::time_point views_at{}; // when the host views were last brought up to date (lazy_views)
    std::vector<std::unique_ptr<PropState>> props;
    std::vector<std::unique_ptr<Selection>> sels;
    hipStream_t stream = nullptr;
    int device = 0;
    Profiler prof;
    Profiler prof_copy;                               // events on copy_stream (device decode of staged batches): resolved by settle_stage
    // batch scratch.  Two stages: while the kernels of batch k run, the host loads batch k+1 through load_frame into
    // the other pinned buffer and its H2D copy runs on copy_stream (trajectory staging).
    struct Stage {
        float* h = nullptr; size_t hcap = 0;     // pinned host frames [nb][3][npad]
        DevBuf<float> d;                         // their device copy
        // compressed frames for the device decoder (load_raw): pinned bit streams + per-frame records, device copies, status
        unsigned char* hraw = nullptr; size_t hraw_cap = 0;
        DevBuf<unsigned char> d_raw;
        std::vector<vmd_xtc_frame_t> raw_info;
        DevBuf<vmd_xtc_frame_t> d_raw_info;
        DevBuf<uint32_t> d_raw_status;
        DevBuf<uint64_t> d_raw_scratch;          // checkpoints of the two-pass decoder
        uint32_t* h_raw_status = nullptr; size_t h_raw_status_cap = 0;
        bool raw_pending = false;                // a device decode is queued behind `ready`: its status words are checked before use
        bool sectioned = false;                  // that decode ran from checkpoints (sections), not from bit 0
        bool* rec_failed = nullptr;              // that decode placed its groups from records: where to note that they were rejected
        // that decode also writes the frames' checkpoints: mark them valid (ck_mark[0 .. nb)) when it succeeded
        uint8_t* ck_mark = nullptr;
        // that decode entered the frames at their checkpoints: forget them (ck_clear[0 .. nb)) when it was rejected
        uint8_t* ck_clear = nullptr;
        std::shared_ptr<CkCache> ck_hold;        // the table those three point into, for as long as the decode is pending
        DevBuf<float> d_boxes;
        std::vector<float> h_boxes;              // [nb][6]: L, 1/L
        std::vector<vmd_unitcell_t> cells;
        // batches with open (non-periodic) axes: per-frame bounding box -> the boxes the pencil grid uses (extent, 1/extent, origin)
        DevBuf<float> d_bbox, d_gboxes;
        std::vector<float> h_bbox, h_gboxes;
        bool gboxes_ready = false;
        hipEvent_t ready = nullptr;
        const float* base = nullptr; size_t frame_stride = 0, row_stride = 0;   // where the kernels read the batch
        size_t f0 = 0, nb = 0;
        // device views: the range whose cells / boxes this stage holds (host vectors and d_boxes), 0 = none
        const vmd_unitcell_t* boxes_cells = nullptr; size_t boxes_f0 = 0, boxes_nb = 0; uint64_t boxes_version = 0;
    };
    static constexpr size_t kDecodeStreams = 4;
    Stage stages[kDecodeStreams + 1];        // batch k evaluated, k + 1 (first pass of a compressed file: up to k + 4) being staged
    // compressed batches on their way to the device decoder, a ring of three: while the pair kernels of batch k run, batch k + 1 is
    // being decompressed (decode_stream) and the bit streams of batch k + 2 cross PCIe (copy_stream) - three engines, three batches
    struct RawSlot {
        unsigned char* h = nullptr; size_t hcap = 0;     // pinned bit streams
        DevBuf<unsigned char> d;                         // [frame table, info_bytes][bit streams]: one DMA per batch
        std::vector<vmd_xtc_frame_t> info;
        uint32_t codec = VMD_RAW_CODEC_XTC;              // what the slot holds: XTC bit streams (info) or plain floats (f32)
        std::vector<vmd_f32_frame_t> f32;
        size_t info_bytes = 0;
        const vmd_xtc_frame_t* d_info() const { return (const vmd_xtc_frame_t*)d.p; }
        const unsigned char* d_streams() const { return d.p + info_bytes; }
        const unsigned char* h_streams = nullptr;        // host: where info[b].offset counts from (the pinned block, or the mapped file)
        std::vector<vmd_unitcell_t> cells;
        hipEvent_t uploaded = nullptr;
        size_t f0 = 0, nb = 0;
        int state = 0;                                   // 1 = [f0, f0 + nb) uploaded (event recorded), 0 = nothing, -1 = not available raw
    };
    static constexpr size_t kRawSlots = 6;
    RawSlot raw_slots[kRawSlots];
    std::shared_ptr<CkCache> ck_cache;       // the decoder checkpoints of the trajectory being evaluated (process-wide store)
    std::atomic<size_t> frames_section_decoded{0};
    std::atomic<size_t> frames_mapped{0};
    bool raw_skip = false;                   // the range being evaluated does not use the raw ring (set by the leader in evaluate_range)
    hipStream_t decode_stream = nullptr;     // = decode_streams[0]
    // A first pass over compressed frames walks every bit stream from its start: one dependent chain per frame, ~7 ms for a c2 frame
    // however many frames the launch holds.  Several batches walk side by side on their own streams while their successors cross PCIe.
    hipStream_t decode_streams[kDecodeStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipStream_t aux_stream = nullptr;        // background work nothing else queues behind (the clearing DMA of a volume's host view)
    DevBuf<uint64_t> d_partial;
    DevBuf<uint64_t> d_partial2;             // partial rows of the pair launches on pair_stream (batches of frame blocks)
    hipStream_t pair_stream = nullptr;       // every other block of a batch of frame blocks runs its pair kernel here
    hipEvent_t pair_fork = nullptr, pair_join = nullptr;
    std::vector<RdfGroup> rdf_groups;
    // [passes of the batch][bins]: scratch histogram of every pair pass, committed at the batch's end
    DevBuf<uint64_t> d_pass;
    DevBuf<uint32_t> d_overflow;             // device flag raised by the two-level cell build when a pencil bucket is full
    uint32_t* h_overflow = nullptr;          // pinned host copies (one per batch in flight), read where a batch is completed
    hipEvent_t batch_done[2] = {nullptr, nullptr};   // end of a queued batch (deferred completion: process_range)
    uint64_t* h_snap = nullptr; size_t h_snap_cap = 0;   // pinned: the RDF counts behind the commits of the two batches in flight
    std::vector<double> w_snap;              // ... and the weights that go with them
    std::vector<float> h_temporal_slot[2];
    DevBuf<uint32_t> d_pen_sample;
    // filtered evaluation: per-block partial accumulators and the eval whose blocks this one may reuse
    size_t block_frames = 0;
    std::unique_ptr<std::atomic<uint8_t>[]> block_ready;
    size_t num_blocks = 0;
    vmd_script_eval_t* source = nullptr;
    // the trajectory instance this eval's block partials were evaluated from: a user of this eval as a SOURCE takes blocks only while it is
    // itself evaluating the same instance (two evals of one script over different trajectories of equal length must not trade
    // blocks)
    TrajId blocks_inst;
    std::atomic<bool> defer_volume_views{false};   // vmd_eval_defer_volume_views
    std::atomic<size_t> frames_computed{0}, frames_reused{0}, frames_device_decoded{0};
    // ---- read-ahead.  Block states move NONE -> PENDING -> READY under queue_mtx (the region leader), READY -> COMMITTED /
    // DIRECT and NONE -> DIRECT under queue_mtx + mtx (settle / the direct path); the fast path of a call only READS a state and sets
    // frame_req of its frames (compare-exchange: a frame is requested once).
    enum : uint8_t { RA_NONE = 0, RA_PENDING = 1, RA_READY = 2, RA_COMMITTED = 3, RA_DIRECT = 4 };
    struct ReadAhead {
        std::atomic<bool> on{false};                 // states allocated, block partials exist: small calls take the read-ahead path
        bool own_blocks = false;                     // block_frames was set by read-ahead itself (not by vmd_eval_set_block_frames)
        std::unique_ptr<std::atomic<uint8_t>[]> blk_state;
        // requested by a call (committed or not).  Sixteen pool threads mark sixteen consecutive frames at the same instant: frame f lives
        // at slot (f % 64) * req_stride + f / 64, so neighbours in time are at least a cache line apart
        std::unique_ptr<std::atomic<uint8_t>[]> frame_req;
        size_t req_stride = 64;
        std::atomic<uint8_t>& req(size_t f) const { return frame_req[(f & 63) * req_stride + (f >> 6)]; }
        // calls inside vmd_eval_frame_range (low half) and calls that ever arrived (high half) in ONE word: a call costs this line one
        // read-modify-write when it enters and one when it leaves - with 16 threads and 10 000 one-frame calls every further shared
        // counter on the path showed up in the total (r04b: 12.4 ms for the 10 000-frame SDF, 9.4 at grain 64)
        alignas(64) std::atomic<uint64_t> flight{0};
        alignas(64) std::atomic<bool> marks_pending{false};   // frames were marked since the last full settle began
        std::atomic<bool> views_dirty{false};
        std::atomic<bool> concurrent{false};         // this evaluation (since clear_data) has seen two calls at once: it is a pool
        bool spec_active = false;                    // a region is being evaluated (queue_mtx)
        bool lonely = false;                         // a first call has waited for company in vain (queue_mtx)
        // settles keep finding partly requested blocks (three strikes): the callers do not arrive the way read-ahead assumes
        std::atomic<bool> disabled{false};
        int strikes = 0;
        size_t next_region = 0;                      // frames of the next region
        bool failed = false; std::string error;      // a region failed: every waiting call reports it
        std::mutex settle_mtx;                       // one settle at a time
        int combining = 0;                           // calls inside the combining queue that entered before the states existed (queue_mtx)
        size_t bmax = 0;                             // frames of one kernel batch for this eval and trajectory
        TrajId traj_inst;                            // the trajectory the regions are evaluated from
        // statistics (vmd_eval_readahead_stats)
        std::atomic<uint64_t> regions{0}, region_frames{0}, slow_calls{0}, settles{0}, direct_frames{0}, committed_blocks{0};
        // deferred settle (option readahead_lone): decided per evaluation at its first small call
        std::atomic<bool> lone{false};
        // vmd_eval_set_deferred_settle: -1 = the process-wide option readahead_lone, 0 / 1 = this eval's own choice
        std::atomic<int> lone_pref{-1};
        struct Helper {
            std::thread th;
            std::mutex mtx;
            std::condition_variable cv, idle_cv;
            bool started = false, quit = false, busy = false, have = false;      // (mtx)
            // (mtx) bumped by every cancel: a settle that was running then does not re-arm itself
            uint64_t cancel_seq = 0;
            std::atomic<bool> armed{false};                 // a settle is owed once the eval has been quiet long enough
            std::atomic<int64_t> last_leave_ns{0};          // when the last call left (steady clock)
            std::atomic<uint64_t> settles{0};
            vmd_system_t sys; vmd_trajectory_i traj;        // (mtx) copies of the caller's records: what the deferred settle evaluates from
            // vmd_eval_set_settled_callback: told after every settle the helper (or vmd_eval_wait_settled) has performed, without any lock
            // of the eval held.  Written before the evaluation's calls (like lone_pref), read by the helper: atomics, not a lock
            std::atomic<void (*)(void*)> on_settled{nullptr};
            std::atomic<void*> on_settled_user{nullptr};
        } helper;
    } ra;
    vmd_reduce_stats_t reduce_stats = {};
    // fixed at creation
    struct Spec { bool rdf_closed = false, sdf_include_self = false, sdf_density = false, dist_geometric_com = false, rdf_raw = false;
            int rdf_norm = 0; } spec;
    size_t atoms_checked = (size_t)-1;       // trajectory atom count the properties' indices were validated against (under mtx)
};

typedef vmd_script_eval_t::Stage Stage;

typedef vmd_script_eval_t::RawSlot RawSlot;

PropState* find_prop(const vmd_script_eval_t* e, const char* name);

int intern_selection(vmd_script_eval_t* e, const std::vector<int32_t>& idx);

void build_rdf_plan(vmd_script_eval_t* e);

void lone_stop(vmd_script_eval_t* e);

// Published scalars.  VIAMD's GUI thread reads a property's record while pool threads are inside frame_range (src/main.cpp:1508-1524):
// by the reference's contract a reader may see old and new fields side by side, never a crash.  The scalar fields are therefore
// written with relaxed atomic stores - plain moves on x86-64 - so that the contract is also what the C++ memory model and
// ThreadSanitizer (scripts/tsan_emu.sh) see; the shim's refresh() loads them the same way.  The arrays behind `values` / `weights`
// are written by DMA, memcpy and fills: a reader of those runs under the reference's "torn data is tolerated" rule only.
template <class T> static inline void pub(T& dst, T v) { __atomic_store(&dst, &v, __ATOMIC_RELAXED); }

static inline void pub_touch(uint64_t& fingerprint) { uint64_t v; __atomic_load(&fingerprint, &v, __ATOMIC_RELAXED); v += 1;
        __atomic_store(&fingerprint, &v, __ATOMIC_RELAXED); }

void ra_reset(vmd_script_eval_t* e);

void lone_cancel(vmd_script_eval_t* e);

void refresh_distribution_from(PropState* p, const uint64_t* counts, const double* weights64);

bool refresh_distribution(vmd_script_eval_t* e, PropState* p);

bool refresh_volume(vmd_script_eval_t* e, PropState* p);

void refresh_temporal_stats(vmd_script_eval_t* e, PropState* p);

extern "C" bool vmd_eval_wait_settled(vmd_script_eval_t* eval);

bool upload_static(vmd_script_eval_t* e, const vmd_system_t* sys, size_t traj_atoms);

bool check_atoms(vmd_script_eval_t* e, size_t num_atoms);

struct BatchSrc {
    const float* base = nullptr;   // device
    size_t frame_stride = 0, row_stride = 0;
};

int launch_raw_decode(vmd_script_eval_t* e, Stage& st, const unsigned char* d_raw, const vmd_xtc_frame_t* d_info, size_t num_atoms,
                             size_t nb, size_t npad, hipStream_t stream, vmd_xtc_ck_t* ck = nullptr, uint32_t* nck = nullptr,
                             uint8_t* ck_have = nullptr, uint16_t* rec = nullptr, uint32_t* nrec = nullptr, size_t rec_stride = 0,
                             bool* rec_failed = nullptr);

int raw_upload_f32(vmd_script_eval_t* e, vmd_script_eval_t::RawSlot& rs, vmd_trajectory_i* traj, const std::vector<vmd_raw_frame_t>& infos,
                          size_t num_atoms, size_t f0, size_t nb);

int raw_upload(vmd_script_eval_t* e, RawSlot& rs, vmd_trajectory_i* traj, size_t num_atoms, size_t f0, size_t nb);

bool fetch_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, bool force_host = false, RawSlot* pre = nullptr);

bool settle_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, size_t num_atoms);

bool fetch_batch(vmd_script_eval_t* e, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, BatchSrc* src);

uint32_t batch_pbc(const Stage& st);

bool prepare_open_boxes(vmd_script_eval_t* e, Stage& st, size_t nb, uint32_t pbc, size_t num_atoms);

bool choose_grid(const std::vector<float>& boxes, uint32_t pbc, size_t nb, float rmax, vmd_grid_t* g, bool dense_lanes = false);

bool ensure_pencil_caps(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb,
        const vmd_grid_t& g);

bool build_selection(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb,
        const vmd_grid_t& g);

size_t auto_batch(const vmd_script_eval_t* e, size_t num_atoms, bool staged);

// one kernel batch: frames [f0, f0 + nb); blk >= 0 when the batch is made of the whole frame blocks blk .. blk + nblk - 1 of this eval
// (filtered evaluation: each block accumulates into its own partial; they share the batch's cell build and synchronisation)
struct Batch { size_t f0, nb; long blk; size_t nblk; };

void plan_batches(const vmd_script_eval_t* e, size_t beg, size_t end, size_t Bmax, std::vector<Batch>* out);

// filtered evaluation: merge every ready block of the source eval that lies inside [beg, end) into this eval's accumulators and return the
// sub-ranges that still have to be computed.
// block_ready[b] != 0: block b's partial (d_blocks, block_weights64, temporal rows) is complete.
// Where its temporal rows are: a block evaluated by a plain call has them in `values`; a block evaluated AHEAD (read-ahead, spec) or
// adopted from a source has them in the side buffer `ahead_values` until it is committed - `values` only ever shows frames somebody asked
// for.  An eval that takes blocks from a source (reuse_blocks, ra_adopt_blocks) must read the rows where they are: a filtered evaluation
// running BESIDE its source (src/main.cpp:982-1039 enqueues both) used to copy rows of blocks the source had evaluated ahead but not yet
// committed out of `values` - zeros (tests/native/stress_readahead.cpp, "beside").
enum : uint8_t { BLOCK_ROWS_IN_PLACE = 1, BLOCK_ROWS_AHEAD = 2 };

const float* block_rows(const vmd_script_eval_t* src, const PropState* q, size_t blk);

bool reuse_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t beg, size_t end, std::vector<std::pair<size_t, size_t>>* todo);

bool view_sharded(const vmd_device_view_t& view);

bool view_holds(bool have_view, const vmd_device_view_t& view, size_t frame);

bool process_range_locked(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end,
        bool views, bool spec);

bool process_range(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end,
        bool views = true);

bool refresh_views_locked(vmd_script_eval_t* e);

bool refresh_views(vmd_script_eval_t* e);

bool combine_call(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end);

// ---------------------------------------------------------------------------------------------------------------------------------
// Read-ahead: VIAMD's own call pattern at the one-call rate, without editing VIAMD.
//
// VIAMD evaluates a script as a pool task over [0, num_frames) with grain 1 (src/main.cpp:993-997, src/task_system.cpp:73-81): every pool
// thread calls md_script_eval_frame_range with a frame or a few and blocks until they are evaluated, and the results must be final when
// the last call returns - which no call knows to be.  A round of the combining queue above can therefore never hold more than
// threads x grain frames (27 ms instead of 8 for the 100k-atom RDF, 130 instead of 7 for the 10 000-frame SDF).
//
// Here the first small call that needs a frame nobody has evaluated becomes the leader of a REGION: a run of whole frame blocks starting
// at its frame (128 frames at first, four times as many each time, up to one kernel batch), evaluated in one go into the blocks' partial
// accumulators (the filtered-evaluation machinery: process_range(spec)).  Nothing of a region is visible in the results.  A call whose
// frames lie in an evaluated region only marks them REQUESTED (a compare-exchange per frame, no lock, no device work) and returns.  A block
// joins the totals when every one of its frames has been requested - one k_add_u64 per accumulator, by the next region's leader or by the
// settle below - so frames nobody asked for are never counted: a filtered range that ends inside a block, or an interrupt, leaves
// the rest of the region unused.
//
// "Final when the last call returns": a call that leaves while no other call is inside the function waits a moment (readahead_linger_us)
// for the next one to arrive - the pool threads of a running task come back within a microsecond - and if nobody comes it SETTLES the
// eval before it returns: whole requested blocks are committed, requested frames of partly requested blocks are evaluated directly
// (those blocks stay direct from then on), the host views are brought up to date.  Whoever returns last has either settled or handed
// that duty to a call that arrived later.
//
// Callers that do not arrive like a pool (one thread calling frame by frame: every call is "the last") are recognised - the first call of
// an evaluation waits readahead_company_us for a second caller - and served by the combining queue as before; so are large ranges, evals
// with a source (filtered evaluation out of another eval's blocks) and evals whose settles keep finding partly requested blocks.
typedef vmd_script_eval_t::ReadAhead ReadAhead;

void ra_reset(vmd_script_eval_t* e);

size_t ra_block_frames(const vmd_script_eval_t* e, size_t Bmax);

bool ra_engage(vmd_script_eval_t* e, vmd_trajectory_i* traj);

bool ra_adopt_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t b0, size_t b1, std::vector<char>* adopted);

bool ra_commit_block(vmd_script_eval_t* e, size_t blk);

bool ra_settle(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, bool full);

bool ra_fast(vmd_script_eval_t* e, uint32_t beg, uint32_t end);

bool ra_direct_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end);

// A bounded wait on a condition variable.  libstdc++ waits on the steady clock through pthread_cond_clockwait, which the
// ThreadSanitizer runtime of this toolchain does not intercept (it then believes the waiter kept the mutex): instrumented
// builds wait on the system clock (pthread_cond_timedwait) so that the TSan runs of scripts/tsan_emu.sh see every hand-over.
template <class Pred>
static bool cv_wait_us(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, int us, Pred pred) {
#if defined(__SANITIZE_THREAD__)
    return cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us), pred);
#else
    return cv.wait_for(lk, std::chrono::microseconds(us), pred);
#endif
}

bool ra_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end);

int64_t steady_ns();

void lone_helper_main(vmd_script_eval_t* e);

void lone_arm(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj);

void lone_cancel(vmd_script_eval_t* e);

void lone_stop(vmd_script_eval_t* e);

bool ra_leave(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj);

uint64_t next_cells_version();

struct vmd_devtraj_t {
    size_t num_frames = 0, num_atoms = 0, npad = 0;
    size_t first = 0, resident = 0;     // frames [first, first + resident) are in HBM (a rank's shard; the whole trajectory otherwise)
    float* d = nullptr;                 // frame `first`
    // shards that do not start at frame 0 keep a copy of it: the SDF reference pose is taken there (SPEC S5)
    float* d0 = nullptr;
    bool has(size_t beg, size_t end) const { return (beg >= first && end <= first + resident && beg <= end) || (d0 && beg == 0
            && end == 1); }
    float* frame(size_t f) const { return (d0 && f == 0) ? d0 : d + (f - first) * 3 * npad; }
    int device = 0;
    std::vector<vmd_unitcell_t> cells;
    uint64_t cells_version = next_cells_version();   // a new process-wide number for every change of `cells` or of the coordinates: two
                                                     // trajectories (one freed, one created at the same address) never share one (ADVICE
                                                     // r02)
    vmd_trajectory_i iface;
};

size_t dt_num_frames(void* inst);

size_t dt_num_atoms(void* inst);

bool dt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z);

bool dt_device_view(void* inst, vmd_device_view_t* out);

// ------------------------------------------------------------------------------------------------ compressed trajectory in HBM
// Every frame of a trajectory that offers load_raw (today: XTC), still compressed, in ONE device allocation + the decoder records
// as a device array.  Evaluations decode batches straight from it (fetch_stage: raw_device_view), so a file-backed trajectory is
// read and crosses PCIe once, not once per evaluation - VIAMD keeps a host-side cache of decoded frames for the same reason
// (/root/reference/src/loader.cpp:111-159); here the cache is the compressed stream and it lives next to the kernels.
struct vmd_rawtraj_t {
    vmd_trajectory_i* src = nullptr;     // borrowed: must outlive this object (load_frame of single frames, fallbacks)
    size_t num_frames = 0, num_atoms = 0, bytes = 0;
    int device = 0;
    unsigned char* d_raw = nullptr;
    vmd_xtc_frame_t* d_info = nullptr;
    vmd_xtc_ck_t* d_ck = nullptr;        // decoder checkpoints, filled by the first evaluation of each frame
    uint32_t* d_nck = nullptr;
    std::vector<uint8_t> ck_have;
    uint16_t* d_rec = nullptr;           // group records, written next to the checkpoints (rec_stride entries per frame; 0 = none)
    uint32_t* d_nrec = nullptr;
    size_t rec_stride = 0;
    bool rec_failed = false;
    std::vector<vmd_unitcell_t> cells;
    vmd_trajectory_i iface;
};

size_t rt_num_frames(void* inst);

size_t rt_num_atoms(void* inst);

bool rt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z);

bool rt_load_raw(void* inst, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst, size_t cap);

bool rt_raw_device_view(void* inst, vmd_raw_device_view_t* out);

// ------------------------------------------------------------------------------------------------ host trajectory (pinned)
struct vmd_hosttraj_t {
    size_t num_frames = 0, num_atoms = 0, npad = 0;
    float* h = nullptr;
    std::vector<vmd_unitcell_t> cells;
    vmd_trajectory_i iface;
};

size_t ht_num_frames(void* inst);

size_t ht_num_atoms(void* inst);

bool ht_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z);

bool ht_host_view(void* inst, vmd_host_view_t* out);
