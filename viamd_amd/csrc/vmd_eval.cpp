// viamd_amd/csrc/vmd_eval.cpp — C++ host side of the drop-in boundary (include/vmd_eval.h).
//
// Mirrors the md_script_eval_* lifecycle VIAMD drives (/root/reference/src/main.cpp:951-1039): create ->
// clear_data -> frame_range from pool threads -> property_data / frame_mask polled by the GUI thread.
// All arithmetic happens in the HIP kernels of vmd_kernels.hip; this file only batches frames, owns the
// device buffers and keeps the md_script_property_data_t views up to date.  There is no CPU compute path.
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

static thread_local std::string g_last_error;

// md_log_register analogue (VIAMD installs a logger that turns messages into toasts, src/main.cpp:384-420): failures go to
// the registered callback, or to stderr when there is none.  The callback may be invoked from any thread that calls the API.
static std::mutex g_log_mtx;
static vmd_log_fn g_log_fn = nullptr;
static void* g_log_user = nullptr;
extern "C" void vmd_log_register(vmd_log_fn fn, void* user) {
    std::lock_guard<std::mutex> l(g_log_mtx);
    g_log_fn = fn;
    g_log_user = user;
}
static void vmd_log(int level, const char* msg) {
    vmd_log_fn fn;
    void* user;
    {
        std::lock_guard<std::mutex> l(g_log_mtx);
        fn = g_log_fn; user = g_log_user;
    }
    if (fn) fn(level, msg, user);
    else fprintf(stderr, "[viamd_amd] %s: %s\n", level >= VMD_LOG_ERROR ? "error" : "info", msg);
}

// a host-side layer (include/vmd_md_script_shim.h) reports through the same channel as the library
extern "C" void vmd_log_message(int level, const char* message) { if (message) vmd_log(level, message); }

static bool vmd_fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    vmd_log(VMD_LOG_ERROR, buf);
    return false;
}

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
    // pencils of cross-section rmax/split (walk reach = split).  z: explicit (A/B).  y: 0 = by density - selections of >= 0.08 atoms / A^3 in the
    // lanes of every pass of a group (all heavy atoms of a liquid; SURVEY 8d's C3-dense) walk half-width pencils in y: a 64-atom chunk of such
    // a selection is only ~4 A long, so the x windows are dominated by the 2 r_max of padding and thinner pencils pay (c3d 2 028 -> 2 131
    // frames/s, profiles/r05d_pencil_split_by_density.txt; at c3's 0.033 / A^3 the same split costs 10 %, at c5's mix 6 %); 1 / 2 / .. = fixed
    std::atomic<int> pencil_split_y{0}, pencil_split_z{1};
    std::atomic<int> nxf_divisor{16};    // fine x cell = rmax / nxf_divisor (8 / 12 / 16 / 24 / 32 measured: 16 is +0.8 % on c3, profiles/r02l_ab_fine_cells.txt)
    std::atomic<int> cells_aos{1};       // sort through 16-byte records + repack
    std::atomic<int> xtc_device_decode{3};   // frames offered raw (load_raw) are decompressed on the device (0 = on the host threads): 1 = one thread per
                                             // frame (k_xtc_decode), 2 = index pass + one thread per chunk (k_xtc_index / k_xtc_chunks),
                                             // 3 = one wave per frame (k_xtc_wave)
    std::atomic<int> xtc_chunk{256};         // atoms per chunk of variant 2
    std::atomic<int> pool_mb{16384};         // process-wide cache of device blocks freed by evals (MB; pinned host blocks: a quarter of it); 0 = off
    std::atomic<int> gather_us{150};         // combining queue: how long the leader waits for the other pool threads of the previous round to come back with their next ranges (0 = take what is there)
    std::atomic<int> lazy_views{1};          // combining queue: the host views are refreshed when no call is waiting (and every lazy_views_ms at the latest), not after every batch
    std::atomic<int> lazy_views_ms{20};
    std::atomic<int> defer_sync{0};          // the next batch is queued before the host waits for the current one (evals without block partials); measured r03ad: no gain (the per-batch host gap is ~0.06 ms; the next decode then lands on the cell build), off
    std::atomic<int> block_superbatch{1};    // filtered evaluation: consecutive frame blocks share ONE batch (one cell build, one synchronisation; a pair launch per block)
    std::atomic<int> block_two_streams{1};   // ... and the blocks' pair launches alternate between two streams, so that the tail of one runs under the head of the next
    std::atomic<int> xtc_ramp{0};            // file-backed device decode: small first and last batches (pipeline fill / drain); r03o: no gain, off
    std::atomic<int> xtc_decode_ahead{1};    // batches the device decoder runs ahead of the kernels (1 or 2); r03n: 2 changes nothing
    std::atomic<int> raw_f32_device{1};      // TRR / DCD: frames DMA'd out of the mapped file, swapped / scaled / transposed by k_raw_f32 (0: host threads)
    std::atomic<int> xtc_cold_streams{1};    // first pass out of a mapped file: up to four batches walked side by side on their own streams
    std::atomic<int> xtc_mapped{1};          // variant 3: DMA the compressed frames straight out of the mapped file (raw_mapped_view), no host copy
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
    std::atomic<int> spec_rdf_raw{0};             // D-WRAP: positions enter rdf() as they are, minimum image by rounding - evaluated by k_rdf_brute (all pairs: a
                                                  // setting for matching an mdlib that does it this way, not a fast path)
    std::atomic<int> spec_rdf_norm{0};            // D-RDF-NORM: 0 = cell volume when fully periodic, else the cutoff sphere; 1 = always the cutoff sphere;
                                                  // 2 = per reference atom (the weights do not carry N_ref)
    std::atomic<int> sdf_direct_view{1};          // k_counts_to_float writes the volume's float view into its pinned host pages itself
    std::atomic<int> stage_frames{128};      // frames per staged batch of a host / file trajectory (batch_frames <= 0)
    std::atomic<int> rdf_blocks_decode{1536};   // pair-kernel grid while batches are decompressed on the device: 6 blocks per CU leave every
                                                // SIMD a wave slot and 80 VGPRs, so k_xtc_wave of batch k + 1 (wave priority 3) runs under the pair
                                                // kernel of batch k; costs the pair kernel ~4 % (0 = leave the grid alone)
    std::atomic<int> sdf_dense{0};       // dense-target SDF scatter (stream whole frames, select by tag): measured slower, off
    std::atomic<int> sdf_arith{1};       // SDF target lists that are arithmetic progressions are generated on the device (0 = always load the index list)
    std::atomic<int> cells_small{8192};  // selections of at most this many atoms are sorted by one block per frame (k_cells_fused), never through pencil buckets (0: buckets for everyone)
    std::atomic<int> rdf_classes{1};     // co-evaluated RDFs of one range share pair passes through disjoint atom classes (0 = one pass per property)
    // read-ahead under VIAMD's call pattern (many pool threads, ranges of a frame or a few): the first small call that finds
    // company evaluates a whole REGION of frame blocks ahead into block partials, later calls for those frames only mark them requested
    std::atomic<int> readahead{1};           // 0 = every call is evaluated when it arrives (the combining queue of round 3)
    std::atomic<int> readahead_frames{128};  // frames of the first region of an evaluation (rounded to whole blocks) ...
    std::atomic<int> readahead_growth{4};    // ... every further region is this many times larger (up to one kernel batch)
    std::atomic<int> readahead_small{64};    // calls of at most this many frames take part; larger ranges are evaluated directly
    std::atomic<int> readahead_block{0};     // frames per block partial (0 = by script: 256 without pair passes, 16 - 128 by selection size with)
    std::atomic<int> readahead_linger_us{60};// a call that leaves alone waits this long for another call before it settles the eval (commit + views)
    std::atomic<int> readahead_company_us{80};// the FIRST call of an evaluation waits this long for a second caller before it decides it is alone
    std::atomic<int> readahead_fail_alloc{0}; // test hook: the block partials' allocation "fails" (the eval must fall back to the combining queue)
    // Opt-in: small calls are served by read-ahead even when they come from ONE thread (a host that walks a range frame by frame), and the
    // settle a pool's last leaver performs is DEFERRED to a helper thread that runs once the eval has been quiet for readahead_lone_settle_us.
    // The price is the contract: results then trail the last call by that long (a polling reader like VIAMD's GUI does not notice;
    // vmd_eval_wait_settled / finalize / reduce / the exporters wait for them), and system + trajectory must stay valid until then.
    std::atomic<int> readahead_lone{0};
    std::atomic<int> readahead_lone_settle_us{300};
};
static Options g_opt;

// One process per GPU and 8 GPUs per node share the host: an eighth of the hardware threads, at least 8 (memcpy-bound DCD
// frames saturate there), at most 32 (compressed XTC frames scale further).
static size_t load_threads() {
    const int v = g_opt.load_threads.load();
    if (v > 0) return (size_t)v;
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min<size_t>(32, std::max<size_t>(8, hw / 8));
}

// k_rdf_pencil can deal a chunk's neighbour pencils to separate work items in small launches (vmd_hip_set_rdf_nsplit): measured
// r03ar - a one-frame launch gains (170 -> 133 us), a four-frame launch loses (194 -> 252 us) - so it is switched off when the library
// loads (here, not in vmd_kernels.hip: that file is byte for byte what the committed PMC passes were collected on)
static const int g_rdf_nsplit_at_load = vmd_hip_set_rdf_nsplit(0);

extern "C" int vmd_set_option(const char* key, int value) {
    std::atomic<int>* o = nullptr;
    if (!strcmp(key, "rdf_variant")) o = &g_opt.rdf_variant;
    else if (!strcmp(key, "batch_frames")) o = &g_opt.batch_frames;
    else if (!strcmp(key, "force_brute")) o = &g_opt.force_brute;
    else if (!strcmp(key, "load_threads")) o = &g_opt.load_threads;
    else if (!strcmp(key, "nxf_divisor")) o = &g_opt.nxf_divisor;
    else if (!strcmp(key, "pencil_split_y")) o = &g_opt.pencil_split_y;
    else if (!strcmp(key, "pencil_split_z")) o = &g_opt.pencil_split_z;
    else if (!strcmp(key, "cells_aos")) o = &g_opt.cells_aos;
    else if (!strcmp(key, "sdf_dense")) o = &g_opt.sdf_dense;
    else if (!strcmp(key, "xtc_device_decode")) o = &g_opt.xtc_device_decode;
    else if (!strcmp(key, "xtc_chunk")) o = &g_opt.xtc_chunk;
    else if (!strcmp(key, "xtc_checkpoints")) o = &g_opt.xtc_checkpoints;
    else if (!strcmp(key, "xtc_records")) o = &g_opt.xtc_records;
    else if (!strcmp(key, "xtc_record_mb")) o = &g_opt.xtc_record_mb;
    else if (!strcmp(key, "xtc_mapped")) o = &g_opt.xtc_mapped;
    else if (!strcmp(key, "xtc_cold_streams")) o = &g_opt.xtc_cold_streams;
    else if (!strcmp(key, "raw_f32_device")) o = &g_opt.raw_f32_device;
    else if (!strcmp(key, "xtc_ramp")) o = &g_opt.xtc_ramp;
    else if (!strcmp(key, "block_superbatch")) o = &g_opt.block_superbatch;
    else if (!strcmp(key, "defer_sync")) o = &g_opt.defer_sync;
    else if (!strcmp(key, "gather_us")) o = &g_opt.gather_us;
    else if (!strcmp(key, "lazy_views")) o = &g_opt.lazy_views;
    else if (!strcmp(key, "lazy_views_ms")) o = &g_opt.lazy_views_ms;
    else if (!strcmp(key, "pool_mb")) { const int old = g_opt.pool_mb.exchange(value < 0 ? 0 : value); vmd_pool_trim(); return old; }
    else if (!strcmp(key, "block_two_streams")) o = &g_opt.block_two_streams;
    else if (!strcmp(key, "xtc_decode_ahead")) o = &g_opt.xtc_decode_ahead;
    else if (!strcmp(key, "xtc_map_limit_mb")) o = &g_opt.xtc_map_limit_mb;
    else if (!strcmp(key, "xtc_waves")) return vmd_hip_set_xtc_waves(value);
    else if (!strcmp(key, "stage_frames")) o = &g_opt.stage_frames;
    else if (!strcmp(key, "sdf_direct_view")) o = &g_opt.sdf_direct_view;
    else if (!strcmp(key, "sdf_nt")) return vmd_hip_set_sdf_nt(value);
    else if (!strcmp(key, "spec_rdf_closed")) o = &g_opt.spec_rdf_closed;
    else if (!strcmp(key, "spec_sdf_include_self")) o = &g_opt.spec_sdf_include_self;
    else if (!strcmp(key, "spec_sdf_density")) o = &g_opt.spec_sdf_density;
    else if (!strcmp(key, "spec_dist_geometric_com")) o = &g_opt.spec_dist_geometric_com;
    else if (!strcmp(key, "spec_rdf_raw")) o = &g_opt.spec_rdf_raw;
    else if (!strcmp(key, "spec_rdf_norm")) o = &g_opt.spec_rdf_norm;
    else if (!strcmp(key, "rdf_blocks_decode")) o = &g_opt.rdf_blocks_decode;
    else if (!strcmp(key, "rdf_classes")) o = &g_opt.rdf_classes;
    else if (!strcmp(key, "cells_small")) o = &g_opt.cells_small;
    else if (!strcmp(key, "readahead")) o = &g_opt.readahead;
    else if (!strcmp(key, "readahead_frames")) o = &g_opt.readahead_frames;
    else if (!strcmp(key, "readahead_growth")) o = &g_opt.readahead_growth;
    else if (!strcmp(key, "readahead_small")) o = &g_opt.readahead_small;
    else if (!strcmp(key, "readahead_block")) o = &g_opt.readahead_block;
    else if (!strcmp(key, "readahead_linger_us")) o = &g_opt.readahead_linger_us;
    else if (!strcmp(key, "readahead_company_us")) o = &g_opt.readahead_company_us;
    else if (!strcmp(key, "readahead_fail_alloc")) o = &g_opt.readahead_fail_alloc;
    else if (!strcmp(key, "readahead_lone")) o = &g_opt.readahead_lone;
    else if (!strcmp(key, "readahead_lone_settle_us")) o = &g_opt.readahead_lone_settle_us;
    else if (!strcmp(key, "sdf_arith")) o = &g_opt.sdf_arith;
    else if (!strcmp(key, "sdf_ilp")) return vmd_hip_set_sdf_ilp(value);
    else if (!strcmp(key, "sdf_rows")) return vmd_hip_set_sdf_rows(value);
    else if (!strcmp(key, "sdf_wave")) return vmd_hip_set_sdf_wave(value);
    else if (!strcmp(key, "rdf_nsplit")) return vmd_hip_set_rdf_nsplit(value);
    else if (!strcmp(key, "cells_pencil")) return vmd_hip_set_cells_pencil(value);
    else if (!strcmp(key, "cells_rec3")) return vmd_hip_set_cells_rec3(value);
    else if (!strcmp(key, "cells_bin_lds")) return vmd_hip_set_cells_bin_lds(value);
    else if (!strcmp(key, "cells_fused")) return vmd_hip_set_cells_fused(value);
    else if (!strcmp(key, "cells_split")) return vmd_hip_set_cells_split(value);
    else if (!strcmp(key, "rdf_blocks")) return vmd_hip_set_rdf_blocks(value);
    else if (!strcmp(key, "rdf_shared_hist")) return vmd_hip_set_rdf_shared_hist(value);
    else if (!strcmp(key, "rdf_nsub")) return vmd_hip_set_rdf_nsub(value);
    else if (!strcmp(key, "rdf_nsub_pct")) return vmd_hip_set_rdf_nsub_pct(value);
    if (!o) return -1;
    return o->exchange(value);
}

extern "C" const char* vmd_last_error(void) { return g_last_error.c_str(); }
// Where the evaluator is (process-wide, last writer wins): a static string set at every stage of a batch.  Costs one relaxed store; a
// crash handler (tests/native/stress_eval.cpp installs one for SIGABRT / SIGSEGV) can print it when the process dies inside the HIP
// runtime without a message - round 2 saw one such abort and could not say where.
static std::atomic<const char*> g_stage{"idle"};
#define VMD_STAGE(text) g_stage.store(text, std::memory_order_relaxed)
extern "C" const char* vmd_last_stage(void) { return g_stage.load(std::memory_order_relaxed); }
// for the other translation units of the library (not part of the public headers)
extern "C" void vmd_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
extern "C" void vmd_clear_last_error(void) { g_last_error.clear(); }
extern "C" const char* vmd_version(void) { return "viamd_amd 0.1 (gfx950)"; }

extern "C" int vmd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
extern "C" bool vmd_set_device(int device) {
    HIP_OK(hipSetDevice(device));
    return true;
}

// ------------------------------------------------------------------------------------------------ profiling (hipEvents)

struct ProfEntry { double ms = 0.0; uint64_t launches = 0; };
static std::mutex g_prof_mtx;
static std::map<std::string, ProfEntry> g_prof;
static std::atomic<bool> g_prof_on{false};

extern "C" void vmd_profile_enable(bool on) { g_prof_on = on; }
extern "C" void vmd_profile_reset(void) { std::lock_guard<std::mutex> l(g_prof_mtx); g_prof.clear(); }
extern "C" double vmd_profile_ms(const char* which, uint64_t* launches) {
    std::lock_guard<std::mutex> l(g_prof_mtx);
    auto it = g_prof.find(which);
    if (it == g_prof.end()) { if (launches) *launches = 0; return 0.0; }
    if (launches) *launches = it->second.launches;
    return it->second.ms;
}

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
static ResourcePool& pool() { static ResourcePool* p = new ResourcePool(); return *p; }      // never destroyed: the HIP runtime may be gone first
static thread_local int t_pool_idle = 0;
struct PoolIdle { PoolIdle() { ++t_pool_idle; } ~PoolIdle() { --t_pool_idle; } };
static const int kPinned = 64;
static int pool_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { (void)hipGetLastError(); d = 0; } return d; }
static hipError_t pool_raw_alloc(int kind, void** p, size_t bytes) {
    return kind == kPinned ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
}
static void pool_raw_free(int kind, void* p) { if (kind == kPinned) (void)hipHostFree(p); else (void)hipFree(p); }
extern "C" void vmd_pool_trim(void) {
    ResourcePool& P = pool();
    std::vector<std::pair<int, void*>> victims;
    { std::lock_guard<std::mutex> l(P.mtx);
      for (int k = 0; k <= kPinned; ++k) { for (auto& b : P.blocks[k]) { victims.push_back({k, b.second}); P.owner.erase(b.second); } P.blocks[k].clear(); }
      P.pooled[0] = P.pooled[1] = 0; }
    if (victims.empty()) return;
    int prev = pool_device();
    for (auto& v : victims) { if (v.first != kPinned) (void)hipSetDevice(v.first); pool_raw_free(v.first, v.second); }
    (void)hipSetDevice(prev);
}
extern "C" void vmd_pool_stats(size_t* device_bytes, size_t* pinned_bytes, size_t* blocks) {
    ResourcePool& P = pool();
    std::lock_guard<std::mutex> l(P.mtx);
    if (device_bytes) *device_bytes = P.pooled[0];
    if (pinned_bytes) *pinned_bytes = P.pooled[1];
    size_t n = 0;
    for (int k = 0; k <= kPinned; ++k) n += P.blocks[k].size();
    if (blocks) *blocks = n;
}
// kind: kPinned, or -1 = the current device
static hipError_t pool_take(int kind, void** out, size_t bytes) {
    if (kind < 0) kind = pool_device();
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    ResourcePool& P = pool();
    if (g_opt.pool_mb.load() > 0) {
        std::lock_guard<std::mutex> l(P.mtx);
        auto it = P.blocks[kind].lower_bound(bytes);
        if (it != P.blocks[kind].end() && it->first <= bytes + bytes / 4 + ((size_t)1 << 20)) {
            *out = it->second;
            P.pooled[kind == kPinned] -= it->first;
            P.blocks[kind].erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = pool_raw_alloc(kind, out, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); vmd_pool_trim(); e = pool_raw_alloc(kind, out, bytes); }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> l(P.mtx);
    P.owner[*out] = {bytes, kind};
    return hipSuccess;
}
static void pool_give(void* p) {
    if (!p) return;
    ResourcePool& P = pool();
    size_t bytes = 0; int kind = -2;
    { std::lock_guard<std::mutex> l(P.mtx);
      auto it = P.owner.find(p);
      if (it != P.owner.end()) { bytes = it->second.first; kind = it->second.second; } }
    if (kind == -2) { (void)hipFree(p); return; }                  // not ours (never happens: every block comes through pool_take)
    // a device block belongs to ITS device, whatever the calling thread's current one is (a reader closing on another thread)
    const int prev = kind != kPinned ? pool_device() : 0;
    if (kind != kPinned && prev != kind) (void)hipSetDevice(kind);
    if (kind != kPinned && !t_pool_idle) (void)hipDeviceSynchronize();     // queued work may still touch it (hipFree's implicit guarantee)
    const size_t limit = ((size_t)std::max(0, g_opt.pool_mb.load()) << 20) / (kind == kPinned ? 4 : 1);
    bool kept = false;
    { std::lock_guard<std::mutex> l(P.mtx);
      if (P.pooled[kind == kPinned] + bytes <= limit) {
          P.blocks[kind].insert({bytes, p});
          P.pooled[kind == kPinned] += bytes;
          kept = true;
      } else P.owner.erase(p); }
    if (!kept) pool_raw_free(kind, p);
    if (kind != kPinned && prev != kind) (void)hipSetDevice(prev);
}
static hipStream_t pool_stream(bool high_priority) {
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.streams[d][high_priority ? 1 : 0];
      if (!v.empty() && g_opt.pool_mb.load() > 0) { hipStream_t s = v.back(); v.pop_back(); return s; } }
    hipStream_t s = nullptr;
    if (high_priority) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
// the stream must be idle (synchronised by its owner)
static void pool_stream_give(hipStream_t s, bool high_priority) {
    if (!s) return;
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.streams[d][high_priority ? 1 : 0];
      if (g_opt.pool_mb.load() > 0 && v.size() < 64) { v.push_back(s); return; } }
    (void)hipStreamDestroy(s);
}
static hipEvent_t pool_event(bool timing) {
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.events[d][timing ? 0 : 1];
      if (!v.empty() && g_opt.pool_mb.load() > 0) { hipEvent_t e = v.back(); v.pop_back(); return e; } }
    hipEvent_t e = nullptr;
    if ((timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) return nullptr;
    return e;
}
static void pool_event_give(hipEvent_t e, bool timing) {
    if (!e) return;
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.events[d][timing ? 0 : 1];
      if (g_opt.pool_mb.load() > 0 && v.size() < 512) { v.push_back(e); return; } }
    (void)hipEventDestroy(e);
}

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
            if (rc == hipErrorNotReady) { (void)hipGetLastError(); later.push_back(p); continue; }     // queued on another stream, still running
            if (rc == hipSuccess) { g_prof[p.name].ms += ms; g_prof[p.name].launches += 1; }
            pool.push_back(p.a); pool.push_back(p.b);
        }
        pending.swap(later);
    }
    ~Profiler() { for (auto e : pool) pool_event_give(e, true); for (auto& p : pending) { (void)hipEventSynchronize(p.b); pool_event_give(p.a, true); pool_event_give(p.b, true); } }
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

static uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}

extern "C" vmd_script_ir_t* vmd_ir_create(void) { return new vmd_script_ir_t(); }
// atom pairs one frame of this script asks for (rdf: |ref| x |target|; sdf: K x |target| + K m for the alignment; distance: |a| x |b| of every
// context): what a host compares with its threshold before it sends a SMALL script to the GPU at all (include/vmd_md_script_shim.h,
// vmd_shim_set_min_work; VIAMD's default dataset is ~1e2 atoms, src/main.cpp:522-528)
extern "C" uint64_t vmd_ir_work_per_frame(const vmd_script_ir_t* ir) {
    if (!ir) return 0;
    uint64_t w = 0;
    for (const Property& p : ir->props) {
        if (p.kind == PROP_RDF) w += (uint64_t)p.a.size() * (uint64_t)p.b.size();
        else if (p.kind == PROP_SDF) w += (uint64_t)p.K * ((uint64_t)p.b.size() + (uint64_t)p.m);
        else if (p.aoff.size() > 1) { for (size_t c = 0; c + 1 < p.aoff.size(); ++c) w += (uint64_t)(p.aoff[c + 1] - p.aoff[c]) * (uint64_t)(p.boff[c + 1] - p.boff[c]); }
        else w += (uint64_t)p.a.size() * (uint64_t)p.b.size();
    }
    return w;
}
extern "C" void vmd_ir_free(vmd_script_ir_t* ir) { delete ir; }

static bool ir_name_ok(vmd_script_ir_t* ir, const char* name) {
    if (!ir) return vmd_fail("ir is NULL");
    if (!name || !*name) return vmd_fail("property name is empty");
    for (auto& p : ir->props) if (p.name == name) return vmd_fail("property '%s' already defined", name);
    return true;
}
static bool idx_ok(const int32_t* idx, size_t n, const char* what) {
    if (n == 0 || !idx) return vmd_fail("%s is empty", what);
    for (size_t i = 0; i < n; ++i) if (idx[i] < 0) return vmd_fail("%s contains a negative atom index", what);
    return true;
}

extern "C" bool vmd_ir_add_rdf(vmd_script_ir_t* ir, const char* name, const int32_t* ref, size_t nref,
                               const int32_t* target, size_t ntarget, float rmin, float rmax) {
    if (!ir_name_ok(ir, name) || !idx_ok(ref, nref, "rdf reference set") || !idx_ok(target, ntarget, "rdf target set")) return false;
    if (!(rmin >= 0.0f) || !(rmax > rmin)) return vmd_fail("rdf range must satisfy 0 <= rmin < rmax");
    Property p;
    p.name = name; p.kind = PROP_RDF; p.flags = VMD_PROPERTY_FLAG_DISTRIBUTION;
    p.a.assign(ref, ref + nref); p.b.assign(target, target + ntarget);
    p.rmin = rmin; p.rmax = rmax;
    ir->props.push_back(std::move(p));
    ir->rebuild_names();
    return true;
}

extern "C" bool vmd_ir_add_sdf(vmd_script_ir_t* ir, const char* name, const int32_t* structures, size_t K, size_t m,
                               const int32_t* target, size_t ntarget, float cutoff) {
    if (!ir_name_ok(ir, name) || !idx_ok(structures, K * m, "sdf reference structures") || !idx_ok(target, ntarget, "sdf target set")) return false;
    if (!(cutoff > 0.0f)) return vmd_fail("sdf cutoff must be positive");
    Property p;
    p.name = name; p.kind = PROP_SDF; p.flags = VMD_PROPERTY_FLAG_VOLUME;
    p.a.assign(structures, structures + K * m); p.b.assign(target, target + ntarget);
    p.K = K; p.m = m; p.rmax = cutoff;
    ir->props.push_back(std::move(p));
    ir->rebuild_names();
    return true;
}

extern "C" bool vmd_ir_add_distance(vmd_script_ir_t* ir, const char* name, vmd_distance_kind_t kind,
                                    const int32_t* a, size_t na, const int32_t* b, size_t nb) {
    if (!ir_name_ok(ir, name) || !idx_ok(a, na, "distance set a") || !idx_ok(b, nb, "distance set b")) return false;
    if ((int)kind < 0 || (int)kind > 3) return vmd_fail("unknown distance kind %d", (int)kind);
    Property p;
    p.name = name; p.kind = PROP_DIST; p.flags = VMD_PROPERTY_FLAG_TEMPORAL;
    p.a.assign(a, a + na); p.b.assign(b, b + nb);
    p.aoff = {0, (int32_t)na}; p.boff = {0, (int32_t)nb};
    p.dist_kind = (int)kind;
    ir->props.push_back(std::move(p));
    ir->rebuild_names();
    return true;
}

extern "C" bool vmd_ir_add_distance_population(vmd_script_ir_t* ir, const char* name, vmd_distance_kind_t kind, size_t P,
                                               const int32_t* a, const int32_t* a_offsets, const int32_t* b, const int32_t* b_offsets) {
    if (!ir_name_ok(ir, name)) return false;
    if (P == 0 || !a_offsets || !b_offsets) return vmd_fail("distance population is empty");
    if ((int)kind < 0 || (int)kind > 3) return vmd_fail("unknown distance kind %d", (int)kind);
    if (a_offsets[0] != 0 || b_offsets[0] != 0) return vmd_fail("context offsets must start at 0");
    for (size_t c = 0; c < P; ++c) {
        if (a_offsets[c + 1] <= a_offsets[c] || b_offsets[c + 1] <= b_offsets[c]) return vmd_fail("distance context %zu has an empty set", c);
        if (kind == VMD_DISTANCE_PAIR && ((a_offsets[c + 1] - a_offsets[c]) != a_offsets[1] || (b_offsets[c + 1] - b_offsets[c]) != b_offsets[1]))
            return vmd_fail("distance_pair needs contexts of equal size");
    }
    if (!idx_ok(a, (size_t)a_offsets[P], "distance set a") || !idx_ok(b, (size_t)b_offsets[P], "distance set b")) return false;
    Property p;
    p.name = name; p.kind = PROP_DIST; p.flags = VMD_PROPERTY_FLAG_TEMPORAL;
    p.a.assign(a, a + a_offsets[P]); p.b.assign(b, b + b_offsets[P]);
    p.aoff.assign(a_offsets, a_offsets + P + 1); p.boff.assign(b_offsets, b_offsets + P + 1);
    p.dist_kind = (int)kind;
    ir->props.push_back(std::move(p));
    ir->rebuild_names();
    return true;
}

extern "C" bool vmd_ir_valid(const vmd_script_ir_t* ir) { return ir != nullptr; }

extern "C" uint64_t vmd_ir_fingerprint(const vmd_script_ir_t* ir) {
    if (!ir) return 0;
    const uint64_t cached = ir->fingerprint.load();
    if (cached) return cached;
    uint64_t h = 0xCBF29CE484222325ull;
    for (auto& p : ir->props) {
        h = fnv1a(h, p.name.data(), p.name.size());
        h = fnv1a(h, &p.kind, sizeof(p.kind));
        h = fnv1a(h, p.a.data(), p.a.size() * sizeof(int32_t));
        h = fnv1a(h, p.b.data(), p.b.size() * sizeof(int32_t));
        h = fnv1a(h, &p.rmin, sizeof(float)); h = fnv1a(h, &p.rmax, sizeof(float));
        h = fnv1a(h, &p.K, sizeof(p.K)); h = fnv1a(h, &p.m, sizeof(p.m)); h = fnv1a(h, &p.dist_kind, sizeof(int));
        h = fnv1a(h, p.aoff.data(), p.aoff.size() * sizeof(int32_t)); h = fnv1a(h, p.boff.data(), p.boff.size() * sizeof(int32_t));
    }
    h = h ? h : 1;
    ir->fingerprint = h;
    return h;
}
extern "C" size_t vmd_ir_property_count(const vmd_script_ir_t* ir) { return ir ? ir->props.size() : 0; }
extern "C" const char* const* vmd_ir_property_names(const vmd_script_ir_t* ir) { return ir ? ir->names.data() : nullptr; }
extern "C" vmd_property_flags_t vmd_ir_property_flags(const vmd_script_ir_t* ir, const char* name) {
    if (!ir || !name) return VMD_PROPERTY_FLAG_NONE;
    for (auto& p : ir->props) if (p.name == name) return p.flags;
    return VMD_PROPERTY_FLAG_NONE;
}

// ------------------------------------------------------------------------------------------------ eval

// a distinct atom selection that needs a cell-sorted copy per frame batch (shared between RDF properties)
struct Selection {
    std::vector<int32_t> idx;
    DevBuf<int32_t> d_idx;
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

// what `values` of a volume points at between clear_data and the evaluation's first view: zero pages shared by every volume of the process,
// mapped read-only (a write through the pointer is a bug and faults loudly) and never backed by memory of their own (anonymous pages that
// are only ever read all alias the kernel's zero page)
static float* zero_volume_view(size_t nfloats) {
    static std::mutex mtx;
    static float* view = nullptr;
    static size_t cap = 0;
    std::lock_guard<std::mutex> l(mtx);
    if (nfloats > cap) {
        void* m = mmap(nullptr, nfloats * sizeof(float), PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return nullptr;
        view = (float*)m; cap = nfloats;        // (an earlier, smaller mapping stays: readers may still hold it)
    }
    return view;
}

struct PropState {
    Property prop;                      // private copy of the descriptor
    vmd_script_property_data_t data;
    vmd_script_aggregate_t aggregate;
    HostBuf<float> values;              // what data.values points at
    std::vector<float> ahead_values;    // DIST: temporal rows of frames evaluated ahead, copied into `values` when their block is committed (read-ahead)
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
static size_t record_stride_for(size_t frames, size_t atoms, int level = 1) {
    if (g_opt.xtc_records.load() < level || g_opt.xtc_device_decode.load() != 3) return 0;
    const size_t stride = (atoms + 63) & ~(size_t)63;
    const size_t budget = (size_t)std::max(0, g_opt.xtc_record_mb.load()) << 20;
    return (frames && stride * 2 <= budget / frames) ? stride : 0;
}
static std::mutex g_ck_mtx;
// keyed by (trajectory instance, device): two devices decoding the same file keep a table each instead of replacing each other's on
// every batch; stages with a decode in flight hold their own reference (Stage::ck_hold), so an eviction never frees what they point into
typedef std::pair<const void*, int> CkKey;
static std::map<CkKey, std::shared_ptr<CkCache>> g_ck_store;
static std::shared_ptr<CkCache> ckcache_for(const void* inst_, size_t frames, size_t atoms, int device) {
    std::lock_guard<std::mutex> l(g_ck_mtx);
    const CkKey inst(inst_, device);
    std::shared_ptr<CkCache>& c = g_ck_store[inst];
    if (!c || c->frames != frames || c->atoms != atoms || c->device != device) {
        c = std::make_shared<CkCache>();
        c->frames = frames; c->atoms = atoms; c->device = device;
        c->have.assign(frames, 0);
        c->sig.assign(frames, 0);
        if (!c->ck.ensure(std::max<size_t>(frames, 1) * VMD_XTC_CK_MAX) || !c->nck.ensure(std::max<size_t>(frames, 1))) { g_ck_store.erase(inst); return nullptr; }
        c->rec_stride = record_stride_for(frames, atoms);
        if (c->rec_stride && (!c->rec.ensure(frames * c->rec_stride) || !c->nrec.ensure(frames))) { (void)hipGetLastError(); c->rec.release(); c->nrec.release(); c->rec_stride = 0; }
        if (g_ck_store.size() > 16) {                       // a handful of open trajectories at most: forget the others
            for (auto it = g_ck_store.begin(); it != g_ck_store.end();) it = it->first == inst ? std::next(it) : g_ck_store.erase(it);
        }
    }
    return c;
}
extern "C" void vmd_ckcache_drop(const void* inst) {
    std::lock_guard<std::mutex> l(g_ck_mtx);
    for (auto it = g_ck_store.lower_bound(CkKey(inst, INT_MIN)); it != g_ck_store.end() && it->first.first == inst;) it = g_ck_store.erase(it);
}

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

extern "C" bool vmd_ckcache_save(const vmd_trajectory_i* traj, const char* path) {
    g_last_error.clear();
    if (!traj || !path) return vmd_fail("vmd_ckcache_save: NULL argument");
    std::shared_ptr<CkCache> c;
    { std::lock_guard<std::mutex> l(g_ck_mtx);
      auto it = g_ck_store.lower_bound(CkKey(traj->inst, INT_MIN));       // whichever device decoded it: the table describes the file
      if (it != g_ck_store.end() && it->first.first == traj->inst) c = it->second; }
    if (!c || c->frames == 0) return vmd_fail("vmd_ckcache_save: no decoder checkpoints exist for this trajectory (nothing of it was decoded on the device yet)");
    int prev = 0;
    HIP_OK(hipGetDevice(&prev));
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipDeviceSynchronize());                       // the tables are written by decode kernels on the evals' streams
    std::vector<uint32_t> nck(c->frames);
    std::vector<vmd_xtc_ck_t> ck(c->frames * VMD_XTC_CK_MAX);
    const bool copied = hipMemcpy(nck.data(), c->nck.p, nck.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess &&
                        hipMemcpy(ck.data(), c->ck.p, ck.size() * sizeof(vmd_xtc_ck_t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipSetDevice(prev);
    if (!copied) return vmd_fail("vmd_ckcache_save: reading the checkpoint tables back failed");
    CkFileHeader h;
    memcpy(h.magic, kCkMagic, 8);
    h.version = 1; h.ck_max = VMD_XTC_CK_MAX; h.frames = c->frames; h.atoms = c->atoms;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return vmd_fail("vmd_ckcache_save: cannot create %s", tmp.c_str());
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(c->have.data(), 1, c->frames, f) == c->frames &&
              fwrite(c->sig.data(), sizeof(uint64_t), c->frames, f) == c->frames && fwrite(nck.data(), sizeof(uint32_t), nck.size(), f) == nck.size() &&
              fwrite(ck.data(), sizeof(vmd_xtc_ck_t), ck.size(), f) == ck.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return vmd_fail("vmd_ckcache_save: writing %s failed", path); }
    return true;
}

// -> number of frames whose checkpoints were installed (0: the file does not describe this trajectory), -1 on error
extern "C" long vmd_ckcache_load(const vmd_trajectory_i* traj, const char* path, int device) {
    g_last_error.clear();
    if (!traj || !path) { vmd_fail("vmd_ckcache_load: NULL argument"); return -1; }
    FILE* f = fopen(path, "rb");
    if (!f) { vmd_fail("vmd_ckcache_load: cannot open %s", path); return -1; }
    CkFileHeader h;
    const size_t frames = traj->num_frames(traj->inst), atoms = traj->num_atoms(traj->inst);
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, kCkMagic, 8) != 0 || h.version != 1) { fclose(f); vmd_fail("vmd_ckcache_load: %s is not a checkpoint file", path); return -1; }
    if (h.ck_max != VMD_XTC_CK_MAX || h.frames != frames || h.atoms != atoms || frames == 0) { fclose(f); return 0; }
    std::vector<uint8_t> have(frames);
    std::vector<uint64_t> sig(frames);
    std::vector<uint32_t> nck(frames);
    std::vector<vmd_xtc_ck_t> ck(frames * VMD_XTC_CK_MAX);
    const bool ok = fread(have.data(), 1, frames, f) == frames && fread(sig.data(), sizeof(uint64_t), frames, f) == frames &&
                    fread(nck.data(), sizeof(uint32_t), frames, f) == frames && fread(ck.data(), sizeof(vmd_xtc_ck_t), ck.size(), f) == ck.size();
    fclose(f);
    if (!ok) { vmd_fail("vmd_ckcache_load: %s is truncated", path); return -1; }
    long n = 0;
    for (size_t i = 0; i < frames; ++i) {
        if (have[i] && (nck[i] < 1 || nck[i] > VMD_XTC_CK_MAX)) have[i] = 0;       // nothing the kernels would accept anyway
        n += have[i] ? 1 : 0;
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { vmd_fail("vmd_ckcache_load: no such device"); return -1; }
    std::shared_ptr<CkCache> c = ckcache_for(traj->inst, frames, atoms, device);
    bool up = c != nullptr;
    if (up) {
        (void)hipDeviceSynchronize();                     // an eval may be decoding this trajectory from the tables being replaced
        up = hipMemcpy(c->nck.p, nck.data(), nck.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->ck.p, ck.data(), ck.size() * sizeof(vmd_xtc_ck_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    (void)hipSetDevice(prev);
    if (!up) { vmd_fail("vmd_ckcache_load: uploading the checkpoint tables failed"); return -1; }
    std::lock_guard<std::mutex> l(g_ck_mtx);
    std::copy(have.begin(), have.end(), c->have.begin());        // in place: stages of a running eval point into these vectors
    std::copy(sig.begin(), sig.end(), c->sig.begin());
    c->rec_failed = true;                                  // no records came with them: sections from the checkpoints
    return n;
}
// Mapped trajectory files (vmd_trajectory_i::raw_mapped_view), pinned for the copy engine in windows of 1 GiB on first use
// (hipHostRegister on the mapping: 5 ms per 512 MB once, then DMA at the rate of hipHostMalloc memory - profiles/r03c_hostio.txt).
// Process-wide, keyed by the mapping's base; the reader that owns the mapping calls vmd_mapreg_drop before it unmaps.  A window
// that cannot be pinned (limit reached, the driver refuses) stays unpinned: its batches take the load_raw copy instead.
struct MapReg {
    size_t bytes = 0;
    std::vector<uint8_t> state;              // per window: 0 = not tried, 1 = pinned, 2 = refused
};
static std::mutex g_map_mtx;
static std::map<const unsigned char*, MapReg> g_map_store;
static size_t g_map_pinned = 0;
static const size_t kMapWindow = (size_t)1 << 30;
static void mapreg_release(const unsigned char* base, MapReg& m) {
    for (size_t w = 0; w < m.state.size(); ++w) {
        if (m.state[w] != 1) continue;
        (void)hipHostUnregister((void*)(base + w * kMapWindow));
        g_map_pinned -= std::min(kMapWindow, m.bytes - w * kMapWindow);
    }
    m.state.clear();
}
static bool mapreg_pin(const unsigned char* base, size_t bytes, size_t lo, size_t hi) {
    std::lock_guard<std::mutex> l(g_map_mtx);
    MapReg& m = g_map_store[base];
    if (m.bytes != bytes) {                  // a new mapping at a recycled address whose owner never dropped the old one
        mapreg_release(base, m);
        m.bytes = bytes;
        m.state.assign((bytes + kMapWindow - 1) / kMapWindow, 0);
    }
    size_t limit = (size_t)std::max(0, g_opt.xtc_map_limit_mb.load()) << 20;
    if (!limit) {
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
        limit = (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz / 2 : ((size_t)8 << 30);
    }
    for (size_t w = lo / kMapWindow; w <= (hi - 1) / kMapWindow; ++w) {
        if (m.state[w] == 1) continue;
        if (m.state[w] == 2) return false;
        const size_t len = std::min(kMapWindow, bytes - w * kMapWindow);
        if (g_map_pinned + len > limit || hipHostRegister((void*)(base + w * kMapWindow), len, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            m.state[w] = 2;
            return false;
        }
        m.state[w] = 1;
        g_map_pinned += len;
    }
    return true;
}
extern "C" void vmd_mapreg_drop(const void* base) {
    std::lock_guard<std::mutex> l(g_map_mtx);
    auto it = g_map_store.find((const unsigned char*)base);
    if (it == g_map_store.end()) return;
    mapreg_release(it->first, it->second);
    g_map_store.erase(it);
}

static uint64_t frame_signature(const vmd_xtc_frame_t& fi, const unsigned char* bytes) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ fi.nbytes;
    auto mix = [&](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); };
    uint32_t p; memcpy(&p, &fi.precision, 4);
    mix(p); mix((uint32_t)fi.smallidx);
    for (int k = 0; k < 3; ++k) { mix((uint32_t)fi.minint[k]); mix((uint32_t)fi.maxint[k]); }
    const size_t n = fi.nbytes < 64 ? (size_t)fi.nbytes : 64;
    for (size_t i = 0; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, bytes + i, 8); mix(w); }
    if (fi.nbytes >= 72) { uint64_t w; memcpy(&w, bytes + fi.nbytes - 8, 8); mix(w); }
    return h | 1ull;
}

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
static TrajId traj_id(const vmd_trajectory_i* t) {
    TrajId id;
    if (t) { id.inst = t->inst; id.fn = t->load_frame ? reinterpret_cast<const void*>(t->load_frame) : reinterpret_cast<const void*>(t->device_view); }
    return id;
}

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
        uint8_t* ck_mark = nullptr;              // that decode also writes the frames' checkpoints: mark them valid (ck_mark[0 .. nb)) when it succeeded
        uint8_t* ck_clear = nullptr;             // that decode entered the frames at their checkpoints: forget them (ck_clear[0 .. nb)) when it was rejected
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
    DevBuf<uint64_t> d_pass;                 // [passes of the batch][bins]: scratch histogram of every pair pass, committed at the batch's end
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
    // itself evaluating the same instance (two evals of one script over different trajectories of equal length must not trade blocks)
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
        std::atomic<bool> disabled{false};           // settles keep finding partly requested blocks (three strikes): the callers do not arrive the way read-ahead assumes
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
        std::atomic<int> lone_pref{-1};              // vmd_eval_set_deferred_settle: -1 = the process-wide option readahead_lone, 0 / 1 = this eval's own choice
        struct Helper {
            std::thread th;
            std::mutex mtx;
            std::condition_variable cv, idle_cv;
            bool started = false, quit = false, busy = false, have = false;      // (mtx)
            uint64_t cancel_seq = 0;                        // (mtx) bumped by every cancel: a settle that was running then does not re-arm itself
            std::atomic<bool> armed{false};                 // a settle is owed once the eval has been quiet long enough
            std::atomic<int64_t> last_leave_ns{0};          // when the last call left (steady clock)
            std::atomic<uint64_t> settles{0};
            vmd_system_t sys; vmd_trajectory_i traj;        // (mtx) copies of the caller's records: what the deferred settle evaluates from
            // vmd_eval_set_settled_callback: told after every settle the helper (or vmd_eval_wait_settled) has performed, without any lock of
            // the eval held.  Written before the evaluation's calls (like lone_pref), read by the helper: atomics, not a lock
            std::atomic<void (*)(void*)> on_settled{nullptr};
            std::atomic<void*> on_settled_user{nullptr};
        } helper;
    } ra;
    vmd_reduce_stats_t reduce_stats = {};
    struct Spec { bool rdf_closed = false, sdf_include_self = false, sdf_density = false, dist_geometric_com = false, rdf_raw = false; int rdf_norm = 0; } spec;   // fixed at creation
    size_t atoms_checked = (size_t)-1;       // trajectory atom count the properties' indices were validated against (under mtx)
};
typedef vmd_script_eval_t::Stage Stage;
typedef vmd_script_eval_t::RawSlot RawSlot;

static PropState* find_prop(const vmd_script_eval_t* e, const char* name) {
    if (!e || !name) return nullptr;
    for (auto& p : e->props) if (p->prop.name == name) return p.get();
    return nullptr;
}

static int intern_selection(vmd_script_eval_t* e, const std::vector<int32_t>& idx) {
    for (size_t i = 0; i < e->sels.size(); ++i) if (e->sels[i]->idx == idx) return (int)i;
    auto s = std::make_unique<Selection>();
    s->idx = idx;
    s->overflow_bit = 1u << (e->sels.size() % 32);
    e->sels.push_back(std::move(s));
    return (int)e->sels.size() - 1;
}

// Groups the RDF properties by range and decides, per group, between one pair pass per property and the class decomposition
// (see PairPass).  Classes are used when every set is duplicate-free, there are at most 8 of them, and the pair work
// (sum over passes of n_a * n_b, halved for same-set passes) drops by at least 10 %.
static void build_rdf_plan(vmd_script_eval_t* e) {
    e->rdf_groups.clear();
    for (size_t i = 0; i < e->props.size(); ++i) {
        const Property& d = e->props[i]->prop;
        if (d.kind != PROP_RDF) continue;
        RdfGroup* g = nullptr;
        for (auto& q : e->rdf_groups) if (memcmp(&q.rmin, &d.rmin, sizeof(float)) == 0 && memcmp(&q.rmax, &d.rmax, sizeof(float)) == 0) g = &q;
        if (!g) { e->rdf_groups.emplace_back(); g = &e->rdf_groups.back(); g->rmin = d.rmin; g->rmax = d.rmax; }
        g->props.push_back((int)i);
    }
    // every property keeps its own sets as selections (index lists only; sorted copies exist for the selections passes use):
    // the all-pairs kernel, which takes over when a batch cannot use the grid, works per property
    for (auto& p : e->props) {
        if (p->prop.kind != PROP_RDF) continue;
        p->sel_a = intern_selection(e, p->prop.a);
        p->sel_b = p->same_set ? p->sel_a : intern_selection(e, p->prop.b);
    }
    for (auto& g : e->rdf_groups) {
        auto direct = [&]() {
            g.passes.clear(); g.classes = false;
            for (int pi : g.props) {
                PropState* p = e->props[pi].get();
                PairPass ps;
                ps.sel_a = p->sel_a; ps.sel_b = p->sel_b;
                ps.same = p->same_set;
                ps.targets.push_back({pi, 1});
                g.passes.push_back(std::move(ps));
            }
        };
        const size_t np = g.props.size();
        if (np < 2 || np > 30 || !g_opt.rdf_classes) { direct(); continue; }
        // signature of every atom: bit 2k = in the reference set of the group's k-th property, bit 2k + 1 = in its target set
        int32_t amax = 0;
        for (int pi : g.props) { for (int32_t a : e->props[pi]->prop.a) amax = std::max(amax, a); for (int32_t b : e->props[pi]->prop.b) amax = std::max(amax, b); }
        std::vector<uint64_t> sig((size_t)amax + 1, 0);
        bool dup = false;
        for (size_t k = 0; k < np && !dup; ++k) {
            const Property& d = e->props[g.props[k]]->prop;
            for (int side = 0; side < 2 && !dup; ++side) {
                const uint64_t bit = 1ull << (2 * k + side);
                for (int32_t a : (side ? d.b : d.a)) { if (sig[a] & bit) { dup = true; break; } sig[a] |= bit; }
            }
        }
        if (dup) { direct(); continue; }      // a set that lists an atom twice counts it twice: only the direct passes reproduce that
        std::vector<uint64_t> csig;
        std::vector<std::vector<int32_t>> cidx;
        bool too_many = false;
        for (int32_t a = 0; a <= amax && !too_many; ++a) {
            if (!sig[a]) continue;
            size_t c = 0;
            while (c < csig.size() && csig[c] != sig[a]) ++c;
            if (c == csig.size()) { if (csig.size() == 8) { too_many = true; break; } csig.push_back(sig[a]); cidx.emplace_back(); }
            cidx[c].push_back(a);
        }
        if (too_many) { direct(); continue; }
        std::vector<PairPass> passes;
        double cost_classes = 0.0, cost_direct = 0.0;
        for (size_t c = 0; c < csig.size(); ++c)
            for (size_t d = c; d < csig.size(); ++d) {
                PairPass ps;
                for (size_t k = 0; k < np; ++k) {
                    const uint64_t X = 1ull << (2 * k), Y = 1ull << (2 * k + 1);
                    uint64_t mult;
                    if (c == d) mult = ((csig[c] & X) && (csig[c] & Y)) ? 1 : 0;
                    else mult = (((csig[c] & X) && (csig[d] & Y)) ? 1 : 0) + (((csig[d] & X) && (csig[c] & Y)) ? 1 : 0);
                    if (mult) ps.targets.push_back({g.props[k], mult});
                }
                if (ps.targets.empty()) continue;
                ps.same = c == d;
                ps.sel_a = (int)c; ps.sel_b = (int)d;          // class indices for now
                cost_classes += (double)cidx[c].size() * (double)cidx[d].size() * (c == d ? 0.5 : 1.0);
                passes.push_back(std::move(ps));
            }
        for (int pi : g.props) {
            const PropState* p = e->props[pi].get();
            cost_direct += (double)p->prop.a.size() * (double)p->prop.b.size() * (p->same_set ? 0.5 : 1.0);
        }
        if (!(cost_classes < 0.9 * cost_direct)) { direct(); continue; }
        std::vector<int> csel(csig.size());
        for (size_t c = 0; c < csig.size(); ++c) csel[c] = intern_selection(e, cidx[c]);
        for (auto& ps : passes) { ps.sel_a = csel[ps.sel_a]; ps.sel_b = csel[ps.sel_b]; }
        g.passes = std::move(passes);
        g.classes = true;
    }
}

extern "C" vmd_script_eval_t* vmd_eval_create(size_t num_frames, const vmd_script_ir_t* ir) {
    if (!ir) { vmd_fail("vmd_eval_create: ir is NULL"); return nullptr; }
    if (vmd_device_count() <= 0) { vmd_fail("vmd_eval_create: no usable HIP device (the evaluator has no CPU path)"); return nullptr; }
    auto e = std::make_unique<vmd_script_eval_t>();
    if (hipGetDevice(&e->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    // streams and events come out of the process-wide cache (an eval has nine streams and ~20 events; VIAMD makes one per script edit)
    if (!(e->stream = pool_stream(false)) || !(e->copy_stream = pool_stream(false)) || !(e->aux_stream = pool_stream(false)) ||
        !(e->pair_stream = pool_stream(false))) { vmd_fail("hipStreamCreate failed"); return nullptr; }
    if (!(e->pair_fork = pool_event(false)) || !(e->pair_join = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    // the decoder's waves should get the wave slots the pair kernel leaves free as soon as a batch has arrived: highest priority
    for (auto& ds : e->decode_streams) if (!(ds = pool_stream(true))) { vmd_fail("hipStreamCreate failed"); return nullptr; }
    e->decode_stream = e->decode_streams[0];
    for (auto& rs : e->raw_slots) if (!(rs.uploaded = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    for (auto& st : e->stages) if (!(st.ready = pool_event(true))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    e->ir_fingerprint = vmd_ir_fingerprint(ir);
    e->num_frames = num_frames;
    e->spec.rdf_closed = g_opt.spec_rdf_closed.load() != 0;
    e->spec.sdf_include_self = g_opt.spec_sdf_include_self.load() != 0;
    e->spec.sdf_density = g_opt.spec_sdf_density.load() != 0;
    e->spec.rdf_raw = g_opt.spec_rdf_raw.load() != 0;
    e->spec.rdf_norm = g_opt.spec_rdf_norm.load();
    e->spec.dist_geometric_com = g_opt.spec_dist_geometric_com.load() != 0;
    e->frame_mask.assign(num_frames, 0);
    for (auto& p : ir->props) {
        auto st = std::make_unique<PropState>();
        st->prop = p;
        memset(&st->data, 0, sizeof(st->data));
        memset(&st->aggregate, 0, sizeof(st->aggregate));
        switch (p.kind) {
        case PROP_RDF:
            st->ncounts = VMD_RDF_NUM_BINS;
            st->values.assign(st->ncounts, 0.0f); st->weights.assign(st->ncounts, 0.0f);
            st->counts.assign(st->ncounts, 0); st->weights64.assign(st->ncounts, 0.0);
            st->data.dim[0] = 1; st->data.dim[1] = 1; st->data.dim[2] = (int32_t)st->ncounts; st->data.dim[3] = 0;
            st->data.weights = st->weights.data();
            st->data.weights64 = st->weights64.data();
            st->data.min_range[0] = p.rmin; st->data.max_range[0] = p.rmax;
            st->data.unit_str[0] = "\xC3\x85"; st->data.unit_str[1] = "";
            st->same_set = (p.a == p.b);         // selections are interned by build_rdf_plan (own sets, or the classes they split into)
            break;
        case PROP_SDF:
            st->ncounts = (size_t)VMD_VOLUME_DIM * VMD_VOLUME_DIM * VMD_VOLUME_DIM;
            // the 8 + 17 MB host views of a volume are pinned: their D2H refresh runs at PCIe speed
            st->values.assign(st->ncounts, 0.0f, true);
            st->counts.assign(st->ncounts, 0, true);
            st->pinned = st->values.pinned && st->counts.pinned;
            st->data.dim[0] = 1; st->data.dim[1] = st->data.dim[2] = st->data.dim[3] = VMD_VOLUME_DIM;
            st->data.min_range[0] = -p.rmax; st->data.max_range[0] = p.rmax;
            st->data.unit_str[0] = ""; st->data.unit_str[1] = "";
            break;
        case PROP_DIST:
            st->dist_P = p.aoff.size() - 1;
            st->dist_per = p.dist_kind == VMD_DISTANCE_PAIR ? (size_t)p.aoff[1] * (size_t)p.boff[1] : 1;
            st->dim1 = st->dist_P * st->dist_per;
            st->values.assign(num_frames * st->dim1, 0.0f);
            st->data.dim[0] = (int32_t)num_frames; st->data.dim[1] = (int32_t)st->dim1;
            st->data.unit_str[0] = ""; st->data.unit_str[1] = "\xC3\x85";
            if (st->dim1 > 1) {
                st->agg_mean.assign(num_frames, 0.0f); st->agg_var.assign(num_frames, 0.0f); st->agg_ext.assign(num_frames * 2, 0.0f);
                st->aggregate.num_values = num_frames;
                st->aggregate.population_mean = st->agg_mean.data();
                st->aggregate.population_var = st->agg_var.data();
                st->aggregate.population_ext = (float(*)[2])st->agg_ext.data();
                st->data.aggregate = &st->aggregate;
            }
            break;
        }
        st->data.values = st->values.data();
        st->data.num_values = st->values.size();
        st->data.counts = st->counts.empty() ? nullptr : st->counts.data();
        st->data.fingerprint = 1;
        if (st->ncounts) {
            if (!st->d_counts.ensure(st->ncounts)) return nullptr;
            if (hipMemsetAsync(st->d_counts.p, 0, st->ncounts * sizeof(uint64_t), e->stream) != hipSuccess) { vmd_fail("hipMemset failed"); return nullptr; }
        }
        e->props.push_back(std::move(st));
    }
    build_rdf_plan(e.get());
    if (!e->d_overflow.ensure(1) || hipMemsetAsync(e->d_overflow.p, 0, sizeof(uint32_t), e->stream) != hipSuccess ||
        pool_take(kPinned, (void**)&e->h_overflow, 2 * sizeof(uint32_t)) != hipSuccess) { vmd_fail("allocating the overflow flag failed"); return nullptr; }
    e->h_overflow[0] = e->h_overflow[1] = 0;
    for (auto& ev : e->batch_done) if (!(ev = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { vmd_fail("hipStreamSynchronize failed"); return nullptr; }
    return e.release();
}

static void lone_stop(vmd_script_eval_t* e);
extern "C" void vmd_eval_free(vmd_script_eval_t* eval) {
    if (!eval) return;
    VMD_STAGE("vmd_eval_free");
    lone_stop(eval);                     // the helper thread of a deferred-settle eval finishes what it is doing and ends
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(eval->device);
    {
        std::lock_guard<std::mutex> l(eval->mtx);
        // everything this eval ever queued ran on its own streams: once they are idle its blocks, streams and events can go back to
        // the process-wide cache without another synchronisation (PoolIdle)
        if (eval->stream) { (void)hipStreamSynchronize(eval->stream); }
        if (eval->copy_stream) { (void)hipStreamSynchronize(eval->copy_stream); }
        if (eval->aux_stream) { (void)hipStreamSynchronize(eval->aux_stream); }
        if (eval->pair_stream) { (void)hipStreamSynchronize(eval->pair_stream); }
        for (auto& ds : eval->decode_streams) { if (ds) { (void)hipStreamSynchronize(ds); pool_stream_give(ds, true); } ds = nullptr; }
        eval->decode_stream = nullptr;
        PoolIdle idle;
        for (auto& rs : eval->raw_slots) {
            if (rs.h) pool_give(rs.h);
            rs.h = nullptr;
            rs.d.release();
            pool_event_give(rs.uploaded, false);
            rs.uploaded = nullptr;
        }
        for (auto& st : eval->stages) {
            if (st.h) pool_give(st.h);
            st.h = nullptr;
            if (st.hraw) pool_give(st.hraw);
            st.hraw = nullptr;
            if (st.h_raw_status) pool_give(st.h_raw_status);
            st.h_raw_status = nullptr;
            st.d.release(); st.d_boxes.release(); st.d_bbox.release(); st.d_gboxes.release();
            st.d_raw.release(); st.d_raw_info.release(); st.d_raw_status.release(); st.d_raw_scratch.release();
            pool_event_give(st.ready, true);
            st.ready = nullptr;
        }
        pool_stream_give(eval->copy_stream, false); eval->copy_stream = nullptr;
        pool_stream_give(eval->aux_stream, false); eval->aux_stream = nullptr;
        pool_stream_give(eval->pair_stream, false); eval->pair_stream = nullptr;
        pool_event_give(eval->pair_fork, false); pool_event_give(eval->pair_join, false);
        eval->pair_fork = eval->pair_join = nullptr;
        eval->props.clear();
        eval->sels.clear();
        if (eval->h_overflow) pool_give(eval->h_overflow);
        eval->h_overflow = nullptr;
        if (eval->h_snap) pool_give(eval->h_snap);
        eval->h_snap = nullptr;
        for (auto& ev : eval->batch_done) { pool_event_give(ev, false); ev = nullptr; }
        eval->d_partial.release(); eval->d_partial2.release(); eval->d_pass.release(); eval->d_overflow.release(); eval->d_pen_sample.release();
        pool_stream_give(eval->stream, false);
        eval->stream = nullptr;
    }
    {
        PoolIdle idle;              // whatever the destructors still hold (the profilers' events, the remaining buffers)
        delete eval;
    }
    (void)hipSetDevice(prev_dev);
}

// Published scalars.  VIAMD's GUI thread reads a property's record while pool threads are inside frame_range (src/main.cpp:1508-1524):
// by the reference's contract a reader may see old and new fields side by side, never a crash.  The scalar fields are therefore
// written with relaxed atomic stores - plain moves on x86-64 - so that the contract is also what the C++ memory model and
// ThreadSanitizer (scripts/tsan_emu.sh) see; the shim's refresh() loads them the same way.  The arrays behind `values` / `weights`
// are written by DMA, memcpy and fills: a reader of those runs under the reference's "torn data is tolerated" rule only.
template <class T> static inline void pub(T& dst, T v) { __atomic_store(&dst, &v, __ATOMIC_RELAXED); }
static inline void pub_touch(uint64_t& fingerprint) { uint64_t v; __atomic_load(&fingerprint, &v, __ATOMIC_RELAXED); v += 1; __atomic_store(&fingerprint, &v, __ATOMIC_RELAXED); }

static void ra_reset(vmd_script_eval_t* e);
static void lone_cancel(vmd_script_eval_t* e);
extern "C" void vmd_eval_clear_data(vmd_script_eval_t* eval) {
    VMD_STAGE("vmd_eval_clear_data");
    if (!eval) return;
    lone_cancel(eval);                   // a deferred settle of the evaluation that ends here must neither start nor be running
    std::lock_guard<std::mutex> l(eval->mtx);
    eval->interrupt = false;
    std::fill(eval->frame_mask.begin(), eval->frame_mask.end(), (uint8_t)0);
    eval->frames_done = 0;
    eval->frames_computed = 0; eval->frames_reused = 0; eval->frames_device_decoded = 0; eval->frames_section_decoded = 0; eval->frames_mapped = 0;
    for (size_t b = 0; b < eval->num_blocks; ++b) eval->block_ready[b] = 0;
    eval->blocks_inst = TrajId();
    ra_reset(eval);
    for (auto& p : eval->props) {
        // The float view of a volume (8.4 MB, pinned) is NOT zeroed: `data.values` is pointed at a shared, read-only page range of zeros until
        // the next view of this evaluation has been written - k_counts_to_float rewrites every voxel of the real view, then the pointer
        // flips back (refresh_volume).  Round 6: the zeroing was a second 8.4 MB pass over PCIe per evaluation - 0.15 ms
        // that the kernel trace showed IN FRONT of the evaluation's kernels, not under them (profiles/r06a_c4_1250_timeline.txt) - a fifth of
        // a rank's 1 250-frame share of configs[3].  A reader polling `fingerprint` (src/main.cpp:1508; density_volume.cpp:159-163, 279-283) sees
        // zeros under the new fingerprint at once, never the previous run's voxels; VIAMD dereferences prop_data->values when it uploads.
        if (p->prop.kind == PROP_SDF) pub(p->data.values, zero_volume_view(p->ncounts));
        else std::fill(p->values.begin(), p->values.end(), 0.0f);
        std::fill(p->weights.begin(), p->weights.end(), 0.0f);
        // the 17 MB u64 mirror of a volume is only ever read after vmd_eval_refresh_counts: mark it stale instead of zeroing it
        if (p->prop.kind == PROP_SDF) p->counts_stale = true;
        else std::fill(p->counts.begin(), p->counts.end(), (uint64_t)0);
        std::fill(p->weights64.begin(), p->weights64.end(), 0.0);
        std::fill(p->agg_mean.begin(), p->agg_mean.end(), 0.0f);
        std::fill(p->agg_var.begin(), p->agg_var.end(), 0.0f);
        std::fill(p->agg_ext.begin(), p->agg_ext.end(), 0.0f);
        if (p->ncounts) (void)hipMemsetAsync(p->d_counts.p, 0, p->ncounts * sizeof(uint64_t), eval->stream);
        pub(p->data.max_value, 0.0f); pub(p->data.min_value, 0.0f);
        pub(p->data.max_range[1], 0.0f);
        p->dirty = false;
        if (p->prop.kind != PROP_SDF) p->counts_stale = false;
        pub_touch(p->data.fingerprint);
    }
    (void)hipStreamSynchronize(eval->stream);
}

extern "C" void vmd_eval_interrupt(vmd_script_eval_t* eval) {
    if (!eval) return;
    eval->interrupt = true;
    // deferred-settle mode: a settle that is owed is dropped, one that is running ends at its next batch boundary - and has ended when this
    // returns: VIAMD resets the arena that holds molecule and trajectory right after interrupt_async_tasks (src/viamd.cpp:234-241, 624-630)
    if (eval->ra.lone.load()) lone_cancel(eval);
}
extern "C" uint64_t vmd_eval_ir_fingerprint(const vmd_script_eval_t* eval) { return eval ? eval->ir_fingerprint : 0; }
extern "C" const vmd_script_property_data_t* vmd_eval_property_data(const vmd_script_eval_t* eval, const char* name) {
    PropState* p = find_prop(eval, name);
    return p ? &p->data : nullptr;
}
extern "C" const uint8_t* vmd_eval_frame_mask(const vmd_script_eval_t* eval) { return eval ? eval->frame_mask.data() : nullptr; }
extern "C" size_t vmd_eval_frame_mask_bits(const vmd_script_eval_t* eval, uint64_t* words, size_t cap) {
    if (!eval) return 0;
    const size_t nw = (eval->num_frames + 63) / 64;
    for (size_t w = 0; w < nw && w < cap && words; ++w) {
        uint64_t v = 0;
        const size_t f1 = std::min(eval->num_frames, (w + 1) * 64);
        for (size_t f = w * 64; f < f1; ++f) if (eval->frame_mask[f]) v |= 1ull << (f & 63);
        words[w] = v;
    }
    return nw;
}
extern "C" size_t vmd_eval_num_frames(const vmd_script_eval_t* eval) { return eval ? eval->num_frames : 0; }
extern "C" size_t vmd_eval_frames_done(const vmd_script_eval_t* eval) { return eval ? eval->frames_done.load() : 0; }

// ---- host views -------------------------------------------------------------------------------------------------


// the float views of a distribution from integer counts and fp64 weights (the device accumulators, or a snapshot of them)
static void refresh_distribution_from(PropState* p, const uint64_t* counts, const double* weights64) {
    float ymax = 0.0f, vmax = 0.0f;
    for (size_t b = 0; b < p->ncounts; ++b) {
        p->counts[b] = counts[b];
        const float v = (float)counts[b];
        const float w = (float)weights64[b];
        p->values[b] = v; p->weights[b] = w;
        vmax = std::max(vmax, v);
        if (w > 0.0f) ymax = std::max(ymax, v / w);
    }
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub(p->data.min_range[1], 0.0f); pub(p->data.max_range[1], ymax);
    pub_touch(p->data.fingerprint);
}

static bool refresh_distribution(vmd_script_eval_t* e, PropState* p) {
    HIP_OK(hipMemcpyAsync(p->counts.data(), p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    float ymax = 0.0f, vmax = 0.0f;
    for (size_t b = 0; b < p->ncounts; ++b) {
        const float v = (float)p->counts[b];
        const float w = (float)p->weights64[b];
        p->values[b] = v; p->weights[b] = w;
        vmax = std::max(vmax, v);
        if (w > 0.0f) ymax = std::max(ymax, v / w);
    }
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub(p->data.min_range[1], 0.0f); pub(p->data.max_range[1], ymax);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
    return true;
}

static bool refresh_volume(vmd_script_eval_t* e, PropState* p) {
    VMD_STAGE("refresh_volume: counts -> float view, D2H");
    if (!p->d_max.ensure(1)) return false;
    float scale = 1.0f;
    if (e->spec.sdf_density) {
        // DECISION(D-SDF-NORM) flipped: number density per cubic Angstrom, averaged over the frames evaluated so far
        const double edge = 2.0 * (double)p->prop.rmax / (double)VMD_VOLUME_DIM;
        const size_t nf = e->frames_done.load();
        scale = nf ? (float)(1.0 / ((double)nf * edge * edge * edge)) : 0.0f;
    }
    float vmax = 0.0f;
    float* host_view_dev = nullptr;
    if (g_opt.sdf_direct_view.load() && p->values.pinned && hipHostGetDevicePointer((void**)&host_view_dev, p->values.data(), 0) != hipSuccess) {
        (void)hipGetLastError();
        host_view_dev = nullptr;
    }
    if (host_view_dev) {
        // the conversion kernel writes the float view VIAMD reads straight into its pinned host pages (8.4 MB over PCIe at the
        // DMA's rate): no device-side copy of the view, no separate DMA behind the kernel
        KRN_OK(vmd_hip_counts_to_float(e->stream, p->d_counts.p, p->ncounts, host_view_dev, p->d_max.p, scale));
    } else {
        if (!p->d_values.ensure(p->ncounts)) return false;
        KRN_OK(vmd_hip_counts_to_float(e->stream, p->d_counts.p, p->ncounts, p->d_values.p, p->d_max.p, scale));
        HIP_OK(hipMemcpyAsync(p->values.data(), p->d_values.p, p->ncounts * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    }
    // the 17 MB u64 mirror behind `counts` is an extension VIAMD never reads: it is synchronised on demand
    // (vmd_eval_refresh_counts), only the float view travels after every range
    p->counts_stale = true;
    HIP_OK(hipMemcpyAsync(&vmax, p->d_max.p, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    pub(p->data.values, p->values.data());       // every voxel of the real view has just been rewritten: readers leave the shared zeros (clear_data)
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
    return true;
}

static void refresh_temporal_stats(vmd_script_eval_t* e, PropState* p) {
    float lo = 3.4e38f, hi = -3.4e38f;
    bool any = false;
    for (size_t f = 0; f < e->num_frames; ++f) {
        if (!e->frame_mask[f]) continue;
        const float* row = &p->values[f * p->dim1];
        float rlo = row[0], rhi = row[0];
        double s = 0.0;
        for (size_t i = 0; i < p->dim1; ++i) { rlo = std::min(rlo, row[i]); rhi = std::max(rhi, row[i]); s += row[i]; }
        if (p->dim1 > 1) {
            const double mean = s / (double)p->dim1;
            double v = 0.0;
            for (size_t i = 0; i < p->dim1; ++i) { const double d = row[i] - mean; v += d * d; }
            p->agg_mean[f] = (float)mean;
            p->agg_var[f] = (float)std::sqrt(v / (double)p->dim1);   // VIAMD plots mean +- var as a band (src/main.cpp:1409-1424)
            p->agg_ext[2 * f] = rlo; p->agg_ext[2 * f + 1] = rhi;
        }
        lo = std::min(lo, rlo); hi = std::max(hi, rhi);
        any = true;
    }
    if (!any) { lo = hi = 0.0f; }
    pub(p->data.min_value, lo); pub(p->data.max_value, hi);
    pub(p->data.min_range[0], lo); pub(p->data.max_range[0], hi);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
}

extern "C" bool vmd_eval_defer_volume_views(vmd_script_eval_t* eval, bool defer) {
    if (!eval) return vmd_fail("eval is NULL");
    eval->defer_volume_views.store(defer, std::memory_order_relaxed);
    return true;
}

extern "C" bool vmd_eval_wait_settled(vmd_script_eval_t* eval);
extern "C" bool vmd_eval_finalize(vmd_script_eval_t* eval) {
    if (!eval) return vmd_fail("eval is NULL");
    if (!vmd_eval_wait_settled(eval)) return false;       // deferred-settle evals: the totals first
    std::lock_guard<std::mutex> l(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    for (auto& p : eval->props) {
        bool ok = true;
        if (p->prop.kind == PROP_RDF) ok = refresh_distribution(eval, p.get());
        else if (p->prop.kind == PROP_SDF) ok = refresh_volume(eval, p.get());
        else refresh_temporal_stats(eval, p.get());
        if (!ok) return false;
    }
    return true;
}

extern "C" bool vmd_eval_refresh_counts(vmd_script_eval_t* eval, const char* name) {
    PropState* p = find_prop(eval, name);
    if (!p) return vmd_fail("vmd_eval_refresh_counts: no property '%s'", name ? name : "(null)");
    std::lock_guard<std::mutex> l(eval->mtx);
    if (!p->counts_stale || !p->ncounts) return true;
    HIP_OK(hipSetDevice(eval->device));
    HIP_OK(hipMemcpyAsync(p->counts.data(), p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, eval->stream));
    HIP_OK(hipStreamSynchronize(eval->stream));
    p->counts_stale = false;
    return true;
}

extern "C" bool vmd_eval_set_block_frames(vmd_script_eval_t* eval, size_t block_frames) {
    if (!eval) return vmd_fail("eval is NULL");
    std::lock_guard<std::mutex> l(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    if (eval->frames_done.load() != 0) return vmd_fail("vmd_eval_set_block_frames: call before the first frame_range or right after clear_data");
    eval->ra.on = false; eval->ra.own_blocks = false; eval->ra.blk_state.reset(); eval->ra.frame_req.reset();     // read-ahead re-engages on the new blocks
    eval->block_frames = 0; eval->num_blocks = 0; eval->block_ready.reset();
    for (auto& p : eval->props) { p->d_blocks.release(); p->block_weights64.clear(); }
    if (block_frames == 0) return true;
    const size_t nblocks = (eval->num_frames + block_frames - 1) / block_frames;
    size_t bytes = 0;
    for (auto& p : eval->props) bytes += nblocks * p->ncounts * sizeof(uint64_t);
    if (bytes > ((size_t)96 << 30))
        return vmd_fail("vmd_eval_set_block_frames: %zu blocks need %.1f GB of block partials; use larger blocks", nblocks, (double)bytes / 1073741824.0);
    for (auto& p : eval->props) {
        if (!p->ncounts) continue;
        if (!p->d_blocks.ensure(nblocks * p->ncounts)) return false;
        if (p->prop.kind == PROP_RDF) p->block_weights64.assign(nblocks * p->ncounts, 0.0);
    }
    eval->block_ready.reset(new std::atomic<uint8_t>[nblocks]);
    for (size_t b = 0; b < nblocks; ++b) eval->block_ready[b] = 0;
    eval->num_blocks = nblocks;
    eval->block_frames = block_frames;
    return true;
}

extern "C" bool vmd_eval_set_source(vmd_script_eval_t* eval, vmd_script_eval_t* source) {
    if (!eval) return vmd_fail("eval is NULL");
    std::lock_guard<std::mutex> l(eval->mtx);
    if (!source) { eval->source = nullptr; return true; }
    if (source == eval) return vmd_fail("vmd_eval_set_source: an eval cannot be its own source");
    if (source->ir_fingerprint != eval->ir_fingerprint || source->num_frames != eval->num_frames || source->props.size() != eval->props.size())
        return vmd_fail("vmd_eval_set_source: source was created from a different script or frame count");
    if (source->device != eval->device) return vmd_fail("vmd_eval_set_source: source lives on another device");
    // (a source without block partials is accepted since round 4: read-ahead gives an eval driven by pool threads block partials of its own
    // accord, and whether the source has any is looked up, under its mutex, whenever a range is served)
    eval->source = source;
    return true;
}

extern "C" size_t vmd_eval_frames_device_decoded(const vmd_script_eval_t* eval) { return eval ? eval->frames_device_decoded.load() : 0; }
extern "C" size_t vmd_eval_frames_section_decoded(const vmd_script_eval_t* eval) { return eval ? eval->frames_section_decoded.load() : 0; }
extern "C" size_t vmd_eval_frames_mapped(const vmd_script_eval_t* eval) { return eval ? eval->frames_mapped.load() : 0; }

extern "C" void vmd_eval_cell_build_stats(const vmd_script_eval_t* eval, size_t* bucket_overflows, size_t* selections_off_buckets) {
    size_t ov = 0, off = 0;
    if (eval) for (auto& s : eval->sels) { ov += (size_t)std::min(s->overflows, 98); off += s->overflows >= 3 ? 1 : 0; }
    if (bucket_overflows) *bucket_overflows = ov;
    if (selections_off_buckets) *selections_off_buckets = off;
}

extern "C" void vmd_eval_frame_stats(const vmd_script_eval_t* eval, size_t* frames_computed, size_t* frames_reused) {
    if (frames_computed) *frames_computed = eval ? eval->frames_computed.load() : 0;
    if (frames_reused) *frames_reused = eval ? eval->frames_reused.load() : 0;
}

extern "C" void vmd_eval_set_frame_mask(vmd_script_eval_t* eval, const uint8_t* mask, size_t n) {
    if (!eval || !mask) return;
    std::lock_guard<std::mutex> l(eval->mtx);
    size_t done = 0;
    for (size_t f = 0; f < eval->num_frames; ++f) {
        if (f < n) eval->frame_mask[f] = mask[f] ? 1 : 0;
        done += eval->frame_mask[f] ? 1 : 0;
    }
    eval->frames_done = done;
}

extern "C" size_t vmd_eval_accum_views(vmd_script_eval_t* eval, vmd_accum_view_t* out, size_t cap) {
    if (!eval) return 0;
    size_t n = 0;
    for (auto& p : eval->props) {
        if (n < cap && out) {
            vmd_accum_view_t v;
            memset(&v, 0, sizeof(v));
            v.name = p->prop.name.c_str();
            v.flags = p->prop.flags;
            if (p->ncounts) { v.counts_dev = p->d_counts.p; v.num_counts = p->ncounts; }
            // a voxel receives at most one count per (frame, structure, target atom)
            if (p->prop.kind == PROP_SDF) {
                const long double b = (long double)eval->num_frames * (long double)p->prop.K * (long double)p->prop.b.size();
                v.count_bound = b < 1.8e19L ? (uint64_t)b : 0;
            }
            if (!p->weights64.empty()) { v.weights64 = p->weights64.data(); v.num_weights = p->weights64.size(); }
            if (p->prop.kind == PROP_DIST) { v.temporal = p->values.data(); v.num_temporal = p->values.size(); }
            out[n] = v;
        }
        n += 1;
    }
    return n;
}

// hooks for vmd_reduce.cpp (same library, not part of the public headers)
extern "C" int vmd_eval_internal_device(const vmd_script_eval_t* eval) { return eval ? eval->device : 0; }
extern "C" void vmd_eval_internal_lock(vmd_script_eval_t* eval, int lock) { if (eval) { if (lock) eval->mtx.lock(); else eval->mtx.unlock(); } }
extern "C" vmd_reduce_stats_t* vmd_eval_internal_reduce_stats(vmd_script_eval_t* eval) { return eval ? &eval->reduce_stats : nullptr; }
extern "C" void vmd_eval_reduce_stats(const vmd_script_eval_t* eval, vmd_reduce_stats_t* out) {
    if (!out) return;
    if (eval) *out = eval->reduce_stats; else memset(out, 0, sizeof(*out));
}

// ---- the hot call -----------------------------------------------------------------------------------------------

static bool upload_static(vmd_script_eval_t* e, const vmd_system_t* sys, size_t traj_atoms) {
    for (auto& s : e->sels) {
        if (!s->d_idx.p) { if (!s->d_idx.upload(s->idx.data(), s->idx.size(), e->stream)) return false; }
    }
    for (auto& p : e->props) {
        if (p->uploaded) continue;
        const Property& d = p->prop;
        auto masses = [&](const std::vector<int32_t>& idx, std::vector<float>& out) {
            out.resize(idx.size());
            for (size_t i = 0; i < idx.size(); ++i)
                out[i] = (sys && sys->mass && (size_t)idx[i] < sys->atom_count) ? sys->mass[idx[i]] : 1.0f;
            if (d.kind == PROP_DIST && e->spec.dist_geometric_com) std::fill(out.begin(), out.end(), 1.0f);   // D-DIST-COM flipped
        };
        std::vector<float> tmp;
        if (d.kind == PROP_SDF) {
            if (!p->d_structs.upload(d.a.data(), d.a.size(), e->stream)) return false;
            if (!p->d_tgt.upload(d.b.data(), d.b.size(), e->stream)) return false;
            masses(d.a, tmp);
            if (!p->d_mass.upload(tmp.data(), tmp.size(), e->stream)) return false;
            if (!p->d_ref_pose.ensure(d.m * 3)) return false;
            p->have_tree = false;
            if (sys && sys->bonds && sys->bond_count) {
                // D-SDF-UNWRAP with bonds: breadth-first from local atom 0 over the bonds among the structure's atoms, neighbours in
                // increasing local index; atoms the walk does not reach hang on their index predecessor (oracle: vo_bond_tree)
                std::vector<int32_t> order(d.K * d.m), parent(d.K * d.m);
                std::map<int32_t, std::vector<int32_t>> adj;            // only atoms of reference structures matter
                std::map<int32_t, char> member;
                for (int32_t a : d.a) member[a] = 1;
                for (size_t b = 0; b < sys->bond_count; ++b) {
                    const int32_t i = sys->bonds[b][0], j = sys->bonds[b][1];
                    if (member.count(i) && member.count(j)) { adj[i].push_back(j); adj[j].push_back(i); }
                }
                for (size_t k = 0; k < d.K; ++k) {
                    const int32_t* idx = &d.a[k * d.m];
                    int32_t* ord = &order[k * d.m];
                    int32_t* par = &parent[k * d.m];
                    std::map<int32_t, int32_t> local;
                    for (size_t a = 0; a < d.m; ++a) local.emplace(idx[a], (int32_t)a);
                    std::vector<char> seen(d.m, 0);
                    size_t head = 0, tail = 0;
                    ord[tail++] = 0; seen[0] = 1; par[0] = -1;
                    while (head < tail) {
                        const int32_t a = ord[head++];
                        std::vector<int32_t> nb;
                        auto it = adj.find(idx[a]);
                        if (it != adj.end()) for (int32_t g : it->second) { auto l = local.find(g); if (l != local.end()) nb.push_back(l->second); }
                        std::sort(nb.begin(), nb.end());
                        for (int32_t c : nb) if (!seen[c]) { seen[c] = 1; par[c] = a; ord[tail++] = c; }
                    }
                    for (size_t a = 1; a < d.m; ++a) if (!seen[a]) { par[a] = (int32_t)a - 1; ord[tail++] = (int32_t)a; }
                }
                if (!p->d_tree_order.upload(order.data(), order.size(), e->stream) || !p->d_tree_parent.upload(parent.data(), parent.size(), e->stream)) return false;
                HIP_OK(hipStreamSynchronize(e->stream));                 // the vectors go out of scope
                p->have_tree = true;
            }
            // owner[t]: the structure target t is a member of (exclusion rule); only valid when memberships are unique
            std::vector<int8_t> owner(d.b.size(), (int8_t)-1);
            bool unique = d.K <= 127;
            if (unique) {
                std::map<int32_t, int> where;
                for (size_t k = 0; k < d.K && unique; ++k)
                    for (size_t a = 0; a < d.m; ++a) {
                        auto it = where.find(d.a[k * d.m + a]);
                        if (it != where.end() && it->second != (int)k) { unique = false; break; }
                        where[d.a[k * d.m + a]] = (int)k;
                    }
                if (unique) for (size_t t = 0; t < d.b.size(); ++t) { auto it = where.find(d.b[t]); if (it != where.end()) owner[t] = (int8_t)it->second; }
            }
            p->have_owner = unique;
            if (unique && !p->d_owner.upload(owner.data(), owner.size(), e->stream)) return false;
            p->unowned = unique && std::all_of(owner.begin(), owner.end(), [](int8_t o) { return o < 0; });
            // an arithmetic progression (every water oxygen of a regular solvent box: first + 3 t) needs no index list on the device
            p->tgt_first = d.b[0]; p->tgt_stride = 0;
            if (d.b.size() >= 2 && d.b[1] > d.b[0] && g_opt.sdf_arith != 0) {
                const int64_t st = (int64_t)d.b[1] - d.b[0];
                bool ok = true;
                for (size_t t = 2; t < d.b.size() && ok; ++t) ok = (int64_t)d.b[t] - d.b[t - 1] == st;
                if (ok && st < (1 << 20)) p->tgt_stride = (int)st;
            }
            // dense targets: stream whole frames and select by a per-atom tag instead of gathering through the index list
            // sized from the TRAJECTORY's atom count (the target indices were validated against it, check_atoms), never from
            // sys->atom_count, which a host may leave unset or out of step
            const size_t natoms = traj_atoms;
            p->have_tag = unique && d.K <= 253 && natoms > 0 && d.b.size() * 8 >= natoms && g_opt.sdf_dense != 0;
            if (p->have_tag) {
                p->tag_len = (natoms + 63) & ~(size_t)63;
                std::vector<uint8_t> tag(p->tag_len, (uint8_t)255);
                for (size_t t = 0; t < d.b.size(); ++t) tag[d.b[t]] = owner[t] < 0 ? (uint8_t)254 : (uint8_t)owner[t];
                if (!p->d_tag.upload(tag.data(), tag.size(), e->stream)) return false;
                HIP_OK(hipStreamSynchronize(e->stream));
            }
            HIP_OK(hipStreamSynchronize(e->stream));
        } else if (d.kind == PROP_DIST) {
            if (!p->d_a.upload(d.a.data(), d.a.size(), e->stream)) return false;
            if (!p->d_b.upload(d.b.data(), d.b.size(), e->stream)) return false;
            if (!p->d_aoff.upload(d.aoff.data(), d.aoff.size(), e->stream)) return false;
            if (!p->d_boff.upload(d.boff.data(), d.boff.size(), e->stream)) return false;
            masses(d.a, tmp);
            if (!p->d_ma.upload(tmp.data(), tmp.size(), e->stream)) return false;
            masses(d.b, tmp);
            if (!p->d_mb.upload(tmp.data(), tmp.size(), e->stream)) return false;
        }
        HIP_OK(hipStreamSynchronize(e->stream));   // tmp goes out of scope
        p->uploaded = true;
    }
    return true;
}

static bool check_atoms(vmd_script_eval_t* e, size_t num_atoms) {
    if (e->atoms_checked == num_atoms) return true;          // the index lists never change: one pass per trajectory size
    for (auto& p : e->props) {
        for (int32_t i : p->prop.a) if ((size_t)i >= num_atoms) return vmd_fail("property '%s' references atom %d but the trajectory has %zu atoms", p->prop.name.c_str(), i, num_atoms);
        for (int32_t i : p->prop.b) if ((size_t)i >= num_atoms) return vmd_fail("property '%s' references atom %d but the trajectory has %zu atoms", p->prop.name.c_str(), i, num_atoms);
    }
    e->atoms_checked = num_atoms;
    return true;
}

struct BatchSrc {
    const float* base = nullptr;   // device
    size_t frame_stride = 0, row_stride = 0;
};

// Device-side decompression of a staged batch (vmd_trajectory_i::load_raw + k_xtc_wave): the compressed bit streams are read
// into pinned memory on the decode threads, cross PCIe as they are (0.4x the float bytes for water) and are decompressed on the
// copy stream, i.e. under the kernels of the previous batch.  Nothing here waits for the device: the status words come back
// with the batch's `ready` event and are looked at when the batch is about to be used (settle_stage).  Returns 1 when the decode
// is queued into st.d, 0 when the batch has to go through load_frame (a frame is not available raw), -1 on error.
static int launch_raw_decode(vmd_script_eval_t* e, Stage& st, const unsigned char* d_raw, const vmd_xtc_frame_t* d_info, size_t num_atoms,
                             size_t nb, size_t npad, hipStream_t stream, vmd_xtc_ck_t* ck = nullptr, uint32_t* nck = nullptr,
                             uint8_t* ck_have = nullptr, uint16_t* rec = nullptr, uint32_t* nrec = nullptr, size_t rec_stride = 0,
                             bool* rec_failed = nullptr) {
    st.ck_mark = nullptr;
    st.ck_clear = nullptr;
    st.rec_failed = nullptr;
    st.sectioned = false;
    if (nb > st.h_raw_status_cap) {
        if (st.h_raw_status) pool_give(st.h_raw_status);
        st.h_raw_status = nullptr; st.h_raw_status_cap = 0;
        if (pool_take(kPinned, (void**)&st.h_raw_status, nb * sizeof(uint32_t)) != hipSuccess) { vmd_fail("hipHostMalloc failed"); return -1; }
        st.h_raw_status_cap = nb;
    }
    if (!st.d.ensure(nb * 3 * npad) || !st.d_raw_status.ensure(nb)) return -1;
    int rc;
    const int mode = g_opt.xtc_device_decode.load();
    e->prof_copy.begin("xtc_decode", stream);
    if (mode == 2) {
        const int chunk = std::max(64, g_opt.xtc_chunk.load());
        if (!st.d_raw_scratch.ensure((vmd_hip_xtc_scratch_bytes((int)nb, (int)num_atoms, chunk) + 7) / 8)) return -1;
        rc = vmd_hip_xtc_decode_chunked(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad,
                                        st.d_raw_status.p, chunk, st.d_raw_scratch.p);
    } else if (mode == 1) {
        rc = vmd_hip_xtc_decode(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p);
    } else if (ck && nck && ck_have && g_opt.xtc_checkpoints.load()) {
        bool all = true;
        for (size_t b = 0; b < nb; ++b) all = all && flag_get(&ck_have[b]) != 0;
        // every frame of the batch has been decoded before: sections from its checkpoints; otherwise decode and leave checkpoints.
        // With group records next to the checkpoints (the first pass writes both) a later pass walks nothing at all.
        const bool recs = rec && nrec && rec_stride >= num_atoms && rec_failed && !flag_get(rec_failed) && g_opt.xtc_records.load();
        if (recs) {
            rc = vmd_hip_xtc_decode_wave_rec(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p, all ? 1 : 0, ck, nck, rec, nrec, rec_stride);
            if (all) st.rec_failed = rec_failed;
        } else
        rc = vmd_hip_xtc_decode_wave_ck(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p, all ? 1 : 0, ck, nck);
        if (!all) st.ck_mark = ck_have;
        else st.ck_clear = ck_have;
        st.sectioned = all;
    } else {
        rc = vmd_hip_xtc_decode_wave(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p);
    }
    e->prof_copy.end(stream);
    if (rc != 0) { vmd_fail("XTC decode kernel launch failed"); return -1; }
    for (size_t b = 0; b < nb; ++b) st.h_raw_status[b] = 99u;
    if (hipMemcpyAsync(st.h_raw_status, st.d_raw_status.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) { vmd_fail("device XTC decode failed"); return -1; }
    st.raw_pending = true;
    return 1;
}

// Frames stored as plain floats: the copy engine takes the batch's span of the mapped file, k_raw_f32 turns it into the frame layout.
// 1 = queued, 0 = not this way (no mapping, not pinnable, option off), -1 error.
static int raw_upload_f32(vmd_script_eval_t* e, vmd_script_eval_t::RawSlot& rs, vmd_trajectory_i* traj, const std::vector<vmd_raw_frame_t>& infos,
                          size_t num_atoms, size_t f0, size_t nb) {
    vmd_raw_mapped_view_t mv;
    if (!g_opt.raw_f32_device.load() || !g_opt.xtc_mapped.load() || !traj->raw_mapped_view || !traj->raw_mapped_view(traj->inst, &mv) ||
        mv.codec != VMD_RAW_CODEC_F32 || !mv.base || !mv.stream_offset) return 0;
    uint64_t lo64 = ~(uint64_t)0, hi64 = 0;
    for (size_t b = 0; b < nb; ++b) {
        const uint64_t so = mv.stream_offset[f0 + b];
        const vmd_raw_frame_t& fi = infos[b];
        if (fi.f32_stride == 0 || so + fi.nbytes > mv.bytes) return 0;
        for (int c = 0; c < 3; ++c)
            if (((so + fi.f32_offset[c]) & 3u) != 0 || fi.f32_offset[c] + 4ull * fi.f32_stride * (num_atoms - 1) + 4 > fi.nbytes) return 0;
        lo64 = std::min(lo64, so);
        hi64 = std::max(hi64, so + fi.nbytes);
    }
    const size_t lo = (size_t)(lo64 & ~(uint64_t)7), hi = (size_t)hi64;
    if (hi <= lo || !mapreg_pin(mv.base, mv.bytes, lo, hi)) return 0;
    HostTimer map_timer("host_raw_map");
    rs.f32.resize(nb);
    for (size_t b = 0; b < nb; ++b) {
        vmd_f32_frame_t& o = rs.f32[b];
        memset(&o, 0, sizeof(o));
        for (int c = 0; c < 3; ++c) o.offset[c] = mv.stream_offset[f0 + b] - lo + infos[b].f32_offset[c];
        o.stride = infos[b].f32_stride; o.flags = infos[b].f32_flags; o.scale = infos[b].f32_scale;
    }
    rs.info_bytes = (nb * sizeof(vmd_f32_frame_t) + 255) & ~(size_t)255;
    if (rs.info_bytes > rs.hcap) {
        if (rs.h) pool_give(rs.h);
        rs.h = nullptr; rs.hcap = 0;
        if (pool_take(kPinned, (void**)&rs.h, 2 * rs.info_bytes) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", 2 * rs.info_bytes); return -1; }
        rs.hcap = 2 * rs.info_bytes;
    }
    memcpy(rs.h, rs.f32.data(), nb * sizeof(vmd_f32_frame_t));
    rs.h_streams = mv.base + lo;
    const size_t span = hi - lo;
    if (!rs.d.ensure(rs.info_bytes + span + span / 8 + 64)) return -1;
    if (hipMemcpyAsync(rs.d.p, rs.h, nb * sizeof(vmd_f32_frame_t), hipMemcpyHostToDevice, e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the frame table failed"); return -1; }
    for (size_t a = lo; a < hi;) {                          // one copy per pinned window the span touches
        const size_t stop = std::min(hi, (a / kMapWindow + 1) * kMapWindow);
        if (hipMemcpyAsync(rs.d.p + rs.info_bytes + (a - lo), mv.base + a, stop - a, hipMemcpyHostToDevice, e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync from the mapped trajectory file failed"); return -1; }
        a = stop;
    }
    if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
    e->frames_mapped += nb;
    rs.state = 1;
    return 1;
}

// First half of the compressed path: read the bit streams of frames [f0, f0 + nb) into the slot's pinned block (load threads) and queue
// their DMA on copy_stream.  1 = queued (slot.uploaded recorded), 0 = a frame is not available raw, -1 error.
static int raw_upload(vmd_script_eval_t* e, RawSlot& rs, vmd_trajectory_i* traj, size_t num_atoms, size_t f0, size_t nb) {
    HostTimer host_timer("host_raw_upload");
    rs.state = 0; rs.f0 = f0; rs.nb = nb;
    rs.info.resize(nb);
    rs.cells.resize(nb);
    std::vector<vmd_raw_frame_t> infos(nb);
    size_t total = 0;
    for (size_t b = 0; b < nb; ++b) {                      // sizes first (no payload), then one pinned block for the batch
        vmd_frame_header_t hdr;
        if (!traj->load_raw(traj->inst, (int64_t)(f0 + b), &hdr, &infos[b], nullptr, 0) || hdr.num_atoms != num_atoms || infos[b].codec != infos[0].codec ||
            (infos[b].codec != VMD_RAW_CODEC_XTC && infos[b].codec != VMD_RAW_CODEC_F32)) { rs.state = -1; return 0; }
        rs.cells[b] = hdr.unitcell;
        vmd_xtc_frame_t& fi = rs.info[b];
        fi.precision = infos[b].precision;
        for (int k = 0; k < 3; ++k) { fi.minint[k] = infos[b].minint[k]; fi.maxint[k] = infos[b].maxint[k]; }
        fi.smallidx = infos[b].smallidx;
        fi.offset = total;
        fi.nbytes = infos[b].nbytes;
        total += ((size_t)infos[b].nbytes + 32 + 63) & ~(size_t)63;     // >= 32 readable bytes behind every stream, 64-byte aligned starts
    }
    rs.codec = infos[0].codec;
    if (rs.codec == VMD_RAW_CODEC_F32) {
        // plain floats (TRR, DCD): only out of the mapped file - copying them through a pinned block first is what load_frame does
        const int up = raw_upload_f32(e, rs, traj, infos, num_atoms, f0, nb);
        if (up <= 0) rs.state = -1;
        return up;
    }
    rs.info_bytes = (nb * sizeof(vmd_xtc_frame_t) + 255) & ~(size_t)255;
    // The file is mapped: the copy engine takes the batch's span of it as it lies there (frame headers in between and all), this
    // thread only writes the frame table.  r03m: reading the streams into the pinned block took 9.7 ms of a 15.8 ms c2 step (1 000
    // frames, 0.51 GB, ~53 GB/s whatever the thread count) and sat on the eval thread's critical path.
    vmd_raw_mapped_view_t mv;
    if (g_opt.xtc_mapped.load() && g_opt.xtc_device_decode.load() == 3 && traj->raw_mapped_view && traj->raw_mapped_view(traj->inst, &mv) &&
        mv.codec == VMD_RAW_CODEC_XTC && mv.base && mv.stream_offset) {
        bool usable = true;
        uint64_t prev_end = 0;
        for (size_t b = 0; b < nb && usable; ++b) {
            const uint64_t so = mv.stream_offset[f0 + b];
            usable = (so & 3u) == 0 && so >= prev_end && so + infos[b].nbytes <= mv.bytes;
            prev_end = so + infos[b].nbytes;
        }
        const size_t lo = usable ? (size_t)(mv.stream_offset[f0] & ~(uint64_t)7) : 0;
        const size_t hi = usable ? std::min<size_t>(mv.bytes, (size_t)prev_end + 40) : 0;
        if (usable && hi > lo && mapreg_pin(mv.base, mv.bytes, lo, hi)) {
            HostTimer map_timer("host_raw_map");
            for (size_t b = 0; b < nb; ++b) rs.info[b].offset = mv.stream_offset[f0 + b] - lo;
            if (rs.info_bytes > rs.hcap) {
                if (rs.h) pool_give(rs.h);
                rs.h = nullptr; rs.hcap = 0;
                if (pool_take(kPinned, (void**)&rs.h, 2 * rs.info_bytes) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", 2 * rs.info_bytes); return -1; }
                rs.hcap = 2 * rs.info_bytes;
            }
            memcpy(rs.h, rs.info.data(), nb * sizeof(vmd_xtc_frame_t));
            rs.h_streams = mv.base + lo;
            const size_t span = hi - lo;
            if (!rs.d.ensure(rs.info_bytes + span + span / 8 + 64)) return -1;        // >= 32 readable bytes behind the last stream even at the file's end
            if (hipMemcpyAsync(rs.d.p, rs.h, nb * sizeof(vmd_xtc_frame_t), hipMemcpyHostToDevice, e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the frame table failed"); return -1; }
            for (size_t a = lo; a < hi;) {                  // one copy per pinned window the span touches
                const size_t stop = std::min(hi, (a / kMapWindow + 1) * kMapWindow);
                if (hipMemcpyAsync(rs.d.p + rs.info_bytes + (a - lo), mv.base + a, stop - a, hipMemcpyHostToDevice, e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync from the mapped trajectory file failed"); return -1; }
                a = stop;
            }
            if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
            e->frames_mapped += nb;
            rs.state = 1;
            return 1;
        }
    }
    // the frame table travels at the head of the same pinned block: a second copy from pageable memory would stall this thread
    // behind the DMA already queued on copy_stream (r03m: 11.5 ms of a 17.4 ms c2 step were spent in this function)
    total += rs.info_bytes;
    if (total > rs.hcap) {
        if (rs.h) pool_give(rs.h);
        rs.h = nullptr; rs.hcap = 0;
        const size_t cap = total + total / 8;                            // frames of one trajectory differ by a few per cent
        if (pool_take(kPinned, (void**)&rs.h, cap) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", cap); return -1; }
        rs.hcap = cap;
    }
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), nb / 4));
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nb || !ok.load()) break;
            const vmd_xtc_frame_t& fi = rs.info[b];
            vmd_raw_frame_t info;
            unsigned char* dst = rs.h + rs.info_bytes + fi.offset;
            if (!traj->load_raw(traj->inst, (int64_t)(f0 + b), nullptr, &info, dst, (size_t)fi.nbytes) || info.nbytes != fi.nbytes) { ok = false; break; }
            memset(dst + fi.nbytes, 0, (((size_t)fi.nbytes + 32 + 63) & ~(size_t)63) - (size_t)fi.nbytes);
        }
    };
    {
        HostTimer read_timer("host_raw_read");
        memcpy(rs.h, rs.info.data(), nb * sizeof(vmd_xtc_frame_t));
        if (nthreads == 1) work();
        else {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        }
    }
    if (!ok.load()) { rs.state = -1; return 0; }           // let load_frame produce the real error message
    rs.h_streams = rs.h + rs.info_bytes;
    if (!rs.d.ensure(total + total / 8)) return -1;
    if (hipMemcpyAsync(rs.d.p, rs.h, total, hipMemcpyHostToDevice, e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the compressed batch failed"); return -1; }
    if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
    rs.state = 1;
    return 1;
}

// bring frames [f0, f0+nb) to the device (or alias them in place) through stage `st`: fills st.cells / st.h_boxes, queues
// the copies on copy_stream and records st.ready
static bool fetch_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, bool force_host = false, RawSlot* pre = nullptr) {
    st.raw_pending = false;
    hipStream_t ss = e->copy_stream;         // the stream this stage's `ready` is recorded on
    vmd_host_view_t hv;
    const vmd_host_view_t* hview = (!view && traj->host_view && traj->host_view(traj->inst, &hv)) ? &hv : nullptr;
    st.f0 = f0; st.nb = nb;
    if (view && view->cells_version != 0 && st.boxes_version == view->cells_version && st.boxes_cells == view->cells && st.boxes_f0 == f0 &&
        st.boxes_nb == nb) {
        // the same frames of an unchanged resident trajectory as last time (VIAMD re-evaluates after every script edit; a 10 000-frame
        // SDF step spent 0.1 ms here): cells, boxes (also the bounding-box ones of open axes) and their device copy are still valid
        st.base = view->base + f0 * view->frame_stride;
        st.frame_stride = view->frame_stride;
        st.row_stride = view->row_stride;
        HIP_OK(hipEventRecord(st.ready, e->copy_stream));
        return true;
    }
    st.boxes_version = 0;
    st.gboxes_ready = false;
    st.cells.resize(nb);
    st.h_boxes.resize(nb * 9);
    if (view) {
        st.base = view->base + f0 * view->frame_stride;
        st.frame_stride = view->frame_stride;
        st.row_stride = view->row_stride;
        for (size_t b = 0; b < nb; ++b) st.cells[b] = view->cells[f0 + b];
    } else if (hview) {
        // frames already sit in host memory in our layout: DMA them as one block, no load_frame copies
        const size_t need = nb * hview->frame_stride;
        if (!st.d.ensure(need)) return false;
        HIP_OK(hipMemcpyAsync(st.d.p, hview->base + f0 * hview->frame_stride, need * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
        st.base = st.d.p;
        st.frame_stride = hview->frame_stride;
        st.row_stride = hview->row_stride;
        for (size_t b = 0; b < nb; ++b) st.cells[b] = hview->cells[f0 + b];
    } else {
        const size_t npad = (num_atoms + 63) & ~(size_t)63;
        const size_t need = nb * 3 * npad;
        int raw = 0;
        vmd_raw_device_view_t rv;
        memset(&rv, 0, sizeof(rv));
        if (!force_host && traj->raw_device_view && traj->raw_device_view(traj->inst, &rv) && rv.codec == VMD_RAW_CODEC_XTC && rv.device == e->device) {
            // the compressed trajectory is resident in HBM: no host work, no PCIe - decode the batch where it lies
            for (size_t b = 0; b < nb; ++b) st.cells[b] = rv.cells[f0 + b];
            raw = launch_raw_decode(e, st, rv.base, (const vmd_xtc_frame_t*)rv.info + f0, num_atoms, nb, npad, ss,
                                    rv.ck ? (vmd_xtc_ck_t*)rv.ck + f0 * VMD_XTC_CK_MAX : nullptr, rv.nck ? rv.nck + f0 : nullptr, rv.ck_have ? rv.ck_have + f0 : nullptr,
                                    (rv.rec && rv.rec_stride) ? rv.rec + f0 * rv.rec_stride : nullptr, (rv.rec && rv.rec_stride) ? rv.nrec + f0 : nullptr, rv.rec_stride, rv.rec_failed);
            if (raw < 0) return false;
        } else if (!force_host && g_opt.xtc_device_decode.load() && traj->load_raw && !(e->raw_skip && !pre)) {
            // the bit streams were (or are now) sent ahead through a slot of the ring; decompression runs on its own stream
            RawSlot* rs = (pre && pre->f0 == f0 && pre->nb == nb && pre->state != 0) ? pre : &e->raw_slots[0];
            if (rs != pre && (raw = raw_upload(e, *rs, traj, num_atoms, f0, nb)) < 0) return false;
            if (rs->state == 1) {
                ss = e->decode_streams[pre ? (size_t)(pre - e->raw_slots) % vmd_script_eval_t::kDecodeStreams : 0];
                st.cells = rs->cells;
                HIP_OK(hipStreamWaitEvent(ss, rs->uploaded, 0));
                if (rs->codec == VMD_RAW_CODEC_F32) {
                    if (!st.d.ensure(nb * 3 * npad)) return false;
                    e->prof_copy.begin("raw_f32", ss);
                    KRN_OK(vmd_hip_raw_f32_decode(ss, rs->d_streams(), (const vmd_f32_frame_t*)rs->d.p, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad));
                    e->prof_copy.end(ss);
                    e->frames_device_decoded += nb;
                    raw = 1;
                } else {
                std::shared_ptr<CkCache> cc = ckcache_for(traj->inst, traj->num_frames(traj->inst), num_atoms, e->device);
                e->ck_cache = cc;
                st.ck_hold = cc;
                if (cc) {
                    // a frame's checkpoints count only for the very bytes they were written for
                    for (size_t b = 0; b < nb; ++b) {
                        const uint64_t sg = frame_signature(rs->info[b], rs->h_streams + rs->info[b].offset);
                        if (flag_get(&cc->sig[f0 + b]) != sg) { flag_set(&cc->sig[f0 + b], sg); flag_set(&cc->have[f0 + b], (uint8_t)0); }
                    }
                    raw = launch_raw_decode(e, st, rs->d_streams(), rs->d_info(), num_atoms, nb, npad, ss, cc->ck.p + f0 * VMD_XTC_CK_MAX, cc->nck.p + f0, cc->have.data() + f0,
                                            cc->rec_stride ? cc->rec.p + f0 * cc->rec_stride : nullptr, cc->rec_stride ? cc->nrec.p + f0 : nullptr, cc->rec_stride, &cc->rec_failed);
                } else {
                    raw = launch_raw_decode(e, st, rs->d_streams(), rs->d_info(), num_atoms, nb, npad, ss);
                }
                if (raw < 0) return false;
                }
            } else {
                raw = 0;
            }
        }
        if (raw == 1) {
            st.base = st.d.p;
            st.frame_stride = 3 * npad;
            st.row_stride = npad;
        } else {
            if (need > st.hcap) {
                if (st.h) pool_give(st.h);
                st.h = nullptr; st.hcap = 0;
                HIP_OK(pool_take(kPinned, (void**)&st.h, need * sizeof(float)));
                st.hcap = need;
            }
            if (!st.d.ensure(need)) return false;
            // md_trajectory_load_frame is called from all of VIAMD's pool threads at once (src/main.cpp:995-996 inside the
            // enkiTS range tasks), so the decoder behind it is re-entrant: decode the batch on a few threads
            const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), nb / 4));
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            std::mutex err_mtx;
            std::string err;
            auto work = [&]() {
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb || !ok.load()) break;
                    vmd_frame_header_t hdr;
                    memset(&hdr, 0, sizeof(hdr));
                    float* x = st.h + b * 3 * npad;
                    if (!traj->load_frame(traj->inst, (int64_t)(f0 + b), &hdr, x, x + npad, x + 2 * npad)) {
                        std::lock_guard<std::mutex> l(err_mtx);
                        if (ok.exchange(false)) {
                            char buf[96];
                            snprintf(buf, sizeof(buf), "trajectory load_frame(%zu) failed", f0 + b);
                            err = buf;
                            if (!g_last_error.empty()) err += ": " + g_last_error;     // the decoder's own message (this thread's)
                        }
                        break;
                    }
                    st.cells[b] = hdr.unitcell;
                }
            };
            if (nthreads == 1) work();
            else {
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
                work();
                for (auto& t : pool) t.join();
            }
            if (!ok.load()) return vmd_fail("%s", err.c_str());
            HIP_OK(hipMemcpyAsync(st.d.p, st.h, need * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
            st.base = st.d.p;
            st.frame_stride = 3 * npad;
            st.row_stride = npad;
        }
    }
    for (size_t b = 0; b < nb; ++b) {
        const vmd_unitcell_t& c = st.cells[b];
        const bool tri = c.xy != 0.0f || c.xz != 0.0f || c.yz != 0.0f;
        if (tri && ((c.flags & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL || !(c.x > 0.0f && c.y > 0.0f && c.z > 0.0f)))
            return vmd_fail("frame %zu: a triclinic unit cell must be periodic along all three axes (SPEC S3t)", f0 + b);
        const vmd_unitcell_t& c0 = st.cells[0];
        if (c.flags != c0.flags || tri != (c0.xy != 0.0f || c0.xz != 0.0f || c0.yz != 0.0f))
            return vmd_fail("frame %zu: periodicity / cell type changes inside the trajectory", f0 + b);
        float* hb = &st.h_boxes[9 * b];
        hb[0] = c.x; hb[1] = c.y; hb[2] = c.z;
        hb[3] = 1.0f / c.x; hb[4] = 1.0f / c.y; hb[5] = 1.0f / c.z;      // SPEC S2: invL = fl(1.0f / L)
        hb[6] = c.xy; hb[7] = c.xz; hb[8] = c.yz;
    }
    if (!st.d_boxes.upload(st.h_boxes.data(), nb * 9, ss)) return false;
    HIP_OK(hipEventRecord(st.ready, ss));
    if (view && view->cells_version != 0) { st.boxes_cells = view->cells; st.boxes_f0 = f0; st.boxes_nb = nb; st.boxes_version = view->cells_version; }
    return true;
}

// A stage whose frames were decompressed on the device: wait for its `ready` event (the decode ran under the previous batch's
// kernels, so this rarely waits) and look at the status words.  A stream the device rejects - damaged, or a packed number above
// 2^64 - sends the whole batch through the host reader, which decides and reports.
static bool settle_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, size_t num_atoms) {
    if (!st.raw_pending) return true;
    HIP_OK(hipEventSynchronize(st.ready));
    e->prof_copy.resolve();
    st.raw_pending = false;
    std::shared_ptr<CkCache> hold = std::move(st.ck_hold);      // released when this function is done with ck_mark / ck_clear / rec_failed
    bool good = true;
    for (size_t b = 0; b < st.nb; ++b) if (st.h_raw_status[b] != 0) good = false;
    if (good) {
        if (st.ck_mark) for (size_t b = 0; b < st.nb; ++b) flag_set(&st.ck_mark[b], (uint8_t)1);
        st.ck_mark = nullptr;
        st.ck_clear = nullptr;
        st.rec_failed = nullptr;
        e->frames_device_decoded += st.nb;
        if (st.sectioned) e->frames_section_decoded += st.nb;
        return true;
    }
    st.ck_mark = nullptr;
    // checkpoints that did not describe these streams (a sidecar table that passed the signature test and still lies): the frames
    // walk from bit 0 again next time
    if (st.ck_clear) for (size_t b = 0; b < st.nb; ++b) flag_set(&st.ck_clear[b], (uint8_t)0);
    st.ck_clear = nullptr;
    if (st.rec_failed) flag_set(st.rec_failed, true);          // the records did not describe these streams: never again for this trajectory
    st.rec_failed = nullptr;
    return fetch_stage(e, st, traj, nullptr, num_atoms, st.f0, st.nb, true);
}

// synchronous variant used for single frames (reference pose, vis payload)
static bool fetch_batch(vmd_script_eval_t* e, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, BatchSrc* src) {
    Stage& st = e->stages[0];
    if (!fetch_stage(e, st, traj, view, num_atoms, f0, nb)) return false;
    if (!settle_stage(e, st, traj, num_atoms)) return false;
    HIP_OK(hipEventSynchronize(st.ready));
    src->base = st.base; src->frame_stride = st.frame_stride; src->row_stride = st.row_stride;
    return true;
}

static uint32_t batch_pbc(const Stage& st) {
    const vmd_unitcell_t& c = st.cells[0];
    uint32_t f = c.flags & VMD_UNITCELL_PBC_ALL;
    if (!(c.x > 0.0f)) f &= ~VMD_UNITCELL_PBC_X;
    if (!(c.y > 0.0f)) f &= ~VMD_UNITCELL_PBC_Y;
    if (!(c.z > 0.0f)) f &= ~VMD_UNITCELL_PBC_Z;
    if (c.xy != 0.0f || c.xz != 0.0f || c.yz != 0.0f) f |= 8u;         // triclinic (kernels: VMD_PBC_TRICLINIC)
    return f;
}

// Batches with open axes: the grid spans the bounding box of the batch's atoms.  Fills st.h_gboxes / st.d_gboxes with
// {extent or L, inverse, origin or 0} per frame (one bbox kernel + one small readback per batch).
static bool prepare_open_boxes(vmd_script_eval_t* e, Stage& st, size_t nb, uint32_t pbc, size_t num_atoms) {
    if (st.gboxes_ready) return true;
    if (!st.d_bbox.ensure(nb * 6) || !st.d_gboxes.ensure(nb * 9)) return false;
    st.h_bbox.resize(nb * 6);
    KRN_OK(vmd_hip_bbox(e->stream, st.base, st.frame_stride, st.row_stride, (int)nb, (int)num_atoms, st.d_bbox.p));
    HIP_OK(hipMemcpyAsync(st.h_bbox.data(), st.d_bbox.p, nb * 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    st.h_gboxes = st.h_boxes;
    for (size_t b = 0; b < nb; ++b) {
        float* g = &st.h_gboxes[9 * b];
        for (int a = 0; a < 3; ++a) {
            g[6 + a] = 0.0f;
            if (pbc & (1u << a)) continue;
            const float lo = st.h_bbox[6 * b + a], hi = st.h_bbox[6 * b + 3 + a];
            const float pad = std::max(1.0e-2f, 1.0e-3f * (hi - lo));
            g[6 + a] = lo - pad;                    // origin
            g[a] = (hi - lo) + 2.0f * pad;          // extent
            g[3 + a] = 1.0f / g[a];
        }
    }
    if (!st.d_gboxes.upload(st.h_gboxes.data(), nb * 9, e->stream)) return false;
    st.gboxes_ready = true;
    return true;
}

// pencil grid for a batch and cutoff; false when the batch cannot use the grid kernel.  `boxes` = st.h_boxes, or
// st.h_gboxes when some axes are open (pbc bits clear): those carry the bounding-box extent instead of a cell edge.
static bool choose_grid(const std::vector<float>& boxes, uint32_t pbc, size_t nb, float rmax, vmd_grid_t* g, bool dense_lanes = false) {
    if (g_opt.force_brute) return false;
    const bool tri = (pbc & 8u) != 0;
    if (tri && (pbc & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL) return false;
    // smallest extent per axis over the batch, measured perpendicular to the cell faces (SPEC S3t: a triclinic cell's
    // pencils are sheared, what has to be >= rmax is their width w_k = 1 / |reciprocal vector k|)
    float wmin[3] = {3.4e38f, 3.4e38f, 3.4e38f}, Lxmin = 3.4e38f;
    for (size_t b = 0; b < nb; ++b) {
        const float* q = &boxes[9 * b];
        const double Lx = q[0], Ly = q[1], Lz = q[2];
        const double xy = tri ? q[6] : 0.0, xz = tri ? q[7] : 0.0, yz = tri ? q[8] : 0.0;
        const double wx = Lx / std::sqrt(1.0 + (xy / Ly) * (xy / Ly) + ((xy * yz - Ly * xz) / (Ly * Lz)) * ((xy * yz - Ly * xz) / (Ly * Lz)));
        const double wy = Ly / std::sqrt(1.0 + (yz / Lz) * (yz / Lz));
        wmin[0] = std::min(wmin[0], (float)wx); wmin[1] = std::min(wmin[1], (float)wy); wmin[2] = std::min(wmin[2], (float)Lz);
        Lxmin = std::min(Lxmin, q[0]);
    }
    // periodic axes: the minimum image must be unique for every hit (rmax < w/2 with margin); open axes: no restriction
    for (int a = 0; a < 3; ++a) if ((pbc & (1u << a)) && !(rmax * 2.0f * 1.001f < wmin[a])) return false;
    int n[3];
    const int sy = g_opt.pencil_split_y.load();
    const int split[3] = {1, sy <= 0 ? (dense_lanes ? 2 : 1) : std::min(4, sy), std::max(1, std::min(4, g_opt.pencil_split_z.load()))};
    vmd_hip_set_pencil_reach(split[1], split[2]);
    for (int a = 1; a < 3; ++a) {
        const float redge = rmax / (float)split[a];
        int k = (int)std::floor(wmin[a] / redge);
        // head room between the pencil edge and rmax: wrapped coordinates are exact to ~1e-6 of the edge; on an open axis
        // coordinates keep their raw magnitude (possibly far from the origin), so leave ten times more
        const float edge_margin = (pbc & (1u << a)) ? 0.9999f : 0.999f;
        while (k > 1 && ((float)k / wmin[a]) * redge > edge_margin) k -= 1;
        if (pbc & (1u << a)) { if (k < 2) return false; }
        else k = std::max(k, 1);
        n[a] = std::min(k, 1024);
    }
    const float cx = rmax / (float)std::max(1, g_opt.nxf_divisor.load());
    int nxf = (int)std::floor(Lxmin / cx);
    nxf = std::max(1, std::min(nxf, 4096));
    // keep the cell table small enough for the LDS-resident build (24576 counters) as long as the fine cells stay <= rmax/3
    // (the single-level builds only: the two-level build keeps a table of pencils, not of cells)
    const int nxf_lds = 24575 / (n[1] * n[2]);
    vmd_grid_t probe{nxf, n[1], n[2], 0};
    if (!vmd_hip_cells_pencil_ok(probe) && nxf > nxf_lds && nxf_lds >= (int)std::ceil(3.0f * Lxmin / rmax)) nxf = nxf_lds;
    g->nxf = nxf; g->ny = n[1]; g->nz = n[2];
    const long long ncell = (long long)nxf * n[1] * n[2];
    if (ncell > (1ll << 26)) return false;
    g->ncell = (int32_t)ncell;
    return true;
}

// Bucket capacities of the two-level build for selection `s` on the pencils of grid `g`: per-pencil maximum over the first and
// last (up to) 4 frames of the batch x margin + a few standard deviations.  One small readback, then kept for the eval's
// lifetime (frames of one trajectory look alike; a bucket that overflows later is caught by the device flag and re-measured).
static bool ensure_pencil_caps(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb, const vmd_grid_t& g) {
    if (!s->pen_off.empty() && s->pen_ny == g.ny && s->pen_nz == g.nz) return true;
    if (!s->pen_off.empty()) {                 // keep what was measured for the layout we are leaving
        bool known = false;
        for (auto& c : s->caps_cache) known = known || (c.ny == s->pen_ny && c.nz == s->pen_nz);
        if (!known) {
            if (s->caps_cache.size() >= 4) s->caps_cache.erase(s->caps_cache.begin());
            s->caps_cache.push_back({s->pen_ny, s->pen_nz, s->cap_max, s->total_cap, s->pen_off});
        }
    }
    for (auto& c : s->caps_cache) {
        if (c.ny != g.ny || c.nz != g.nz) continue;
        s->pen_off = c.pen_off; s->cap_max = c.cap_max; s->total_cap = c.total_cap; s->pen_ny = c.ny; s->pen_nz = c.nz;
        return s->d_pen_off.upload(s->pen_off.data(), s->pen_off.size(), e->stream);      // pageable source: the copy is staged before the call returns
    }
    const int npen = g.ny * g.nz, nsel = (int)s->idx.size();
    // after an overflow: every frame of the batch (exact populations), otherwise the first and last 4
    const bool exhaustive = s->overflows > 0 || nb <= 8;
    const size_t S = exhaustive ? nb : 4, rows = exhaustive ? nb : 8;
    if (!e->d_pen_sample.ensure(rows * (size_t)npen)) return false;
    std::vector<uint32_t> h(rows * (size_t)npen);
    KRN_OK(vmd_hip_cells_pencil_count(e->stream, src.base, src.frame_stride, src.row_stride, d_boxes, pbc, (int)S, s->d_idx.p, nsel, g, e->d_pen_sample.p));
    if (!exhaustive) {
        const size_t tail = nb - S;
        KRN_OK(vmd_hip_cells_pencil_count(e->stream, src.base + tail * src.frame_stride, src.frame_stride, src.row_stride, d_boxes + 9 * tail, pbc, (int)S,
                                          s->d_idx.p, nsel, g, e->d_pen_sample.p + S * (size_t)npen));
    }
    HIP_OK(hipMemcpyAsync(h.data(), e->d_pen_sample.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    s->pen_off.assign((size_t)npen + 1, 0);
    s->cap_max = 0;
    uint64_t total = 0;
    for (int p = 0; p < npen; ++p) {
        uint32_t m = 0;
        for (size_t k = 0; k < rows; ++k) m = std::max(m, h[k * npen + p]);
        uint32_t cap = (uint32_t)std::ceil((double)m * s->cap_margin + 6.0 * std::sqrt((double)m)) + 16;
        cap = (cap + 3u) & ~3u;
        s->pen_off[p] = (uint32_t)total;
        total += cap;
        s->cap_max = std::max<int>(s->cap_max, (int)cap);
    }
    if (total > 0x7fffffffull) { s->pen_off.clear(); s->overflows = 99; return true; }     // not a job for the buckets
    s->pen_off[npen] = (uint32_t)total;
    s->total_cap = (int)total;
    s->pen_ny = g.ny; s->pen_nz = g.nz;
    return s->d_pen_off.upload(s->pen_off.data(), s->pen_off.size(), e->stream);
}

static bool build_selection(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb, const vmd_grid_t& g) {
    if (s->built && s->built_grid.nxf == g.nxf && s->built_grid.ny == g.ny && s->built_grid.nz == g.nz) return true;
    const int nsel = (int)s->idx.size();
    s->nsel_pad = (nsel + 63) & ~63;
    if (!s->cell_start.ensure(nb * (size_t)(g.ncell + 1)) || !s->sorted.ensure(nb * 3 * (size_t)s->nsel_pad + 64)) return false;   // +64: the pair kernel prefetches past a segment
    s->used_pencil = false;
    // two-level build through per-pencil buckets (one read of the frame, coalesced sorted rows); single-level builds otherwise
    // A small selection (a solute: the 2 000-atom blob of config 5) is not spread evenly over the pencils and wanders through them as the
    // trajectory goes on: capacities measured on one batch overflow in the next, and every overflow repeats the batch's pair passes.  It is
    // sorted by ONE block per frame in LDS instead (k_cells_fused: no buckets, nothing to overflow), which costs such a selection nothing.
    const bool small = nsel <= g_opt.cells_small.load() && vmd_hip_cells_fused_ok(g, nsel);
    if (vmd_hip_cells_pencil_ok(g) && s->overflows < 3 && !small) {
        if (!ensure_pencil_caps(e, s, src, d_boxes, pbc, nb, g)) return false;
        if (!s->pen_off.empty() && s->cap_max <= vmd_hip_cells_pencil_cap_max()) {
            const size_t npen = (size_t)g.ny * g.nz;
            if (!s->pen_count.ensure(nb * npen) || !s->pen_start.ensure(nb * (npen + 1)) || !s->bucket.ensure(nb * (size_t)s->total_cap * 4)) return false;
            e->prof.begin("cells_build", e->stream);
            vmd_hip_set_cells_overflow_bit(s->overflow_bit);
            KRN_OK(vmd_hip_cells_build_pencil(e->stream, src.base, src.frame_stride, src.row_stride, d_boxes, pbc, (int)nb, s->d_idx.p, nsel, s->nsel_pad, g,
                                              s->d_pen_off.p, s->total_cap, s->cap_max, s->pen_count.p, s->pen_start.p, s->bucket.p, e->d_overflow.p,
                                              s->cell_start.p, s->sorted.p));
            e->prof.end(e->stream);
            s->built = true; s->built_grid = g; s->used_pencil = true;
            return true;
        }
    }
    if (!s->cell_count.ensure(nb * (size_t)(g.ncell + 1)) || !s->rank.ensure(nb * vmd_hip_cells_scratch_words(g, nsel))) return false;
    const bool use_aos = g_opt.cells_aos != 0;
    if (use_aos && !s->aos.ensure(nb * 4 * (size_t)s->nsel_pad)) return false;
    e->prof.begin("cells_build", e->stream);
    KRN_OK(vmd_hip_cells_build(e->stream, src.base, src.frame_stride, src.row_stride, d_boxes, pbc, (int)nb, s->d_idx.p, nsel,
                               s->nsel_pad, g, s->cell_count.p, s->rank.p, s->cell_start.p, s->sorted.p, use_aos ? s->aos.p : nullptr));
    e->prof.end(e->stream);
    s->built = true;
    s->built_grid = g;
    return true;
}

static size_t auto_batch(const vmd_script_eval_t* e, size_t num_atoms, bool staged) {
    const int forced = g_opt.batch_frames;
    if (forced > 0) return (size_t)forced;
    // scratch per frame: a selection that takes part in a pair pass holds ~40 B per atom (sorted rows, bucket records, tables);
    // host trajectories add the staged frame itself; SDF / distance properties need a few hundred bytes
    size_t per_frame = staged ? 12 * num_atoms : 0;
    std::vector<char> used(e->sels.size(), 0);
    for (auto& g : e->rdf_groups) for (auto& ps : g.passes) { used[ps.sel_a] = 1; used[ps.sel_b] = 1; }
    for (size_t i = 0; i < e->sels.size(); ++i) if (used[i]) per_frame += 40 * e->sels[i]->idx.size();
    for (auto& p : e->props) per_frame += p->prop.kind == PROP_SDF ? 64 * p->prop.K : (p->prop.kind == PROP_DIST ? 4 * p->dim1 : 0);
    // 288 GB of HBM: a 16 GB scratch budget holds the 1 000 frames of the 1M-atom RDF (333k selected atoms) in ONE batch
    // (every batch boundary costs ~1 ms of host round trips against ~37 ms of kernels per 500 frames)
    size_t B = (size_t)(16ull << 30) / std::max<size_t>(per_frame, 1);
    // pair passes are long (a 1 024-frame batch of the 1M-atom RDF runs ~90 ms: interrupts are polled between batches); scripts
    // without them stream whole frames at HBM speed and take much larger batches, so that launches, the alignment kernel's
    // latency and the per-batch synchronisation stay small against the stream (grid.y = frames of the batch <= 65535)
    const size_t cap = e->rdf_groups.empty() ? 16384 : 1024;
    B = std::max<size_t>(1, std::min<size_t>(B, cap));
    return B;
}

// one kernel batch: frames [f0, f0 + nb); blk >= 0 when the batch is made of the whole frame blocks blk .. blk + nblk - 1 of this eval
// (filtered evaluation: each block accumulates into its own partial; they share the batch's cell build and synchronisation)
struct Batch { size_t f0, nb; long blk; size_t nblk; };

static void plan_batches(const vmd_script_eval_t* e, size_t beg, size_t end, size_t Bmax, std::vector<Batch>* out) {
    auto even = [&](size_t a, size_t b) {
        const size_t total = b - a;
        if (!total) return;
        const size_t nbatch = (total + Bmax - 1) / Bmax;
        const size_t B = (total + nbatch - 1) / nbatch;
        for (size_t f = a; f < b; f += B) out->push_back({f, std::min(B, b - f), -1, 0});
    };
    const size_t S = e->block_frames;
    if (S == 0) { even(beg, end); return; }
    const bool super = g_opt.block_superbatch.load() != 0;
    // whole blocks that fit one batch become a batch of blocks (their partials are kept), everything else is a plain piece
    size_t run = beg;                          // start of the pending plain piece
    for (size_t f = beg; f < end;) {
        const size_t blk = f / S;
        const size_t bend = std::min((blk + 1) * S, e->num_frames);
        if (f == blk * S && bend <= end && bend - f <= Bmax) {
            even(run, f);
            Batch* last = out->empty() ? nullptr : &out->back();
            if (super && last && last->blk >= 0 && last->f0 + last->nb == f && last->nb + (bend - f) <= Bmax) { last->nb += bend - f; last->nblk += 1; }
            else out->push_back({f, bend - f, (long)blk, 1});
            f = bend; run = f;
        } else {
            f = std::min(bend, end);
        }
    }
    even(run, end);
}

// filtered evaluation: merge every ready block of the source eval that lies inside [beg, end) into this eval's accumulators
// and return the sub-ranges that still have to be computed
// block_ready[b] != 0: block b's partial (d_blocks, block_weights64, temporal rows) is complete.  Where its temporal rows are: a block
// evaluated by a plain call has them in `values`; a block evaluated AHEAD (read-ahead, spec) or adopted from a source has them in the side
// buffer `ahead_values` until it is committed - `values` only ever shows frames somebody asked for.  An eval that takes blocks from a source
// (reuse_blocks, ra_adopt_blocks) must read the rows where they are: a filtered evaluation running BESIDE its source (src/main.cpp:982-1039
// enqueues both) used to copy rows of blocks the source had evaluated ahead but not yet committed out of `values` - zeros
// (tests/native/stress_readahead.cpp, "beside").
enum : uint8_t { BLOCK_ROWS_IN_PLACE = 1, BLOCK_ROWS_AHEAD = 2 };
static const float* block_rows(const vmd_script_eval_t* src, const PropState* q, size_t blk) {
    return src->block_ready[blk].load() == BLOCK_ROWS_AHEAD && q->ahead_values.size() == q->values.size() ? q->ahead_values.data() : q->values.data();
}

static bool reuse_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t beg, size_t end, std::vector<std::pair<size_t, size_t>>* todo) {
    vmd_script_eval_t* src = e->source;
    if (!src) { todo->push_back({beg, end}); return true; }
    std::lock_guard<std::mutex> lock(src->mtx);   // order: own mutex, then the source's (a source never locks its users)
    if (src->block_frames == 0 || src->blocks_inst != traj_inst) { todo->push_back({beg, end}); return true; }      // (looked up under its mutex: read-ahead may be giving it blocks right now)
    const size_t S = src->block_frames;
    size_t run = beg, reused = 0;
    for (size_t f = beg; f < end;) {
        const size_t blk = f / S;
        const size_t bend = std::min((blk + 1) * S, e->num_frames);
        if (f == blk * S && bend <= end && blk < src->num_blocks && src->block_ready[blk]) {
            if (run < f) todo->push_back({run, f});
            for (size_t i = 0; i < e->props.size(); ++i) {
                PropState* p = e->props[i].get();
                const PropState* q = src->props[i].get();
                if (p->ncounts) {
                    KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, q->d_blocks.p + blk * p->ncounts, p->ncounts));
                    if (p->prop.kind == PROP_RDF)
                        for (size_t k = 0; k < p->ncounts; ++k) p->weights64[k] += q->block_weights64[blk * p->ncounts + k];
                } else {
                    memcpy(&p->values[f * p->dim1], block_rows(src, q, blk) + f * p->dim1, (bend - f) * p->dim1 * sizeof(float));
                }
                p->dirty = true;
            }
            for (size_t g = f; g < bend; ++g) e->frame_mask[g] = 1;
            reused += bend - f;
            f = bend; run = f;
        } else {
            f = std::min(bend, end);
        }
    }
    if (run < end) todo->push_back({run, end});
    if (reused) {
        HIP_OK(hipStreamSynchronize(e->stream));   // the source's partials are read before its mutex is released
        e->frames_done += reused;
        e->frames_reused += reused;
        { HostTimer host_timer("host_refresh");
          for (auto& p : e->props) if (p->prop.kind == PROP_RDF) { if (!refresh_distribution(e, p.get())) return false; } }
    }
    return true;
}

// a sharded device trajectory keeps only its block of frames behind the view; other frames (frame 0 for the SDF reference
// pose) come through load_frame
// (resident_beg, resident_end) = (0, 0) means "every frame"; any other pair is a shard, and beg == end != 0 is an EMPTY shard (a rank
// that owns no frame: 4 ranks on 5 frames) - nothing is resident then, not everything
static bool view_sharded(const vmd_device_view_t& view) { return view.resident_beg != 0 || view.resident_end != 0; }
static bool view_holds(bool have_view, const vmd_device_view_t& view, size_t frame) {
    return have_view && (!view_sharded(view) || (frame >= view.resident_beg && frame < view.resident_end));
}

// evaluates frames [frame_beg, frame_end) in large batches; returns false on interrupt (empty error) or failure
// views: bring the host views (values / weights / volume / aggregates) up to date before returning; false = the caller does it later
// (refresh_views), the device accumulators and the frame mask are complete either way
// spec (read-ahead): [frame_beg, frame_end) is a run of whole frame blocks; every block is evaluated into its own partial and
// NOTHING else changes - no add into the totals, no frame mask, no frames_done, no normalisation weights outside the block's own, no view
// (temporal rows are written: a frame's row is the same whenever it is computed, and nobody reads it before its mask bit is set)
static bool process_range_locked(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end, bool views, bool spec) {
    HIP_OK(hipSetDevice(eval->device));
    vmd_script_eval_t* e = eval;
    const size_t num_atoms = traj->num_atoms(traj->inst);
    if (traj->num_frames(traj->inst) < frame_end) return vmd_fail("trajectory has fewer frames than the requested range");
    if (!check_atoms(e, num_atoms)) return false;
    if (!upload_static(e, sys, num_atoms)) return false;

    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    if (have_view && view_sharded(view) && frame_beg < frame_end && (frame_beg < view.resident_beg || frame_end > view.resident_end))
        return vmd_fail("frames [%u, %u) are not resident on this rank (its shard holds [%zu, %zu))", frame_beg, frame_end, view.resident_beg, view.resident_end);

    // SDF reference pose: structure 0 at trajectory frame 0 (SPEC S5)
    for (auto& p : e->props) {
        if (p->prop.kind != PROP_SDF || p->ref_pose_ready) continue;
        BatchSrc src;
        if (!fetch_batch(e, traj, view_holds(have_view, view, 0) ? &view : nullptr, num_atoms, 0, 1, &src)) return false;
        KRN_OK(vmd_hip_sdf_ref_pose(e->stream, src.base, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), p->d_structs.p, p->d_mass.p,
                                    (int)p->prop.m, p->d_ref_pose.p, p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr));
        HIP_OK(hipStreamSynchronize(e->stream));
        p->ref_pose_ready = true;
    }

    // frames served from the block partials of the source eval (filtered evaluation), the rest is computed
    std::vector<std::pair<size_t, size_t>> segments;
    if (spec) segments.push_back({frame_beg, frame_end});         // a region's blocks are adopted from the source by the region leader, or evaluated here
    else if (!reuse_blocks(e, traj_id(traj), frame_beg, frame_end, &segments)) return false;
    if (e->block_frames) e->blocks_inst = traj_id(traj);

    // compressed frames for the device decoder travel two batches ahead through a ring of three slots (RawSlot)
    vmd_host_view_t hv_probe;
    vmd_raw_device_view_t rv_probe;
    bool raw_ring = !have_view && traj->load_raw && g_opt.xtc_device_decode.load() != 0 &&
                    !(traj->host_view && traj->host_view(traj->inst, &hv_probe)) &&
                    !(traj->raw_device_view && traj->raw_device_view(traj->inst, &rv_probe));
    vmd_raw_mapped_view_t mv_probe;
    memset(&mv_probe, 0, sizeof(mv_probe));
    const bool have_map = raw_ring && g_opt.xtc_mapped.load() && traj->raw_mapped_view && traj->raw_mapped_view(traj->inst, &mv_probe);
    // plain-float files (TRR, DCD) take the ring only out of a mapping: without one their frames go through load_frame as before
    bool f32_ring = false;
    if (raw_ring && frame_beg < frame_end) {
        vmd_raw_frame_t probe;
        memset(&probe, 0, sizeof(probe));
        if (!traj->load_raw(traj->inst, (int64_t)frame_beg, nullptr, &probe, nullptr, 0)) raw_ring = false;
        else if (probe.codec == VMD_RAW_CODEC_F32) { f32_ring = have_map && mv_probe.codec == VMD_RAW_CODEC_F32 && g_opt.raw_f32_device.load(); raw_ring = f32_ring; }
    }
    e->raw_skip = traj->load_raw && !raw_ring;              // fetch_stage: do not ask this trajectory for raw frames batch by batch
    // how many batches the bit streams run ahead of the kernels (one more than the decoder, which runs two ahead).  Copied through
    // pinned blocks (host threads read them, this thread waits): 3.  Taken out of the mapped file by the copy engine alone: as many as the ring holds minus the one being decoded - the
    // DMAs then queue back to back and PCIe never waits for this thread (r03n: 12.4 ms per c2 step against 9.4 ms of transfers).
    const size_t raw_ahead = (raw_ring && have_map && (f32_ring || g_opt.xtc_device_decode.load() == 3)) ? vmd_script_eval_t::kRawSlots - 1 : 3;      // always > stage_ahead
    auto slot_of = [&](size_t bi) -> RawSlot* { return raw_ring ? &e->raw_slots[bi % vmd_script_eval_t::kRawSlots] : nullptr; };
    // batches decompressed on the device while the previous batch is in the pair kernel: the persistent pair grid leaves room for them
    const bool device_decode = raw_ring || (!have_view && traj->raw_device_view && traj->raw_device_view(traj->inst, &rv_probe));

    bool cold_walk = false;
    // frames per launch: as many as the scratch budget allows, split evenly so that no small tail batch is left
    size_t Bmax = auto_batch(e, num_atoms, !have_view);
    const vmd_device_view_t* vw = have_view ? &view : nullptr;
    // host trajectories are staged in smaller batches so that load_frame of batch k+1 overlaps the kernels of batch k
    // ... except when the batches are decompressed on the device (profiles/r03_xtc_device_decode.txt).  The FIRST decode of a frame
    // walks its whole bit stream - a latency-bound chain, 7 ms per batch whether it holds 64 or 1 000 synthetic frames - so first passes
    // use large batches (4 x stage_frames).  It leaves checkpoints; every later pass decodes in sections at 2 - 3 us per frame, and
    // then small batches win from a file (the PCIe trip of batch k + 1 hides under decode + pair kernel of batch k: 64.6k frames/s
    // with batches of 128 against 51.3k with 512) and one large batch from HBM (103.8k against 98.8k).
    if (!have_view && g_opt.batch_frames <= 0) {
        // stage_frames is quoted for a 100 000-atom system (a 154 MB float stage); larger systems get proportionally fewer frames per
        // batch - r03u: 1M atoms in batches of 128 frames (0.64 GB of bit streams each) spent 15 of 34 ms waiting for the first batch
        const size_t npad_s = (num_atoms + 63) & ~(size_t)63;
        const size_t S0 = (size_t)std::max(1, g_opt.stage_frames.load());
        const size_t S = std::max<size_t>(1, std::min<size_t>(S0, S0 * 100032 / std::max<size_t>(npad_s, 1)));
        bool warm = f32_ring;                               // does the first frame of the range have checkpoints already?  (plain floats need none)
        if (!f32_ring && device_decode && g_opt.xtc_checkpoints.load() && frame_beg < frame_end) {
            if (raw_ring) {
                std::lock_guard<std::mutex> l(g_ck_mtx);
                auto it = g_ck_store.find(CkKey(traj->inst, e->device));
                warm = it != g_ck_store.end() && it->second && it->second->frames == traj->num_frames(traj->inst) && it->second->atoms == num_atoms &&
                       it->second->device == e->device && it->second->have.size() > frame_beg && flag_get(&it->second->have[frame_beg]);
            }
            else warm = rv_probe.ck_have && flag_get(&rv_probe.ck_have[frame_beg]);
        }
        // a first pass out of a mapped file keeps four walks in flight (stage_ahead below): half-size batches, twice as many
        cold_walk = raw_ring && !f32_ring && !warm && have_map && g_opt.xtc_device_decode.load() == 3 && g_opt.xtc_cold_streams.load() != 0;
        // (a walk takes as long for one frame as for a thousand - 7 ms for a c2 frame, 70 ms for 1M atoms -: never more launches than
        // decode streams for a short range)
        const size_t cold_b = std::max<size_t>(2 * S, (frame_end - frame_beg + vmd_script_eval_t::kDecodeStreams - 1) / vmd_script_eval_t::kDecodeStreams);
        Bmax = std::min<size_t>(Bmax, !device_decode ? S : (raw_ring ? (warm ? S : (cold_walk ? cold_b : 4 * S)) : (warm ? 8 * S : 4 * S)));
    }
    std::vector<Batch> batches;
    for (auto& sg : segments) plan_batches(e, sg.first, sg.second, Bmax, &batches);
    if (spec) for (auto& b : batches) if (b.blk < 0) return vmd_fail("read-ahead: region [%u, %u) is not made of whole frame blocks", frame_beg, frame_end);
    // From a file the first batch has to cross PCIe and be decompressed before any kernel can start, and nothing overlaps the last
    // batch's kernels (r03p timeline: 1.7 ms of a 12.3 ms c2 step before the first pair kernel, one DMA = 1.13 ms per 128 frames).
    // Option xtc_ramp: the run starts with an eighth and a quarter of a batch and ends with a quarter.  Measured (r03o): the shorter
    // fill is paid back by the pair kernel's lower efficiency on small launches - 81.5k frames/s either way, so it is off.
    if (raw_ring && g_opt.xtc_ramp.load() && batches.size() >= 3) {
        std::vector<Batch> ramped;
        auto carve_front = [&](Batch& b, size_t n) { ramped.push_back({b.f0, n, -1, 0}); b.f0 += n; b.nb -= n; };
        Batch first = batches.front(), last = batches.back();
        if (first.blk < 0 && first.nb >= 64) { carve_front(first, first.nb / 8); carve_front(first, first.nb / 3); }
        ramped.push_back(first);
        for (size_t i = 1; i + 1 < batches.size(); ++i) ramped.push_back(batches[i]);
        if (last.blk < 0 && last.nb >= 64) { const size_t tail = last.nb / 4; ramped.push_back({last.f0, last.nb - tail, -1, 0}); ramped.push_back({last.f0 + last.nb - tail, tail, -1, 0}); }
        else ramped.push_back(last);
        batches.swap(ramped);
    }

    bool completed = true;
    // Batches staged ahead of the one being evaluated: one; optionally two when they are decompressed on the device (the decoder runs
    // UNDER the pair kernel, in the wave slots that kernel leaves).  r03n/r03p: the wait in settle_stage is the pipeline filling at the
    // start of a range, not a late decoder - two ahead measures the same, so one is the default.
    // A FIRST pass (no checkpoints yet) out of a mapped file: the walks of up to four batches run side by side (decode_streams).
    size_t stage_ahead = (raw_ring && g_opt.xtc_device_decode.load() == 3 && g_opt.xtc_decode_ahead.load() >= 2) ? 2 : 1;
    if (cold_walk) stage_ahead = std::min<size_t>(vmd_script_eval_t::kDecodeStreams, raw_ahead - 1);
    auto stage_of = [&](size_t bi) -> Stage& { return e->stages[bi % (stage_ahead + 1)]; };
    struct BlocksGuard {
        int old = -1;
        ~BlocksGuard() { if (old > 0) vmd_hip_set_rdf_blocks(old); }
    } blocks_guard;
    if (device_decode && batches.size() > 1 && g_opt.rdf_blocks_decode.load() >= 8) {
        blocks_guard.old = vmd_hip_set_rdf_blocks(g_opt.rdf_blocks_decode.load());
        if (blocks_guard.old < g_opt.rdf_blocks_decode.load()) vmd_hip_set_rdf_blocks(blocks_guard.old);      // never raise a smaller setting
    }
    if (raw_ring) {
        for (auto& rs : e->raw_slots) rs.state = 0;
        for (size_t bi = 0; bi < std::min<size_t>(raw_ahead, batches.size()); ++bi)
            if (raw_upload(e, *slot_of(bi), traj, num_atoms, batches[bi].f0, batches[bi].nb) < 0) return false;
    }
    for (size_t bi = 0; bi < std::min(stage_ahead, batches.size()); ++bi)
        if (!fetch_stage(e, stage_of(bi), traj, vw, num_atoms, batches[bi].f0, batches[bi].nb, false, slot_of(bi))) return false;
    // ---- one batch in flight, one being queued.  The kernels of batch k + 1 are queued BEFORE the host waits for batch k (on an event,
    // not on the stream): the device never idles across the host's per-batch work - the wait itself, the bookkeeping, the ~15 launches
    // of the next batch (~0.15 ms per boundary, a tenth of a step when batches are the 128 frames a file-backed pass stages).  Everything
    // a batch hands to the host has two slots (overflow flag, temporal rows, a snapshot of the RDF counts behind its commits); a batch
    // whose cell build overflowed still voids itself AND whatever was queued behind it (the flag is sticky): the later batch is marked
    // and repeats its RDF part when its turn comes.  Evals that keep block partials complete every batch before the next is queued.
    struct Sub { size_t off, nb; long blk; };
    struct BatchCtx {
        Batch bt{0, 0, -1, 0};
        Stage* src = nullptr;
        size_t f0 = 0, nb = 0;
        uint32_t pbc = 0;
        std::vector<Sub> subs;
        bool two_streams = false;
        int slot = 0;
        bool active = false;        // queued, not completed
        bool poisoned = false;      // queued behind a batch that overflowed: its RDF part saw the flag and did nothing
        bool snapshot = false;      // h_snap[slot] holds the RDF counts behind this batch's commits (+ w_snap: the weights)
    };
    BatchCtx ctx[2];
    const bool defer = g_opt.defer_sync.load() != 0 && e->block_frames == 0 && batches.size() > 1;
    size_t rdf_counts = 0;
    for (auto& p : e->props) if (p->prop.kind == PROP_RDF) rdf_counts += p->ncounts;
    if (defer && rdf_counts) {
        if (e->h_snap_cap < 2 * rdf_counts) {
            if (e->h_snap) pool_give(e->h_snap);
            e->h_snap = nullptr; e->h_snap_cap = 0;
            HIP_OK(pool_take(kPinned, (void**)&e->h_snap, 2 * rdf_counts * sizeof(uint64_t)));
            e->h_snap_cap = 2 * rdf_counts;
        }
        e->w_snap.resize(2 * rdf_counts);
    }
    auto acc_of = [&](PropState* p, const Sub& sb) -> uint64_t* {
        return (sb.blk >= 0 && p->ncounts) ? p->d_blocks.p + (size_t)sb.blk * p->ncounts : p->d_counts.p;
    };
    // ---- RDF: one pair pass per (group, pass); launch_rdf may run again for this batch when a cell-build bucket overflowed
    // Every pass accumulates into its own scratch row and the rows are committed to the properties' accumulators by ONE
    // group of k_axpy_u64 launches at the very end, behind the overflow flag: by then every cell build of the batch has run,
    // so the flag is final and the batch's RDF part is all-or-nothing (a bucket of a LATER build may overflow after earlier
    // passes have long finished; nothing of them may stay behind when the batch is repeated).
    auto launch_rdf = [&](BatchCtx& c) -> bool {
        VMD_STAGE("batch: cell build + pair kernels");
        vmd_hip_set_rdf_closed(e->spec.rdf_closed ? 1 : 0);
        vmd_hip_set_rdf_raw(e->spec.rdf_raw ? 1 : 0);
        size_t scratch_rows = 0;
        for (auto& g : e->rdf_groups) scratch_rows += std::max(g.passes.size(), g.props.size());
        scratch_rows *= c.subs.size();
        if (!e->d_pass.ensure(std::max<size_t>(scratch_rows, 1) * VMD_RDF_NUM_BINS)) return false;
        HIP_OK(hipMemsetAsync(e->d_pass.p, 0, scratch_rows * VMD_RDF_NUM_BINS * sizeof(uint64_t), e->stream));
        struct Commit { uint64_t* dst; const uint64_t* src; uint64_t mult; };
        std::vector<Commit> commits;
        size_t row = 0;
        bool forked = false;
        for (auto& g : e->rdf_groups) {
            vmd_grid_t grid;
            // fully periodic cells use the frame boxes; open axes (non-periodic systems, slabs) span the batch's bounding box
            const bool open_axes = (c.pbc & 8u) == 0 && (c.pbc & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL;
            if (open_axes && !g_opt.force_brute && !prepare_open_boxes(e, *c.src, c.nb, c.pbc, num_atoms)) return false;
            const std::vector<float>& gb = (open_axes && c.src->gboxes_ready) ? c.src->h_gboxes : c.src->h_boxes;
            const float* d_gb = (open_axes && c.src->gboxes_ready) ? c.src->d_gboxes.p : c.src->d_boxes.p;
            // density of the sparsest selection any pass of this group puts in the lanes (the denser of its two), against the first frame's cell
            bool dense_lanes = !open_axes && !g.passes.empty();
            if (dense_lanes) {
                const float* q = gb.data();
                const double vol = (double)q[0] * q[1] * q[2];
                for (auto& ps : g.passes) {
                    const size_t lanes = std::max(e->sels[ps.sel_a]->idx.size(), e->sels[ps.sel_b]->idx.size());
                    dense_lanes = dense_lanes && vol > 0.0 && (double)lanes / vol >= 0.08;
                }
            }
            if (e->spec.rdf_raw || !choose_grid(gb, c.pbc, c.nb, g.rmax, &grid, dense_lanes)) {
                // no grid for this batch (cutoff >= half the cell width, ...): all pairs, per property
                for (int pi : g.props) {
                    PropState* p = e->props[pi].get();
                    Selection* sa = e->sels[p->sel_a].get();
                    Selection* sb = e->sels[p->sel_b].get();
                    for (auto& su : c.subs) {
                        uint64_t* dst = e->d_pass.p + (row++) * VMD_RDF_NUM_BINS;
                        e->prof.begin("rdf_brute", e->stream);
                        KRN_OK(vmd_hip_rdf_brute(e->stream, c.src->base + su.off * c.src->frame_stride, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p + 9 * su.off, c.pbc, (int)su.nb,
                                                 sa->d_idx.p, (int)sa->idx.size(), sb->d_idx.p, (int)sb->idx.size(),
                                                 g.rmin, g.rmax, VMD_RDF_NUM_BINS, dst));
                        e->prof.end(e->stream);
                        commits.push_back({acc_of(p, su), dst, 1});
                    }
                }
                continue;
            }
            if (!e->d_partial.ensure(vmd_hip_rdf_partial_words())) return false;
            if (c.two_streams && !e->d_partial2.ensure(vmd_hip_rdf_partial_words())) return false;
            for (auto& ps : g.passes) {
                Selection* sa = e->sels[ps.sel_a].get();
                Selection* sb = e->sels[ps.sel_b].get();
                // passes with the same cutoff share the sorted copies; build_selection re-sorts when the grid differs
                if (forked) {     // the second stream still reads the sorted copies of the previous pass
                    HIP_OK(hipEventRecord(e->pair_join, e->pair_stream));
                    HIP_OK(hipStreamWaitEvent(e->stream, e->pair_join, 0));
                    forked = false;
                }
                if (!build_selection(e, sa, *c.src, d_gb, c.pbc, c.nb, grid)) return false;
                if (sb != sa && !build_selection(e, sb, *c.src, d_gb, c.pbc, c.nb, grid)) return false;
                // the pair set is symmetric in (ref, target): put the denser selection in the lanes - 64 of its atoms span a
                // shorter stretch of the pencil, so the x window of every segment carries less padding
                if (sb->idx.size() > sa->idx.size()) std::swap(sa, sb);
                if (c.two_streams) {
                    HIP_OK(hipEventRecord(e->pair_fork, e->stream));
                    HIP_OK(hipStreamWaitEvent(e->pair_stream, e->pair_fork, 0));
                    forked = true;
                }
                size_t si = 0;
                for (auto& su : c.subs) {
                    uint64_t* dst = e->d_pass.p + (row++) * VMD_RDF_NUM_BINS;
                    const bool second = c.two_streams && (si++ & 1);
                    hipStream_t ks = second ? e->pair_stream : e->stream;
                    if (!second) e->prof.begin("rdf_pencil", ks);
                    KRN_OK(vmd_hip_rdf_pencil(ks, sa->sorted.p + su.off * 3 * (size_t)sa->nsel_pad, sa->cell_start.p + su.off * (size_t)(grid.ncell + 1), (int)sa->idx.size(), sa->nsel_pad,
                                              sb->sorted.p + su.off * 3 * (size_t)sb->nsel_pad, sb->cell_start.p + su.off * (size_t)(grid.ncell + 1), (int)sb->idx.size(), sb->nsel_pad,
                                              d_gb + 9 * su.off, (int)su.nb, grid, g.rmin, g.rmax, VMD_RDF_NUM_BINS,
                                              ps.same ? 1 : 0, g_opt.rdf_variant, c.pbc, second ? e->d_partial2.p : e->d_partial.p, dst, e->d_overflow.p));
                    if (!second) e->prof.end(ks);
                    if (e->spec.rdf_closed && ps.same && g.rmin <= 0.0f && 0.0f <= g.rmax) {
                        // closed interval: d = 0 is a hit, but a same-set pass walks the half shell (j > i, every hit twice) and never
                        // meets the pairs (i, i) - one per list entry and frame, all in the bin of d = 0 (SPEC S4 binning of 0)
                        int bin0 = (int)(((0.0f - g.rmin) * (1.0f / (g.rmax - g.rmin))) * (float)VMD_RDF_NUM_BINS);
                        bin0 = std::min(std::max(bin0, 0), VMD_RDF_NUM_BINS - 1);
                        KRN_OK(vmd_hip_bump_u64(ks, dst + bin0, (uint64_t)su.nb * (uint64_t)sa->idx.size()));
                    }
                    for (auto& tg : ps.targets) commits.push_back({acc_of(e->props[tg.first].get(), su), dst, tg.second});
                }
            }
        }
        if (forked) {
            HIP_OK(hipEventRecord(e->pair_join, e->pair_stream));
            HIP_OK(hipStreamWaitEvent(e->stream, e->pair_join, 0));
        }
        for (auto& cm : commits) KRN_OK(vmd_hip_axpy_u64(e->stream, cm.dst, cm.src, VMD_RDF_NUM_BINS, cm.mult, e->d_overflow.p));
        HIP_OK(hipMemcpyAsync(&e->h_overflow[c.slot], e->d_overflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        return true;
    };

    // waits for a queued batch (`later`: the batch already queued behind it, if any), repeats its RDF part when a bucket overflowed, books its frames
    auto complete_batch = [&](BatchCtx& c, BatchCtx* later) -> bool {
        if (!c.active) return true;
        c.active = false;
        const bool behind = later && later->active;
        VMD_STAGE("batch: waiting for its kernels");
        { HostTimer host_timer("host_sync_wait");
          if (behind) HIP_OK(hipEventSynchronize(e->batch_done[c.slot]));
          else HIP_OK(hipStreamSynchronize(e->stream)); }
        VMD_STAGE("batch: host bookkeeping");
        // a bucket of the two-level cell build was too small: nothing reached the histograms (every consumer saw the flag).
        // Re-measure the selections that used buckets with more head room and evaluate the RDF part of this batch again.
        bool repeated = false;
        for (int attempt = 0; e->h_overflow[c.slot] != 0; ++attempt) {
            if (attempt >= 4) return vmd_fail("cell build: pencil buckets keep overflowing");
            if (behind) {          // the batch behind this one saw the flag too: let it drain, it repeats its RDF part at its own completion
                HIP_OK(hipStreamSynchronize(e->stream));
                later->poisoned = true;
            }
            const bool own = !(c.poisoned && attempt == 0);      // a poisoned batch did not overflow itself (as far as anyone knows)
            const uint32_t who = e->h_overflow[c.slot];           // one bit per selection (Selection::overflow_bit; selections beyond 32 share)
            for (auto& sl : e->sels) {
                sl->built = false;
                // only the selection whose buckets were too small gets wider ones.  (Its bit, not used_pencil, says so: a selection can be
                // sorted through buckets on one group's grid and by the single-block build on another's within ONE batch - co-evaluated RDFs
                // with different cutoffs - and used_pencil only remembers the last of them; fuzz seed 8941, round 4.)
                if (!own || !(who & sl->overflow_bit)) continue;
                sl->pen_off.clear();
                sl->caps_cache.clear();
                sl->cap_margin *= 1.6f;
                sl->overflows += 1;
            }
            e->h_overflow[c.slot] = 0;
            HIP_OK(hipMemsetAsync(e->d_overflow.p, 0, sizeof(uint32_t), e->stream));
            if (!launch_rdf(c)) return false;
            HIP_OK(hipStreamSynchronize(e->stream));
            repeated = true;
        }
        if (c.bt.blk >= 0 && !spec)
            for (auto& su : c.subs)
                for (auto& p : e->props) if (p->ncounts) KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, acc_of(p.get(), su), p->ncounts));
        e->prof.resolve();
        if (g_prof_on) { std::lock_guard<std::mutex> l(g_prof_mtx); g_prof["batches"].launches += 1; }
        size_t toff = 0;
        for (auto& p : e->props) {
            if (p->prop.kind != PROP_DIST) continue;
            // evaluated ahead: the rows wait beside the view until their block is committed (a reader of the values array never sees a
            // frame nobody asked for)
            if (spec && p->ahead_values.size() != p->values.size()) p->ahead_values.assign(p->values.size(), 0.0f);
            memcpy(spec ? &p->ahead_values[c.f0 * p->dim1] : &p->values[c.f0 * p->dim1], e->h_temporal_slot[c.slot].data() + toff, c.nb * p->dim1 * sizeof(float));
            toff += c.nb * p->dim1;
        }
        e->frames_computed += c.nb;
        if (c.bt.blk >= 0) for (auto& su : c.subs) e->block_ready[su.blk] = spec ? BLOCK_ROWS_AHEAD : BLOCK_ROWS_IN_PLACE;
        if (spec) return true;
        for (size_t b = 0; b < c.nb; ++b) e->frame_mask[c.f0 + b] = 1;
        e->frames_done += c.nb;
        // cheap views are refreshed every batch so a polling GUI sees progress (src/main.cpp:1508-1524): from the device when nothing
        // is queued behind this batch, from the snapshot taken behind its commits otherwise
        size_t soff = (size_t)c.slot * rdf_counts;
        for (auto& p : e->props) {
            if (p->prop.kind != PROP_RDF) continue;
            if (behind && c.snapshot && !repeated && !(later && later->poisoned)) refresh_distribution_from(p.get(), e->h_snap + soff, e->w_snap.data() + soff);
            else if (!behind && views) { if (!refresh_distribution(e, p.get())) return false; }
            soff += p->ncounts;
        }
        return true;
    };

    for (size_t bi = 0; bi < batches.size(); ++bi) {
        if (e->interrupt) { completed = false; break; }
        BatchCtx& c = ctx[bi & 1];
        BatchCtx& prev = ctx[(bi & 1) ^ 1];
        c = BatchCtx{};
        c.bt = batches[bi]; c.f0 = c.bt.f0; c.nb = c.bt.nb; c.slot = (int)(bi & 1);
        c.src = &stage_of(bi);
        { HostTimer host_timer("host_settle"); if (!settle_stage(e, *c.src, traj, num_atoms)) return false; }
        VMD_STAGE("batch: kernels queued");
        HostTimer queue_timer("host_queue_to_sync");
        HIP_OK(hipStreamWaitEvent(e->stream, c.src->ready, 0));
        c.pbc = batch_pbc(*c.src);
        for (auto& s : e->sels) s->built = false;

        size_t temporal_floats = 0;
        for (auto& p : e->props) if (p->prop.kind == PROP_DIST) temporal_floats += c.nb * p->dim1;
        e->h_temporal_slot[c.slot].resize(temporal_floats);
        size_t toff = 0;

        // a whole frame block accumulates into its own partial first and is merged into the totals afterwards.  A batch of blocks
        // (filtered evaluation) is evaluated block by block - `subs` - behind one cell build and in front of one synchronisation.
        if (c.bt.blk >= 0 && c.bt.nblk > 1) {
            const size_t S = e->block_frames;
            for (size_t j = 0; j < c.bt.nblk; ++j) c.subs.push_back({j * S, std::min(S, c.nb - j * S), c.bt.blk + (long)j});
        } else c.subs.push_back({0, c.nb, c.bt.blk});
        if (c.bt.blk >= 0)
            for (auto& sb : c.subs)
                for (auto& p : e->props) if (p->ncounts) HIP_OK(hipMemsetAsync(acc_of(p.get(), sb), 0, p->ncounts * sizeof(uint64_t), e->stream));
        // the blocks' pair launches alternate between the eval's stream and a second one (own partial rows): a 50-frame launch of a
        // 100k-atom system is ~3 work items per resident wave, and the tail of one launch then runs under the head of the next
        c.two_streams = c.subs.size() > 1 && g_opt.block_two_streams.load() != 0;

        e->h_overflow[c.slot] = 0;
        if (!e->rdf_groups.empty() && !launch_rdf(c)) return false;

        for (auto& p : e->props) {
            const Property& d = p->prop;
            if (d.kind == PROP_RDF) {
                // SPEC S4 normalisation, fp64 on the host (needs only the box)
                for (auto& su : c.subs) {
                    double* bw = su.blk >= 0 ? &p->block_weights64[(size_t)su.blk * p->ncounts] : nullptr;
                    if (bw) std::fill(bw, bw + p->ncounts, 0.0);
                    for (size_t b = su.off; b < su.off + su.nb; ++b) {
                        const float* L = &c.src->h_boxes[9 * b];
                        double V;
                        if ((c.pbc & VMD_UNITCELL_PBC_ALL) == VMD_UNITCELL_PBC_ALL && e->spec.rdf_norm != 1) V = (double)L[0] * (double)L[1] * (double)L[2];   // also the triclinic volume
                        else V = (4.0 / 3.0) * M_PI * (double)d.rmax * (double)d.rmax * (double)d.rmax;
                        const double rho = (e->spec.rdf_norm == 2 ? 1.0 : (double)d.a.size()) * (double)d.b.size() / V;
                        const double w = ((double)d.rmax - (double)d.rmin) / (double)p->ncounts;
                        for (size_t k = 0; k < p->ncounts; ++k) {
                            const double r0 = (double)d.rmin + w * (double)k;
                            const double r1 = (double)d.rmin + w * (double)(k + 1);
                            const double wk = rho * (4.0 / 3.0) * M_PI * (r1 * r1 * r1 - r0 * r0 * r0);
                            if (!spec) p->weights64[k] += wk;
                            if (bw) bw[k] += wk;
                        }
                    }
                }
                p->dirty = p->dirty || !spec;
            } else if (d.kind == PROP_SDF) {
                if (!p->d_R32.ensure(c.nb * d.K * 9) || !p->d_c32.ensure(c.nb * d.K * 3) || !p->d_group.ensure(c.nb * 4)) return false;
                VMD_STAGE("batch: sdf align + scatter");
                e->prof.begin("sdf_align", e->stream);
                if (p->have_tree && !p->d_tree_pos.ensure(c.nb * d.K * d.m * 3)) return false;
                KRN_OK(vmd_hip_sdf_align(e->stream, c.src->base, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p, c.pbc, (int)c.nb,
                                         p->d_structs.p, p->d_mass.p, (int)d.K, (int)d.m, p->d_ref_pose.p, p->d_R32.p, p->d_c32.p, nullptr, p->d_group.p,
                                         p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr, p->have_tree ? p->d_tree_pos.p : nullptr));
                e->prof.end(e->stream);
                e->prof.begin("sdf_scatter", e->stream);
                for (auto& su : c.subs)
                    KRN_OK(vmd_hip_sdf_scatter(e->stream, c.src->base + su.off * c.src->frame_stride, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p + 9 * su.off, c.pbc, (int)su.nb,
                                               p->d_structs.p, (int)d.K, (int)d.m, p->d_R32.p + su.off * d.K * 9, p->d_c32.p + su.off * d.K * 3, p->d_tgt.p,
                                               (p->have_owner && !e->spec.sdf_include_self) ? p->d_owner.p : nullptr, (int)d.b.size(),
                                               d.rmax, VMD_VOLUME_DIM, acc_of(p.get(), su), p->d_group.p + 4 * su.off,
                                               (p->have_tag && p->tag_len == c.src->row_stride && !e->spec.sdf_include_self) ? p->d_tag.p : nullptr,
                                               p->tgt_first, p->tgt_stride, (p->unowned || e->spec.sdf_include_self) ? 1 : 0));
                e->prof.end(e->stream);
                p->dirty = p->dirty || !spec;
            } else {
                if (!p->d_out.ensure(c.nb * p->dim1)) return false;
                e->prof.begin("distance", e->stream);
                KRN_OK(vmd_hip_distance(e->stream, c.src->base, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p, c.pbc, (int)c.nb, d.dist_kind,
                                        (int)p->dist_P, (int)p->dist_per, p->d_a.p, p->d_ma.p, p->d_aoff.p, p->d_b.p, p->d_mb.p, p->d_boff.p,
                                        p->d_out.p));
                e->prof.end(e->stream);
                HIP_OK(hipMemcpyAsync(e->h_temporal_slot[c.slot].data() + toff, p->d_out.p, c.nb * p->dim1 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
                toff += c.nb * p->dim1;
                p->dirty = p->dirty || !spec;
            }
        }
        if (defer) {
            // what the host will want from this batch once a later one is queued behind it: the RDF counts as they stand behind its
            // commits (the weights as they stand now), and an event to wait on
            size_t soff = (size_t)c.slot * rdf_counts;
            for (auto& p : e->props) {
                if (p->prop.kind != PROP_RDF) continue;
                HIP_OK(hipMemcpyAsync(e->h_snap + soff, p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
                memcpy(e->w_snap.data() + soff, p->weights64.data(), p->ncounts * sizeof(double));
                soff += p->ncounts;
            }
            c.snapshot = true;
            HIP_OK(hipEventRecord(e->batch_done[c.slot], e->stream));
        }
        c.active = true;
        // deferred: the batch in front of this one is completed now that the device has this one to go on with (its stage is free
        // for the staging below only then)
        if (defer && !complete_batch(prev, &c)) return false;
        VMD_STAGE("batch: staging the next batch (fetch_stage)");
        // the kernels of this batch are queued: load the next batch on the host while they run
        if (bi + 1 < batches.size() && !e->interrupt) {
            if (bi + stage_ahead < batches.size()) {
                HostTimer host_timer("host_fetch_stage");
                const size_t nx = bi + stage_ahead;
                if (!fetch_stage(e, stage_of(nx), traj, vw, num_atoms, batches[nx].f0, batches[nx].nb, false, slot_of(nx))) return false;
            }
            // ... and send the bit streams of the batch after that on their way (its slot held batch bi - 1: decoded long ago)
            if (raw_ring && bi + raw_ahead < batches.size() &&
                raw_upload(e, *slot_of(bi + raw_ahead), traj, num_atoms, batches[bi + raw_ahead].f0, batches[bi + raw_ahead].nb) < 0) return false;
        }
        if (!defer && !complete_batch(c, nullptr)) return false;
    }
    // whatever is still in flight (deferred: the last batch queued; after an interrupt: the one before the break)
    { BatchCtx& a = ctx[0].active && ctx[1].active ? (ctx[0].f0 < ctx[1].f0 ? ctx[0] : ctx[1]) : ctx[0];
      BatchCtx& b = &a == &ctx[0] ? ctx[1] : ctx[0];
      if (!complete_batch(a, b.active ? &b : nullptr)) return false;
      if (!complete_batch(b, nullptr)) return false; }
    if (views) {
        for (auto& p : e->props) {
            if (!p->dirty) continue;
            if (p->prop.kind == PROP_SDF) { if (!refresh_volume(e, p.get())) return false; }
            else if (p->prop.kind == PROP_DIST) refresh_temporal_stats(e, p.get());
        }
        e->views_at = std::chrono::steady_clock::now();
    }
    return completed;
}

static bool process_range(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end, bool views = true) {
    g_last_error.clear();
    if (eval->interrupt) return false;
    std::lock_guard<std::mutex> lock(eval->mtx);
    return process_range_locked(eval, sys, traj, frame_beg, frame_end, views, false);
}

// the host views of every property whose accumulators changed since its last refresh (the combining queue calls this when no call
// is waiting, process_range(views = true) does the same at its end)
static bool refresh_views_locked(vmd_script_eval_t* e) {
    HIP_OK(hipSetDevice(e->device));
    for (auto& p : e->props) {
        if (!p->dirty) continue;
        if (p->prop.kind == PROP_RDF) { if (!refresh_distribution(e, p.get())) return false; }
        else if (p->prop.kind == PROP_SDF) {
            // vmd_eval_defer_volume_views: a rank of a multi-GPU evaluation does not materialise ITS partial volume's float view (8.4 MB over
            // PCIe after every range) - the merge re-derives the view of the merged counts (vmd_eval_reduce -> vmd_eval_finalize); the volume
            // stays dirty until then
            if (e->defer_volume_views.load(std::memory_order_relaxed)) continue;
            if (!refresh_volume(e, p.get())) return false;
        }
        else refresh_temporal_stats(e, p.get());
    }
    e->views_at = std::chrono::steady_clock::now();
    return true;
}
static bool refresh_views(vmd_script_eval_t* e) {
    std::lock_guard<std::mutex> lock(e->mtx);
    return refresh_views_locked(e);
}

// The hot call.  VIAMD invokes it from N pool threads with small disjoint ranges (grain 1, src/main.cpp:993-997,
// src/task_system.cpp:73-81).  Launching kernels per call would drown the GPU in tiny batches, so calls COMBINE: the first
// caller becomes the leader, later callers queue their range and sleep; the leader repeatedly takes everything queued so far,
// merges adjacent ranges into long runs and evaluates those in large frame batches, then wakes the owners.
static bool combine_call(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    RangeRequest me;
    me.beg = frame_beg; me.end = frame_end; me.sys = sys; me.traj = traj;
    std::unique_lock<std::mutex> ql(eval->queue_mtx);
    eval->queue.push_back(&me);
    if (eval->leader_active) {
        eval->queue_cv.wait(ql, [&] { return me.done; });
        if (!me.ok) g_last_error = me.error;
        return me.ok;
    }
    eval->leader_active = true;
    while (!eval->queue.empty()) {
        // VIAMD's pool threads pull ranges of a few frames each (enkiTS: num_frames / (threads x (threads - 1)), at least 1) and every
        // one of them blocks in here, so a round can never hold more than threads x grain frames - and far fewer if the leader runs
        // off with whatever is queued the instant it looks: the threads it has just released are back with their next ranges within
        // microseconds.  It waits for them (gather_us at most, only while requests keep arriving) - a batch of 16 x 4 frames costs the
        // same ~0.15 ms of launches and round trips as a batch of 4.
        // Waiting is only worth a fraction of what a round costs: the slowest of the released threads needs 50 - 100 us to come back,
        // which a round of the 10 000-frame SDF (0.08 ms for 16 frames) cannot afford and a round of the 100k-atom RDF (0.3 ms) can:
        // at most half the previous round's time.  A large pool brings enough frames per round by itself, and on an oversubscribed
        // host waiting for 128 threads costs more than it gathers: pools of up to 32 callers only.
        const int gather = (int)std::min<long>(g_opt.gather_us.load(), eval->last_round_us / 2);
        // Scripts without pair passes (SDF / distance only: 0.7 us of kernels per frame) never gain from it - measured r03an: 129 ms
        // without, 195 ms with, for the 10 000 frames of config 4 from 16 threads - so only evals with RDF groups wait.
        if (gather >= 20 && !eval->rdf_groups.empty() && eval->queue.size() < eval->last_round && eval->last_round <= 32) {
            const auto t0 = std::chrono::steady_clock::now();
            const auto deadline = t0 + std::chrono::microseconds(gather);
            auto last_arrival = t0;
            size_t seen = eval->queue.size();
            while (eval->queue.size() < eval->last_round && !eval->interrupt) {
                ql.unlock();
                std::this_thread::yield();
                ql.lock();
                const auto now = std::chrono::steady_clock::now();
                if (eval->queue.size() != seen) { seen = eval->queue.size(); last_arrival = now; }
                // nobody new for a third of the window: the task is running out of ranges (its tail), or the pool is busy elsewhere
                if (now >= deadline || now - last_arrival > std::chrono::microseconds(gather / 3 + 1)) break;
            }
        }
        std::vector<RangeRequest*> taken;
        taken.swap(eval->queue);
        eval->last_round = taken.size();
        ql.unlock();
        // the views are for readers, and a reader only needs them final when the LAST call returns: while other calls are waiting they
        // are brought up to date at most every lazy_views_ms (a polling GUI still sees progress), and always before a round whose end
        // finds the queue empty hands its callers back
        const bool lazy = g_opt.lazy_views.load() != 0;
        bool all_ok = true;
        const auto round_t0 = std::chrono::steady_clock::now();
        // requests for the same trajectory, sorted by first frame; touching ranges fuse into one run
        // "the same trajectory" = the same instance behind the same callbacks, not the same interface STRUCT: a host that wraps its own
        // trajectory type per call (include/vmd_md_script_shim.h did, from every pool thread) presents a different address each time
        // (such ranges never fused)
        auto same_traj = [](const vmd_trajectory_i* a, const vmd_trajectory_i* b) {
            return a == b || (a->inst == b->inst && a->load_frame == b->load_frame && a->device_view == b->device_view && a->load_raw == b->load_raw);
        };
        std::sort(taken.begin(), taken.end(), [](const RangeRequest* a, const RangeRequest* b) {
            return a->traj->inst != b->traj->inst ? a->traj->inst < b->traj->inst : a->beg < b->beg; });
        size_t i = 0;
        while (i < taken.size()) {
            size_t j = i + 1;
            uint32_t run_end = taken[i]->end;
            while (j < taken.size() && same_traj(taken[j]->traj, taken[i]->traj) && taken[j]->beg == run_end) { run_end = taken[j]->end; ++j; }
            const bool ok = process_range(eval, taken[i]->sys, taken[i]->traj, taken[i]->beg, run_end, !lazy);
            const std::string err = ok ? std::string() : g_last_error;
            for (size_t k = i; k < j; ++k) { taken[k]->ok = ok; taken[k]->error = err; }
            all_ok = all_ok && ok;
            i = j;
        }
        const long round_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - round_t0).count();
        ql.lock();
        eval->last_round_us = round_us;
        if (lazy) {
            const bool overdue = std::chrono::steady_clock::now() - eval->views_at > std::chrono::milliseconds(std::max(1, g_opt.lazy_views_ms.load()));
            if (eval->queue.empty() || overdue) {
                ql.unlock();
                const bool vok = refresh_views(eval);
                if (!vok && all_ok) { const std::string err = g_last_error; for (RangeRequest* r : taken) { r->ok = false; r->error = err; } }
                ql.lock();
            }
        }
        for (RangeRequest* r : taken) r->done = true;
        eval->queue_cv.notify_all();
    }
    eval->leader_active = false;
    ql.unlock();
    if (!me.ok) g_last_error = me.error;
    return me.ok;
}

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

static void ra_reset(vmd_script_eval_t* e) {          // clear_data (mtx held): a new evaluation starts
    ReadAhead& ra = e->ra;
    if (ra.blk_state) for (size_t b = 0; b < e->num_blocks; ++b) ra.blk_state[b] = vmd_script_eval_t::RA_NONE;
    if (ra.frame_req) for (size_t i = 0; i < 64 * ra.req_stride; ++i) ra.frame_req[i] = 0;
    ra.marks_pending = false; ra.views_dirty = false;
    ra.concurrent = false; ra.lonely = false; ra.disabled = false; ra.strikes = 0; ra.next_region = 0; ra.failed = false; ra.error.clear();
    ra.lone.store(false);
}

static size_t ra_block_frames(const vmd_script_eval_t* e, size_t Bmax) {
    size_t G = (size_t)std::max(0, g_opt.readahead_block.load());
    if (!G) {
        if (e->rdf_groups.empty()) G = 1024;          // streaming scripts: 16.8 MB of memset + add per volume and block - few, large blocks (r04c, 10 000-frame SDF at grain 1: 10.8 ms with 256, 9.7 with 512, 9.2 with 1 024; one call 7.0)
        else {
            // pair passes: one pair launch per block; it needs ~4M selected atoms to fill the chip (50-frame launches of the
            // 100k-atom box cost +12 %, 125-frame launches +5 %), and a block is also the most a ragged range end evaluates directly
            size_t sel = 1;
            for (auto& g : e->rdf_groups) for (auto& ps : g.passes) sel = std::max(sel, std::max(e->sels[ps.sel_a]->idx.size(), e->sels[ps.sel_b]->idx.size()));
            G = 16;
            while (G < 128 && G * sel < 4000000) G *= 2;
        }
    }
    return std::max<size_t>(1, std::min(G, Bmax));
}

// queue_mtx held by the caller (and no combining call in flight): allocate the block partials and the states
static bool ra_engage(vmd_script_eval_t* e, vmd_trajectory_i* traj) {
    ReadAhead& ra = e->ra;
    std::lock_guard<std::mutex> lock(e->mtx);
    HIP_OK(hipSetDevice(e->device));
    const size_t num_atoms = traj->num_atoms(traj->inst);
    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    const size_t Bmax = auto_batch(e, num_atoms, !have_view);
    if (e->block_frames == 0) {
        // a filtered eval (src/main.cpp:1014-1039) adopts whole blocks from its source's partials: same blocks as the source
        // (the source may be engaging at this very moment - "Eval Full" and "Eval Filt" side by side: its block size is read under its mutex;
        // order: own mutex, then the source's, as everywhere)
        size_t src_S = 0;
        if (e->source) { std::lock_guard<std::mutex> sl(e->source->mtx); src_S = e->source->block_frames; }
        const size_t S = src_S ? std::min(src_S, std::max<size_t>(Bmax, 1)) : ra_block_frames(e, Bmax);
        const size_t nblocks = (e->num_frames + S - 1) / S;
        size_t bytes = 0;
        for (auto& p : e->props) bytes += nblocks * p->ncounts * sizeof(uint64_t);
        // not worth a ninth of the HBM, nor more than half of what is free right now (a trajectory resident in HBM may have taken most of
        // it): the combining queue serves this eval, as it did before read-ahead existed
        size_t cap_bytes = (size_t)32 << 30, free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_bytes = std::min(cap_bytes, free_b / 2);
        if (bytes > cap_bytes) { ra.disabled = true; return true; }
        for (auto& p : e->props) {
            if (!p->ncounts) continue;
            if (!p->d_blocks.ensure(nblocks * p->ncounts) || g_opt.readahead_fail_alloc.load()) {
                // an allocation that fails here must not fail the evaluation (and every later call with it) - read-ahead is an
                // optimisation; give back what was taken and let the combining queue serve the calls
                for (auto& q : e->props) { q->d_blocks.release(); q->block_weights64.clear(); q->block_weights64.shrink_to_fit(); }
                g_last_error.clear();
                ra.disabled = true;
                return true;
            }
            if (p->prop.kind == PROP_RDF) p->block_weights64.assign(nblocks * p->ncounts, 0.0);
        }
        e->block_ready.reset(new std::atomic<uint8_t>[nblocks]);
        for (size_t b = 0; b < nblocks; ++b) e->block_ready[b] = 0;
        e->num_blocks = nblocks;
        e->block_frames = S;
        ra.own_blocks = true;
    }
    const size_t S = e->block_frames;
    ra.blk_state.reset(new std::atomic<uint8_t>[e->num_blocks]);
    ra.req_stride = std::max<size_t>(64, (e->num_frames + 63) / 64);
    ra.frame_req.reset(new std::atomic<uint8_t>[64 * ra.req_stride]);
    for (size_t i = 0; i < 64 * ra.req_stride; ++i) ra.frame_req[i] = 0;
    // a rank's shard of a device trajectory: only blocks that lie inside it can be evaluated ahead
    size_t lo = 0, hi = e->num_frames;
    if (have_view && view_sharded(view)) { lo = view.resident_beg; hi = view.resident_end; }
    for (size_t b = 0; b < e->num_blocks; ++b) {
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        bool done = false;
        for (size_t f = f0; f < f1; ++f) { ra.req(f) = e->frame_mask[f] ? 1 : 0; done = done || e->frame_mask[f]; }
        ra.blk_state[b] = (done || f0 < lo || f1 > hi || f1 - f0 > Bmax) ? vmd_script_eval_t::RA_DIRECT : vmd_script_eval_t::RA_NONE;
    }
    ra.bmax = Bmax;
    ra.traj_inst = traj_id(traj);
    ra.on.store(true, std::memory_order_release);
    return true;
}

// The filtered evaluation under VIAMD's call pattern: blocks of a region that the source eval has finished are not evaluated again - their
// partials (and temporal rows) are copied from the source into this eval's own block partials, where they wait to be requested like any
// block evaluated ahead.  mtx held, device set.  adopted[b - b0] = 1 for the blocks taken.
static bool ra_adopt_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t b0, size_t b1, std::vector<char>* adopted) {
    adopted->assign(b1 - b0, 0);
    vmd_script_eval_t* src = e->source;
    if (!src || src->props.size() != e->props.size()) return true;
    std::lock_guard<std::mutex> lock(src->mtx);       // order: own mutex, then the source's (as reuse_blocks)
    if (src->block_frames != e->block_frames || src->blocks_inst != traj_inst) return true;
    const size_t S = e->block_frames;
    size_t taken = 0;
    for (size_t b = b0; b < b1; ++b) {
        if (b >= src->num_blocks || !src->block_ready[b]) continue;
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        for (size_t i = 0; i < e->props.size(); ++i) {
            PropState* p = e->props[i].get();
            const PropState* q = src->props[i].get();
            if (p->ncounts) {
                HIP_OK(hipMemcpyAsync(p->d_blocks.p + b * p->ncounts, q->d_blocks.p + b * p->ncounts, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream));
                if (p->prop.kind == PROP_RDF) memcpy(&p->block_weights64[b * p->ncounts], &q->block_weights64[b * p->ncounts], p->ncounts * sizeof(double));
            } else {
                if (p->ahead_values.size() != p->values.size()) p->ahead_values.assign(p->values.size(), 0.0f);
                memcpy(&p->ahead_values[f0 * p->dim1], block_rows(src, q, b) + f0 * p->dim1, (f1 - f0) * p->dim1 * sizeof(float));
            }
        }
        e->block_ready[b] = BLOCK_ROWS_AHEAD;           // (the rows went into the side buffer above)
        (*adopted)[b - b0] = 1;
        taken += f1 - f0;
    }
    if (taken) {
        HIP_OK(hipStreamSynchronize(e->stream));          // the source's partials are read before its mutex is released
        e->frames_reused += taken;
    }
    return true;
}

// mtx held, device set: the block's partial joins the totals
static bool ra_commit_block(vmd_script_eval_t* e, size_t blk) {
    const size_t S = e->block_frames, f0 = blk * S, f1 = std::min(f0 + S, e->num_frames);
    for (auto& p : e->props) {
        if (p->ncounts) {
            KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, p->d_blocks.p + blk * p->ncounts, p->ncounts));
            if (p->prop.kind == PROP_RDF) for (size_t k = 0; k < p->ncounts; ++k) p->weights64[k] += p->block_weights64[blk * p->ncounts + k];
        } else if (p->ahead_values.size() == p->values.size()) {
            memcpy(&p->values[f0 * p->dim1], &p->ahead_values[f0 * p->dim1], (f1 - f0) * p->dim1 * sizeof(float));
        }
        p->dirty = true;
    }
    for (size_t f = f0; f < f1; ++f) e->frame_mask[f] = 1;
    e->frames_done += f1 - f0;
    if (e->block_ready[blk]) e->block_ready[blk] = BLOCK_ROWS_IN_PLACE;
    e->ra.blk_state[blk].store(vmd_script_eval_t::RA_COMMITTED, std::memory_order_release);
    e->ra.committed_blocks += 1;
    e->ra.views_dirty = true;
    return true;
}

// Brings the accumulators up to what has been requested.  full = false (a region leader, before its region): whole requested blocks are
// committed, requested frames of direct blocks evaluated.  full = true (a call that leaves alone): also the requested frames of partly
// requested blocks - evaluated directly, the block is direct from then on - and the host views.
static bool ra_settle(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, bool full) {
    ReadAhead& ra = e->ra;
    std::lock_guard<std::mutex> sl(ra.settle_mtx);
    std::lock_guard<std::mutex> lock(e->mtx);
    HIP_OK(hipSetDevice(e->device));
    const size_t S = e->block_frames;
    if (full) (void)ra.marks_pending.exchange(false, std::memory_order_seq_cst);    // before the scan: whoever marks after this point sets it again (ra_fast)
    std::vector<std::pair<uint32_t, uint32_t>> runs;          // frames to evaluate directly
    bool tainted = false;
    for (size_t b = 0; b < e->num_blocks; ++b) {
        const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
        if (st != vmd_script_eval_t::RA_READY && st != vmd_script_eval_t::RA_DIRECT) continue;
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        if (st == vmd_script_eval_t::RA_READY) {
            size_t req = 0;
            for (size_t f = f0; f < f1; ++f) req += ra.req(f).load(std::memory_order_seq_cst) ? 1 : 0;     // seq_cst: ordered after the exchange of marks_pending above
            if (req == f1 - f0) { if (!ra_commit_block(e, b)) return false; continue; }
            if (req == 0 || !full) continue;
            ra.blk_state[b].store(vmd_script_eval_t::RA_DIRECT, std::memory_order_release);       // partly requested: its frames are evaluated one by one from now on
            tainted = true;
        }
        for (size_t f = f0; f < f1; ++f) {
            if (!ra.req(f).load(std::memory_order_seq_cst) || e->frame_mask[f]) continue;
            if (!runs.empty() && runs.back().second == f) runs.back().second = (uint32_t)f + 1;
            else runs.push_back({(uint32_t)f, (uint32_t)f + 1});
        }
    }
    for (auto& r : runs) {
        g_last_error.clear();
        if (e->interrupt) return false;
        if (!process_range_locked(e, sys, traj, r.first, r.second, false, false)) return false;
        ra.direct_frames += r.second - r.first;
        ra.views_dirty = true;
    }
    if (tainted && ++ra.strikes >= 3) ra.disabled = true;     // (settle_mtx) callers that keep leaving blocks half requested are not a pool walking a range
    const bool overdue = std::chrono::steady_clock::now() - e->views_at > std::chrono::milliseconds(std::max(1, g_opt.lazy_views_ms.load()));
    if ((full || overdue) && ra.views_dirty.exchange(false)) { if (!refresh_views_locked(e)) return false; }
    ra.settles += full ? 1 : 0;
    return true;
}

// no lock: 1 = every frame of [beg, end) lies in an evaluated (or direct) block and is now marked requested
static bool ra_fast(vmd_script_eval_t* e, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    const size_t S = e->block_frames;
    for (size_t b = beg / S; b <= (size_t)(end - 1) / S; ++b) {
        const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
        if (st != vmd_script_eval_t::RA_READY && st != vmd_script_eval_t::RA_DIRECT) return false;
    }
    for (uint32_t f = beg; f < end; ++f) if (ra.req(f).load(std::memory_order_relaxed)) return false;      // asked for twice: the slow path sorts that out
    // the marks FIRST, then the flag, both sequentially consistent (the other order lost marks - B sees or sets the flag, the
    // settling A clears it and scans B's block before B's CAS lands, B marks, and B's ra_leave finds the flag clear: requested frames that
    // nobody commits).  A settle clears the flag with a seq_cst exchange and scans after it: a mark that the scan misses is followed by a
    // store of the flag that the exchange did not clear, so the marker's own ra_leave (or a later caller's) settles again.
    for (uint32_t f = beg; f < end; ++f) { uint8_t z = 0; (void)ra.req(f).compare_exchange_strong(z, 1, std::memory_order_seq_cst); }   // a lost race = another call for the same frame owns it
    ra.marks_pending.store(true, std::memory_order_seq_cst);
    return true;
}

// the eval is not evaluating ahead for this call (a large range, a lone caller, read-ahead given up): the combining queue evaluates it when
// it arrives.  With block states in place the blocks it touches become direct FIRST, so that no partial of theirs is committed later.
static bool ra_direct_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    if (!ra.on.load(std::memory_order_acquire)) {
        { std::lock_guard<std::mutex> ql(e->queue_mtx); ra.combining += 1; }
        const bool ok = combine_call(e, sys, traj, beg, end);
        { std::lock_guard<std::mutex> ql(e->queue_mtx); ra.combining -= 1; }
        e->queue_cv.notify_all();
        return ok;
    }
    const size_t S = e->block_frames;
    const size_t b0 = beg / S, b1 = (size_t)(end - 1) / S;
    for (;;) {
        {
            std::unique_lock<std::mutex> ql(e->queue_mtx);
            e->queue_cv.wait(ql, [&] {
                if (e->interrupt) return true;
                for (size_t b = b0; b <= b1; ++b) if (ra.blk_state[b].load() == vmd_script_eval_t::RA_PENDING) return false;
                return true; });
        }
        if (e->interrupt) { g_last_error.clear(); return false; }
        // whatever has been requested in those blocks so far is settled first (committed whole, or evaluated), then they are direct
        bool ready = false;
        for (size_t b = b0; b <= b1; ++b) ready = ready || ra.blk_state[b].load() == vmd_script_eval_t::RA_READY;
        if (ready && !ra_settle(e, sys, traj, true)) return false;
        std::lock_guard<std::mutex> sl(ra.settle_mtx);
        std::lock_guard<std::mutex> ql(e->queue_mtx);
        bool pending = false;          // a region leader took one of them in the meantime: wait for it, or its partial would count these frames again
        for (size_t b = b0; b <= b1; ++b) pending = pending || ra.blk_state[b].load() == vmd_script_eval_t::RA_PENDING;
        if (pending) continue;
        for (size_t b = b0; b <= b1; ++b) {
            const uint8_t st = ra.blk_state[b].load();
            if (st == vmd_script_eval_t::RA_READY || st == vmd_script_eval_t::RA_NONE) ra.blk_state[b].store(vmd_script_eval_t::RA_DIRECT, std::memory_order_release);
        }
        break;
    }
    const bool ok = combine_call(e, sys, traj, beg, end);
    if (ok) for (uint32_t f = beg; f < end; ++f) ra.req(f).store(1, std::memory_order_release);
    return ok;
}

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

static bool ra_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    const bool small = (int)(end - beg) <= g_opt.readahead_small.load();
    if (small && !ra.disabled && ra.on.load(std::memory_order_acquire) && ra.concurrent.load(std::memory_order_relaxed) && ra.traj_inst == traj_id(traj) && ra_fast(e, beg, end)) return true;
    std::unique_lock<std::mutex> ql(e->queue_mtx);
    if ((uint32_t)ra.flight.load() >= 2 && !ra.concurrent) { ra.concurrent = true; e->queue_cv.notify_all(); }
    if (small && !ra.disabled) {
        // (also on an eval whose blocks exist from an earlier evaluation: whether THIS evaluation is driven by a pool is found out anew)
        if (!ra.concurrent && !ra.lonely) {
            const int pref = ra.lone_pref.load(std::memory_order_relaxed);
            if (pref < 0 ? g_opt.readahead_lone.load() > 0 : pref > 0) {
                // opted in: every small call is part of a walk, whoever makes it - served like a pool's, settled by the helper thread
                ra.lone.store(true);
                ra.concurrent = true;
            } else {
                // the first call of an evaluation: is this a pool?  Its other threads are microseconds behind
                cv_wait_us(e->queue_cv, ql, std::max(0, g_opt.readahead_company_us.load()), [&] { return ra.concurrent || e->interrupt.load(); });
                if (!ra.concurrent) ra.lonely = true;
            }
        }
        if (ra.concurrent && !ra.on.load()) {
            e->queue_cv.wait(ql, [&] { return ra.combining == 0 || ra.on.load(); });      // calls that went to the combining queue before anyone knew
            if (!ra.on.load() && !ra_engage(e, traj)) return false;
        }
    }
    if (!(small && !ra.disabled && ra.concurrent && ra.on.load() && ra.traj_inst == traj_id(traj))) {
        ql.unlock();
        return ra_direct_call(e, sys, traj, beg, end);
    }
    ra.slow_calls += 1;
    const size_t S = e->block_frames;
    const size_t b0 = beg / S, b1 = (size_t)(end - 1) / S;
    for (;;) {
        if (e->interrupt) { g_last_error.clear(); return false; }
        if (ra.failed) { g_last_error = ra.error; return false; }
        size_t need = (size_t)-1;
        bool pending = false;
        for (size_t b = b0; b <= b1; ++b) {
            const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
            if (st == vmd_script_eval_t::RA_NONE) { need = b; break; }
            pending = pending || st == vmd_script_eval_t::RA_PENDING;
        }
        if (need == (size_t)-1 && !pending) break;
        if (need != (size_t)-1 && !ra.spec_active) {
            // this call leads a region: whole blocks from `need` on, as far as nobody has touched them
            size_t want = ra.next_region ? ra.next_region : (size_t)std::max(1, g_opt.readahead_frames.load());
            want = std::min(std::max(want, S), std::max(ra.bmax, S));
            size_t e1 = need, frames = 0;
            while (e1 < e->num_blocks && ra.blk_state[e1].load() == vmd_script_eval_t::RA_NONE && frames < want) {
                frames += std::min((e1 + 1) * S, e->num_frames) - e1 * S;
                ++e1;
            }
            // a caller that walks the range downwards (enkiTS: the thread that owns the task set pops its partitions from the far end while
            // the others steal from the near end) finds everything above its block taken: the region grows towards lower frames instead
            // (only when the way up is blocked - by evaluated blocks or the end of the trajectory - and never across blocks that are not free)
            while (frames < want && need > 0 && ra.blk_state[need - 1].load() == vmd_script_eval_t::RA_NONE) {
                --need;
                frames += S;
            }
            ra.next_region = std::min(std::max(ra.bmax, S), want * (size_t)std::max(1, g_opt.readahead_growth.load()));
            for (size_t b = need; b < e1; ++b) ra.blk_state[b].store(vmd_script_eval_t::RA_PENDING, std::memory_order_release);
            ra.spec_active = true;
            ql.unlock();
            const uint32_t f_lo = (uint32_t)(need * S), f_hi = (uint32_t)std::min(e1 * S, e->num_frames);
            bool ok = ra_settle(e, sys, traj, false);         // what the callers have asked for so far joins the totals: progress for a polling GUI
            if (ok) {
                g_last_error.clear();
                std::lock_guard<std::mutex> lock(e->mtx);
                std::vector<char> adopted;
                ok = hipSetDevice(e->device) == hipSuccess && ra_adopt_blocks(e, traj_id(traj), need, e1, &adopted);
                for (size_t b = need; b < e1 && ok;) {           // what the source could not supply: evaluated, in runs of blocks
                    if (adopted[b - need]) { ++b; continue; }
                    size_t r1 = b;
                    while (r1 < e1 && !adopted[r1 - need]) ++r1;
                    ok = !e->interrupt && process_range_locked(e, sys, traj, (uint32_t)(b * S), (uint32_t)std::min(r1 * S, e->num_frames), false, true);
                    b = r1;
                }
            }
            const std::string err = ok ? std::string() : g_last_error;
            ql.lock();
            ra.spec_active = false;
            for (size_t b = need; b < e1; ++b) ra.blk_state[b].store(ok ? vmd_script_eval_t::RA_READY : vmd_script_eval_t::RA_NONE, std::memory_order_release);
            if (ok) { ra.regions += 1; ra.region_frames += f_hi - f_lo; }
            else if (!e->interrupt) { ra.failed = true; ra.error = err; }
            e->queue_cv.notify_all();
            if (!ok) { g_last_error = err; return false; }
            continue;
        }
        e->queue_cv.wait(ql);
    }
    ql.unlock();
    // every block is evaluated (READY), direct or already committed: mark what can be marked, evaluate the rest now (frames asked for
    // twice - the combining queue counts them twice, as it always has)
    std::vector<std::pair<uint32_t, uint32_t>> again;
    for (uint32_t f = beg; f < end; ++f) {
        uint8_t z = 0;
        const bool committed = ra.blk_state[f / S].load(std::memory_order_acquire) == vmd_script_eval_t::RA_COMMITTED;
        if (!committed && ra.req(f).compare_exchange_strong(z, 1, std::memory_order_seq_cst)) continue;
        if (!again.empty() && again.back().second == f) again.back().second = f + 1;
        else again.push_back({f, f + 1});
    }
    ra.marks_pending.store(true, std::memory_order_seq_cst);      // after the marks, as in ra_fast
    for (auto& r : again) if (!combine_call(e, sys, traj, r.first, r.second)) return false;
    return true;
}

// ---- deferred settle (option readahead_lone) ----------------------------------------------------------------------------------
static int64_t steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The helper thread of an eval in deferred-settle mode: sleeps until a settle is owed (armed) and the eval has been quiet for
// readahead_lone_settle_us since the last call left, then does what a pool's last leaver does - as a call of its own (flight + 1), so every
// hand-over rule of ra_settle / ra_leave holds unchanged.  Marks that arrive while it settles keep it armed.
static void lone_helper_main(vmd_script_eval_t* e) {
    ReadAhead& ra = e->ra;
    ReadAhead::Helper& h = ra.helper;
    std::unique_lock<std::mutex> lk(h.mtx);
    for (;;) {
        h.cv.wait(lk, [&] { return h.quit || h.armed.load(); });
        if (h.quit) break;
        for (;;) {
            const int64_t due = h.last_leave_ns.load() + (int64_t)std::max(1, g_opt.readahead_lone_settle_us.load()) * 1000;
            const int64_t now = steady_ns();
            if (h.quit || !h.armed.load() || now >= due) break;
            cv_wait_us(h.cv, lk, (int)((due - now) / 1000 + 1), [&] { return h.quit || !h.armed.load(); });
        }
        if (h.quit) break;
        if (!h.armed.load() || !h.have) continue;        // cancelled (clear_data, wait_settled)
        if (e->interrupt.load()) { h.armed.store(false); h.idle_cv.notify_all(); continue; }     // an interrupted evaluation is not completed behind the host's back
        h.busy = true;
        const uint64_t seq = h.cancel_seq;
        vmd_system_t sys = h.sys;
        vmd_trajectory_i traj = h.traj;
        lk.unlock();
        bool retry = false;
        {
            const uint64_t w = ra.flight.fetch_add(1, std::memory_order_acq_rel);
            if ((uint32_t)w == 0) {
                g_last_error.clear();
                const bool ok = ra_settle(e, &sys, &traj, true);
                if (!ok && !e->interrupt && !g_last_error.empty()) {
                    std::lock_guard<std::mutex> ql(e->queue_mtx);
                    ra.failed = true; ra.error = g_last_error;                 // the next call reports it
                }
                h.settles += 1;
            } else {
                retry = true;                                                     // a call is inside: it stamps last_leave when it goes
            }
            ra.flight.fetch_sub(1, std::memory_order_acq_rel);
            // the host's records of this eval follow NOW (the shim re-publishes fingerprint / ranges / max_value: ADVICE r05 #1) - after the
            // settle, before `busy` drops: clear_data / interrupt / free wait for the callback too, it never runs on a freed host object
            if (!retry) if (auto cb = h.on_settled.load(std::memory_order_acquire)) cb(h.on_settled_user.load(std::memory_order_acquire));
        }
        lk.lock();
        h.busy = false;
        if (h.cancel_seq != seq) {
            // cancelled while it ran (interrupt, clear_data, wait_settled): whatever is marked from now on belongs to calls that arm afresh -
            // with THEIR system and trajectory (lone_arm copies them only when it arms)
            h.armed.store(false, std::memory_order_seq_cst);
        } else if (retry) {
            h.last_leave_ns.store(std::max(h.last_leave_ns.load(), steady_ns()));
        } else {
            // Disarm, THEN look at the marks (both seq_cst) - the mirror image of a leaving call, which marks and then looks at `armed`
            // (lone_arm): at least one of the two sees the other, so a mark made while this settle ran is never left without an owner
            h.armed.store(false, std::memory_order_seq_cst);
            if (ra.marks_pending.load(std::memory_order_seq_cst) || ra.views_dirty.load(std::memory_order_seq_cst)) h.armed.store(true, std::memory_order_seq_cst);
        }
        h.idle_cv.notify_all();
    }
}

// a call that leaves last in deferred-settle mode: stamp the time, make sure the helper knows a settle is owed
static void lone_arm(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj) {
    ReadAhead::Helper& h = e->ra.helper;
    h.last_leave_ns.store(steady_ns(), std::memory_order_relaxed);
    if (h.armed.load(std::memory_order_seq_cst)) return;          // (the caller's marks are seq_cst stores before this load: see lone_helper_main)
    std::lock_guard<std::mutex> l(h.mtx);
    if (sys) h.sys = *sys; else memset(&h.sys, 0, sizeof(h.sys));
    h.traj = *traj;
    h.have = true;
    if (!h.started) { h.started = true; h.th = std::thread(lone_helper_main, e); }
    h.armed.store(true, std::memory_order_release);
    h.cv.notify_one();
}
// clear_data / wait_settled: no settle may start from now on, and none is running when this returns (call WITHOUT e->mtx held)
static void lone_cancel(vmd_script_eval_t* e) {
    ReadAhead::Helper& h = e->ra.helper;
    std::unique_lock<std::mutex> lk(h.mtx);
    if (!h.started) return;
    h.cancel_seq += 1;
    h.armed.store(false);
    h.cv.notify_one();
    h.idle_cv.wait(lk, [&] { return !h.busy; });
}
static void lone_stop(vmd_script_eval_t* e) {           // vmd_eval_free
    ReadAhead::Helper& h = e->ra.helper;
    {
        std::lock_guard<std::mutex> l(h.mtx);
        if (!h.started) return;
        e->interrupt = true;                            // a settle that is running ends at its next batch boundary: nobody will read its results
        h.quit = true;
        h.cv.notify_one();
    }
    h.th.join();
}

// the end of every call: whoever leaves last settles (or hands the duty to a call that has arrived since)
static bool ra_leave(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj) {
    ReadAhead& ra = e->ra;
    const bool lone = ra.lone.load(std::memory_order_relaxed);
    if (lone) ra.helper.last_leave_ns.store(steady_ns(), std::memory_order_relaxed);
    for (;;) {
        const uint64_t w = ra.flight.fetch_sub(1, std::memory_order_acq_rel);
        if ((uint32_t)w != 1) return true;
        if (!ra.on.load(std::memory_order_acquire) || (!ra.marks_pending.load() && !ra.views_dirty.load())) return true;
        if (e->interrupt) return true;
        if (lone) { lone_arm(e, sys, traj); return true; }      // deferred: the helper settles once the eval has been quiet
        const uint64_t a0 = w >> 32;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(std::max(0, g_opt.readahead_linger_us.load()));
        while (std::chrono::steady_clock::now() < deadline) {
            if ((ra.flight.load(std::memory_order_acquire) >> 32) != a0) return true;
            std::this_thread::yield();
        }
        if ((ra.flight.load(std::memory_order_acquire) >> 32) != a0) return true;
        ra.flight.fetch_add(1, std::memory_order_acq_rel);
        if (!ra_settle(e, sys, traj, true)) { ra.flight.fetch_sub(1, std::memory_order_acq_rel); return false; }
    }
}

extern "C" bool vmd_eval_frame_range(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys,
                                     vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    g_last_error.clear();   // a false return with an empty message means "interrupted"
    if (!eval || !traj) return vmd_fail("vmd_eval_frame_range: NULL argument");
    if (ir && vmd_ir_fingerprint(ir) != eval->ir_fingerprint) return vmd_fail("vmd_eval_frame_range: eval was created from a different ir");
    if (frame_end > eval->num_frames) frame_end = (uint32_t)eval->num_frames;
    if (frame_beg >= frame_end) return true;
    if (eval->interrupt) return false;
    if (!g_opt.readahead.load()) {
        if (!eval->ra.on.load()) return combine_call(eval, sys, traj, frame_beg, frame_end);
        eval->ra.flight.fetch_add(((uint64_t)1 << 32) | 1, std::memory_order_acq_rel);
        const bool ok = ra_direct_call(eval, sys, traj, frame_beg, frame_end);
        const std::string err = ok ? std::string() : g_last_error;
        const bool lok = ra_leave(eval, sys, traj);
        if (!ok) g_last_error = err;
        return ok && lok;
    }
    eval->ra.flight.fetch_add(((uint64_t)1 << 32) | 1, std::memory_order_acq_rel);
    const bool ok = ra_call(eval, sys, traj, frame_beg, frame_end);
    const std::string err = ok ? std::string() : g_last_error;
    const bool lok = ra_leave(eval, sys, traj);
    if (!ok) g_last_error = err;
    return ok && lok;
}

// VIAMD's call pattern as a utility (src/main.cpp:993-997, src/task_system.cpp:73-81): `num_threads` pool threads pull ranges of `grain`
// frames off [frame_beg, frame_end) and call vmd_eval_frame_range on the ONE eval, each blocking until its frames are evaluated.  What
// bench.py times VIAMD's pattern with (native threads: a Python thread per call costs more than a small call does), and what a host
// without a task system of its own can use as is.  Returns false if any call failed or was interrupted.
extern "C" bool vmd_eval_frame_range_pooled(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys, vmd_trajectory_i* traj,
                                            uint32_t frame_beg, uint32_t frame_end, int num_threads, uint32_t grain) {
    if (!eval || !traj) return vmd_fail("vmd_eval_frame_range_pooled: NULL argument");
    if (num_threads < 1) num_threads = 1;
    if (grain < 1) grain = 1;
    std::atomic<uint32_t> next{frame_beg};
    std::atomic<bool> ok{true};
    std::mutex err_mtx;
    std::string err;
    auto work = [&] {
        for (;;) {
            const uint32_t b = next.fetch_add(grain, std::memory_order_relaxed);
            if (b >= frame_end || b < frame_beg) break;           // (b < frame_beg: the counter wrapped)
            const uint32_t e = frame_end - b < grain ? frame_end : b + grain;
            if (!vmd_eval_frame_range(eval, ir, sys, traj, b, e)) {
                std::lock_guard<std::mutex> l(err_mtx);
                if (err.empty()) err = g_last_error;               // thread-local in the worker: carried to the caller below
                ok.store(false);
                break;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < num_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (!ok.load()) g_last_error = err;
    return ok.load();
}

extern "C" bool vmd_eval_set_settled_callback(vmd_script_eval_t* eval, void (*fn)(void*), void* user) {
    if (!eval) return vmd_fail("eval is NULL");
    eval->ra.helper.on_settled_user.store(user, std::memory_order_release);
    eval->ra.helper.on_settled.store(fn, std::memory_order_release);
    return true;
}
extern "C" bool vmd_eval_set_deferred_settle(vmd_script_eval_t* eval, int mode) {
    if (!eval) return vmd_fail("eval is NULL");
    eval->ra.lone_pref.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed);       // takes effect at the next clear_data / first small call of an evaluation
    return true;
}

// Deferred-settle mode (option readahead_lone): everything the calls so far have asked for joins the totals and the views NOW, on the
// calling thread, instead of when the helper's quiet period is over.  Call after the last vmd_eval_frame_range has returned; a no-op for
// every other eval (their last call has settled before it returned).
extern "C" bool vmd_eval_wait_settled(vmd_script_eval_t* eval) {
    if (!eval) return vmd_fail("eval is NULL");
    ReadAhead& ra = eval->ra;
    if (!ra.lone.load()) return true;
    lone_cancel(eval);
    ReadAhead::Helper& h = ra.helper;
    vmd_system_t sys; vmd_trajectory_i traj;
    {
        std::lock_guard<std::mutex> l(h.mtx);
        if (!h.have) return true;
        sys = h.sys; traj = h.traj;
    }
    g_last_error.clear();
    bool ok = true;
    ra.flight.fetch_add(1, std::memory_order_acq_rel);
    if (ra.on.load(std::memory_order_acquire) && (ra.marks_pending.load() || ra.views_dirty.load())) ok = ra_settle(eval, &sys, &traj, true);
    ra.flight.fetch_sub(1, std::memory_order_acq_rel);
    if (ok) { std::lock_guard<std::mutex> ql(eval->queue_mtx); if (ra.failed) { g_last_error = ra.error; ok = false; } }
    if (auto cb = h.on_settled.load(std::memory_order_acquire)) cb(h.on_settled_user.load(std::memory_order_acquire));
    return ok;
}

extern "C" void vmd_eval_readahead_stats(const vmd_script_eval_t* eval, vmd_readahead_stats_t* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    if (!eval) return;
    const ReadAhead& ra = eval->ra;
    out->engaged = ra.on.load() ? 1 : 0;
    out->block_frames = ra.on.load() ? (uint32_t)eval->block_frames : 0;
    out->regions = ra.regions.load(); out->region_frames = ra.region_frames.load();
    out->slow_calls = ra.slow_calls.load();
    out->settles = ra.settles.load(); out->direct_frames = ra.direct_frames.load(); out->committed_blocks = ra.committed_blocks.load();
}

extern "C" const int32_t* vmd_eval_sdf_structures(const vmd_script_eval_t* eval, const char* name, size_t* num_structures, size_t* atoms_per_structure) {
    PropState* p = find_prop(eval, name);
    if (!p || p->prop.kind != PROP_SDF) { vmd_fail("'%s' is not an sdf property", name ? name : "(null)"); return nullptr; }
    if (num_structures) *num_structures = p->prop.K;
    if (atoms_per_structure) *atoms_per_structure = p->prop.m;
    return p->prop.a.data();
}

extern "C" bool vmd_eval_sdf_matrices(vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys,
                                      vmd_trajectory_i* traj, uint32_t frame, float* matrices, size_t* K_out, float* extent_out) {
    if (!eval || !traj) return vmd_fail("vmd_eval_sdf_matrices: NULL argument");
    PropState* p = find_prop(eval, name);
    if (!p || p->prop.kind != PROP_SDF) return vmd_fail("'%s' is not an sdf property", name ? name : "(null)");
    std::lock_guard<std::mutex> lock(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    vmd_script_eval_t* e = eval;
    const size_t num_atoms = traj->num_atoms(traj->inst);
    if (!check_atoms(e, num_atoms) || !upload_static(e, sys, num_atoms)) return false;
    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    BatchSrc src;
    if (!p->ref_pose_ready) {
        if (!fetch_batch(e, traj, view_holds(have_view, view, 0) ? &view : nullptr, num_atoms, 0, 1, &src)) return false;
        KRN_OK(vmd_hip_sdf_ref_pose(e->stream, src.base, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), p->d_structs.p, p->d_mass.p,
                                    (int)p->prop.m, p->d_ref_pose.p, p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr));
        HIP_OK(hipStreamSynchronize(e->stream));
        p->ref_pose_ready = true;
    }
    if (!fetch_batch(e, traj, view_holds(have_view, view, frame) ? &view : nullptr, num_atoms, frame, 1, &src)) return false;
    const size_t K = p->prop.K;
    DevBuf<double> dM;
    if (!dM.ensure(K * 12) || !p->d_R32.ensure(K * 9) || !p->d_c32.ensure(K * 3)) return false;
    if (p->have_tree && !p->d_tree_pos.ensure(K * p->prop.m * 3)) return false;
    KRN_OK(vmd_hip_sdf_align(e->stream, src.base, src.frame_stride, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), 1,
                             p->d_structs.p, p->d_mass.p, (int)K, (int)p->prop.m, p->d_ref_pose.p, p->d_R32.p, p->d_c32.p, dM.p, nullptr,
                             p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr, p->have_tree ? p->d_tree_pos.p : nullptr));
    std::vector<double> M(K * 12);
    HIP_OK(hipMemcpyAsync(M.data(), dM.p, K * 12 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (matrices) {
        for (size_t k = 0; k < K; ++k) {
            float* o = matrices + 16 * k;   // column-major mat4
            const double* r = &M[12 * k];
            for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 3; ++rr) o[4 * c + rr] = (float)r[4 * rr + c];
            o[3] = 0.0f; o[7] = 0.0f; o[11] = 0.0f; o[15] = 1.0f;
        }
    }
    if (K_out) *K_out = K;
    if (extent_out) *extent_out = p->prop.rmax;
    return true;
}

// ------------------------------------------------------------------------------------------------ device trajectory

static uint64_t next_cells_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1) + 1;
}
struct vmd_devtraj_t {
    size_t num_frames = 0, num_atoms = 0, npad = 0;
    size_t first = 0, resident = 0;     // frames [first, first + resident) are in HBM (a rank's shard; the whole trajectory otherwise)
    float* d = nullptr;                 // frame `first`
    float* d0 = nullptr;                // shards that do not start at frame 0 keep a copy of it: the SDF reference pose is taken there (SPEC S5)
    bool has(size_t beg, size_t end) const { return (beg >= first && end <= first + resident && beg <= end) || (d0 && beg == 0 && end == 1); }
    float* frame(size_t f) const { return (d0 && f == 0) ? d0 : d + (f - first) * 3 * npad; }
    int device = 0;
    std::vector<vmd_unitcell_t> cells;
    uint64_t cells_version = next_cells_version();   // a new process-wide number for every change of `cells` or of the coordinates: two
                                                     // trajectories (one freed, one created at the same address) never share one
    vmd_trajectory_i iface;
};

static size_t dt_num_frames(void* inst) { return ((vmd_devtraj_t*)inst)->num_frames; }
static size_t dt_num_atoms(void* inst) { return ((vmd_devtraj_t*)inst)->num_atoms; }
static bool dt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_devtraj_t* t = (vmd_devtraj_t*)inst;
    if (idx < 0 || !t->has((size_t)idx, (size_t)idx + 1)) return vmd_fail("devtraj: frame %lld is not resident on this rank", (long long)idx);
    const float* f = t->frame((size_t)idx);
    if (x) HIP_OK(hipMemcpy(x, f, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (y) HIP_OK(hipMemcpy(y, f + t->npad, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (z) HIP_OK(hipMemcpy(z, f + 2 * t->npad, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (hdr) { hdr->num_atoms = t->num_atoms; hdr->index = idx; hdr->timestamp = (double)idx; hdr->unitcell = t->cells[idx]; }
    return true;
}
static bool dt_device_view(void* inst, vmd_device_view_t* out) {
    vmd_devtraj_t* t = (vmd_devtraj_t*)inst;
    // frame f sits at base + f * frame_stride: for a shard the base lies `first` frames before the allocation and is only ever
    // used with resident frame indices (the evaluator is handed ranges inside the shard)
    out->base = t->d - t->first * 3 * t->npad; out->frame_stride = 3 * t->npad; out->row_stride = t->npad; out->cells = t->cells.data(); out->device = t->device;
    out->resident_beg = t->first; out->resident_end = t->first + t->resident;
    out->cells_version = t->cells_version;
    return true;
}

extern "C" vmd_devtraj_t* vmd_devtraj_create_shard(size_t num_frames, size_t frame_beg, size_t frame_end, size_t num_atoms) {
    if (vmd_device_count() <= 0) { vmd_fail("vmd_devtraj_create: no usable HIP device"); return nullptr; }
    if (frame_beg > frame_end || frame_end > num_frames) { vmd_fail("vmd_devtraj_create_shard: bad frame range"); return nullptr; }
    auto t = std::make_unique<vmd_devtraj_t>();
    t->num_frames = num_frames; t->num_atoms = num_atoms; t->npad = (num_atoms + 63) & ~(size_t)63;
    t->first = frame_beg; t->resident = frame_end - frame_beg;
    if (hipGetDevice(&t->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    const size_t bytes = std::max<size_t>(t->resident * 3 * t->npad, 1) * sizeof(float);
    hipError_t err = hipMalloc((void**)&t->d, bytes);
    if (err == hipSuccess && frame_beg > 0) err = hipMalloc((void**)&t->d0, 3 * t->npad * sizeof(float));
    if (err != hipSuccess) { vmd_fail("vmd_devtraj_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err)); return nullptr; }
    vmd_unitcell_t none;
    memset(&none, 0, sizeof(none));
    t->cells.assign(num_frames, none);
    t->iface.inst = t.get();
    t->iface.num_frames = dt_num_frames; t->iface.num_atoms = dt_num_atoms;
    t->iface.load_frame = dt_load_frame; t->iface.device_view = dt_device_view; t->iface.host_view = nullptr;
    t->iface.load_raw = nullptr;
    t->iface.raw_device_view = nullptr;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}
extern "C" vmd_devtraj_t* vmd_devtraj_create(size_t num_frames, size_t num_atoms) { return vmd_devtraj_create_shard(num_frames, 0, num_frames, num_atoms); }
extern "C" void vmd_devtraj_free(vmd_devtraj_t* t) {
    if (!t) return;
    if (t->d) (void)hipFree(t->d);
    if (t->d0) (void)hipFree(t->d0);
    delete t;
}
extern "C" vmd_trajectory_i* vmd_devtraj_interface(vmd_devtraj_t* t) { return t ? &t->iface : nullptr; }

extern "C" bool vmd_devtraj_upload_frame(vmd_devtraj_t* t, size_t frame, const vmd_unitcell_t* cell,
                                         const float* x, const float* y, const float* z) {
    if (!t || !t->has(frame, frame + 1)) return vmd_fail("vmd_devtraj_upload_frame: bad frame");
    float* f = t->frame(frame);
    HIP_OK(hipMemcpy(f, x, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f + t->npad, y, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f + 2 * t->npad, z, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    if (cell) t->cells[frame] = *cell;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" bool vmd_devtraj_upload_atoms(vmd_devtraj_t* t, size_t frame_beg, size_t frame_count, size_t first_atom, size_t atom_count,
                                        const float* xyz /* [frame_count][3][atom_count] */) {
    if (!t || !t->has(frame_beg, frame_beg + frame_count) || first_atom + atom_count > t->num_atoms) return vmd_fail("vmd_devtraj_upload_atoms: bad range");
    for (size_t f = 0; f < frame_count; ++f)
        for (int c = 0; c < 3; ++c)
            HIP_OK(hipMemcpyAsync(t->frame(frame_beg + f) + (size_t)c * t->npad + first_atom,
                                  xyz + (f * 3 + c) * atom_count, atom_count * sizeof(float), hipMemcpyHostToDevice, nullptr));
    HIP_OK(hipDeviceSynchronize());
    t->cells_version = next_cells_version();       // coordinates changed: bounding boxes cached per range (open axes) are stale too
    return true;
}

extern "C" bool vmd_devtraj_synth(vmd_devtraj_t* t, uint64_t seed, float L, float sigma, uint32_t n_blob,
                                  size_t frame_beg, size_t frame_end) {
    if (!t || !t->has(frame_beg, frame_end)) return vmd_fail("vmd_devtraj_synth: bad range");
    vmd_unitcell_t c;
    memset(&c, 0, sizeof(c));
    c.x = c.y = c.z = L; c.flags = VMD_UNITCELL_PBC_ALL;
    for (size_t f0 = frame_beg; f0 < frame_end; f0 += 1024) {
        const size_t nb = std::min<size_t>(1024, frame_end - f0);
        KRN_OK(vmd_hip_synth_frames(nullptr, t->frame(f0), 3 * t->npad, t->npad, (int)nb, (uint32_t)f0, seed,
                                    (uint32_t)t->num_atoms, n_blob, L, sigma));
    }
    if (t->d0) {    // the shard's copy of frame 0
        KRN_OK(vmd_hip_synth_frames(nullptr, t->d0, 3 * t->npad, t->npad, 1, 0u, seed, (uint32_t)t->num_atoms, n_blob, L, sigma));
        t->cells[0] = c;
    }
    HIP_OK(hipDeviceSynchronize());
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = c;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" bool vmd_devtraj_set_cell(vmd_devtraj_t* t, size_t frame_beg, size_t frame_end, const vmd_unitcell_t* cell) {
    if (!t || !cell || frame_end > t->num_frames || frame_beg > frame_end) return vmd_fail("vmd_devtraj_set_cell: bad frame range");
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = *cell;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" float* vmd_devtraj_device_ptr(vmd_devtraj_t* t, size_t* frame_stride, size_t* row_stride) {
    if (!t) return nullptr;
    if (frame_stride) *frame_stride = 3 * t->npad;
    if (row_stride) *row_stride = t->npad;
    return t->d;
}

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
static size_t rt_num_frames(void* inst) { return ((vmd_rawtraj_t*)inst)->num_frames; }
static size_t rt_num_atoms(void* inst) { return ((vmd_rawtraj_t*)inst)->num_atoms; }
static bool rt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    return t->src->load_frame(t->src->inst, idx, hdr, x, y, z);
}
static bool rt_load_raw(void* inst, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst, size_t cap) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    return t->src->load_raw(t->src->inst, idx, hdr, info, dst, cap);
}
static bool rt_raw_device_view(void* inst, vmd_raw_device_view_t* out) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    out->base = t->d_raw; out->info = t->d_info; out->cells = t->cells.data(); out->codec = VMD_RAW_CODEC_XTC; out->device = t->device;
    out->ck = t->d_ck; out->nck = t->d_nck; out->ck_have = t->d_ck ? t->ck_have.data() : nullptr;
    out->rec = t->d_rec; out->nrec = t->d_nrec; out->rec_stride = t->d_rec ? t->rec_stride : 0; out->rec_failed = &t->rec_failed;
    return true;
}

extern "C" void vmd_rawtraj_free(vmd_rawtraj_t* t) {
    if (!t) return;
    if (t->d_raw) (void)hipFree(t->d_raw);
    if (t->d_info) (void)hipFree(t->d_info);
    if (t->d_ck) (void)hipFree(t->d_ck);
    if (t->d_nck) (void)hipFree(t->d_nck);
    if (t->d_rec) (void)hipFree(t->d_rec);
    if (t->d_nrec) (void)hipFree(t->d_nrec);
    delete t;
}

extern "C" vmd_rawtraj_t* vmd_rawtraj_create(vmd_trajectory_i* src) {
    if (!src || !src->load_raw) { vmd_fail("vmd_rawtraj_create: the trajectory does not offer its frames compressed (load_raw)"); return nullptr; }
    if (vmd_device_count() <= 0) { vmd_fail("vmd_rawtraj_create: no usable HIP device"); return nullptr; }
    std::unique_ptr<vmd_rawtraj_t, void (*)(vmd_rawtraj_t*)> t(new vmd_rawtraj_t(), vmd_rawtraj_free);
    t->src = src;
    t->num_frames = src->num_frames(src->inst);
    t->num_atoms = src->num_atoms(src->inst);
    if (hipGetDevice(&t->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    const size_t F = t->num_frames;
    std::vector<vmd_xtc_frame_t> info(F);
    t->cells.resize(F);
    size_t total = 0;
    for (size_t f = 0; f < F; ++f) {
        vmd_frame_header_t hdr;
        vmd_raw_frame_t ri;
        if (!src->load_raw(src->inst, (int64_t)f, &hdr, &ri, nullptr, 0) || ri.codec != VMD_RAW_CODEC_XTC || hdr.num_atoms != t->num_atoms) {
            vmd_fail("vmd_rawtraj_create: frame %zu is not available compressed", f);
            return nullptr;
        }
        t->cells[f] = hdr.unitcell;
        info[f].precision = ri.precision;
        for (int k = 0; k < 3; ++k) { info[f].minint[k] = ri.minint[k]; info[f].maxint[k] = ri.maxint[k]; }
        info[f].smallidx = ri.smallidx;
        info[f].offset = total;
        info[f].nbytes = ri.nbytes;
        total += ((size_t)ri.nbytes + 32 + 63) & ~(size_t)63;            // the layout the decode kernels expect (vmd_hip.h)
    }
    t->bytes = total;
    hipError_t err = hipMalloc((void**)&t->d_raw, std::max<size_t>(total, 64));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_info, std::max<size_t>(F, 1) * sizeof(vmd_xtc_frame_t));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_ck, std::max<size_t>(F, 1) * VMD_XTC_CK_MAX * sizeof(vmd_xtc_ck_t));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_nck, std::max<size_t>(F, 1) * sizeof(uint32_t));
    t->ck_have.assign(F, 0);
    if (err != hipSuccess) { vmd_fail("vmd_rawtraj_create: hipMalloc(%zu) failed: %s", total, hipGetErrorString(err)); return nullptr; }
    t->rec_stride = record_stride_for(F, t->num_atoms, 2);
    if (t->rec_stride && (hipMalloc((void**)&t->d_rec, F * t->rec_stride * sizeof(uint16_t)) != hipSuccess || hipMalloc((void**)&t->d_nrec, F * sizeof(uint32_t)) != hipSuccess)) {
        (void)hipGetLastError();                   // no room for the records: the sections are walked from their checkpoints as before
        if (t->d_rec) (void)hipFree(t->d_rec);
        t->d_rec = nullptr; t->rec_stride = 0;
    }
    if (F && hipMemcpy(t->d_info, info.data(), F * sizeof(vmd_xtc_frame_t), hipMemcpyHostToDevice) != hipSuccess) { vmd_fail("vmd_rawtraj_create: upload failed"); return nullptr; }
    // upload in pinned pieces of <= 256 MB, each filled by the load threads
    const size_t piece_cap = std::min<size_t>(std::max<size_t>(total, 64), (size_t)256 << 20);
    unsigned char* pin = nullptr;
    size_t pin_cap = 0;
    auto grow = [&](size_t need) {
        if (need <= pin_cap) return true;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; pin_cap = 0;
        if (hipHostMalloc((void**)&pin, need, hipHostMallocDefault) != hipSuccess) return false;
        pin_cap = need;
        return true;
    };
    bool good = true;
    for (size_t f0 = 0; f0 < F && good;) {
        size_t f1 = f0, piece = 0;
        while (f1 < F && (f1 == f0 || piece + (info[f1].offset + (((size_t)info[f1].nbytes + 32 + 63) & ~(size_t)63) - info[f1].offset) <= piece_cap)) {
            piece = info[f1].offset + (((size_t)info[f1].nbytes + 32 + 63) & ~(size_t)63) - info[f0].offset;
            ++f1;
        }
        if (!grow(piece)) { vmd_fail("hipHostMalloc(%zu bytes) failed", piece); good = false; break; }
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), (f1 - f0) / 4));
        std::atomic<size_t> next{f0};
        std::atomic<bool> ok{true};
        auto work = [&]() {
            for (;;) {
                const size_t f = next.fetch_add(1);
                if (f >= f1 || !ok.load()) break;
                unsigned char* dst = pin + (info[f].offset - info[f0].offset);
                vmd_raw_frame_t ri;
                if (!src->load_raw(src->inst, (int64_t)f, nullptr, &ri, dst, (size_t)info[f].nbytes) || ri.nbytes != info[f].nbytes) { ok = false; break; }
                memset(dst + info[f].nbytes, 0, (((size_t)info[f].nbytes + 32 + 63) & ~(size_t)63) - (size_t)info[f].nbytes);
            }
        };
        if (nthreads == 1) work();
        else {
            std::vector<std::thread> pool;
            for (size_t k = 1; k < nthreads; ++k) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
        }
        if (!ok.load()) { if (g_last_error.empty()) vmd_fail("vmd_rawtraj_create: reading the compressed frames failed"); good = false; break; }
        if (hipMemcpy(t->d_raw + info[f0].offset, pin, piece, hipMemcpyHostToDevice) != hipSuccess) { vmd_fail("vmd_rawtraj_create: upload failed"); good = false; break; }
        f0 = f1;
    }
    if (pin) (void)hipHostFree(pin);
    if (!good) return nullptr;
    t->iface.inst = t.get();
    t->iface.num_frames = rt_num_frames; t->iface.num_atoms = rt_num_atoms;
    t->iface.load_frame = rt_load_frame; t->iface.device_view = nullptr; t->iface.host_view = nullptr;
    t->iface.load_raw = rt_load_raw;
    t->iface.raw_device_view = rt_raw_device_view;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}
extern "C" vmd_trajectory_i* vmd_rawtraj_interface(vmd_rawtraj_t* t) { return t ? &t->iface : nullptr; }
extern "C" size_t vmd_rawtraj_device_bytes(const vmd_rawtraj_t* t) {
    if (!t) return 0;           // everything the object keeps in HBM: bit streams, frame table, checkpoints, group records
    return t->bytes + t->num_frames * (sizeof(vmd_xtc_frame_t) + VMD_XTC_CK_MAX * sizeof(vmd_xtc_ck_t) + sizeof(uint32_t)) +
           (t->d_rec ? t->num_frames * (t->rec_stride * sizeof(uint16_t) + sizeof(uint32_t)) : 0);
}

// ------------------------------------------------------------------------------------------------ host trajectory (pinned)

struct vmd_hosttraj_t {
    size_t num_frames = 0, num_atoms = 0, npad = 0;
    float* h = nullptr;
    std::vector<vmd_unitcell_t> cells;
    vmd_trajectory_i iface;
};
static size_t ht_num_frames(void* inst) { return ((vmd_hosttraj_t*)inst)->num_frames; }
static size_t ht_num_atoms(void* inst) { return ((vmd_hosttraj_t*)inst)->num_atoms; }
static bool ht_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_hosttraj_t* t = (vmd_hosttraj_t*)inst;
    if (idx < 0 || (size_t)idx >= t->num_frames) return vmd_fail("hosttraj: frame %lld out of range", (long long)idx);
    const float* f = t->h + (size_t)idx * 3 * t->npad;
    if (x) memcpy(x, f, t->num_atoms * sizeof(float));
    if (y) memcpy(y, f + t->npad, t->num_atoms * sizeof(float));
    if (z) memcpy(z, f + 2 * t->npad, t->num_atoms * sizeof(float));
    if (hdr) { hdr->num_atoms = t->num_atoms; hdr->index = idx; hdr->timestamp = (double)idx; hdr->unitcell = t->cells[idx]; }
    return true;
}
static bool ht_host_view(void* inst, vmd_host_view_t* out) {
    vmd_hosttraj_t* t = (vmd_hosttraj_t*)inst;
    out->base = t->h; out->frame_stride = 3 * t->npad; out->row_stride = t->npad; out->cells = t->cells.data();
    return true;
}
extern "C" vmd_hosttraj_t* vmd_hosttraj_create(size_t num_frames, size_t num_atoms) {
    auto t = std::make_unique<vmd_hosttraj_t>();
    t->num_frames = num_frames; t->num_atoms = num_atoms; t->npad = (num_atoms + 63) & ~(size_t)63;
    const size_t bytes = std::max<size_t>(num_frames * 3 * t->npad, 1) * sizeof(float);
    if (hipHostMalloc((void**)&t->h, bytes, hipHostMallocDefault) != hipSuccess) { vmd_fail("vmd_hosttraj_create: hipHostMalloc(%zu) failed", bytes); return nullptr; }
    vmd_unitcell_t none;
    memset(&none, 0, sizeof(none));
    t->cells.assign(num_frames, none);
    t->iface.inst = t.get();
    t->iface.num_frames = ht_num_frames; t->iface.num_atoms = ht_num_atoms; t->iface.load_frame = ht_load_frame;
    t->iface.device_view = nullptr; t->iface.host_view = ht_host_view;
    t->iface.load_raw = nullptr;
    t->iface.raw_device_view = nullptr;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}
extern "C" void vmd_hosttraj_free(vmd_hosttraj_t* t) { if (!t) return; if (t->h) (void)hipHostFree(t->h); delete t; }
extern "C" vmd_trajectory_i* vmd_hosttraj_interface(vmd_hosttraj_t* t) { return t ? &t->iface : nullptr; }
extern "C" float* vmd_hosttraj_frame_ptr(vmd_hosttraj_t* t, size_t frame, size_t* row_stride) {
    if (!t || frame >= t->num_frames) return nullptr;
    if (row_stride) *row_stride = t->npad;
    return t->h + frame * 3 * t->npad;
}
extern "C" bool vmd_hosttraj_set_cell(vmd_hosttraj_t* t, size_t frame, const vmd_unitcell_t* cell) {
    if (!t || !cell || frame >= t->num_frames) return vmd_fail("vmd_hosttraj_set_cell: bad frame");
    t->cells[frame] = *cell;
    return true;
}
extern "C" bool vmd_hosttraj_copy_from_device(vmd_hosttraj_t* t, vmd_devtraj_t* src, size_t frame_beg, size_t frame_end) {
    if (!t || !src || frame_end > t->num_frames || frame_end > src->num_frames || src->num_atoms != t->num_atoms) return vmd_fail("vmd_hosttraj_copy_from_device: shape mismatch");
    if (frame_beg >= frame_end) return true;
    if (!src->has(frame_beg, frame_end)) return vmd_fail("vmd_hosttraj_copy_from_device: frames are not resident on this rank");
    HIP_OK(hipMemcpy(t->h + frame_beg * 3 * t->npad, src->frame(frame_beg), (frame_end - frame_beg) * 3 * t->npad * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = src->cells[f];
    return true;
}

// ------------------------------------------------------------------------------------------------ consumer post-processing

// what VIAMD does with a distribution before plotting it (/root/reference/src/main.cpp:232-250)
extern "C" void vmd_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights,
                                         int num_src_bins) {
    const int factor = std::max(1, num_src_bins / std::max(1, num_dst_bins));
    for (int d = 0; d < num_dst_bins; ++d) {
        double bin = 0.0, weight = 0.0;
        for (int i = 0; i < factor; ++i) {
            const int s = d * factor + i;
            if (s >= num_src_bins) break;
            bin += src_bins[s];
            weight += src_weights ? src_weights[s] : 1.0;
        }
        dst_bins[d] = (float)(bin / weight);
    }
}

// temporal -> distribution as VIAMD builds it from the frame mask (/root/reference/src/main.cpp:172-230); y_range = the
// Histogram's y_min / y_max (:212-229; untouched when no frame is set, as there)
extern "C" void vmd_compute_histogram_masked_y(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                               int dim, const uint8_t* frame_mask, int num_frames, bool aggregate, float* y_range) {
    const int hdim = aggregate ? 1 : dim;
    std::fill(bins, bins + (size_t)hdim * num_bins, 0.0f);
    const float ext = range_max - range_min;
    const float inv = ext > 0.0f ? 1.0f / ext : 0.0f;
    std::vector<int> count(hdim, 0);
    bool any = false;
    for (int f = 0; f < num_frames; ++f) {
        if (!frame_mask[f]) continue;
        any = true;
        for (int i = 0; i < dim; ++i) {
            const float v = values[(size_t)f * dim + i];
            if (v < range_min || range_max < v) continue;
            const int b = std::min(std::max((int)(((v - range_min) * inv) * num_bins), 0), num_bins - 1);
            const int row = aggregate ? 0 : i;
            bins[(size_t)row * num_bins + b] += 1.0f;
            count[row] += 1;
        }
    }
    if (!any || dim <= 0) return;
    float lo = FLT_MAX, hi = -FLT_MAX;
    const float width = ext / num_bins;
    for (int r = 0; r < hdim; ++r) {
        const float scl = 1.0f / (width * count[r]);
        for (int j = 0; j < num_bins; ++j) {
            float& v = bins[(size_t)r * num_bins + j];
            v *= scl;
            lo = lo < v ? lo : v;          // MIN(min_bin, val) / MAX(max_bin, val) as the reference's macros order them: a NaN bin
            hi = hi > v ? hi : v;          // (a row without samples: 0 * inf) REPLACES the running value
        }
    }
    if (y_range) { y_range[0] = lo; y_range[1] = hi; }
}
extern "C" void vmd_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                             int dim, const uint8_t* frame_mask, int num_frames, bool aggregate) {
    vmd_compute_histogram_masked_y(bins, num_bins, range_min, range_max, values, dim, frame_mask, num_frames, aggregate, nullptr);
}

// the unmasked form (/root/reference/src/main.cpp:139-170): normalised by 1 / (bin width x samples inside the range)
extern "C" void vmd_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values, int num_values,
                                      float* bin_val_min, float* bin_val_max) {
    std::fill(bins, bins + num_bins, 0.0f);
    const float ext = range_max - range_min;
    const float inv = 1.0f / ext;
    int count = 0;
    for (int i = 0; i < num_values; ++i) {
        if (values[i] < range_min || range_max < values[i]) continue;
        const int b = std::min(std::max((int)(((values[i] - range_min) * inv) * num_bins), 0), num_bins - 1);
        bins[b] += 1.0f;
        count += 1;
    }
    if (count == 0) {
        if (bin_val_min) *bin_val_min = 0;
        if (bin_val_max) *bin_val_max = 0;
        return;
    }
    float lo = FLT_MAX, hi = -FLT_MAX;
    const float width = ext / num_bins;
    const float scl = 1.0f / (width * count);
    for (int i = 0; i < num_bins; ++i) {
        bins[i] *= scl;
        lo = lo < bins[i] ? lo : bins[i];
        hi = hi > bins[i] ? hi : bins[i];
    }
    if (bin_val_min) *bin_val_min = lo;
    if (bin_val_max) *bin_val_max = hi;
}

// bins[i] /= weights[i] where the weight is not zero (/root/reference/src/main.cpp:252-261)
extern "C" void vmd_scale_histogram(float* bins, const float* weights, int num_bins) {
    for (int i = 0; i < num_bins; ++i) if (weights[i]) bins[i] /= weights[i];
}
