// viamd_amd/csrc/vmd_eval_runtime.cpp - what every part of the evaluator stands on: the thread-local error string and the log hook
// (md_log_register analogue, /root/reference/src/main.cpp:384-420), the tuning knobs behind vmd_set_option, hipEvent profiling, and the
// process-wide resource pool - device blocks, pinned blocks, streams and events an eval gives up are handed to the next one (VIAMD
// creates a fresh eval per script edit, src/main.cpp:960-972).  Shared declarations: vmd_eval_internal.h.
#include "vmd_eval_internal.h"

// ------------------------------------------------------------------------------------------------ errors / options
thread_local std::string g_last_error;

// md_log_register analogue (VIAMD installs a logger that turns messages into toasts, src/main.cpp:384-420): failures go to
// the registered callback, or to stderr when there is none.  The callback may be invoked from any thread that calls the API.
std::mutex g_log_mtx;

vmd_log_fn g_log_fn = nullptr;

void* g_log_user = nullptr;

extern "C" void vmd_log_register(vmd_log_fn fn, void* user) {
    std::lock_guard<std::mutex> l(g_log_mtx);
    g_log_fn = fn;
    g_log_user = user;
}

void vmd_log(int level, const char* msg) {
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

bool vmd_fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    vmd_log(VMD_LOG_ERROR, buf);
    return false;
}

Options g_opt;

// One process per GPU and 8 GPUs per node share the host: an eighth of the hardware threads, at least 8 (memcpy-bound DCD
// frames saturate there), at most 32 (compressed XTC frames scale further).
size_t load_threads() {
    const int v = g_opt.load_threads.load();
    if (v > 0) return (size_t)v;
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min<size_t>(32, std::max<size_t>(8, hw / 8));
}

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
    else if (!strcmp(key, "cells_sel_pattern")) o = &g_opt.cells_sel_pattern;
    else if (!strcmp(key, "cells_cap_sample")) o = &g_opt.cells_cap_sample;
    else if (!strcmp(key, "cells_cap_floor")) o = &g_opt.cells_cap_floor;
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
    else if (!strcmp(key, "rdf_pop")) return vmd_hip_set_rdf_pop(value);
    else if (!strcmp(key, "rdf_nsub")) return vmd_hip_set_rdf_nsub(value);
    else if (!strcmp(key, "rdf_nsub_pct")) return vmd_hip_set_rdf_nsub_pct(value);
    if (!o) return -1;
    return o->exchange(value);
}

extern "C" const char* vmd_last_error(void) { return g_last_error.c_str(); }

// Where the evaluator is (process-wide, last writer wins): a static string set at every stage of a batch.  Costs one relaxed store; a
// crash handler (tests/native/stress_eval.cpp installs one for SIGABRT / SIGSEGV) can print it when the process dies inside the HIP
// runtime without a message - round 2 saw one such abort and could not say where.
std::atomic<const char*> g_stage{"idle"};
extern "C" const char* vmd_last_stage(void) { return g_stage.load(std::memory_order_relaxed); }

// k_rdf_pencil can deal a chunk's neighbour pencils to separate work items in small launches (vmd_hip_set_rdf_nsplit): measured
// r03ar - a one-frame launch gains (170 -> 133 us), a four-frame launch loses (194 -> 252 us) - so it is switched off when the library
// loads (here, not in vmd_kernels.hip: that file is byte for byte what the committed PMC passes were collected on)
static const int g_rdf_nsplit_at_load = vmd_hip_set_rdf_nsplit(0);

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

std::mutex g_prof_mtx;

std::map<std::string, ProfEntry> g_prof;

std::atomic<bool> g_prof_on{false};

extern "C" void vmd_profile_enable(bool on) { g_prof_on = on; }

extern "C" void vmd_profile_reset(void) { std::lock_guard<std::mutex> l(g_prof_mtx); g_prof.clear(); }

extern "C" double vmd_profile_ms(const char* which, uint64_t* launches) {
    std::lock_guard<std::mutex> l(g_prof_mtx);
    auto it = g_prof.find(which);
    if (it == g_prof.end()) { if (launches) *launches = 0; return 0.0; }
    if (launches) *launches = it->second.launches;
    return it->second.ms;
}

ResourcePool& pool() { static ResourcePool* p = new ResourcePool(); return *p; }      // never destroyed: the HIP runtime may be gone first

thread_local int t_pool_idle = 0;

int pool_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { (void)hipGetLastError(); d = 0; } return d; }

hipError_t pool_raw_alloc(int kind, void** p, size_t bytes) {
    return kind == kPinned ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
}

void pool_raw_free(int kind, void* p) { if (kind == kPinned) (void)hipHostFree(p); else (void)hipFree(p); }

extern "C" void vmd_pool_trim(void) {
    ResourcePool& P = pool();
    std::vector<std::pair<int, void*>> victims;
    { std::lock_guard<std::mutex> l(P.mtx);
      for (int k = 0; k <= kPinned; ++k) { for (auto& b : P.blocks[k]) { victims.push_back({k, b.second}); P.owner.erase(b.second);
              } P.blocks[k].clear(); }
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
hipError_t pool_take(int kind, void** out, size_t bytes) {
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

void pool_give(void* p) {
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

hipStream_t pool_stream(bool high_priority) {
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
void pool_stream_give(hipStream_t s, bool high_priority) {
    if (!s) return;
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.streams[d][high_priority ? 1 : 0];
      if (g_opt.pool_mb.load() > 0 && v.size() < 64) { v.push_back(s); return; } }
    (void)hipStreamDestroy(s);
}

hipEvent_t pool_event(bool timing) {
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.events[d][timing ? 0 : 1];
      if (!v.empty() && g_opt.pool_mb.load() > 0) { hipEvent_t e = v.back(); v.pop_back(); return e; } }
    hipEvent_t e = nullptr;
    if ((timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) return nullptr;
    return e;
}

void pool_event_give(hipEvent_t e, bool timing) {
    if (!e) return;
    ResourcePool& P = pool();
    const int d = pool_device();
    { std::lock_guard<std::mutex> l(P.mtx);
      auto& v = P.events[d][timing ? 0 : 1];
      if (g_opt.pool_mb.load() > 0 && v.size() < 512) { v.push_back(e); return; } }
    (void)hipEventDestroy(e);
}
