// tests/native/viamd_host_double.h - TEST ONLY.  What VIAMD's own evaluation call sites need from the REST of VIAMD / mdlib, so that
// the slices oracle/make_ref.py cuts out of /root/reference/src (oracle/_ref/*.inc, verbatim, git-ignored) compile and run as a host
// of include/vmd_md_script_shim.h (tests/native/ref_callsites.cpp).  Nothing here computes a result that is compared: it is files ->
// stdio, logging -> stderr, allocators -> malloc, the handful of ApplicationState fields the slices touch, and a thread pool behind
// the task_system declarations of the reference.  Include after md_mock.h, md_mock_eval.h and the shim; ImGui / ImPlot TYPES come
// from the reference's vendored headers where they lie (<imgui.h>, <implot.h>).
//
//   macros            ASSERT MEMCPY MEMSET MIN MAX CLAMP ARRAY_SIZE ALIGN_TO STATIC_ASSERT STR_FMT STR_ARG defer   (mdlib core/md_common.h, md_str.h)
//   md_file_*         src/main.cpp:5646-5682 (open / printf / close with a by-value handle)
//   VIAMD_LOG_*       src/main.cpp:5648, 5682; MD_LOG_DEBUG :1289
//   md_alloc, md_array_push / _create / _bytes, md_temp_*, md_vm_arena_push_zero_array      :178-191, 1339, 5733, 5835
//   md_atom_coord / md_atom_atomic_number     :5788-5791
//   md_time_now / md_time_as_seconds          :1000-1004
//   ApplicationState  src/viamd.h:1026-1401, ONLY the fields the slices read or write, each with its line
//   task_system       src/task_system.h (declarations sliced); the double follows src/task_system.cpp:73-81 (a range task is cut into
//                     partitions that pool threads pull; ExecuteRange hands [start * grain, min(size, end * grain)) to the lambda)
#pragma once
#include <float.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <bitset>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// ---- macros of mdlib's core headers, as the slices use them
#define ASSERT(x) ((void)0)
#define STATIC_ASSERT(cond, msg) static_assert(cond, msg)
#define MEMCPY memcpy
#define MEMSET memset
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define CLAMP(v, lo, hi) MIN(MAX((v), (lo)), (hi))
#define ARRAY_SIZE(a) (sizeof(a) / sizeof((a)[0]))
#define ALIGN_TO(x, a) (((x) + ((a) - 1)) / (a) * (a))
#define STR_FMT "%.*s"
#define STR_ARG(s) (int)(s).len, (s).ptr
enum { MD_SCRIPT_PROPERTY_FLAG_NONE = 0 };                       // src/viamd.h:348 (the default of DisplayProperty::prop_flags)

template <typename F>
struct host_defer_t { F f; ~host_defer_t() { f(); } };
struct host_defer_tag {};
template <typename F>
static host_defer_t<F> operator+(host_defer_tag, F f) { return host_defer_t<F>{f}; }
#define HOST_CAT2(a, b) a##b
#define HOST_CAT(a, b) HOST_CAT2(a, b)
#define defer auto HOST_CAT(host_defer_, __LINE__) = host_defer_tag{} + [&]()

// ---- logging: stderr, one counter per level so that a test can see an error was reported
static std::atomic<long> host_log_errors{0};
static inline void host_log(const char* level, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[viamd %s] ", level);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
#define VIAMD_LOG_ERROR(...) (host_log_errors += 1, host_log("error", __VA_ARGS__))
#define VIAMD_LOG_SUCCESS(...) host_log("success", __VA_ARGS__)
#define VIAMD_LOG_INFO(...) host_log("info", __VA_ARGS__)
#define MD_LOG_DEBUG(...) host_log("debug", __VA_ARGS__)

// ---- md_file_*: stdio
enum { MD_FILE_READ = 1, MD_FILE_WRITE = 2, MD_FILE_APPEND = 4, MD_FILE_CREATE = 8, MD_FILE_TRUNCATE = 16 };
struct md_file_t { FILE* f; };
static inline bool md_file_open(md_file_t* file, str_t path, int flags) {
    const std::string p(path.ptr, path.len);
    file->f = fopen(p.c_str(), (flags & MD_FILE_WRITE) ? "w" : "r");
    return file->f != nullptr;
}
static inline size_t md_file_printf(md_file_t file, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    const int n = vfprintf(file.f, fmt, ap);
    va_end(ap);
    return n > 0 ? (size_t)n : 0;
}
static inline void md_file_close(md_file_t* file) { if (file->f) fclose(file->f); file->f = nullptr; }

// ---- allocators: the frame arena lives until host_frame_reset() (VIAMD resets it once per GUI frame), the persistent one is the heap
struct host_arena_t { std::mutex mtx; std::vector<void*> blocks; };
static host_arena_t host_frame_arena;
static md_allocator_i host_frame_alloc_obj{&host_frame_arena}, host_persistent_alloc_obj{nullptr};
static md_allocator_i* frame_alloc = &host_frame_alloc_obj;           // src/main.cpp:75-76
static md_allocator_i* persistent_alloc = &host_persistent_alloc_obj;
static inline void* md_alloc(md_allocator_i* alloc, size_t bytes) {
    void* p = calloc(bytes ? bytes : 1, 1);
    if (alloc && alloc->inst) { host_arena_t* a = (host_arena_t*)alloc->inst; std::lock_guard<std::mutex> l(a->mtx); a->blocks.push_back(p); }
    return p;
}
static inline void host_frame_reset() {
    std::lock_guard<std::mutex> l(host_frame_arena.mtx);
    for (void* p : host_frame_arena.blocks) free(p);
    host_frame_arena.blocks.clear();
}
struct md_temp_scope_t { int unused; };
static inline md_temp_scope_t md_temp_begin_in(md_allocator_i*) { return md_temp_scope_t{0}; }
static inline void md_temp_end(md_temp_scope_t) {}
#define md_vm_arena_push_zero_array(arena, type, n) ((type*)md_alloc((arena), sizeof(type) * (size_t)(n)))
#define md_array_bytes(a) (md_array_size(a) * sizeof(*(a)))
#define md_array_push(a, item, alloc) (md_array_resize((a), md_array_size(a) + 1, (alloc)), (a)[md_array_size(a) - 1] = (item))
#define md_array_create(T, n, alloc) ((void)(alloc), (T*)md_mock_array_resize(nullptr, (size_t)(n), sizeof(T)))

// ---- atoms: coordinates from md_atom_data_t, atomic numbers from a side table of the test (the mock molecule carries none)
static const uint8_t* host_atomic_numbers = nullptr;
static inline vec3_t md_atom_coord(const md_atom_data_t* atom, size_t i) { return vec3_t{atom->x[i], atom->y[i], atom->z[i]}; }
static inline int md_atom_atomic_number(const md_atom_data_t*, size_t i) { return host_atomic_numbers ? (int)host_atomic_numbers[i] : 0; }

// ---- time
static inline uint64_t md_time_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline double md_time_as_seconds(uint64_t t) { return (double)t * 1e-9; }

#ifndef VIAMD_HOST_DOUBLE_EXPORT_ONLY        // (tests/native/shim_callsites.cpp includes the exporters' slice alone: no ImGui, no task system)
// ---- the one ImGui FUNCTION the slices call (src/main.cpp:1299: the colour of a plot line); imgui.cpp is not linked
ImVec4 ImGui::ColorConvertU32ToFloat4(ImU32 in) {
    const float s = 1.0f / 255.0f;
    return ImVec4((float)((in >> 0) & 0xFF) * s, (float)((in >> 8) & 0xFF) * s, (float)((in >> 16) & 0xFF) * s, (float)((in >> 24) & 0xFF) * s);
}

// ---- task_system: the reference's declarations ...
#include "_ref/task_system_h_slices.inc"

// ... and a double behind the five the slices call.  A range task of `range_size` items with grain g becomes ceil(range_size / g)
// set items (task_system.cpp:55-56), handed out in partitions of max(1, set / (T * (T - 1))) items as enkiTS does for T threads; every
// partition is one call of the lambda with [start * g, min(range_size, end * g)) (task_system.cpp:73-81).  A task with a dependency
// starts when the dependency completes (the "##Time Eval Full" task of src/main.cpp:1001-1007 is never enqueued by hand).
namespace task_system {
struct HostTask {
    std::string label;
    RangeTask range_func;
    Task func;
    uint32_t range_size = 0, grain = 1;
    std::atomic<bool> running{false};
    std::atomic<uint32_t> next{0};
    ID dependent = INVALID_ID;                 // started when this one completes
    std::thread driver;
    std::atomic<long> calls{0};
};
struct HostPool {
    std::mutex mtx;
    std::vector<std::unique_ptr<HostTask>> tasks;      // ID = index + 1; never reused within a test
    size_t num_threads = 4;
};
static HostPool& host_pool() { static HostPool p; return p; }
static HostTask* host_task(ID id) {
    HostPool& p = host_pool();
    std::lock_guard<std::mutex> l(p.mtx);
    return id != INVALID_ID && id <= p.tasks.size() ? p.tasks[id - 1].get() : nullptr;
}
static ID host_add(std::unique_ptr<HostTask> t) {
    HostPool& p = host_pool();
    std::lock_guard<std::mutex> l(p.mtx);
    p.tasks.push_back(std::move(t));
    return (ID)p.tasks.size();
}
void initialize(size_t num_threads) { host_pool().num_threads = num_threads < 2 ? 2 : num_threads; }     // src/main.cpp:494-495 clamps to >= 2
size_t pool_num_threads() { return host_pool().num_threads; }
ID create_pool_task(str_t label, Task task) {
    std::unique_ptr<HostTask> t(new HostTask());
    t->label.assign(label.ptr, label.len); t->func = task;
    return host_add(std::move(t));
}
ID create_pool_task(str_t label, uint32_t range_size, RangeTask task, uint32_t grain_size) {
    std::unique_ptr<HostTask> t(new HostTask());
    t->label.assign(label.ptr, label.len); t->range_func = task; t->range_size = range_size; t->grain = grain_size ? grain_size : 1;
    return host_add(std::move(t));
}
void set_task_dependency(ID task, ID dependency) { if (HostTask* d = host_task(dependency)) d->dependent = task; }
void enqueue_task(ID id) {
    HostTask* t = host_task(id);
    if (!t) return;
    if (t->driver.joinable()) t->driver.join();
    t->running = true;
    t->driver = std::thread([t] {
        if (t->range_func) {
            const size_t T = host_pool().num_threads;
            const uint32_t set_size = (t->range_size + t->grain - 1) / t->grain;
            const uint32_t part = (uint32_t)MAX((size_t)1, (size_t)set_size / (T * (T - 1)));
            t->next = 0;
            std::vector<std::thread> workers;
            for (size_t w = 0; w < T; ++w)
                workers.emplace_back([t, set_size, part, w] {
                    for (;;) {
                        const uint32_t start = t->next.fetch_add(part);
                        if (start >= set_size) break;
                        const uint32_t end = MIN(set_size, start + part);
                        t->calls += 1;
                        t->range_func(start * t->grain, MIN(t->range_size, end * t->grain), (uint32_t)w);
                    }
                });
            for (auto& w : workers) w.join();
        } else if (t->func) {
            t->func();
        }
        const ID dep = t->dependent;
        t->running = false;
        if (dep != INVALID_ID) enqueue_task(dep);
    });
}
bool task_is_running(ID id) { HostTask* t = host_task(id); return t && t->running.load(); }
void task_wait_for(ID id) {
    HostTask* t = host_task(id);
    if (!t) return;
    while (t->running.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
}
void shutdown() {
    HostPool& p = host_pool();
    for (size_t i = 0; i < p.tasks.size(); ++i) {       // a dependent may be enqueued by a finishing task: index loop, joined in order
        while (p.tasks[i]->running.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (p.tasks[i]->driver.joinable()) p.tasks[i]->driver.join();
    }
}
}  // namespace task_system

// ---- DisplayProperty: the reference's own struct (src/viamd.h:272-370), verbatim
#include "_ref/viamd_h_slices.inc"
#else
struct DisplayProperty;
namespace task_system { typedef uint64_t ID; constexpr ID INVALID_ID = 0; }
#endif

// ---- ApplicationState: only what the slices touch
struct ApplicationState {
    struct {
        md_allocator_i* frame = nullptr;                 // src/viamd.h:1035
        md_allocator_i* persistent = nullptr;            // :1036
    } allocator;
    struct {
        md_system_t sys = {};                            // :1093
    } mold;
    DisplayProperty* display_properties = nullptr;       // :1110
    struct {
        task_system::ID evaluate_full = task_system::INVALID_ID;      // :1117
        task_system::ID evaluate_filt = task_system::INVALID_ID;      // :1118
    } tasks;
    struct {
        struct {
            bool enabled = false;                        // :1194
            double beg_frame = 0;                        // :1195
            double end_frame = 1;                        // :1196
        } filter;
        md_array(float) x_values = 0;                    // :1212
    } timeline;
    struct {
        md_script_ir_t* ir = nullptr;                    // :1385
        md_script_ir_t* eval_ir = nullptr;               // :1386
        md_script_eval_t* full_eval = nullptr;           // :1388
        md_script_eval_t* filt_eval = nullptr;           // :1389
        bool eval_init = false;                          // :1396
        bool evaluate_full = false;                      // :1397
        bool evaluate_filt = false;                      // :1398
    } script;
};
