// oracle/ref_stubs.cpp - TEST INFRASTRUCTURE.  The few lines of the reference that CAN be compiled for this path.
//
// VIAMD's consumer-side post-processing of the evaluator's results - compute_histogram, compute_histogram_masked,
// downsample_histogram, scale_histogram - is plain C++ inside /root/reference/src/main.cpp:139-261 (the arithmetic of rdf / sdf /
// distance itself lives in the empty submodule ext/mdlib and cannot be built).  `make_ref.py` cuts those line ranges out of the
// reference WHERE IT LIES into oracle/_ref/viamd_main_slices.inc (generated, git-ignored: no reference source enters this
// repository) and compiles this file around them into oracle/_ref/libviamd_ref.so.  Everything below is the minimum those functions
// need from mdlib - an array with a length header, a bitfield with an iterator, a scratch arena, `defer` - written here from their
// use in the slices, plus extern "C" entry points for the tests (tests/test_oracle.py checks vo_* and vmd_* against them).
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define ASSERT(x) ((void)0)
#define MEMSET memset
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define CLAMP(v, lo, hi) MIN(MAX((v), (lo)), (hi))

// ---- md_allocator_i / md_array stand-in: the element count sits in front of the data
struct md_allocator_i {};
static md_allocator_i g_heap;
template <typename T>
static size_t stub_array_size(T* a) { return a ? ((size_t*)a)[-2] : 0; }
template <typename T>
static T* stub_array_resize(T* a, size_t n) {
    size_t* raw = a ? (size_t*)a - 2 : nullptr;
    raw = (size_t*)realloc(raw, 2 * sizeof(size_t) + n * sizeof(T));
    raw[0] = n;
    return (T*)(raw + 2);
}
#define md_array_resize(a, n, alloc) ((a) = stub_array_resize((a), (n)))
#define md_array_bytes(a) (stub_array_size(a) * sizeof(*(a)))
#define md_array_free(a, alloc) do { if (a) free((size_t*)(a) - 2); } while (0)

// ---- md_bitfield_t stand-in: one byte per frame
struct md_bitfield_t { const uint8_t* bits; int64_t n; };
struct md_bitfield_iter_t { const md_bitfield_t* bf; int64_t idx; };
static size_t md_bitfield_popcount(const md_bitfield_t* bf) { size_t c = 0; for (int64_t i = 0; i < bf->n; ++i) c += bf->bits[i] != 0; return c; }
static md_bitfield_iter_t md_bitfield_iter_create(const md_bitfield_t* bf) { return md_bitfield_iter_t{bf, -1}; }
static bool md_bitfield_iter_next(md_bitfield_iter_t* it) {
    for (++it->idx; it->idx < it->bf->n; ++it->idx) if (it->bf->bits[it->idx]) return true;
    return false;
}
static uint64_t md_bitfield_iter_idx(const md_bitfield_iter_t* it) { return (uint64_t)it->idx; }

// ---- frame arena + temp scope + defer
struct stub_arena { std::vector<void*> blocks; };
static stub_arena g_frame_arena;
static stub_arena* frame_alloc = &g_frame_arena;
struct md_temp_scope_t { size_t mark; };
static md_temp_scope_t md_temp_begin_in(stub_arena* a) { return md_temp_scope_t{a->blocks.size()}; }
static void md_temp_end(md_temp_scope_t t) {
    while (g_frame_arena.blocks.size() > t.mark) { free(g_frame_arena.blocks.back()); g_frame_arena.blocks.pop_back(); }
}
#define md_vm_arena_push_zero_array(arena, type, n) ((type*)stub_arena_push((arena), sizeof(type) * (size_t)(n)))
static void* stub_arena_push(stub_arena* a, size_t bytes) { void* p = calloc(bytes ? bytes : 1, 1); a->blocks.push_back(p); return p; }
template <typename F>
struct stub_defer { F f; ~stub_defer() { f(); } };
struct stub_defer_tag {};
template <typename F>
static stub_defer<F> operator+(stub_defer_tag, F f) { return stub_defer<F>{f}; }
#define STUB_CAT2(a, b) a##b
#define STUB_CAT(a, b) STUB_CAT2(a, b)
#define defer auto STUB_CAT(stub_defer_, __LINE__) = stub_defer_tag{} + [&]()

// ---- the part of DisplayProperty the slices touch (src/main.cpp: struct DisplayProperty::Histogram)
struct DisplayProperty {
    struct Histogram {
        int num_bins = 0;
        float x_min = 0, x_max = 0, y_min = 0, y_max = 0;
        int dim = 0;
        float* bins = nullptr;
        md_allocator_i* alloc = nullptr;
    };
};

#include "_ref/viamd_main_slices.inc"

extern "C" {
void ref_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values, int num_values,
                           float* bin_val_min, float* bin_val_max) {
    compute_histogram(bins, num_bins, range_min, range_max, values, num_values, bin_val_min, bin_val_max);
}
// bins: [dim (or 1 when aggregate)][num_bins]; y_range: {y_min, y_max} as VIAMD stores them in the Histogram
void ref_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values, int dim,
                                  const uint8_t* frame_mask, int num_frames, int aggregate, float* y_range) {
    DisplayProperty::Histogram h;
    h.alloc = &g_heap;
    md_bitfield_t bf{frame_mask, num_frames};
    compute_histogram_masked(&h, num_bins, range_min, range_max, values, dim, &bf, aggregate != 0);
    memcpy(bins, h.bins, sizeof(float) * (size_t)h.dim * (size_t)num_bins);
    if (y_range) { y_range[0] = h.y_min; y_range[1] = h.y_max; }
    free_histogram(&h);
}
void ref_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights, int num_src_bins) {
    downsample_histogram(dst_bins, num_dst_bins, src_bins, src_weights, num_src_bins);
}
void ref_scale_histogram(float* bins, const float* weights, int num_bins) { scale_histogram(bins, weights, num_bins); }
}
