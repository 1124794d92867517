// tests/native/md_mock.h - TEST ONLY.  The slice of mdlib's declarations that VIAMD's evaluation call sites touch, re-declared from
// their USE in /root/reference/src (mdlib itself is an empty submodule there): enough to compile include/vmd_md_script_shim.h and
// a re-typed copy of VIAMD's call sequence (tests/native/shim_callsites.cpp).  Field names and call shapes follow the call sites:
//   md_system_t: atom.{count,x,y,z,mass}, unitcell, trajectory           src/main.cpp:642, 995-996; src/viamd.cpp:465-467, 2253
//   md_unitcell_t: x, y, z, xy, xz, yz, flags                              src/viamd.cpp:1837-1842; src/main.cpp:6255
//   md_trajectory_load_frame(traj, idx, &header, x, y, z)                  src/viamd.cpp:465-467
//   md_script_property_data_t: dim, unit, values, weights, aggregate, ranges, fingerprint     src/main.cpp:1286-1524
//   md_bitfield_t + iterator                                               src/main.cpp:194-210
//   md_script_vis_t {alloc, atom_mask, sdf.{extent, matrices, structures}}, md_script_vis_ctx_t {ir, mol, traj}, flags
//                                                                          src/components/density_volume/density_volume.cpp:179-204, 263-269; src/main.cpp:5746-5757
//   md_array_size / md_array_resize, mat4_t, md_bitfield_scan / _popcount  src/main.cpp:5768-5793
//   md_unit_t + md_unit_print / _none / _equal                             src/main.cpp:1300-1324, 4395-4401
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct str_t { const char* ptr; size_t len; };
#define STR_LIT(s) (str_t{(s), sizeof(s) - 1})
static inline bool str_empty(str_t s) { return s.len == 0; }

#include <stdio.h>
#include <stdlib.h>
struct md_allocator_i { void* inst; };
// md_unit_t: the mock keeps one code per unit (0 = none, 1 = Angstrom); mdlib's has base dimensions and a multiplier
struct md_unit_t { uint32_t base; float mult; };
static inline md_unit_t md_unit_none(void) { return md_unit_t{0u, 1.0f}; }
static inline md_unit_t md_unit_angstrom(void) { return md_unit_t{1u, 1e-10f}; }
static inline bool md_unit_is_none(md_unit_t u) { return u.base == 0u; }
static inline bool md_unit_equal(md_unit_t a, md_unit_t b) { return a.base == b.base && a.mult == b.mult; }
static inline size_t md_unit_print(char* buf, size_t cap, md_unit_t u) { return (size_t)snprintf(buf, cap, "%s", u.base == 1u ? "\xC3\x85" : ""); }

// md_array: mdlib's stretchy buffer (a header in front of the elements); the mock's grows with realloc and ignores the allocator
struct md_mock_array_header_t { size_t size, capacity; };
#define md_array(T) T*
static inline size_t md_mock_array_size(const void* a) { return a ? ((const md_mock_array_header_t*)a)[-1].size : 0; }
static inline void* md_mock_array_resize(void* a, size_t n, size_t elem) {
    md_mock_array_header_t* h = a ? (md_mock_array_header_t*)a - 1 : nullptr;
    const size_t old = h ? h->size : 0;
    if (!h || h->capacity < n) {
        h = (md_mock_array_header_t*)realloc(h, sizeof(md_mock_array_header_t) + (n ? n : 1) * elem);
        h->capacity = n ? n : 1;
    }
    h->size = n;
    if (n > old) memset((char*)(h + 1) + old * elem, 0, (n - old) * elem);
    return h + 1;
}
#define md_array_size(a) md_mock_array_size(a)
#define md_array_resize(a, n, alloc) ((void)(alloc), *(void**)&(a) = md_mock_array_resize((a), (n), sizeof(*(a))))
#define md_array_free(a, alloc) ((void)(alloc), (a) ? free((md_mock_array_header_t*)(a) - 1) : (void)0, *(void**)&(a) = nullptr)

struct vec3_t { float x, y, z; };
struct vec4_t { float x, y, z, w; };
struct mat4_t { vec4_t col[4]; };          // column-major, as VIAMD multiplies it (mat4_mul_vec3(M, coord, 1.0f), src/main.cpp:5790)
static inline vec3_t mat4_mul_vec3(mat4_t M, vec3_t v, float w) {
    return vec3_t{M.col[0].x * v.x + M.col[1].x * v.y + M.col[2].x * v.z + M.col[3].x * w,
                  M.col[0].y * v.x + M.col[1].y * v.y + M.col[2].y * v.z + M.col[3].y * w,
                  M.col[0].z * v.x + M.col[1].z * v.y + M.col[2].z * v.z + M.col[3].z * w};
}
static inline mat4_t mat4_scale(float x, float y, float z) { mat4_t M = {}; M.col[0].x = x; M.col[1].y = y; M.col[2].z = z; M.col[3].w = 1.0f; return M; }
static inline mat4_t mat4_mul(mat4_t A, mat4_t B) {
    mat4_t C;
    const float* a = &A.col[0].x; const float* b = &B.col[0].x; float* c = &C.col[0].x;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) { float s = 0.0f; for (int k = 0; k < 4; ++k) s += a[k * 4 + i] * b[j * 4 + k]; c[j * 4 + i] = s; }
    return C;
}

struct md_unitcell_t {
    float x, y, z, xy, xz, yz;
    uint32_t flags;
};
static inline uint32_t md_unitcell_flags(const md_unitcell_t* c) { return c->flags; }

struct md_trajectory_frame_header_t {
    size_t num_atoms;
    int64_t index;
    double timestamp;
    md_unitcell_t unitcell;
};
struct md_trajectory_header_t { size_t num_frames, num_atoms; };
struct md_trajectory_i {
    void* inst;
    bool (*get_header)(void* inst, md_trajectory_header_t* header);
    bool (*load_frame)(void* inst, int64_t idx, md_trajectory_frame_header_t* header, float* x, float* y, float* z);
};
static inline bool md_trajectory_load_frame(md_trajectory_i* t, int64_t idx, md_trajectory_frame_header_t* h, float* x, float* y, float* z) {
    return t->load_frame(t->inst, idx, h, x, y, z);
}
static inline size_t md_trajectory_num_frames(md_trajectory_i* t) { md_trajectory_header_t h = {}; t->get_header(t->inst, &h); return h.num_frames; }
static inline size_t md_trajectory_num_atoms(md_trajectory_i* t) { md_trajectory_header_t h = {}; t->get_header(t->inst, &h); return h.num_atoms; }

struct md_atom_data_t {
    size_t count;
    float *x, *y, *z;
    float* mass;
};
struct md_system_t {
    md_atom_data_t atom;
    md_unitcell_t unitcell;
    md_trajectory_i* trajectory;
};

struct md_bitfield_t {
    uint64_t* bits;
    uint32_t beg_bit, end_bit;
};
struct md_bitfield_iter_t { const md_bitfield_t* bf; int64_t idx; };
static inline bool md_bitfield_test_bit(const md_bitfield_t* bf, uint64_t i) { return i >= bf->beg_bit && i < bf->end_bit && ((bf->bits[i >> 6] >> (i & 63)) & 1ull); }
static inline size_t md_bitfield_popcount(const md_bitfield_t* bf) { size_t c = 0; for (uint64_t i = bf->beg_bit; i < bf->end_bit; ++i) c += md_bitfield_test_bit(bf, i); return c; }
static inline md_bitfield_iter_t md_bitfield_iter_create(const md_bitfield_t* bf) { return md_bitfield_iter_t{bf, (int64_t)bf->beg_bit - 1}; }
static inline bool md_bitfield_iter_next(md_bitfield_iter_t* it) {
    for (++it->idx; it->idx < (int64_t)it->bf->end_bit; ++it->idx) if (md_bitfield_test_bit(it->bf, (uint64_t)it->idx)) return true;
    return false;
}
static inline uint64_t md_bitfield_iter_idx(const md_bitfield_iter_t* it) { return (uint64_t)it->idx; }
// growable: the mock owns `bits` (heap), enough for the vis payload's atom sets
static inline void md_bitfield_init(md_bitfield_t* bf, md_allocator_i* alloc) { (void)alloc; bf->bits = nullptr; bf->beg_bit = 0; bf->end_bit = 0; }
static inline void md_bitfield_free(md_bitfield_t* bf) { free(bf->bits); bf->bits = nullptr; bf->beg_bit = bf->end_bit = 0; }
static inline void md_bitfield_set_bit(md_bitfield_t* bf, uint64_t i) {
    if (i >= bf->end_bit) {
        const size_t old_words = ((size_t)bf->end_bit + 63) / 64, new_words = (size_t)(i + 64) / 64;
        if (new_words > old_words) { bf->bits = (uint64_t*)realloc(bf->bits, new_words * 8); memset(bf->bits + old_words, 0, (new_words - old_words) * 8); }
        bf->end_bit = (uint32_t)(i + 1);
    }
    bf->bits[i >> 6] |= 1ull << (i & 63);
}
static inline bool md_bitfield_empty(const md_bitfield_t* bf) { return md_bitfield_popcount(bf) == 0; }
// 1-based index of the first set bit in [beg, end), 0 when there is none (src/main.cpp:5786-5788)
static inline size_t md_bitfield_scan(const md_bitfield_t* bf, size_t beg, size_t end) {
    for (size_t i = beg; i < end; ++i) if (md_bitfield_test_bit(bf, i)) return i + 1;
    return 0;
}

typedef uint32_t md_script_property_flags_t;
enum { MD_SCRIPT_PROPERTY_FLAG_TEMPORAL = 1, MD_SCRIPT_PROPERTY_FLAG_DISTRIBUTION = 2, MD_SCRIPT_PROPERTY_FLAG_VOLUME = 4 };

struct vec2_t { float x, y; };
struct md_script_aggregate_t {
    size_t num_values;
    float* population_mean;
    float* population_var;
    vec2_t* population_ext;
};
struct md_script_property_data_t {
    int32_t dim[4];
    md_unit_t unit[2];
    float* values;
    float* weights;
    size_t num_values;
    md_script_aggregate_t* aggregate;
    float min_value, max_value;
    float min_range[2], max_range[2];
    uint64_t fingerprint;
};

struct md_script_ir_t;      // opaque: the script compiler's product
struct md_script_vis_payload_o;   // opaque: what md_script_ir_property_vis_payload hands out (src/main.cpp:1304)
typedef uint32_t md_script_vis_flags_t;
enum { MD_SCRIPT_VISUALIZE_DEFAULT = 0, MD_SCRIPT_VISUALIZE_GEOMETRY = 1, MD_SCRIPT_VISUALIZE_ATOMS = 2, MD_SCRIPT_VISUALIZE_SDF = 4 };
struct md_script_vis_ctx_t {
    const md_script_ir_t* ir;
    const md_system_t* mol;
    md_trajectory_i* traj;
};
struct md_script_vis_t {
    md_allocator_i* alloc;
    md_bitfield_t atom_mask;
    md_array(md_bitfield_t) structure;
    struct {
        md_array(mat4_t) matrices;
        md_array(md_bitfield_t) structures;
        float extent;
    } sdf;
};
static inline void md_script_vis_init(md_script_vis_t* vis, md_allocator_i* alloc) { memset(vis, 0, sizeof(*vis)); vis->alloc = alloc; md_bitfield_init(&vis->atom_mask, alloc); }
static inline void md_script_vis_free(md_script_vis_t* vis) {
    for (size_t i = 0; i < md_array_size(vis->sdf.structures); ++i) md_bitfield_free(&vis->sdf.structures[i]);
    md_array_free(vis->sdf.structures, vis->alloc); md_array_free(vis->sdf.matrices, vis->alloc);
    md_bitfield_free(&vis->atom_mask);
}
struct md_script_eval_t;    // defined by whoever implements the evaluator (mdlib, or include/vmd_md_script_shim.h)
