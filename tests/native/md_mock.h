// tests/native/md_mock.h - TEST ONLY.  The slice of mdlib's declarations that VIAMD's evaluation call sites touch, re-declared from
// their USE in /root/reference/src (mdlib itself is an empty submodule there): enough to compile include/vmd_md_script_shim.h, VIAMD's
// own call sites cut verbatim out of the reference (tests/native/ref_callsites.cpp) and the older re-typed sequences.
//
// Every declaration carries one of two tags:
//   [file:line]      the name / field / call shape is OBSERVABLE at that line of /root/reference/src - a real mdlib must match it
//   [RECOLLECTION]   nothing in the reference shows it: it is the mock's own choice (storage layouts, helper members, what a function
//                    does inside).  include/vmd_md_script_shim.h never touches a [RECOLLECTION] field directly - it goes through the
//                    VMD_SHIM_* hooks or through compile-time "if the field exists" helpers; `grep RECOLLECTION` lists them all.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct str_t { const char* ptr; size_t len; };                               // [src/main.cpp:5682 filename.len / .ptr; :1295 STR_ARG]
#define STR_LIT(s) (str_t{(s), sizeof(s) - 1})                                 // [src/main.cpp:996 STR_LIT("Eval Full")]
static inline bool str_empty(str_t s) { return s.len == 0; }                   // [src/main.cpp:1294]

#include <stdio.h>
#include <stdlib.h>
struct md_allocator_i { void* inst; };                                          // name [src/main.cpp:75-76]; member [RECOLLECTION]
// md_unit_t: the mock keeps one code per unit (0 = none, 1 = Angstrom); mdlib's has base dimensions and a multiplier.
// name [src/viamd.h:343]; members [RECOLLECTION] - the shim only calls VMD_SHIM_UNIT / md_unit_* functions
struct md_unit_t { uint32_t base; float mult; };
static inline md_unit_t md_unit_none(void) { return md_unit_t{0u, 1.0f}; }                         // [src/main.cpp:1324; src/viamd.h:343]
static inline md_unit_t md_unit_angstrom(void) { return md_unit_t{1u, 1e-10f}; }                   // [RECOLLECTION] (the default of the VMD_SHIM_UNIT hook)
static inline bool md_unit_is_none(md_unit_t u) { return u.base == 0u; }                           // [src/main.cpp:5965]
static inline bool md_unit_equal(md_unit_t a, md_unit_t b) { return a.base == b.base && a.mult == b.mult; }      // [src/main.cpp:4401]
// [src/main.cpp:1314-1315 md_unit_print(item.unit_str[0], sizeof(...), item.unit[0]); :5972 returns the length]
static inline size_t md_unit_print(char* buf, size_t cap, md_unit_t u) { return (size_t)snprintf(buf, cap, "%s", u.base == 1u ? "\xC3\x85" : ""); }

// md_array: mdlib's stretchy buffer; the mock's grows with realloc and ignores the allocator.  The MACROS md_array(T), md_array_size(a),
// md_array_resize(a, n, alloc), md_array_free(a, alloc) are [src/main.cpp:135, 182, 1498; src/viamd.h:308]; the header in front of the
// elements is [RECOLLECTION] - the shim only uses the macros
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

struct vec3_t { float x, y, z; };          // [src/main.cpp:5801 coord.x / .y / .z]
struct vec4_t { float x, y, z, w; };       // [src/viamd.h:1376 vec4_t text_color = {1,1,1,1}]
struct mat4_t { vec4_t col[4]; };          // name [src/main.cpp:5774]; 16 floats, column-major as VIAMD multiplies it [mat4_mul_vec3(M, coord, 1.0f), src/main.cpp:5800]; member name [RECOLLECTION]
// [src/main.cpp:5800]
static inline vec3_t mat4_mul_vec3(mat4_t M, vec3_t v, float w) {
    return vec3_t{M.col[0].x * v.x + M.col[1].x * v.y + M.col[2].x * v.z + M.col[3].x * w,
                  M.col[0].y * v.x + M.col[1].y * v.y + M.col[2].y * v.z + M.col[3].y * w,
                  M.col[0].z * v.x + M.col[1].z * v.y + M.col[2].z * v.z + M.col[3].z * w};
}
/* [src/main.cpp:5793] */ static inline mat4_t mat4_scale(float x, float y, float z) { mat4_t M = {}; M.col[0].x = x; M.col[1].y = y; M.col[2].z = z; M.col[3].w = 1.0f; return M; }
/* [src/main.cpp:5793] */ static inline mat4_t mat4_mul(mat4_t A, mat4_t B) {
    mat4_t C;
    const float* a = &A.col[0].x; const float* b = &B.col[0].x; float* c = &C.col[0].x;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) { float s = 0.0f; for (int k = 0; k < 4; ++k) s += a[k * 4 + i] * b[j * 4 + k]; c[j * 4 + i] = s; }
    return C;
}

struct md_unitcell_t {
    float x, y, z, xy, xz, yz;             // [src/viamd.cpp:1837-1842; components/dataset/dataset.cpp:448-456]; float vs double [RECOLLECTION] (the shim casts)
    uint32_t flags;                        // [src/main.cpp:6255, 6261; dataset.cpp:442 md_unitcell_flags_t]
};
static inline uint32_t md_unitcell_flags(const md_unitcell_t* c) { return c->flags; }      // [src/viamd.cpp:2284]

struct md_trajectory_frame_header_t {      // name [src/viamd.cpp:465]
    size_t num_atoms;                      // [RECOLLECTION] (read by the shim only if present)
    int64_t index;                         // [RECOLLECTION] (read by the shim only if present)
    double timestamp;                      // [RECOLLECTION] (read by the shim only if present)
    md_unitcell_t unitcell;                // [src/viamd.cpp:467 data->mold.sys.unitcell = frame_header.unitcell]
};
struct md_trajectory_header_t { size_t num_frames, num_atoms; };      // name [src/viamd.cpp:444]; members [RECOLLECTION] (the reference reads header.frame_times only)
struct md_trajectory_i {                   // name [src/main.cpp:995]; all members [RECOLLECTION] - the shim only calls the three functions below
    void* inst;
    bool (*get_header)(void* inst, md_trajectory_header_t* header);
    bool (*load_frame)(void* inst, int64_t idx, md_trajectory_frame_header_t* header, float* x, float* y, float* z);
};
// [src/viamd.cpp:465-467; src/main.cpp:5741 md_trajectory_load_frame(traj, 0, NULL, x, y, z)]
static inline bool md_trajectory_load_frame(md_trajectory_i* t, int64_t idx, md_trajectory_frame_header_t* h, float* x, float* y, float* z) {
    return t->load_frame(t->inst, idx, h, x, y, z);
}
/* [src/main.cpp:1025] */ static inline size_t md_trajectory_num_frames(md_trajectory_i* t) { md_trajectory_header_t h = {}; t->get_header(t->inst, &h); return h.num_frames; }
/* [components/dataset/dataset.cpp:468] */ static inline size_t md_trajectory_num_atoms(md_trajectory_i* t) { md_trajectory_header_t h = {}; t->get_header(t->inst, &h); return h.num_atoms; }

struct md_atom_data_t {
    size_t count;                          // [src/main.cpp:5736 data.mold.sys.atom.count]
    float *x, *y, *z;                      // [src/main.cpp:5738-5741 mol.atom.x = coords + ...]
    float* mass;                           // [src/main.cpp:2412 mol.atom.mass]
};
struct md_system_t {
    md_atom_data_t atom;                   // [src/main.cpp:5734-5741]
    md_unitcell_t unitcell;                // [src/viamd.cpp:467]
    md_trajectory_i* trajectory;           // [src/main.cpp:995 state.mold.sys.trajectory]
};

struct md_bitfield_t {
    uint64_t* bits;                        // [RECOLLECTION] - the shim never touches it (VMD_SHIM_BITFIELD_* hooks only)
    uint32_t beg_bit, end_bit;             // [src/main.cpp:2511-2512, 5795-5796 bf->beg_bit / bf->end_bit]; width [RECOLLECTION]
};
struct md_bitfield_iter_t { const md_bitfield_t* bf; int64_t idx; };      // name [src/main.cpp:194]; members [RECOLLECTION]
/* [components/ramachandran/ramachandran.cpp:947] */ static inline bool md_bitfield_test_bit(const md_bitfield_t* bf, uint64_t i) {
    // (relaxed atomic word reads: the mock's evaluator sets bits of its frame mask while VIAMD's GUI thread iterates it, as mdlib's does)
    return i >= bf->beg_bit && i < bf->end_bit && ((__atomic_load_n(&bf->bits[i >> 6], __ATOMIC_RELAXED) >> (i & 63)) & 1ull);
}
/* [src/main.cpp:185] */ static inline size_t md_bitfield_popcount(const md_bitfield_t* bf) { size_t c = 0; for (uint64_t i = bf->beg_bit; i < bf->end_bit; ++i) c += md_bitfield_test_bit(bf, i); return c; }
/* [src/main.cpp:194-196: create / next / idx] */ static inline md_bitfield_iter_t md_bitfield_iter_create(const md_bitfield_t* bf) { return md_bitfield_iter_t{bf, (int64_t)bf->beg_bit - 1}; }
static inline bool md_bitfield_iter_next(md_bitfield_iter_t* it) {
    for (++it->idx; it->idx < (int64_t)it->bf->end_bit; ++it->idx) if (md_bitfield_test_bit(it->bf, (uint64_t)it->idx)) return true;
    return false;
}
static inline uint64_t md_bitfield_iter_idx(const md_bitfield_iter_t* it) { return (uint64_t)it->idx; }
// growable: the mock owns `bits` (heap), enough for the vis payload's atom sets and the frame mask
// md_bitfield_init(&bf, alloc) [src/main.cpp:433]; _free(&bf) [src/viamd.cpp:1224]; _clear(&bf) [src/main.cpp:654]; _set_bit(&bf, i) [src/viamd.cpp:2743]
static inline void md_bitfield_clear(md_bitfield_t* bf) { if (bf->bits) memset(bf->bits, 0, (((size_t)bf->end_bit + 63) / 64) * 8); bf->beg_bit = bf->end_bit = 0; }
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
/* [src/main.cpp:1930] */ static inline bool md_bitfield_empty(const md_bitfield_t* bf) { return md_bitfield_popcount(bf) == 0; }
// 1-based index of the first set bit in [beg, end), 0 when there is none [src/main.cpp:5797-5798: while ((beg_bit = md_bitfield_scan(bf, beg_bit, end_bit)) != 0) { i = beg_bit - 1; ...]
static inline size_t md_bitfield_scan(const md_bitfield_t* bf, size_t beg, size_t end) {
    for (size_t i = beg; i < end; ++i) if (md_bitfield_test_bit(bf, i)) return i + 1;
    return 0;
}

typedef uint32_t md_script_property_flags_t;           // name [src/viamd.h:348]; the three flags [src/main.cpp:1317, 1448, 1481]; values [RECOLLECTION]
enum { MD_SCRIPT_PROPERTY_FLAG_TEMPORAL = 1, MD_SCRIPT_PROPERTY_FLAG_DISTRIBUTION = 2, MD_SCRIPT_PROPERTY_FLAG_VOLUME = 4 };

struct vec2_t { float x, y; };                         // [src/main.cpp:1436 y_ext[sample_idx].x / .y]
struct md_script_aggregate_t {             // reached as prop_data->aggregate-> [src/main.cpp:1383-1440]; the type's own name [RECOLLECTION]
    size_t num_values;                     // [RECOLLECTION] (written by the shim only if present)
    float* population_mean;                // [src/main.cpp:1388]
    float* population_var;                 // [src/main.cpp:1406]
    vec2_t* population_ext;                // [src/main.cpp:1433]
};
struct md_script_property_data_t {         // name [src/main.cpp:1286]
    int32_t dim[4];                        // [src/main.cpp:1353 dim[1]; :1524 dim[2]; :5773 dim[1..3]]; element type [RECOLLECTION]
    md_unit_t unit[2];                     // [src/main.cpp:1300-1301]
    float* values;                         // [src/main.cpp:1513, 1524, 5817]
    float* weights;                        // [src/main.cpp:1524]
    size_t num_values;                     // [RECOLLECTION] (written by the shim only if present)
    md_script_aggregate_t* aggregate;      // [src/main.cpp:1378]
    float min_value, max_value;            // max_value [components/density_volume/density_volume.cpp:281]; min_value [RECOLLECTION] (written only if present)
    float min_range[2], max_range[2];      // [src/main.cpp:1513, 1519-1522]
    uint64_t fingerprint;                  // [src/main.cpp:1508-1509]
};

struct md_script_ir_t;      // opaque: the script compiler's product [src/viamd.h:1385]
struct md_script_vis_payload_o;   // opaque: what md_script_ir_property_vis_payload hands out (src/main.cpp:1304)
typedef uint32_t md_script_vis_flags_t;    // flag names [src/main.cpp:72, 5757]; type name and values [RECOLLECTION]
enum { MD_SCRIPT_VISUALIZE_DEFAULT = 0, MD_SCRIPT_VISUALIZE_GEOMETRY = 1, MD_SCRIPT_VISUALIZE_ATOMS = 2, MD_SCRIPT_VISUALIZE_SDF = 4 };
struct md_script_vis_ctx_t {               // [src/main.cpp:5751-5755: designated initialisers .ir, .mol, .traj in this order]
    const md_script_ir_t* ir;
    const md_system_t* mol;
    md_trajectory_i* traj;
};
struct md_script_vis_t {
    md_allocator_i* alloc;                 // [RECOLLECTION] (the default of the VMD_SHIM_VIS_ALLOC hook); `md_script_vis_t vis = {0}` [src/main.cpp:5746] needs a scalar first member
    md_bitfield_t atom_mask;               // [src/viamd.cpp:3205-3207]
    md_array(md_bitfield_t) structure;     // [src/viamd.cpp:1428, 1436]
    struct {
        md_array(mat4_t) matrices;         // [src/main.cpp:5774]
        md_array(md_bitfield_t) structures;// [src/main.cpp:5770, 5775]
        float extent;                      // [src/main.cpp:5778]
    } sdf;
};
// md_script_vis_init(&vis, alloc) [src/main.cpp:5747]; md_script_vis_free(&vis) [:5748]
static inline void md_script_vis_init(md_script_vis_t* vis, md_allocator_i* alloc) { memset(vis, 0, sizeof(*vis)); vis->alloc = alloc; md_bitfield_init(&vis->atom_mask, alloc); }
static inline void md_script_vis_free(md_script_vis_t* vis) {
    for (size_t i = 0; i < md_array_size(vis->sdf.structures); ++i) md_bitfield_free(&vis->sdf.structures[i]);
    md_array_free(vis->sdf.structures, vis->alloc); md_array_free(vis->sdf.matrices, vis->alloc);
    md_bitfield_free(&vis->atom_mask);
}
struct md_script_eval_t;    // defined by whoever implements the evaluator (mdlib, or include/vmd_md_script_shim.h)
