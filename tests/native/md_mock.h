// tests/native/md_mock.h - TEST ONLY.  The slice of mdlib's declarations that VIAMD's evaluation call sites touch, re-declared from
// their USE in /root/reference/src (mdlib itself is an empty submodule there): enough to compile include/vmd_md_script_shim.h and
// a re-typed copy of VIAMD's call sequence (tests/native/shim_callsites.cpp).  Field names and call shapes follow the call sites:
//   md_system_t: atom.{count,x,y,z,mass}, unitcell, trajectory           src/main.cpp:642, 995-996; src/viamd.cpp:465-467, 2253
//   md_unitcell_t: x, y, z, xy, xz, yz, flags                              src/viamd.cpp:1837-1842; src/main.cpp:6255
//   md_trajectory_load_frame(traj, idx, &header, x, y, z)                  src/viamd.cpp:465-467
//   md_script_property_data_t: dim, unit, values, weights, aggregate, ranges, fingerprint     src/main.cpp:1286-1524
//   md_bitfield_t + iterator                                               src/main.cpp:194-210
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct str_t { const char* ptr; size_t len; };
#define STR_LIT(s) (str_t{(s), sizeof(s) - 1})
static inline bool str_empty(str_t s) { return s.len == 0; }

struct md_allocator_i { void* inst; };
struct md_unit_t { uint32_t base; float mult; };

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
struct md_script_eval_t;    // defined by whoever implements the evaluator (mdlib, or include/vmd_md_script_shim.h)
