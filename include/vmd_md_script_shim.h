/* vmd_md_script_shim.h - mdlib's evaluator entry points (md_script_eval_*, as VIAMD calls them) implemented on libviamd_amd.so.
 *
 * C++ header, header-only.  Include it AFTER mdlib's own headers (md_script.h, md_molecule.h, md_trajectory.h, core/md_str.h,
 * core/md_bitfield.h, core/md_unit.h): it uses their types by name and defines the functions VIAMD calls on the evaluation path
 * with the signatures inferred from the call sites:
 *
 *     md_script_eval_create / _free / _clear_data / _interrupt / _ir_fingerprint      /root/reference/src/main.cpp:971,960,990,829,987
 *     md_script_eval_frame_range                                                      /root/reference/src/main.cpp:996,1032
 *     md_script_eval_property_data / md_script_eval_frame_mask                        /root/reference/src/main.cpp:1286,1513
 *     md_script_ir_property_vis_payload                                               /root/reference/src/main.cpp:1304
 *     md_script_vis_eval_payload (MD_SCRIPT_VISUALIZE_SDF / _ATOMS on an sdf property)
 *                                          /root/reference/src/components/density_volume/density_volume.cpp:183-204, 263-269;
 *                                          /root/reference/src/main.cpp:5751-5803 (export_cube); src/viamd.cpp:3197-3210
 *
 * With VMD_SHIM_PREFIX undefined the functions are emitted under those very names (a VIAMD build whose mdlib has its own ten entry points
 * renamed out of the way - see "A DECORATOR" below - or, with VMD_SHIM_NO_FALLBACK, dropped from the link); define VMD_SHIM_PREFIX(name) to
 * put them elsewhere.
 *
 * Hooks (define before including; the defaults name mdlib's own functions): VMD_SHIM_BONDS(sys, vsys) hands md_system_t::bond over;
 * VMD_SHIM_UNIT(dst, str) turns the backend's printed unit ("\xC3\x85" or "") into an md_unit_t (default: md_unit_angstrom() /
 * md_unit_none()); VMD_SHIM_BITFIELD_INIT / _SET / _TEST / _CLEAR / _FREE are how the shim touches ANY md_bitfield_t - the reference
 * structures of a vis payload and the frame mask it hands out (defaults: md_bitfield_init / _set_bit / _test_bit / _clear / _free); md_array_resize
 * / md_array_size are mdlib's stretchy-buffer macros.  The block "what this header requires of mdlib" below lists, and checks at compile time,
 * every name and field used directly, each with the line of the reference where it is observable.
 *
 * A DECORATOR, not a replacement (round 5).  mdlib evaluates EVERY property of the IR in one md_script_eval_frame_range
 * (/root/reference/src/main.cpp:993-997) and VIAMD asks md_script_eval_property_data for every name of md_script_ir_property_names
 * (:1277-1291) - its own default script (:528) carries `a1 = angle(...)` and `{lin,plan,iso} = shape_weights(all)` next to d1 / r / v.
 * The shim therefore keeps mdlib's evaluator BEHIND it: the rdf / sdf / distance properties bound with vmd_shim_bind_ir are evaluated
 * on the GPU, everything else is forwarded to the fallback hooks
 *
 *     VMD_SHIM_FALLBACK(md_script_eval_create | _free | _clear_data | _interrupt | _ir_fingerprint | _frame_range | _property_data |
 *                       _frame_mask)  and  VMD_SHIM_FALLBACK(md_script_ir_property_vis_payload | md_script_vis_eval_payload)
 *
 * whose default names are mdlib's originals under a rename: compile mdlib's md_script.c with
 * -Dmd_script_eval_create=mdlib_md_script_eval_create ... (INTEGRATION.md section 2 lists the ten defines), or define
 * VMD_SHIM_FALLBACK(name) yourself (dlsym(RTLD_NEXT, #name) wrappers, a test double).  Per call: create / free / clear_data / interrupt
 * go to both evaluators; frame_range runs the GPU part, then the fallback on the same range; property_data answers bound names from the
 * GPU eval and every other name from the fallback; frame_mask is the AND of the two masks (a frame counts once both have it);
 * ir_fingerprint is the fallback's - what src/main.cpp:987 compares with md_script_ir_fingerprint(ir) - perturbed while the GPU
 * binding of the ir is not the one the eval was created with (VIAMD then re-creates the evals).  The fallback evaluates the IR it is
 * given: the whole script (the bound properties are then computed twice, their CPU copies ignored), or - vmd_shim_bind_fallback_ir - an
 * IR mdlib compiled from vmd_script_report_fallback_source(), the script text without the statements the GPU took (include/vmd_eval.h).
 * VMD_SHIM_NO_FALLBACK removes all of this: unbound names are then NULL, as in rounds 1 - 4.
 *
 * What the shim needs from the host besides the types: the property DESCRIPTORS.  mdlib's IR is opaque, so the host registers,
 * once per compiled script, the vmd_script_ir_t that carries its rdf / sdf / distance properties (INTEGRATION.md section 3 shows how
 * VIAMD derives them from the evaluated argument bitfields):      vmd_shim_bind_ir(md_ir, vmd_ir);
 *
 * tests/native/ref_callsites.cpp runs VIAMD's OWN call sites - cut verbatim out of /root/reference/src by oracle/make_ref.py - against this
 * header with a test double of mdlib (tests/native/md_mock.h, md_mock_eval.h); shim_default_script.cpp and shim_callsites.cpp cover the
 * fallback / no-fallback builds with re-typed sequences. */
#ifndef VMD_MD_SCRIPT_SHIM_H
#define VMD_MD_SCRIPT_SHIM_H

#include <string.h>

#include <atomic>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "vmd_eval.h"

#ifndef VMD_SHIM_PREFIX
#define VMD_SHIM_PREFIX(name) name
#endif

/* the opaque payload VIAMD keeps per display property (src/viamd.h:350): here, which property of which script */
struct md_script_vis_payload_o { const md_script_ir_t* ir; std::string name; };

#ifndef VMD_SHIM_UNIT
#define VMD_SHIM_UNIT(dst, str) do { (dst) = ((str) && (str)[0]) ? md_unit_angstrom() : md_unit_none(); } while (0)
#endif
#ifndef VMD_SHIM_BITFIELD_INIT
#define VMD_SHIM_BITFIELD_INIT(bf, alloc) md_bitfield_init((bf), (alloc))
#endif
#ifndef VMD_SHIM_BITFIELD_SET
#define VMD_SHIM_BITFIELD_SET(bf, idx) md_bitfield_set_bit((bf), (uint64_t)(idx))
#endif

#ifndef VMD_SHIM_BITFIELD_TEST
#define VMD_SHIM_BITFIELD_TEST(bf, idx) md_bitfield_test_bit((bf), (uint64_t)(idx))
#endif
/* the frame mask the shim hands out (md_script_eval_frame_mask) is an md_bitfield_t the shim OWNS: initialised, emptied, filled and freed
 * through mdlib's own functions only - its storage layout is mdlib's business */
#ifndef VMD_SHIM_BITFIELD_CLEAR
#define VMD_SHIM_BITFIELD_CLEAR(bf) md_bitfield_clear((bf))
#endif
#ifndef VMD_SHIM_BITFIELD_FREE
#define VMD_SHIM_BITFIELD_FREE(bf) md_bitfield_free((bf))
#endif
/* Optional, for speed only: VMD_SHIM_BITFIELD_ASSIGN_WORDS(bf, words, num_bits) - make *bf hold exactly the set bits of the little-endian
 * 64-bit words (bit i of the mask = frame i).  Undefined by default: the mask is rebuilt with CLEAR + one SET per evaluated frame, on the
 * GUI thread, only when a fingerprint moved (src/main.cpp:1508-1513) */

/* ---- what this header requires of mdlib: NAMES, FIELDS and CALL SHAPES, checked at compile time -------------------------------
 * Every line is observable at the cited call site of /root/reference/src (the evaluator's own headers are an empty submodule there);
 * nothing else of mdlib's declarations is touched - storage that is not observable (md_bitfield_t's words, md_unit_t's members, md_array's
 * header) is only ever reached through the hooks above.  A real mdlib that differs fails HERE, with the field's name, not at run time. */
#define VMD_SHIM_REQUIRE_FIELD(T, f) static_assert(sizeof(((T*)nullptr)->f) > 0, #T "::" #f " is required by vmd_md_script_shim.h")
#define VMD_SHIM_REQUIRE_EXPR(expr) static_assert(sizeof(decltype(expr)) > 0, #expr " must be a valid expression for vmd_md_script_shim.h")
namespace vmd_shim_requires {
VMD_SHIM_REQUIRE_FIELD(str_t, ptr); VMD_SHIM_REQUIRE_FIELD(str_t, len);                                   /* src/main.cpp:1295 (STR_ARG), :5682 */
VMD_SHIM_REQUIRE_FIELD(md_system_t, atom.count); VMD_SHIM_REQUIRE_FIELD(md_system_t, atom.x);             /* src/main.cpp:5737-5741 */
VMD_SHIM_REQUIRE_FIELD(md_system_t, atom.y); VMD_SHIM_REQUIRE_FIELD(md_system_t, atom.z);
VMD_SHIM_REQUIRE_FIELD(md_system_t, atom.mass);                                                           /* src/viamd.cpp:2253 */
VMD_SHIM_REQUIRE_FIELD(md_system_t, unitcell); VMD_SHIM_REQUIRE_FIELD(md_system_t, trajectory);           /* src/viamd.cpp:1837; src/main.cpp:995 */
VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, x); VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, y); VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, z);      /* src/viamd.cpp:1837-1842 */
VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, xy); VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, xz); VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, yz);
VMD_SHIM_REQUIRE_FIELD(md_unitcell_t, flags);                                                             /* src/main.cpp:6255 */
VMD_SHIM_REQUIRE_FIELD(md_trajectory_frame_header_t, unitcell);                                            /* src/viamd.cpp:465-467 */
VMD_SHIM_REQUIRE_EXPR(md_trajectory_load_frame((md_trajectory_i*)nullptr, (int64_t)0, (md_trajectory_frame_header_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr));   /* src/viamd.cpp:465-467 */
VMD_SHIM_REQUIRE_EXPR(md_trajectory_num_frames((md_trajectory_i*)nullptr));                               /* src/main.cpp:1022 */
VMD_SHIM_REQUIRE_EXPR(md_trajectory_num_atoms((md_trajectory_i*)nullptr));                                /* src/viamd.cpp:455 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, dim[3]); VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, unit[1]);             /* src/main.cpp:1300-1301, 1353, 5773 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, values); VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, weights);             /* :1513, 1524 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, aggregate);                                             /* :1378 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, max_value);                                             /* density_volume.cpp:281 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, min_range[1]); VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, max_range[1]);  /* :1513, 1519-1522 */
VMD_SHIM_REQUIRE_FIELD(md_script_property_data_t, fingerprint);                                           /* :1508-1509 */
VMD_SHIM_REQUIRE_FIELD(md_script_aggregate_t, population_mean);                                           /* :1383-1440 */
VMD_SHIM_REQUIRE_FIELD(md_script_aggregate_t, population_var); VMD_SHIM_REQUIRE_FIELD(md_script_aggregate_t, population_ext);
VMD_SHIM_REQUIRE_FIELD(md_script_vis_t, atom_mask);                                                       /* src/viamd.cpp:3205-3207 */
VMD_SHIM_REQUIRE_FIELD(md_script_vis_t, sdf.extent); VMD_SHIM_REQUIRE_FIELD(md_script_vis_t, sdf.matrices);                        /* density_volume.cpp:190-204, 263-269 */
VMD_SHIM_REQUIRE_FIELD(md_script_vis_t, sdf.structures);
VMD_SHIM_REQUIRE_FIELD(md_script_vis_ctx_t, ir); VMD_SHIM_REQUIRE_FIELD(md_script_vis_ctx_t, mol); VMD_SHIM_REQUIRE_FIELD(md_script_vis_ctx_t, traj);   /* src/main.cpp:5751-5755 */
static_assert(sizeof(mat4_t) == 16 * sizeof(float), "mat4_t is 16 floats (column-major: mat4_mul_vec3(M, coord, 1.0f), src/main.cpp:5790)");
static_assert(sizeof(md_script_vis_flags_t(MD_SCRIPT_VISUALIZE_ATOMS)) > 0 && sizeof(md_script_vis_flags_t(MD_SCRIPT_VISUALIZE_SDF)) > 0, "MD_SCRIPT_VISUALIZE_ATOMS / _SDF (src/main.cpp:5757)");
}  // namespace vmd_shim_requires
/* Fields this header remembers of mdlib but NO line of the reference shows ([RECOLLECTION]): md_script_property_data_t::num_values and
 * ::min_value, md_script_aggregate_t::num_values, md_trajectory_frame_header_t::{num_atoms, index, timestamp}.  They are written / read
 * IF the host's mdlib has them (detected at compile time) and skipped otherwise - nothing VIAMD reads depends on them.  The allocator
 * mdlib's stretchy buffers of a md_script_vis_t grow with is `vis->alloc` by default ([RECOLLECTION]; md_script_vis_init(&vis, alloc),
 * src/main.cpp:5747, is all the reference shows): define VMD_SHIM_VIS_ALLOC(vis) if it lives elsewhere. */
#ifndef VMD_SHIM_VIS_ALLOC
#define VMD_SHIM_VIS_ALLOC(vis) ((vis)->alloc)
#endif
namespace vmd_shim {
template <class T> inline T peek(const T& v) { T r; __atomic_load(const_cast<T*>(&v), &r, __ATOMIC_RELAXED); return r; }
template <class T, class U> inline void pub(T& d, U v) { T t = (T)v; __atomic_store(&d, &t, __ATOMIC_RELAXED); }
#define VMD_SHIM_OPTIONAL_FIELD(field)                                                                                                  \
    template <class T, class V> inline auto set_if_##field(T& t, V v, int) -> decltype((void)(t.field), void()) { pub(t.field, v); }       \
    template <class T, class V> inline void set_if_##field(T&, V, long) {}                                                                \
    template <class T, class V> inline auto get_if_##field(const T& t, V, int) -> decltype((void)(t.field), V()) { return (V)peek(t.field); } \
    template <class T, class V> inline V get_if_##field(const T&, V fallback, long) { return fallback; }
VMD_SHIM_OPTIONAL_FIELD(num_values)
VMD_SHIM_OPTIONAL_FIELD(min_value)
VMD_SHIM_OPTIONAL_FIELD(num_atoms)
VMD_SHIM_OPTIONAL_FIELD(index)
VMD_SHIM_OPTIONAL_FIELD(timestamp)
}  // namespace vmd_shim

/* ---- the evaluator behind the shim (mdlib's own, renamed) -------------------------------------------------------------------- */
#ifndef VMD_SHIM_NO_FALLBACK
#ifndef VMD_SHIM_FALLBACK
#define VMD_SHIM_FALLBACK(name) mdlib_##name
#endif
#ifndef VMD_SHIM_FALLBACK_DECLARED
/* mdlib's md_script_eval_t and md_script_vis_payload_o are opaque to their callers; behind the rename they are these two names */
struct vmd_shim_fallback_eval_t;
struct vmd_shim_fallback_payload_t;
extern "C" {
vmd_shim_fallback_eval_t* VMD_SHIM_FALLBACK(md_script_eval_create)(size_t num_frames, const md_script_ir_t* ir, md_allocator_i* alloc);
void     VMD_SHIM_FALLBACK(md_script_eval_free)(vmd_shim_fallback_eval_t* eval);
void     VMD_SHIM_FALLBACK(md_script_eval_clear_data)(vmd_shim_fallback_eval_t* eval);
void     VMD_SHIM_FALLBACK(md_script_eval_interrupt)(vmd_shim_fallback_eval_t* eval);
uint64_t VMD_SHIM_FALLBACK(md_script_eval_ir_fingerprint)(const vmd_shim_fallback_eval_t* eval);
bool     VMD_SHIM_FALLBACK(md_script_eval_frame_range)(vmd_shim_fallback_eval_t* eval, const md_script_ir_t* ir, const md_system_t* sys,
                                                       md_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end);
const md_script_property_data_t* VMD_SHIM_FALLBACK(md_script_eval_property_data)(const vmd_shim_fallback_eval_t* eval, str_t name);
const md_bitfield_t* VMD_SHIM_FALLBACK(md_script_eval_frame_mask)(const vmd_shim_fallback_eval_t* eval);
const vmd_shim_fallback_payload_t* VMD_SHIM_FALLBACK(md_script_ir_property_vis_payload)(const md_script_ir_t* ir, str_t name);
bool     VMD_SHIM_FALLBACK(md_script_vis_eval_payload)(md_script_vis_t* vis, const vmd_shim_fallback_payload_t* payload, int subidx,
                                                       const md_script_vis_ctx_t* ctx, md_script_vis_flags_t flags);
}
#endif
#define VMD_SHIM_HAVE_FALLBACK 1
#else
struct vmd_shim_fallback_eval_t;
struct vmd_shim_fallback_payload_t;
#define VMD_SHIM_HAVE_FALLBACK 0
#endif

/* ---- IR registry: md_script_ir_t* -> the descriptors of its properties ------------------------------------------------------ */
namespace vmd_shim {
struct PayloadEval { vmd_script_eval_t* eval = nullptr; size_t num_frames = 0; };
struct Registry {
    std::mutex mtx;
    std::map<const void*, const vmd_script_ir_t*> ir;
    std::map<const void*, const md_script_ir_t*> fallback_ir;        /* md_ir -> the IR the fallback evaluates for it (default: md_ir itself) */
    /* vis payloads handed to VIAMD (one per (ir, property), stable addresses) and one small evaluator per ir that serves them:
     * md_script_vis_eval_payload has no eval argument (density_volume.cpp:188), the reference pose and the per-frame alignment
     * of an sdf() live behind vmd_eval_sdf_payload */
    std::map<std::pair<const void*, std::string>, std::unique_ptr<md_script_vis_payload_o>> payloads;
    std::map<const void*, PayloadEval> payload_evals;
    /* md_trajectory_i* -> the backend's own interface for the same frames (vmd_shim_bind_trajectory) */
    std::map<const void*, vmd_trajectory_i*> native_traj;
    /* live evals per script: VIAMD creates the full and the filtered eval from one ir, in that order (src/main.cpp:971-972); the later
     * ones take the first as their SOURCE, so that a timeline sub-range is served from the block partials the full evaluation left */
    std::map<const void*, std::vector<vmd_script_eval_t*>> evals;
};
inline Registry& registry() { static Registry r; return r; }
inline const vmd_script_ir_t* find_ir(const void* md_ir) {
    Registry& r = registry();
    std::lock_guard<std::mutex> l(r.mtx);
    auto it = r.ir.find(md_ir);
    return it == r.ir.end() ? nullptr : it->second;
}
inline const md_script_ir_t* fallback_ir_of(const md_script_ir_t* md_ir) {
    Registry& r = registry();
    std::lock_guard<std::mutex> l(r.mtx);
    auto it = r.fallback_ir.find(md_ir);
    return it == r.fallback_ir.end() ? md_ir : it->second;
}
inline bool name_is(const std::string& a, str_t b) { return a.size() == (size_t)b.len && memcmp(a.data(), b.ptr, (size_t)b.len) == 0; }
/* is `name` one of the properties the GPU evaluates for this ir? */
inline bool bound_name(const vmd_script_ir_t* vir, str_t name) {
    if (!vir) return false;
    const size_t n = vmd_ir_property_count(vir);
    const char* const* names = vmd_ir_property_names(vir);
    for (size_t i = 0; i < n; ++i) if (strlen(names[i]) == (size_t)name.len && memcmp(names[i], name.ptr, (size_t)name.len) == 0) return true;
    return false;
}

/* md_unitcell_t -> vmd_unitcell_t: the six basis parameters VIAMD itself reads (src/viamd.cpp:1837-1842) + the periodicity bits */
template <class Cell>
inline vmd_unitcell_t unitcell(const Cell& c) {
    vmd_unitcell_t u;
    u.x = (float)c.x; u.y = (float)c.y; u.z = (float)c.z; u.xy = (float)c.xy; u.xz = (float)c.xz; u.yz = (float)c.yz;
    u.flags = (uint32_t)c.flags & VMD_UNITCELL_PBC_ALL;
    return u;
}

/* md_trajectory_i behind vmd_trajectory_i: VIAMD only ever calls md_trajectory_load_frame on it (src/viamd.cpp:465-467) */
inline bool load_frame_adapter(void* inst, int64_t idx, vmd_frame_header_t* h, float* x, float* y, float* z) {
    md_trajectory_i* traj = (md_trajectory_i*)inst;
    md_trajectory_frame_header_t hdr = {};
    if (!md_trajectory_load_frame(traj, idx, &hdr, x, y, z)) return false;
    if (h) {
        h->num_atoms = get_if_num_atoms(hdr, (size_t)md_trajectory_num_atoms(traj), 0);
        h->index = get_if_index(hdr, (int64_t)idx, 0);
        h->timestamp = get_if_timestamp(hdr, (double)idx, 0);
        h->unitcell = unitcell(hdr.unitcell);
    }
    return true;
}
inline size_t num_frames_adapter(void* inst) { return (size_t)md_trajectory_num_frames((md_trajectory_i*)inst); }
inline size_t num_atoms_adapter(void* inst) { return (size_t)md_trajectory_num_atoms((md_trajectory_i*)inst); }
inline vmd_trajectory_i wrap_trajectory(md_trajectory_i* traj) {
    vmd_trajectory_i t;
    {
        /* a trajectory the host opened through the backend's readers (or uploaded to HBM): hand the evaluator THAT interface, with its
         * device view / raw frames / mapped file - behind load_frame alone every frame would be decoded on the host and staged over PCIe */
        Registry& r = registry();
        std::lock_guard<std::mutex> l(r.mtx);
        auto it = r.native_traj.find(traj);
        if (it != r.native_traj.end()) return *it->second;
    }
    memset(&t, 0, sizeof(t));
    t.inst = traj; t.num_frames = num_frames_adapter; t.num_atoms = num_atoms_adapter; t.load_frame = load_frame_adapter;
    return t;
}
inline vmd_system_t wrap_system(const md_system_t* sys) {
    vmd_system_t s;
    memset(&s, 0, sizeof(s));
    s.atom_count = (size_t)sys->atom.count;
    s.x = sys->atom.x; s.y = sys->atom.y; s.z = sys->atom.z; s.mass = sys->atom.mass;
    s.unitcell = unitcell(sys->unitcell);
    /* md_system_t::bond (read while evaluations run, src/viamd.cpp:3088-3091): a host with mdlib's md_bond_data_t at hand sets
     * s.bonds / s.bond_count to its atom index pairs (VMD_SHIM_BONDS(sys, &s) if defined) - sdf() structures are then made whole
     * along the bond graph like md_util_unwrap does (src/viamd.cpp:2257); without them, along their index order */
#ifdef VMD_SHIM_BONDS
    VMD_SHIM_BONDS(sys, &s);
#endif
    return s;
}
}  // namespace vmd_shim

/* Work threshold: VIAMD opens datasets/1ALA-500.pdb - ~1e2 atoms x 500 frames - with the script of src/main.cpp:528,
 * and wants to stay interactive (TODO.md:36).  An evaluation that small costs more to hand to the GPU (batch planning, a dozen launches, the
 * views over PCIe: ~1 ms) than mdlib needs to evaluate it on the pool threads that are calling anyway.  md_script_eval_create therefore leaves
 * a bound ir with the evaluator behind the shim when   vmd_ir_work_per_frame(vmd_ir) x num_frames < min_work   (atom pairs x frames): no GPU
 * eval is created, the fallback evaluates the WHOLE script (not the reduced one), every record is mdlib's.  Default: VMD_SHIM_MIN_WORK_DEFAULT,
 * from bench.py's `secondary.c1`; 0 sends everything that is bound to the GPU.  Without an evaluator behind the shim
 * (VMD_SHIM_NO_FALLBACK) there is nobody to leave it with: the threshold is ignored. */
#ifndef VMD_SHIM_MIN_WORK_DEFAULT
#define VMD_SHIM_MIN_WORK_DEFAULT 1000000ull
#endif
namespace vmd_shim { inline std::atomic<uint64_t>& min_work() { static std::atomic<uint64_t> v{VMD_SHIM_MIN_WORK_DEFAULT}; return v; } }
inline void vmd_shim_set_min_work(uint64_t pairs_times_frames) { vmd_shim::min_work().store(pairs_times_frames, std::memory_order_relaxed); }
inline uint64_t vmd_shim_min_work() { return vmd_shim::min_work().load(std::memory_order_relaxed); }

/* the host's one extra call: which descriptors belong to this compiled script (NULL unbinds; call before md_script_eval_create) */
inline void vmd_shim_bind_ir(const md_script_ir_t* md_ir, const vmd_script_ir_t* vmd_ir) {
    vmd_shim::Registry& r = vmd_shim::registry();
    std::lock_guard<std::mutex> l(r.mtx);
    if (vmd_ir) {
        auto cur = r.ir.find(md_ir);
        const bool same = cur != r.ir.end() && cur->second == vmd_ir;
        r.ir[md_ir] = vmd_ir;
        if (same) return;
        /* a NEW binding under an address that was bound before (mdlib recycles the ir's memory after md_script_ir_free): the payload
         * evaluator and the payloads of the old script must not outlive it */
    } else {
        r.ir.erase(md_ir);
    }
    auto pe = r.payload_evals.find(md_ir);
    if (pe != r.payload_evals.end()) { vmd_eval_free(pe->second.eval); r.payload_evals.erase(pe); }
    for (auto it = r.payloads.begin(); it != r.payloads.end();) it = it->first.first == md_ir ? r.payloads.erase(it) : std::next(it);
}

/* Optional: the IR the fallback evaluator runs for `md_ir` - one mdlib compiled from vmd_script_report_fallback_source() (the script
 * without the statements the GPU evaluates), so that nothing is computed twice.  Default (and fallback_ir = NULL): md_ir itself.  The
 * reduced IR must outlive the evals created from md_ir, like md_ir. */
inline void vmd_shim_bind_fallback_ir(const md_script_ir_t* md_ir, const md_script_ir_t* fallback_ir) {
    vmd_shim::Registry& r = vmd_shim::registry();
    std::lock_guard<std::mutex> l(r.mtx);
    if (fallback_ir) r.fallback_ir[md_ir] = fallback_ir; else r.fallback_ir.erase(md_ir);
}

/* Optional: the frames behind `md_traj` are also available through `native` - a vmd_xdrtraj / vmd_dcdtraj / vmd_rawtraj / vmd_devtraj
 * interface of the same file or of a copy in HBM (a loader shim registers the pair where VIAMD attaches the trajectory,
 * src/loader.cpp:111-159).  md_script_eval_frame_range then evaluates from `native` - frames decompressed on the GPU, DMA'd out of the
 * mapped file, or read in place from HBM - while VIAMD keeps using md_traj for display.  native = NULL unbinds; the native interface
 * must outlive the binding. */
inline void vmd_shim_bind_trajectory(const md_trajectory_i* md_traj, vmd_trajectory_i* native) {
    vmd_shim::Registry& r = vmd_shim::registry();
    std::lock_guard<std::mutex> l(r.mtx);
    if (native) r.native_traj[md_traj] = native; else r.native_traj.erase(md_traj);
}

/* ---- md_script_eval_t ------------------------------------------------------------------------------------------------------ */
struct md_script_eval_t {
    vmd_script_eval_t* eval = nullptr;           /* the GPU evaluator of the bound properties; NULL for a script without any */
    const vmd_script_ir_t* vir = nullptr;        /* what `eval` evaluates: the binding of md_ir at creation, or NULL when the script was too small (vmd_shim_set_min_work) */
    const vmd_script_ir_t* bound_vir = nullptr;  /* the binding of md_ir at creation, whoever evaluates it (a later re-binding makes the eval stale) */
    const md_script_ir_t* md_ir = nullptr;
    vmd_shim_fallback_eval_t* fb = nullptr;      /* mdlib's evaluator of everything else; NULL without fallback hooks */
    const md_script_ir_t* fb_ir = nullptr;       /* the IR `fb` was created from (md_ir, or the reduced one of vmd_shim_bind_fallback_ir) */
    size_t num_frames = 0;
    /* md_script_property_data_t records handed to VIAMD: fetched once and cached by the GUI (src/main.cpp:1286,1303), so their
     * addresses are stable for the eval's lifetime; the arrays they point at are the owning evaluator's (equally stable), the scalar
     * fields (fingerprint, ranges, max_value) are refreshed from it whenever data may have changed.  EVERY record VIAMD gets is one of
     * these - the GPU's (src) and, with an evaluator behind the shim, mdlib's (fb_src) - because the shim must be able to move a
     * record's fingerprint: VIAMD re-reads a property (and md_script_eval_frame_mask with it) only when prop_data->fingerprint changes
     * (:1508-1513), and behind the shim the mask is the AND of two evaluators' masks - it can grow after the owning evaluator's last
     * fingerprint change (the other evaluator finishes the frame later; a deferred settle lands later still).  So the fingerprint handed
     * out is the owner's mixed with `epoch`, which moves whenever either evaluator has finished something. */
    struct Prop {
        std::string name;
        const vmd_script_property_data_t* src = nullptr;        /* the GPU's record ... */
        const md_script_property_data_t* fb_src = nullptr;      /* ... or mdlib's */
        md_script_property_data_t dst;
        md_script_aggregate_t agg;
    };
    std::vector<std::unique_ptr<Prop>> props;
    std::atomic<uint64_t> epoch{0};
    std::vector<uint64_t> mask_words;
    md_bitfield_t mask[2];                           /* double-buffered: the one handed out last stays untouched while the next is built */
    int mask_cur = 0;
    std::mutex mtx;

    /* The evaluators change the scalar fields while other pool threads are inside frame_range, and VIAMD's GUI thread reads the records
     * below at any time (src/main.cpp:1508-1524: old and new fields side by side are tolerated).  Both directions go through relaxed
     * atomic loads / stores - plain moves on x86-64 - so the hand-over is defined behaviour, and ThreadSanitizer-clean. */
    static uint64_t mix(uint64_t fp, uint64_t epoch) { return fp ^ (epoch * 0x9E3779B97F4A7C15ull); }
    void refresh() {                                 /* (mtx) */
        using namespace vmd_shim;
        const uint64_t ep = epoch.load(std::memory_order_acquire);
        for (auto& p : props) {
            md_script_property_data_t& d = p->dst;
            if (p->src) {
                const vmd_script_property_data_t* s = p->src;
                for (int k = 0; k < 4; ++k) pub(d.dim[k], s->dim[k]);
                pub(d.values, peek(s->values)); pub(d.weights, s->weights); set_if_num_values(d, s->num_values, 0);      /* (a volume's `values` moves between the shared zeros and its view: clear_data / first view) */
                set_if_min_value(d, peek(s->min_value), 0); pub(d.max_value, peek(s->max_value));
                for (int k = 0; k < 2; ++k) { pub(d.min_range[k], peek(s->min_range[k])); pub(d.max_range[k], peek(s->max_range[k])); }
                if (s->aggregate) {
                    set_if_num_values(p->agg, s->aggregate->num_values, 0);
                    pub(p->agg.population_mean, s->aggregate->population_mean);
                    pub(p->agg.population_var, s->aggregate->population_var);
                    pub(p->agg.population_ext, (decltype(p->agg.population_ext))s->aggregate->population_ext);
                    pub(d.aggregate, &p->agg);
                } else {
                    pub(d.aggregate, (decltype(d.aggregate))nullptr);
                }
                pub(d.fingerprint, mix(peek(s->fingerprint), ep));      /* last: the GUI compares it to decide whether to re-read (src/main.cpp:1508-1509) */
            } else {
                const md_script_property_data_t* s = p->fb_src;         /* mdlib's record, read the way VIAMD itself reads it */
                for (int k = 0; k < 4; ++k) pub(d.dim[k], peek(s->dim[k]));
                pub(d.values, peek(s->values)); pub(d.weights, peek(s->weights));
                pub(d.aggregate, peek(s->aggregate));
                pub(d.max_value, peek(s->max_value));      /* (fields the shim does not know by name keep the value of the copy made at the first request) */
                for (int k = 0; k < 2; ++k) { pub(d.min_range[k], peek(s->min_range[k])); pub(d.max_range[k], peek(s->max_range[k])); }
                pub(d.fingerprint, mix(peek(s->fingerprint), ep));
            }
        }
    }
    /* either evaluator has finished something (a range, a deferred settle): every record's fingerprint moves, so a polling GUI looks again */
    void advance() {
        epoch.fetch_add(1, std::memory_order_acq_rel);
        std::lock_guard<std::mutex> l(mtx);
        refresh();
    }
    static void on_settled(void* self) { ((md_script_eval_t*)self)->advance(); }       /* vmd_eval_set_settled_callback (deferred-settle mode) */
};

inline md_script_eval_t* VMD_SHIM_PREFIX(md_script_eval_create)(size_t num_frames, const md_script_ir_t* ir, md_allocator_i* alloc) {
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    std::unique_ptr<md_script_eval_t> e(new md_script_eval_t());
    e->bound_vir = vir;
    e->md_ir = ir;
    e->num_frames = num_frames;
#if VMD_SHIM_HAVE_FALLBACK
    /* too small to be worth a trip to the GPU (vmd_shim_set_min_work): this eval is mdlib's alone, whole script */
    const bool too_small = vir && vmd_ir_work_per_frame(vir) * (uint64_t)num_frames < vmd_shim_min_work();
    if (too_small) vir = nullptr;
    e->vir = vir;
    /* mdlib's evaluator of the same script (or of the reduced one): every property the GPU does not evaluate lives there */
    e->fb_ir = too_small ? ir : vmd_shim::fallback_ir_of(ir);
    e->fb = VMD_SHIM_FALLBACK(md_script_eval_create)(num_frames, e->fb_ir, alloc);
#else
    e->vir = vir;                                /* nobody to leave a small script with: the threshold does not apply */
#endif
    (void)alloc;                                 /* host allocations of the GPU part are the library's own */
    if (!vir && !e->fb) return nullptr;          /* nothing bound and nobody to fall back on: as mdlib for an invalid ir */
#ifdef VMD_SHIM_DEFERRED_SETTLE
    /* A VIAMD build whose task pool may have a single worker (src/main.cpp:494-495 clamps to >= 2 today): small calls are evaluated ahead
     * whoever makes them, the final settle trails the last call by a fraction of a millisecond (include/vmd_eval.h, vmd_eval_wait_settled).
     * Safe under VIAMD's teardown order - interrupt_async_tasks interrupts both evals before the system's arena is reset
     * (src/viamd.cpp:234-241, 624-630) and vmd_eval_interrupt drops / awaits the owed settle - but not part of mdlib's contract: opt-in
     * (set per eval right after vmd_eval_create below). */
#endif
    if (vir) {
        e->eval = vmd_eval_create(num_frames, vir);
#ifdef VMD_SHIM_DEFERRED_SETTLE
        if (e->eval) vmd_eval_set_deferred_settle(e->eval, 1);      /* this eval only: nothing process-wide changes */
#endif
        /* whoever settles late (the helper thread of the deferred mode - also when the PROCESS opted in with readahead_lone) tells the shim,
         * which moves the fingerprints VIAMD polls (without this the GUI kept the pre-settle histogram for good) */
        if (e->eval) vmd_eval_set_settled_callback(e->eval, &md_script_eval_t::on_settled, e.get());
#if VMD_SHIM_HAVE_FALLBACK
        if (e->fb && e->fb_ir == ir) {
            /* the fallback evaluates the WHOLE script: rdf / sdf / distance are then computed on the CPU as well, after the GPU part of
             * every call - correct, but no faster than mdlib alone.  Said once per process. */
            static std::atomic<bool> told{false};
            if (!told.exchange(true))
                vmd_log_message(VMD_LOG_INFO, "vmd_md_script_shim: no reduced fallback IR is bound (vmd_shim_bind_fallback_ir): mdlib evaluates the whole script "
                                              "behind the GPU part, hot-path properties included; compile vmd_script_report_fallback_source() and bind it");
        }
#endif
        if (!e->eval) {
#if VMD_SHIM_HAVE_FALLBACK
            if (e->fb) VMD_SHIM_FALLBACK(md_script_eval_free)(e->fb);
#endif
            return nullptr;
        }
        const size_t n = vmd_ir_property_count(vir);
        for (size_t i = 0; i < n; ++i) {
            std::unique_ptr<md_script_eval_t::Prop> p(new md_script_eval_t::Prop());
            p->name = vmd_ir_property_names(vir)[i];
            p->src = vmd_eval_property_data(e->eval, p->name.c_str());
            p->dst = md_script_property_data_t();
            p->agg = md_script_aggregate_t();
            /* unit[2] (src/main.cpp:1300-1301, printed at :1314-1315): the backend carries the printed form, VMD_SHIM_UNIT makes the
             * md_unit_t of it */
            VMD_SHIM_UNIT(p->dst.unit[0], p->src->unit_str[0]);
            VMD_SHIM_UNIT(p->dst.unit[1], p->src->unit_str[1]);
            e->props.push_back(std::move(p));
        }
    }
    e->mask_words.assign((num_frames + 63) / 64 + 1, 0);
    for (md_bitfield_t& m : e->mask) VMD_SHIM_BITFIELD_INIT(&m, alloc);      /* mdlib's own storage, reached through mdlib's own functions only */
    e->refresh();
    if (e->eval) {
        vmd_shim::Registry& r = vmd_shim::registry();
        std::lock_guard<std::mutex> l(r.mtx);
        std::vector<vmd_script_eval_t*>& live = r.evals[ir];
#ifndef VMD_SHIM_NO_AUTO_SOURCE
        /* VIAMD's filtered eval takes the full one as its SOURCE (block partials instead of a second evaluation).  The backend hands
         * blocks over only between evals that are evaluating the same trajectory instance, so a host that runs two evals of one ir over
         * different trajectories of equal length gets two evaluations; a refusal (another device) is not an error of this call */
        if (!live.empty() && vmd_eval_num_frames(live.front()) == num_frames && !vmd_eval_set_source(e->eval, live.front())) vmd_clear_last_error();
#endif
        live.push_back(e->eval);
    }
    return e.release();
}
inline void VMD_SHIM_PREFIX(md_script_eval_free)(md_script_eval_t* e) {
    if (!e) return;
    if (e->eval) {
        /* VIAMD frees the full eval before the filtered one (src/main.cpp:959-964): nobody may keep it as a source */
        vmd_shim::Registry& r = vmd_shim::registry();
        std::lock_guard<std::mutex> l(r.mtx);
        auto it = r.evals.find(e->md_ir);
        if (it != r.evals.end()) {
            std::vector<vmd_script_eval_t*>& live = it->second;
            const bool was_source = !live.empty() && live.front() == e->eval;
            for (size_t i = 0; i < live.size(); ++i) if (live[i] == e->eval) { live.erase(live.begin() + (long)i); break; }
            if (was_source) for (vmd_script_eval_t* o : live) vmd_eval_set_source(o, nullptr);
            if (live.empty()) r.evals.erase(it);
        }
    }
    vmd_eval_free(e->eval);                      /* waits for a deferred settle (and its callback into *e) that is running */
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) VMD_SHIM_FALLBACK(md_script_eval_free)(e->fb);
#endif
    for (md_bitfield_t& m : e->mask) VMD_SHIM_BITFIELD_FREE(&m);
    delete e;
}
inline void VMD_SHIM_PREFIX(md_script_eval_clear_data)(md_script_eval_t* e) {
    if (!e) return;
    if (e->eval) vmd_eval_clear_data(e->eval);
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) VMD_SHIM_FALLBACK(md_script_eval_clear_data)(e->fb);
#endif
    e->advance();
}
inline void VMD_SHIM_PREFIX(md_script_eval_interrupt)(md_script_eval_t* e) {
    if (!e) return;
    if (e->eval) vmd_eval_interrupt(e->eval);
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) VMD_SHIM_FALLBACK(md_script_eval_interrupt)(e->fb);
#endif
}
/* md_script_ir_fingerprint of the bound IR: what a host WITHOUT mdlib behind the shim compares the eval's fingerprint with */
inline uint64_t vmd_shim_ir_fingerprint(const md_script_ir_t* ir) {
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    return vir ? vmd_ir_fingerprint(vir) : 0;
}
/* src/main.cpp:987: md_script_eval_ir_fingerprint(eval) == md_script_ir_fingerprint(ir).  With mdlib behind the shim the right-hand side
 * is mdlib's, so the answer is mdlib's fingerprint of the IR the eval was created from: the fallback eval's own when it evaluates that
 * very IR, VMD_SHIM_IR_FINGERPRINT(md_ir) (default md_script_ir_fingerprint) when it evaluates the reduced one.  While the GPU binding of
 * the ir is no longer the one this eval was created with, the value is perturbed: VIAMD sees a mismatch and re-creates its evals. */
#ifndef VMD_SHIM_IR_FINGERPRINT
#define VMD_SHIM_IR_FINGERPRINT(ir) md_script_ir_fingerprint(ir)
#endif
inline uint64_t VMD_SHIM_PREFIX(md_script_eval_ir_fingerprint)(const md_script_eval_t* e) {
    if (!e) return 0;
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) {
        uint64_t fp = e->fb_ir == e->md_ir ? VMD_SHIM_FALLBACK(md_script_eval_ir_fingerprint)(e->fb) : (uint64_t)VMD_SHIM_IR_FINGERPRINT(e->md_ir);
        if (vmd_shim::find_ir(e->md_ir) != e->bound_vir) fp ^= 0x9E3779B97F4A7C15ull;
        return fp;
    }
#endif
    return e->eval ? vmd_eval_ir_fingerprint(e->eval) : 0;
}

/* the hot call: pool threads, disjoint ranges, one eval (src/main.cpp:993-997).  The GPU part first (the calls of a pool arrive together
 * and are combined / evaluated ahead), then mdlib's evaluator for the same range on the calling thread - the pool's other
 * threads are inside their own ranges meanwhile, exactly as without the shim. */
inline bool VMD_SHIM_PREFIX(md_script_eval_frame_range)(md_script_eval_t* e, const md_script_ir_t* ir, const md_system_t* sys,
                                                        md_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    if (!e || !sys || !traj) return false;
    if (ir != e->md_ir) return false;            /* mdlib compares fingerprints the same way */
    bool ok = true;
    if (e->eval) {
        const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
        if (vir != e->vir) return false;
        const vmd_system_t vsys = vmd_shim::wrap_system(sys);
        vmd_trajectory_i vtraj = vmd_shim::wrap_trajectory(traj);
        ok = vmd_eval_frame_range(e->eval, vir, &vsys, &vtraj, frame_beg, frame_end);
        e->advance();
    }
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb && ok) {
        ok = VMD_SHIM_FALLBACK(md_script_eval_frame_range)(e->fb, e->fb_ir, sys, traj, frame_beg, frame_end);
        /* the AND of the masks has grown by this range only now: the records' fingerprints move once more, or a GUI that built a
         * histogram between the two parts would keep it (found by the reference's own update_display_properties, tests/native/ref_callsites.cpp) */
        e->advance();
    }
#endif
    return ok;
}

/* src/main.cpp:1286, once per name of md_script_ir_property_names: the GPU's record for a bound name, mdlib's for every other one
 * (`a1`, `lin`, `plan`, `iso` of the default script) - NULL only where mdlib says NULL */
inline const md_script_property_data_t* VMD_SHIM_PREFIX(md_script_eval_property_data)(const md_script_eval_t* e, str_t name) {
    if (!e) return nullptr;
    md_script_eval_t* me = const_cast<md_script_eval_t*>(e);
    std::lock_guard<std::mutex> l(me->mtx);
    for (auto& p : me->props) if (vmd_shim::name_is(p->name, name)) return &p->dst;
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) {
        /* mdlib's record, handed out as a copy whose fingerprint the shim can move (see md_script_eval_t::Prop); made at the first request */
        const md_script_property_data_t* rec = VMD_SHIM_FALLBACK(md_script_eval_property_data)(e->fb, name);
        if (!rec) return nullptr;
        std::unique_ptr<md_script_eval_t::Prop> p(new md_script_eval_t::Prop());
        p->name.assign(name.ptr, (size_t)name.len);
        p->fb_src = rec;
        p->dst = *rec;                               /* unit[2] and whatever else mdlib keeps in the record */
        p->agg = md_script_aggregate_t();
        me->props.push_back(std::move(p));
        me->refresh();
        return &me->props.back()->dst;
    }
#endif
    return nullptr;
}

/* frames evaluated so far as the bitfield VIAMD iterates (src/main.cpp:194-210, 1513); refreshed by every call.  With two evaluators a
 * frame is done once BOTH have it: a histogram built over the mask never reads a temporal row one of them has not written yet. */
inline const md_bitfield_t* VMD_SHIM_PREFIX(md_script_eval_frame_mask)(const md_script_eval_t* ce) {
    md_script_eval_t* e = const_cast<md_script_eval_t*>(ce);
    if (!e) return nullptr;
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb && !e->eval) return VMD_SHIM_FALLBACK(md_script_eval_frame_mask)(e->fb);
#endif
    std::lock_guard<std::mutex> l(e->mtx);
    vmd_eval_frame_mask_bits(e->eval, e->mask_words.data(), e->mask_words.size());
#if VMD_SHIM_HAVE_FALLBACK
    if (e->fb) {
        const md_bitfield_t* fm = VMD_SHIM_FALLBACK(md_script_eval_frame_mask)(e->fb);
        for (size_t w = 0; w < e->mask_words.size(); ++w) {
            uint64_t word = e->mask_words[w];
            if (!word) continue;
            for (uint64_t bits = word; bits; bits &= bits - 1) {
                const size_t f = w * 64 + (size_t)__builtin_ctzll(bits);
                if (!fm || !VMD_SHIM_BITFIELD_TEST(fm, f)) word &= ~(1ull << (f & 63));
            }
            e->mask_words[w] = word;
        }
    }
#endif
    /* into the buffer that was NOT handed out last: a reader still iterating the previous answer keeps a consistent bitfield */
    md_bitfield_t* out = &e->mask[e->mask_cur ^= 1];
#ifdef VMD_SHIM_BITFIELD_ASSIGN_WORDS
    VMD_SHIM_BITFIELD_ASSIGN_WORDS(out, e->mask_words.data(), e->num_frames);
#else
    VMD_SHIM_BITFIELD_CLEAR(out);
    for (size_t w = 0; w < e->mask_words.size(); ++w)
        for (uint64_t bits = e->mask_words[w]; bits; bits &= bits - 1) VMD_SHIM_BITFIELD_SET(out, w * 64 + (size_t)__builtin_ctzll(bits));
#endif
    return out;
}

/* ---- vis payloads ----------------------------------------------------------------------------------------------------------- */
/* md_script_ir_property_vis_payload(ir, name), src/main.cpp:1304: NULL for an unknown ir / property */
inline const md_script_vis_payload_o* VMD_SHIM_PREFIX(md_script_ir_property_vis_payload)(const md_script_ir_t* ir, str_t name) {
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    if (!vmd_shim::bound_name(vir, name)) {
#if VMD_SHIM_HAVE_FALLBACK
        /* mdlib's own payload of a property the GPU does not evaluate: an opaque pointer VIAMD only hands back (src/main.cpp:1304) */
        return (const md_script_vis_payload_o*)VMD_SHIM_FALLBACK(md_script_ir_property_vis_payload)(ir, name);
#else
        return nullptr;
#endif
    }
    const std::string nm(name.ptr, (size_t)name.len);
    vmd_shim::Registry& r = vmd_shim::registry();
    std::lock_guard<std::mutex> l(r.mtx);
    std::unique_ptr<md_script_vis_payload_o>& p = r.payloads[std::make_pair((const void*)ir, nm)];
    if (!p) { p.reset(new md_script_vis_payload_o()); p->ir = ir; p->name = nm; }
    return p.get();
}

/* md_script_vis_eval_payload(&vis, payload, subidx, &ctx, flags).  Served for sdf() properties - the one payload on the evaluation
 * path: MD_SCRIPT_VISUALIZE_SDF fills vis->sdf.{extent, matrices, structures} for trajectory frame 0 of ctx->traj (the reference pose
 * VIAMD draws the volume in: density_volume.cpp:190-204, 263-269; export_cube, src/main.cpp:5751-5803), MD_SCRIPT_VISUALIZE_ATOMS adds
 * the atoms of the reference structures to vis->atom_mask (src/viamd.cpp:3205-3207).  subidx >= 0 selects one structure.  Payloads of
 * other property kinds return false (their highlighting is mdlib's own). */
inline bool VMD_SHIM_PREFIX(md_script_vis_eval_payload)(md_script_vis_t* vis, const md_script_vis_payload_o* payload, int subidx,
                                                        const md_script_vis_ctx_t* ctx, md_script_vis_flags_t flags) {
    if (!vis || !payload || !ctx) return false;
    {
        /* a payload this shim did not hand out is mdlib's: its evaluator draws it (angles, planes, selections ...) */
        vmd_shim::Registry& r = vmd_shim::registry();
        bool ours = false;
        {
            std::lock_guard<std::mutex> l(r.mtx);
            for (auto& kv : r.payloads) ours = ours || kv.second.get() == payload;
        }
        if (!ours) {
#if VMD_SHIM_HAVE_FALLBACK
            return VMD_SHIM_FALLBACK(md_script_vis_eval_payload)(vis, (const vmd_shim_fallback_payload_t*)payload, subidx, ctx, flags);
#else
            return false;
#endif
        }
    }
    if (!ctx->mol || !ctx->traj) return false;
    const vmd_script_ir_t* vir = vmd_shim::find_ir(payload->ir);
    if (!vir || !(vmd_ir_property_flags(vir, payload->name.c_str()) & VMD_PROPERTY_FLAG_VOLUME)) return false;
    vmd_shim::Registry& r = vmd_shim::registry();
    vmd_script_eval_t* ev = nullptr;
    const size_t F = (size_t)md_trajectory_num_frames(ctx->traj);
    {
        std::lock_guard<std::mutex> l(r.mtx);
        vmd_shim::PayloadEval& pe = r.payload_evals[payload->ir];
        if (!pe.eval || pe.num_frames != F) { vmd_eval_free(pe.eval); pe.eval = vmd_eval_create(F, vir); pe.num_frames = F; }
        ev = pe.eval;
    }
    if (!ev) return false;
    const vmd_system_t vsys = vmd_shim::wrap_system(ctx->mol);
    vmd_trajectory_i vtraj = vmd_shim::wrap_trajectory(ctx->traj);
    vmd_sdf_payload_t out;
    if (!vmd_eval_sdf_payload(ev, payload->name.c_str(), &vsys, &vtraj, 0, &out)) return false;
    const size_t k0 = subidx >= 0 ? (size_t)subidx : 0, k1 = subidx >= 0 ? (size_t)subidx + 1 : out.num_structures;
    if (k1 > out.num_structures) return false;
    if (flags & MD_SCRIPT_VISUALIZE_SDF) {
        vis->sdf.extent = out.extent;
        md_array_resize(vis->sdf.matrices, k1 - k0, VMD_SHIM_VIS_ALLOC(vis));
        md_array_resize(vis->sdf.structures, k1 - k0, VMD_SHIM_VIS_ALLOC(vis));
        for (size_t k = k0; k < k1; ++k) {
            memcpy(&vis->sdf.matrices[k - k0], out.matrices + 16 * k, 16 * sizeof(float));        /* mat4_t: 16 floats, column-major */
            md_bitfield_t* bf = &vis->sdf.structures[k - k0];
            VMD_SHIM_BITFIELD_INIT(bf, VMD_SHIM_VIS_ALLOC(vis));
            for (size_t a = 0; a < out.atoms_per_structure; ++a) VMD_SHIM_BITFIELD_SET(bf, out.structures[k * out.atoms_per_structure + a]);
        }
    }
    if (flags & MD_SCRIPT_VISUALIZE_ATOMS)
        for (size_t k = k0; k < k1; ++k)
            for (size_t a = 0; a < out.atoms_per_structure; ++a) VMD_SHIM_BITFIELD_SET(&vis->atom_mask, out.structures[k * out.atoms_per_structure + a]);
    return true;
}

#endif /* VMD_MD_SCRIPT_SHIM_H */
