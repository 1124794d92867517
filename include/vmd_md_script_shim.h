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
 * With VMD_SHIM_PREFIX undefined the functions are emitted under those very names (a VIAMD build that drops mdlib's
 * md_script_eval.c from the link); define VMD_SHIM_PREFIX(name) to put them elsewhere (the compile test uses vmdshim_##name
 * next to a mock of mdlib's declarations).
 *
 * Hooks (define before including; the defaults name mdlib's own functions): VMD_SHIM_BONDS(sys, vsys) hands md_system_t::bond over;
 * VMD_SHIM_UNIT(dst, str) turns the backend's printed unit ("\xC3\x85" or "") into an md_unit_t (default: md_unit_angstrom() /
 * md_unit_none()); VMD_SHIM_BITFIELD_INIT / _SET build the md_bitfield_t of a reference structure (default: md_bitfield_init /
 * md_bitfield_set_bit); md_array_resize / md_array_size are mdlib's stretchy-buffer macros.
 *
 * What the shim needs from the host besides the types: the property DESCRIPTORS.  mdlib's IR is opaque, so the host registers,
 * once per compiled script, the vmd_script_ir_t that carries its rdf / sdf / distance properties (INTEGRATION.md section 3 shows how
 * VIAMD derives them from the evaluated argument bitfields):      vmd_shim_bind_ir(md_ir, vmd_ir);
 *
 * tests/native/shim_callsites.cpp re-types VIAMD's call sequence against tests/native/md_mock.h and runs it. */
#ifndef VMD_MD_SCRIPT_SHIM_H
#define VMD_MD_SCRIPT_SHIM_H

#include <string.h>

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

/* ---- IR registry: md_script_ir_t* -> the descriptors of its properties ------------------------------------------------------ */
namespace vmd_shim {
struct PayloadEval { vmd_script_eval_t* eval = nullptr; size_t num_frames = 0; };
struct Registry {
    std::mutex mtx;
    std::map<const void*, const vmd_script_ir_t*> ir;
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
        h->num_atoms = (size_t)hdr.num_atoms; h->index = (int64_t)hdr.index; h->timestamp = (double)hdr.timestamp;
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

/* the host's one extra call: which descriptors belong to this compiled script (NULL unbinds; call before md_script_eval_create) */
inline void vmd_shim_bind_ir(const md_script_ir_t* md_ir, const vmd_script_ir_t* vmd_ir) {
    vmd_shim::Registry& r = vmd_shim::registry();
    std::lock_guard<std::mutex> l(r.mtx);
    if (vmd_ir) { r.ir[md_ir] = vmd_ir; return; }
    r.ir.erase(md_ir);
    auto pe = r.payload_evals.find(md_ir);
    if (pe != r.payload_evals.end()) { vmd_eval_free(pe->second.eval); r.payload_evals.erase(pe); }
    for (auto it = r.payloads.begin(); it != r.payloads.end();) it = it->first.first == md_ir ? r.payloads.erase(it) : std::next(it);
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
    vmd_script_eval_t* eval = nullptr;
    const vmd_script_ir_t* vir = nullptr;
    const md_script_ir_t* md_ir = nullptr;
    /* md_script_property_data_t records handed to VIAMD: fetched once and cached by the GUI (src/main.cpp:1286,1303), so their
     * addresses are stable for the eval's lifetime; the arrays they point at are the backend's own (equally stable), the scalar
     * fields (fingerprint, ranges, max_value) are refreshed from the backend whenever data may have changed */
    struct Prop { std::string name; const vmd_script_property_data_t* src; md_script_property_data_t dst; md_script_aggregate_t agg; };
    std::vector<std::unique_ptr<Prop>> props;
    std::vector<uint64_t> mask_words;
    md_bitfield_t mask;
    std::mutex mtx;

    /* The backend changes the scalar fields while other pool threads are inside frame_range, and VIAMD's GUI thread reads the records
     * below at any time (src/main.cpp:1508-1524: old and new fields side by side are tolerated).  Both directions go through relaxed
     * atomic loads / stores - plain moves on x86-64 - so the hand-over is defined behaviour, and ThreadSanitizer-clean. */
    template <class T> static T peek(const T& v) { T r; __atomic_load(const_cast<T*>(&v), &r, __ATOMIC_RELAXED); return r; }
    template <class T, class U> static void pub(T& d, U v) { T t = (T)v; __atomic_store(&d, &t, __ATOMIC_RELAXED); }
    void refresh() {
        for (auto& p : props) {
            const vmd_script_property_data_t* s = p->src;
            md_script_property_data_t& d = p->dst;
            for (int k = 0; k < 4; ++k) pub(d.dim[k], s->dim[k]);
            pub(d.values, s->values); pub(d.weights, s->weights); pub(d.num_values, s->num_values);
            pub(d.min_value, peek(s->min_value)); pub(d.max_value, peek(s->max_value));
            for (int k = 0; k < 2; ++k) { pub(d.min_range[k], peek(s->min_range[k])); pub(d.max_range[k], peek(s->max_range[k])); }
            if (s->aggregate) {
                pub(p->agg.num_values, s->aggregate->num_values);
                pub(p->agg.population_mean, s->aggregate->population_mean);
                pub(p->agg.population_var, s->aggregate->population_var);
                pub(p->agg.population_ext, (decltype(p->agg.population_ext))s->aggregate->population_ext);
                pub(d.aggregate, &p->agg);
            } else {
                pub(d.aggregate, (decltype(d.aggregate))nullptr);
            }
            pub(d.fingerprint, peek(s->fingerprint));      /* last: the GUI compares it to decide whether to re-read (src/main.cpp:1508-1509) */
        }
    }
};

inline md_script_eval_t* VMD_SHIM_PREFIX(md_script_eval_create)(size_t num_frames, const md_script_ir_t* ir, md_allocator_i* alloc) {
    (void)alloc;                                 /* host allocations are the library's own */
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    if (!vir) return nullptr;
    std::unique_ptr<md_script_eval_t> e(new md_script_eval_t());
    e->vir = vir;
    e->eval = vmd_eval_create(num_frames, vir);
    if (!e->eval) return nullptr;
    const size_t n = vmd_ir_property_count(vir);
    for (size_t i = 0; i < n; ++i) {
        std::unique_ptr<md_script_eval_t::Prop> p(new md_script_eval_t::Prop());
        p->name = vmd_ir_property_names(vir)[i];
        p->src = vmd_eval_property_data(e->eval, p->name.c_str());
        memset(&p->dst, 0, sizeof(p->dst));
        memset(&p->agg, 0, sizeof(p->agg));
        /* unit[2] (src/main.cpp:1300-1301, printed at :1314-1315): the backend carries the printed form, VMD_SHIM_UNIT makes the
         * md_unit_t of it */
        VMD_SHIM_UNIT(p->dst.unit[0], p->src->unit_str[0]);
        VMD_SHIM_UNIT(p->dst.unit[1], p->src->unit_str[1]);
        e->props.push_back(std::move(p));
    }
    e->mask_words.assign((num_frames + 63) / 64 + 1, 0);
    memset(&e->mask, 0, sizeof(e->mask));
    e->mask.bits = e->mask_words.data();
    e->mask.beg_bit = 0;
    e->mask.end_bit = (uint32_t)num_frames;
    e->refresh();
    {
        vmd_shim::Registry& r = vmd_shim::registry();
        std::lock_guard<std::mutex> l(r.mtx);
        std::vector<vmd_script_eval_t*>& live = r.evals[ir];
        if (!live.empty() && vmd_eval_num_frames(live.front()) == num_frames) vmd_eval_set_source(e->eval, live.front());
        live.push_back(e->eval);
        e->md_ir = ir;
    }
    return e.release();
}
inline void VMD_SHIM_PREFIX(md_script_eval_free)(md_script_eval_t* e) {
    if (!e) return;
    {
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
    vmd_eval_free(e->eval);
    delete e;
}
inline void VMD_SHIM_PREFIX(md_script_eval_clear_data)(md_script_eval_t* e) {
    if (!e) return;
    vmd_eval_clear_data(e->eval);
    std::lock_guard<std::mutex> l(e->mtx);
    e->refresh();
}
inline void VMD_SHIM_PREFIX(md_script_eval_interrupt)(md_script_eval_t* e) { if (e) vmd_eval_interrupt(e->eval); }
inline uint64_t VMD_SHIM_PREFIX(md_script_eval_ir_fingerprint)(const md_script_eval_t* e) { return e ? vmd_eval_ir_fingerprint(e->eval) : 0; }
/* md_script_ir_fingerprint of the bound IR: what src/main.cpp:987 compares the eval's fingerprint with */
inline uint64_t vmd_shim_ir_fingerprint(const md_script_ir_t* ir) {
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    return vir ? vmd_ir_fingerprint(vir) : 0;
}

/* the hot call: pool threads, disjoint ranges, one eval (src/main.cpp:993-997) */
inline bool VMD_SHIM_PREFIX(md_script_eval_frame_range)(md_script_eval_t* e, const md_script_ir_t* ir, const md_system_t* sys,
                                                        md_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    if (!e || !sys || !traj) return false;
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    if (vir != e->vir) return false;             /* mdlib compares fingerprints the same way */
    const vmd_system_t vsys = vmd_shim::wrap_system(sys);
    vmd_trajectory_i vtraj = vmd_shim::wrap_trajectory(traj);
    const bool ok = vmd_eval_frame_range(e->eval, vir, &vsys, &vtraj, frame_beg, frame_end);
    std::lock_guard<std::mutex> l(e->mtx);
    e->refresh();
    return ok;
}

inline const md_script_property_data_t* VMD_SHIM_PREFIX(md_script_eval_property_data)(const md_script_eval_t* e, str_t name) {
    if (!e) return nullptr;
    for (auto& p : e->props)
        if (p->name.size() == (size_t)name.len && memcmp(p->name.data(), name.ptr, (size_t)name.len) == 0) return &p->dst;
    return nullptr;
}

/* frames evaluated so far as the bitfield VIAMD iterates (src/main.cpp:194-210, 1513); refreshed by every call */
inline const md_bitfield_t* VMD_SHIM_PREFIX(md_script_eval_frame_mask)(const md_script_eval_t* ce) {
    md_script_eval_t* e = const_cast<md_script_eval_t*>(ce);
    if (!e) return nullptr;
    std::lock_guard<std::mutex> l(e->mtx);
    vmd_eval_frame_mask_bits(e->eval, e->mask_words.data(), e->mask_words.size());
    return &e->mask;
}

/* ---- vis payloads ----------------------------------------------------------------------------------------------------------- */
/* md_script_ir_property_vis_payload(ir, name), src/main.cpp:1304: NULL for an unknown ir / property */
inline const md_script_vis_payload_o* VMD_SHIM_PREFIX(md_script_ir_property_vis_payload)(const md_script_ir_t* ir, str_t name) {
    const vmd_script_ir_t* vir = vmd_shim::find_ir(ir);
    if (!vir) return nullptr;
    const std::string nm(name.ptr, (size_t)name.len);
    bool known = false;
    for (size_t i = 0; i < vmd_ir_property_count(vir); ++i) known = known || nm == vmd_ir_property_names(vir)[i];
    if (!known) return nullptr;
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
    if (!vis || !payload || !ctx || !ctx->mol || !ctx->traj) return false;
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
        md_array_resize(vis->sdf.matrices, k1 - k0, vis->alloc);
        md_array_resize(vis->sdf.structures, k1 - k0, vis->alloc);
        for (size_t k = k0; k < k1; ++k) {
            memcpy(&vis->sdf.matrices[k - k0], out.matrices + 16 * k, 16 * sizeof(float));        /* mat4_t: 16 floats, column-major */
            md_bitfield_t* bf = &vis->sdf.structures[k - k0];
            VMD_SHIM_BITFIELD_INIT(bf, vis->alloc);
            for (size_t a = 0; a < out.atoms_per_structure; ++a) VMD_SHIM_BITFIELD_SET(bf, out.structures[k * out.atoms_per_structure + a]);
        }
    }
    if (flags & MD_SCRIPT_VISUALIZE_ATOMS)
        for (size_t k = k0; k < k1; ++k)
            for (size_t a = 0; a < out.atoms_per_structure; ++a) VMD_SHIM_BITFIELD_SET(&vis->atom_mask, out.structures[k * out.atoms_per_structure + a]);
    return true;
}

#endif /* VMD_MD_SCRIPT_SHIM_H */
