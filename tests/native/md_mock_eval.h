// tests/native/md_mock_eval.h - TEST ONLY.  A stand-in for the part of mdlib that stays BEHIND include/vmd_md_script_shim.h: its script
// IR (property names and flags, src/main.cpp:1277-1285) and its CPU evaluator under the renamed entry points the shim's fallback hooks
// call (VMD_SHIM_FALLBACK(name) = mockmd_##name here, mdlib_##name in a real build).  It "compiles" the statements of VIAMD's default
// script (/root/reference/src/main.cpp:528) well enough to know which properties a script has, and evaluates
//     angle(i, j, k) in <residues>       one value per residue and frame: the angle at atom j (degrees), indices local to the residue
//     {lin, plan, iso} = shape_weights(all)   three temporal properties (the mock's own cheap formula: sorted coordinate variances)
// on the CPU.  The hot-path properties (distance / rdf / sdf) are part of its IR as well - mdlib compiles the whole script - and it
// "evaluates" them to the constant MOCK_CPU_COPY, so that a test sees at once if the shim ever hands out the CPU copy of a bound name.
// Nothing here is mdlib's code (ext/mdlib is an empty submodule in the reference tree); it is the smallest evaluator that lets the call
// sequence of src/main.cpp run with a MIXED script.
#pragma once
#include <math.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "md_mock.h"

static const float MOCK_CPU_COPY = -777.0f;

struct MockProp {
    enum Kind { ANGLE, SHAPE, CPU_COPY } kind;
    std::string name;
    md_script_property_flags_t flags;
    int comp = 0;                                   // SHAPE: 0 lin, 1 plan, 2 iso
    int i = 0, j = 0, k = 0;                        // ANGLE: 0-based indices local to each context
    std::vector<std::vector<int32_t>> contexts;     // ANGLE: atoms of every residue of the `in` selection
    size_t width() const { return kind == ANGLE ? contexts.size() : 1; }
};
struct md_script_ir_t {
    std::vector<MockProp> props;
    std::vector<str_t> names;
    uint64_t fingerprint = 0;
};
static inline size_t md_script_ir_property_count(const md_script_ir_t* ir) { return ir->props.size(); }            // src/main.cpp:992, 1277
static inline const str_t* md_script_ir_property_names(const md_script_ir_t* ir) { return ir->names.data(); }       // :1278
static inline md_script_property_flags_t md_script_ir_property_flags(const md_script_ir_t* ir, str_t name) {        // :1285
    for (auto& p : ir->props) if (p.name.size() == name.len && memcmp(p.name.data(), name.ptr, name.len) == 0) return p.flags;
    return 0;
}
static inline bool md_script_ir_valid(const md_script_ir_t* ir) { return ir != nullptr; }                          // :936
static inline uint64_t md_script_ir_fingerprint(const md_script_ir_t* ir) { return ir ? ir->fingerprint : 0; }      // :937

// the mock's "md_script_ir_compile_from_source": statements split at ';', recognised by the function name on the right-hand side.
// residues_of(resname) -> the atoms of every residue of that name (what `in resname("ALA")` evaluates to)
template <class ResiduesOf>
static inline md_script_ir_t* mock_ir_compile(const char* source, ResiduesOf residues_of) {
    md_script_ir_t* ir = new md_script_ir_t();
    std::string text(source);
    uint64_t h = 1469598103934665603ull;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t end = text.find(';', pos);
        if (end == std::string::npos) end = text.size();
        std::string st = text.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = st.find('=');
        if (eq == std::string::npos) continue;
        auto trim = [](std::string s) { const char* ws = " \t\r\n"; const size_t a = s.find_first_not_of(ws); if (a == std::string::npos) return std::string(); return s.substr(a, s.find_last_not_of(ws) - a + 1); };
        const std::string lhs = trim(st.substr(0, eq)), rhs = trim(st.substr(eq + 1));
        for (char c : lhs + "=" + rhs) { h ^= (unsigned char)c; h *= 1099511628211ull; }
        auto starts = [&](const char* f) { return rhs.compare(0, strlen(f), f) == 0; };
        MockProp p;
        if (starts("angle(")) {
            p.kind = MockProp::ANGLE; p.name = lhs; p.flags = MD_SCRIPT_PROPERTY_FLAG_TEMPORAL;
            if (sscanf(rhs.c_str(), "angle(%d,%d,%d)", &p.i, &p.j, &p.k) != 3) { delete ir; return nullptr; }
            p.i -= 1; p.j -= 1; p.k -= 1;                                       // script indices are 1-based (src/main.cpp:2817)
            const size_t q0 = rhs.find('"'), q1 = rhs.rfind('"');
            p.contexts = residues_of(rhs.substr(q0 + 1, q1 - q0 - 1));
            ir->props.push_back(p);
        } else if (starts("shape_weights(")) {
            // {lin,plan,iso}: one temporal property per tuple member
            std::string names = lhs.substr(1, lhs.size() - 2);
            int comp = 0;
            size_t a = 0;
            while (a <= names.size()) {
                size_t b = names.find(',', a);
                if (b == std::string::npos) b = names.size();
                p = MockProp(); p.kind = MockProp::SHAPE; p.name = trim(names.substr(a, b - a)); p.flags = MD_SCRIPT_PROPERTY_FLAG_TEMPORAL; p.comp = comp++;
                ir->props.push_back(p);
                a = b + 1;
            }
        } else if (starts("distance") || starts("rdf(") || starts("sdf(")) {
            p.kind = MockProp::CPU_COPY; p.name = lhs;
            p.flags = starts("distance") ? MD_SCRIPT_PROPERTY_FLAG_TEMPORAL : starts("rdf(") ? MD_SCRIPT_PROPERTY_FLAG_DISTRIBUTION : MD_SCRIPT_PROPERTY_FLAG_VOLUME;
            ir->props.push_back(p);
        }                                                                       // anything else: a selection (s1 = resname("ALA")[2:8]) - no property
    }
    for (auto& p : ir->props) ir->names.push_back(str_t{p.name.data(), p.name.size()});
    ir->fingerprint = h;
    return ir;
}
static inline void md_script_ir_free(md_script_ir_t* ir) { delete ir; }

// ---- the evaluator behind the shim -------------------------------------------------------------------------------------------
struct vmd_shim_fallback_eval_t {
    const md_script_ir_t* ir;
    size_t num_frames;
    struct Data { md_script_property_data_t rec; std::vector<float> values; };
    std::vector<Data> data;
    std::vector<uint64_t> mask_words;
    md_bitfield_t mask;
    std::atomic<bool> interrupt{false};
    std::atomic<long> frames_evaluated{0}, interrupts{0}, clears{0};
    std::mutex mtx;
};
struct vmd_shim_fallback_payload_t { const md_script_ir_t* ir; std::string name; };
static std::atomic<long> g_mock_live_evals{0};

static inline float mock_angle(const float* x, const float* y, const float* z, int a, int b, int c) {
    const float ux = x[a] - x[b], uy = y[a] - y[b], uz = z[a] - z[b], vx = x[c] - x[b], vy = y[c] - y[b], vz = z[c] - z[b];
    const float d = (ux * vx + uy * vy + uz * vz) / sqrtf((ux * ux + uy * uy + uz * uz) * (vx * vx + vy * vy + vz * vz));
    return acosf(fmaxf(-1.0f, fminf(1.0f, d))) * 57.29577951f;
}
static inline void mock_shape(const float* x, const float* y, const float* z, size_t n, float out[3]) {
    double m[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    const float* c[3] = {x, y, z};
    for (int d = 0; d < 3; ++d) { for (size_t i = 0; i < n; ++i) m[d] += c[d][i]; m[d] /= (double)n; for (size_t i = 0; i < n; ++i) v[d] += (c[d][i] - m[d]) * (c[d][i] - m[d]); }
    std::sort(v, v + 3);
    const double s = v[0] + v[1] + v[2];
    out[0] = (float)((v[2] - v[1]) / s); out[1] = (float)(2.0 * (v[1] - v[0]) / s); out[2] = (float)(3.0 * v[0] / s);
}
// one frame of one property into `row` (width() floats): shared by the evaluator and by the test's expectation
static inline void mock_eval_row(const MockProp& p, const float* x, const float* y, const float* z, size_t n, float* row) {
    if (p.kind == MockProp::ANGLE) { for (size_t c = 0; c < p.contexts.size(); ++c) row[c] = mock_angle(x, y, z, p.contexts[c][(size_t)p.i], p.contexts[c][(size_t)p.j], p.contexts[c][(size_t)p.k]); }
    else if (p.kind == MockProp::SHAPE) { float w[3]; mock_shape(x, y, z, n, w); row[0] = w[p.comp]; }
    else row[0] = MOCK_CPU_COPY;
}

static inline vmd_shim_fallback_eval_t* mockmd_md_script_eval_create(size_t num_frames, const md_script_ir_t* ir, md_allocator_i*) {
    if (!ir) return nullptr;
    vmd_shim_fallback_eval_t* e = new vmd_shim_fallback_eval_t();
    e->ir = ir; e->num_frames = num_frames;
    e->data.resize(ir->props.size());
    for (size_t i = 0; i < ir->props.size(); ++i) {
        const MockProp& p = ir->props[i];
        auto& d = e->data[i];
        d.values.assign(num_frames * p.width(), 0.0f);
        memset(&d.rec, 0, sizeof(d.rec));
        d.rec.dim[0] = (int32_t)num_frames; d.rec.dim[1] = (int32_t)p.width();
        d.rec.values = d.values.data(); d.rec.num_values = d.values.size();
        d.rec.unit[0] = md_unit_none(); d.rec.unit[1] = md_unit_none();
    }
    e->mask_words.assign((num_frames + 63) / 64 + 1, 0);
    e->mask.bits = e->mask_words.data(); e->mask.beg_bit = 0; e->mask.end_bit = (uint32_t)num_frames;
    g_mock_live_evals += 1;
    return e;
}
static inline void mockmd_md_script_eval_free(vmd_shim_fallback_eval_t* e) { if (e) { g_mock_live_evals -= 1; delete e; } }
static inline void mockmd_md_script_eval_clear_data(vmd_shim_fallback_eval_t* e) {
    std::lock_guard<std::mutex> l(e->mtx);
    e->interrupt = false; e->clears += 1; e->frames_evaluated = 0;
    for (auto& w : e->mask_words) __atomic_store_n(&w, 0ull, __ATOMIC_RELAXED);
    for (auto& d : e->data) { std::fill(d.values.begin(), d.values.end(), 0.0f); __atomic_fetch_add(&d.rec.fingerprint, 1, __ATOMIC_RELAXED); }
}
static inline void mockmd_md_script_eval_interrupt(vmd_shim_fallback_eval_t* e) { e->interrupt = true; e->interrupts += 1; }
static inline uint64_t mockmd_md_script_eval_ir_fingerprint(const vmd_shim_fallback_eval_t* e) { return e->ir->fingerprint; }
static inline bool mockmd_md_script_eval_frame_range(vmd_shim_fallback_eval_t* e, const md_script_ir_t* ir, const md_system_t* sys, md_trajectory_i* traj,
                                                     uint32_t frame_beg, uint32_t frame_end) {
    if (ir != e->ir) return false;
    const size_t n = sys->atom.count;
    std::vector<float> x(n), y(n), z(n);                      // per-thread coordinate buffers: sys->atom.x/y/z belong to the display
    for (uint32_t f = frame_beg; f < frame_end; ++f) {
        if (e->interrupt) return false;
        if (!md_trajectory_load_frame(traj, f, nullptr, x.data(), y.data(), z.data())) return false;
        for (size_t i = 0; i < e->ir->props.size(); ++i) {
            const MockProp& p = e->ir->props[i];
            mock_eval_row(p, x.data(), y.data(), z.data(), n, &e->data[i].values[(size_t)f * p.width()]);
        }
        std::lock_guard<std::mutex> l(e->mtx);
        __atomic_fetch_or(&e->mask_words[f >> 6], 1ull << (f & 63), __ATOMIC_RELAXED);      // read by the GUI thread meanwhile (md_bitfield_test_bit)
        for (auto& d : e->data) __atomic_fetch_add(&d.rec.fingerprint, 1, __ATOMIC_RELAXED);
        e->frames_evaluated += 1;
    }
    return true;
}
static inline const md_script_property_data_t* mockmd_md_script_eval_property_data(const vmd_shim_fallback_eval_t* e, str_t name) {
    for (size_t i = 0; i < e->ir->props.size(); ++i)
        if (e->ir->props[i].name.size() == name.len && memcmp(e->ir->props[i].name.data(), name.ptr, name.len) == 0) return &e->data[i].rec;
    return nullptr;                                            // `s1` is a selection, not a property: mdlib has no record for it either
}
static inline const md_bitfield_t* mockmd_md_script_eval_frame_mask(const vmd_shim_fallback_eval_t* e) { return &e->mask; }
static inline const vmd_shim_fallback_payload_t* mockmd_md_script_ir_property_vis_payload(const md_script_ir_t* ir, str_t name) {
    static std::mutex mtx;
    static std::vector<vmd_shim_fallback_payload_t*> all;
    std::lock_guard<std::mutex> l(mtx);
    for (auto* p : all) if (p->ir == ir && p->name.size() == name.len && memcmp(p->name.data(), name.ptr, name.len) == 0) return p;
    if (!md_script_ir_property_flags(ir, name)) return nullptr;
    all.push_back(new vmd_shim_fallback_payload_t{ir, std::string(name.ptr, name.len)});
    return all.back();
}
// highlights the atoms an angle is measured on (MD_SCRIPT_VISUALIZE_ATOMS), nothing for the other kinds
static inline bool mockmd_md_script_vis_eval_payload(md_script_vis_t* vis, const vmd_shim_fallback_payload_t* payload, int subidx, const md_script_vis_ctx_t*, md_script_vis_flags_t flags) {
    for (auto& p : payload->ir->props) {
        if (p.name != payload->name) continue;
        if (p.kind != MockProp::ANGLE) return true;
        if (flags & MD_SCRIPT_VISUALIZE_ATOMS)
            for (size_t c = 0; c < p.contexts.size(); ++c) {
                if (subidx >= 0 && (size_t)subidx != c) continue;
                for (int a : {p.i, p.j, p.k}) md_bitfield_set_bit(&vis->atom_mask, (uint64_t)p.contexts[c][(size_t)a]);
            }
        return true;
    }
    return false;
}
