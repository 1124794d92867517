// viamd_amd/csrc/vmd_eval_ir.cpp - the property descriptors behind vmd_ir_*: what md_script_ir_t carries for the hot-path properties
// (rdf / sdf / distance family; /root/reference/src/main.cpp:528, 2817-2858), their fingerprint and the work estimate a host compares
// with its threshold (include/vmd_md_script_shim.h).
#include "vmd_eval_internal.h"

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}

extern "C" vmd_script_ir_t* vmd_ir_create(void) { return new vmd_script_ir_t(); }

// atom pairs one frame of this script asks for (rdf: |ref| x |target|; sdf: K x |target| + K m for the alignment; distance: |a| x |b| of
// every context): what a host compares with its threshold before it sends a SMALL script to the GPU at all (include/vmd_md_script_shim.h,
// vmd_shim_set_min_work; VIAMD's default dataset is ~1e2 atoms, src/main.cpp:522-528)
extern "C" uint64_t vmd_ir_work_per_frame(const vmd_script_ir_t* ir) {
    if (!ir) return 0;
    uint64_t w = 0;
    for (const Property& p : ir->props) {
        if (p.kind == PROP_RDF) w += (uint64_t)p.a.size() * (uint64_t)p.b.size();
        else if (p.kind == PROP_SDF) w += (uint64_t)p.K * ((uint64_t)p.b.size() + (uint64_t)p.m);
        else if (p.aoff.size() > 1) { for (size_t c = 0; c + 1 < p.aoff.size(); ++c) w += (uint64_t)(p.aoff[c + 1] - p.aoff[c])
                * (uint64_t)(p.boff[c + 1] - p.boff[c]); }
        else w += (uint64_t)p.a.size() * (uint64_t)p.b.size();
    }
    return w;
}

extern "C" void vmd_ir_free(vmd_script_ir_t* ir) { delete ir; }

bool ir_name_ok(vmd_script_ir_t* ir, const char* name) {
    if (!ir) return vmd_fail("ir is NULL");
    if (!name || !*name) return vmd_fail("property name is empty");
    for (auto& p : ir->props) if (p.name == name) return vmd_fail("property '%s' already defined", name);
    return true;
}

bool idx_ok(const int32_t* idx, size_t n, const char* what) {
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
    if (!ir_name_ok(ir, name) || !idx_ok(structures, K * m, "sdf reference structures") || !idx_ok(target, ntarget,
            "sdf target set")) return false;
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
        if (a_offsets[c + 1] <= a_offsets[c] || b_offsets[c + 1] <= b_offsets[c]) return vmd_fail("distance context %zu has an empty set",
                c);
        if (kind == VMD_DISTANCE_PAIR && ((a_offsets[c + 1] - a_offsets[c]) != a_offsets[1] || (b_offsets[c + 1]
                - b_offsets[c]) != b_offsets[1]))
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
