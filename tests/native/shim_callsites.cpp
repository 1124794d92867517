// tests/native/shim_callsites.cpp - VIAMD's evaluation call sequence, re-typed against a mock of mdlib's declarations
// (tests/native/md_mock.h) and bound to libviamd_amd.so through include/vmd_md_script_shim.h: the cheapest proof that the boundary
// drops in.  Every block below cites the lines of /root/reference/src/main.cpp it re-types; names and call shapes are VIAMD's.
//
//   :966-972   eval_init: free old evals, md_script_eval_create(num_frames, eval_ir, alloc) twice (full + filt)
//   :1275-1316 init_display_properties: property_count / names / flags, md_script_eval_property_data once, pointer cached
//   :982-1008  "Eval Full": fingerprint check, clear_data, pool task calling md_script_eval_frame_range on disjoint ranges
//   :1014-1039 "Eval Filt": the same on a sub-range of the timeline with the second eval
//   :1508-1524 update_display_properties: fingerprint compare -> compute_histogram_masked(frame_mask) / downsample_histogram
//   :952-953   interrupt while tasks run; :960-964 free
//   :1300-1315 unit[2] copied and printed; :1304 vis_payload fetched per display property
//   density_volume.cpp:175-204, 263-269  md_script_vis_eval_payload(SDF): extent, one matrix + one atom bitfield per reference structure
//   :5718-5830 export_cube: the reference's own function (sliced verbatim, oracle/_ref/viamd_export_slices.inc); its file must equal
//              vmd_export_cube's byte for byte.  (tests/native/ref_callsites.cpp runs ALL of these call sites verbatim; this program keeps
//              the no-fallback build of the shim - every property bound, no mdlib behind it - covered without ImGui headers)
//
// Prints "OK ..." and exits 0 when the shimmed sequence returns, bit for bit, what direct vmd_* calls return.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "md_mock.h"
#define VMD_SHIM_NO_FALLBACK                // this program binds every property of its script: no mdlib behind the shim (shim_default_script.cpp has one)
#define VMD_SHIM_PREFIX(name) name          // emit the md_script_eval_* names themselves
#include "vmd_md_script_shim.h"
static inline bool md_script_ir_valid(const md_script_ir_t* ir) { return ir != nullptr; }      // src/main.cpp:5750 (this program's ir is an opaque token)
#define VIAMD_HOST_DOUBLE_EXPORT_ONLY
#include "viamd_host_double.h"
#include "_ref/viamd_export_slices.inc"      // export_xvg, export_csv, export_cube, sample_range - VERBATIM from /root/reference/src/main.cpp

static void fail(const char* what) {
    std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error());
    std::exit(1);
}

// ---- a host-memory trajectory behind md_trajectory_i (mdlib's loaders are not on the path under test)
struct MockTraj {
    size_t F, N;
    float L;
    std::vector<float> xyz;      // [F][3][N]
};
static bool mock_get_header(void* inst, md_trajectory_header_t* h) { MockTraj* t = (MockTraj*)inst; h->num_frames = t->F; h->num_atoms = t->N; return true; }
static bool mock_load_frame(void* inst, int64_t idx, md_trajectory_frame_header_t* h, float* x, float* y, float* z) {
    MockTraj* t = (MockTraj*)inst;
    if (idx < 0 || (size_t)idx >= t->F) return false;
    const float* f = t->xyz.data() + (size_t)idx * 3 * t->N;
    if (x) memcpy(x, f, t->N * sizeof(float));
    if (y) memcpy(y, f + t->N, t->N * sizeof(float));
    if (z) memcpy(z, f + 2 * t->N, t->N * sizeof(float));
    if (h) { h->num_atoms = t->N; h->index = idx; h->timestamp = (double)idx; h->unitcell = md_unitcell_t{t->L, t->L, t->L, 0, 0, 0, 7u}; }
    return true;
}

// ---- DisplayProperty as far as the call sites use it (src/viamd.h:300-345)
struct DisplayProperty {
    char label[64];
    md_script_property_flags_t prop_flags;
    const md_script_property_data_t* prop_data;
    const md_script_vis_payload_o* vis_payload;
    md_unit_t unit[2];
    char unit_str[2][32];
    const md_script_eval_t* eval;
    uint64_t prop_fingerprint;
    int num_bins;
    std::vector<float> bins;
};

int main(int argc, char** argv) {
    const size_t F = argc > 1 ? (size_t)std::atoi(argv[1]) : 24;
    const size_t N = 3000;
    const float L = 40.0f;
    if (vmd_device_count() <= 0) fail("no HIP device");

    // the "loaded molecule": coordinates from the library's own synthetic generator, handed over through md_trajectory_i
    MockTraj mt{F, N, L, std::vector<float>(F * 3 * N)};
    {
        vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
        if (!dt || !vmd_devtraj_synth(dt, 11, L, 0.05f, 0, 0, F)) fail("synth");
        vmd_trajectory_i* ti = vmd_devtraj_interface(dt);
        for (size_t f = 0; f < F; ++f) {
            float* p = mt.xyz.data() + f * 3 * N;
            if (!ti->load_frame(ti->inst, (int64_t)f, nullptr, p, p + N, p + 2 * N)) fail("download");
        }
        vmd_devtraj_free(dt);
    }
    md_trajectory_i traj_i{&mt, mock_get_header, mock_load_frame};
    std::vector<float> sx(N), sy(N), sz(N), mass(N, 1.0f);
    md_system_t sys{};
    sys.atom.count = N; sys.atom.x = sx.data(); sys.atom.y = sy.data(); sys.atom.z = sz.data(); sys.atom.mass = mass.data();
    sys.unitcell = md_unitcell_t{L, L, L, 0, 0, 0, 7u};
    sys.trajectory = &traj_i;

    // the compiled script: mdlib's IR is opaque; the host binds the descriptors of its properties once
    std::vector<int32_t> oxy;
    for (size_t i = 0; i < N; i += 3) oxy.push_back((int32_t)i);
    vmd_script_ir_t* vir = vmd_ir_create();
    const int32_t a = 0, b = 300;
    if (!vmd_ir_add_rdf(vir, "r", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 10.0f)) fail("add_rdf");
    if (!vmd_ir_add_distance(vir, "d", VMD_DISTANCE_COM, &a, 1, &b, 1)) fail("add_distance");
    // v = sdf(first three waters, every other water oxygen, 8): K = 3 reference structures of m = 3 atoms
    std::vector<int32_t> sdf_structs;
    for (int32_t i = 0; i < 9; ++i) sdf_structs.push_back(i);
    std::vector<int32_t> sdf_targets(oxy.begin() + 3, oxy.end());
    if (!vmd_ir_add_sdf(vir, "v", sdf_structs.data(), 3, 3, sdf_targets.data(), sdf_targets.size(), 8.0f)) fail("add_sdf");
    const md_script_ir_t* eval_ir = (const md_script_ir_t*)0x1234;      // whatever md_script_ir_compile_from_source returned
    vmd_shim_bind_ir(eval_ir, vir);
    md_allocator_i persistent{nullptr};

    // :966-972
    md_script_eval_t* full_eval = md_script_eval_create(F, eval_ir, &persistent);
    md_script_eval_t* filt_eval = md_script_eval_create(F, eval_ir, &persistent);
    if (!full_eval || !filt_eval) fail("md_script_eval_create");

    // :1275-1316 init_display_properties
    std::vector<DisplayProperty> display_properties;
    const md_script_eval_t* evals[2] = {full_eval, filt_eval};
    for (size_t eval_idx = 0; eval_idx < 2; ++eval_idx) {
        const size_t num_props = vmd_ir_property_count(vir);               // md_script_ir_property_count(ir)
        const char* const* prop_names = vmd_ir_property_names(vir);        // md_script_ir_property_names(ir)
        for (size_t i = 0; i < num_props; ++i) {
            str_t prop_name{prop_names[i], strlen(prop_names[i])};
            md_script_property_flags_t prop_flags = vmd_ir_property_flags(vir, prop_names[i]);   // md_script_ir_property_flags(ir, name)
            const md_script_property_data_t* prop_data = md_script_eval_property_data(evals[eval_idx], prop_name);
            if (!prop_data) fail("md_script_eval_property_data");
            DisplayProperty item{};
            snprintf(item.label, sizeof(item.label), "%.*s%s", (int)prop_name.len, prop_name.ptr, eval_idx ? " filt" : "");
            item.prop_flags = prop_flags; item.prop_data = prop_data; item.eval = evals[eval_idx]; item.prop_fingerprint = 0; item.num_bins = 128;
            item.unit[0] = prop_data->unit[0];                                                     // :1300-1301
            item.unit[1] = prop_data->unit[1];
            item.vis_payload = md_script_ir_property_vis_payload(eval_ir, prop_name);              // :1304
            md_unit_print(item.unit_str[0], sizeof(item.unit_str[0]), item.unit[0]);               // :1314-1315
            md_unit_print(item.unit_str[1], sizeof(item.unit_str[1]), item.unit[1]);
            if (!item.vis_payload) fail("md_script_ir_property_vis_payload");
            display_properties.push_back(item);
        }
    }

    // :982-1008 "Eval Full" - enkiTS hands contiguous sub-ranges to the pool threads
    auto pool_task = [&](md_script_eval_t* eval, uint32_t range_beg, uint32_t range_end, int nthreads) {
        std::atomic<uint32_t> next{range_beg};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const uint32_t frame_beg = next.fetch_add(4);
                    if (frame_beg >= range_end) break;
                    const uint32_t frame_end = frame_beg + 4 < range_end ? frame_beg + 4 : range_end;
                    md_trajectory_i* traj = sys.trajectory;
                    md_script_eval_frame_range(eval, eval_ir, &sys, traj, frame_beg, frame_end);
                }
            });
        for (auto& t : pool) t.join();
    };
    if (md_script_eval_ir_fingerprint(full_eval) != vmd_shim_ir_fingerprint(eval_ir)) fail("fingerprint (full)");
    md_script_eval_clear_data(full_eval);
    pool_task(full_eval, 0, (uint32_t)F, 4);

    // :1014-1039 "Eval Filt" on the timeline sub-range [beg_frame, end_frame)
    const uint32_t beg_frame = (uint32_t)(F / 4), end_frame = (uint32_t)(F - F / 4);
    if (md_script_eval_ir_fingerprint(filt_eval) != vmd_shim_ir_fingerprint(eval_ir)) fail("fingerprint (filt)");
    md_script_eval_clear_data(filt_eval);
    pool_task(filt_eval, beg_frame, end_frame, 3);

    // :1508-1524 update_display_properties
    int refreshed = 0;
    for (DisplayProperty& dp : display_properties) {
        if (dp.prop_fingerprint != dp.prop_data->fingerprint) {
            dp.prop_fingerprint = dp.prop_data->fingerprint;
            dp.bins.assign((size_t)dp.num_bins, 0.0f);
            if (dp.prop_flags & MD_SCRIPT_PROPERTY_FLAG_TEMPORAL) {
                const md_bitfield_t* mask = md_script_eval_frame_mask(dp.eval);
                std::vector<uint8_t> bytes(F, 0);
                md_bitfield_iter_t it = md_bitfield_iter_create(mask);
                while (md_bitfield_iter_next(&it)) bytes[md_bitfield_iter_idx(&it)] = 1;
                vmd_compute_histogram_masked(dp.bins.data(), dp.num_bins, dp.prop_data->min_range[0], dp.prop_data->max_range[0],
                                             dp.prop_data->values, dp.prop_data->dim[1], bytes.data(), (int)F, false);
            } else if (dp.prop_flags & MD_SCRIPT_PROPERTY_FLAG_DISTRIBUTION) {
                vmd_downsample_histogram(dp.bins.data(), dp.num_bins, dp.prop_data->values, dp.prop_data->weights, dp.prop_data->dim[2]);
            }
            ++refreshed;
        }
    }
    if (refreshed != 6) fail("every property's fingerprint must have moved");

    // ---- the same two tasks SIDE BY SIDE: VIAMD enqueues "Eval Full" and "Eval Filt" as two pool tasks (:982-1039) and nothing orders them -
    // the filtered eval walks its sub-range while the full one, its source behind the shim, is still being evaluated.  Same numbers as above.
    {
        std::vector<std::vector<float>> want;
        for (const DisplayProperty& dp : display_properties) want.emplace_back(dp.prop_data->values, dp.prop_data->values + dp.prop_data->num_values);
        std::vector<uint8_t> want_mask[2];
        for (int which = 0; which < 2; ++which) {
            const md_bitfield_t* mask = md_script_eval_frame_mask(which ? filt_eval : full_eval);
            want_mask[which].assign(F, 0);
            md_bitfield_iter_t it = md_bitfield_iter_create(mask);
            while (md_bitfield_iter_next(&it)) want_mask[which][md_bitfield_iter_idx(&it)] = 1;
        }
        md_script_eval_clear_data(full_eval);
        md_script_eval_clear_data(filt_eval);
        std::thread full_task([&] { pool_task(full_eval, 0, (uint32_t)F, 4); });
        std::thread filt_task([&] { pool_task(filt_eval, beg_frame, end_frame, 3); });
        full_task.join(); filt_task.join();
        size_t k = 0;
        for (const DisplayProperty& dp : display_properties) {
            const bool temporal = (dp.prop_flags & MD_SCRIPT_PROPERTY_FLAG_TEMPORAL) != 0;
            const std::vector<uint8_t>& m = want_mask[dp.eval == filt_eval ? 1 : 0];
            const size_t width = temporal ? (size_t)dp.prop_data->dim[1] : 0;
            for (size_t i = 0; i < want[k].size(); ++i) {
                if (temporal && !m[i / width]) continue;                         // rows of frames nobody asked for: whatever
                if (dp.prop_data->values[i] != want[k][i]) { std::fprintf(stderr, "%s (%s eval), value %zu: %g, one after the other %g\n", dp.label, dp.eval == filt_eval ? "filt" : "full", i, dp.prop_data->values[i], want[k][i]); fail("side-by-side evaluation differs"); }
            }
            ++k;
        }
        for (int which = 0; which < 2; ++which) {
            const md_bitfield_t* mask = md_script_eval_frame_mask(which ? filt_eval : full_eval);
            std::vector<uint8_t> got(F, 0);
            md_bitfield_iter_t it = md_bitfield_iter_create(mask);
            while (md_bitfield_iter_next(&it)) got[md_bitfield_iter_idx(&it)] = 1;
            if (got != want_mask[which]) fail("side-by-side evaluation: frame mask differs");
        }
    }
    // :1314-1315 what the property windows print: an rdf's x axis and a distance's y axis are lengths, an sdf has no unit
    for (const DisplayProperty& dp : display_properties) {
        const bool rdf = dp.label[0] == 'r', dist = dp.label[0] == 'd';
        if (strcmp(dp.unit_str[0], rdf ? "\xC3\x85" : "") != 0 || strcmp(dp.unit_str[1], dist ? "\xC3\x85" : "") != 0) fail("unit strings");
        if (md_unit_is_none(dp.unit[0]) == rdf) fail("unit[0]");
    }

    // ---- density_volume.cpp:134-150, 175-204, 263-269: the SDF window asks for the reference structures of the selected volume property
    const DisplayProperty* vol_dp = nullptr;
    for (const DisplayProperty& dp : display_properties) if ((dp.prop_flags & MD_SCRIPT_PROPERTY_FLAG_VOLUME) && dp.eval == full_eval) vol_dp = &dp;
    if (!vol_dp) fail("no volume property among the display properties");
    md_allocator_i frame_alloc{nullptr};
    float sdf_extent = 0.0f;
    std::vector<mat4_t> rep_model_mats;
    std::vector<std::vector<int32_t>> rep_atom_indices;
    {
        const md_script_property_data_t* prop_data = vol_dp->prop_data;
        const md_script_vis_payload_o* vis_payload = vol_dp->vis_payload;
        size_t num_reps = 0;
        bool result = false;
        md_script_vis_t vis = {};
        md_script_vis_init(&vis, &frame_alloc);
        md_script_vis_ctx_t ctx = {eval_ir, &sys, sys.trajectory};
        result = md_script_vis_eval_payload(&vis, vis_payload, 0, &ctx, MD_SCRIPT_VISUALIZE_SDF);
        if (!result) fail("md_script_vis_eval_payload(SDF)");
        if (vis.sdf.extent) {
            sdf_extent = vis.sdf.extent;
            const float voxel_spacing = 2 * sdf_extent / prop_data->dim[1];
            if (!(voxel_spacing > 0.0f)) fail("voxel spacing");
        }
        num_reps = md_array_size(vis.sdf.structures);
        if (num_reps != 1 || md_array_size(vis.sdf.matrices) != 1) fail("subidx 0 must select ONE reference structure");
        md_script_vis_free(&vis);
        // all of them (subidx -1: script_visualize_payload(state, dp.vis_payload, -1, ...), density_volume.cpp:318) + the atoms to highlight
        md_script_vis_init(&vis, &frame_alloc);
        if (!md_script_vis_eval_payload(&vis, vis_payload, -1, &ctx, MD_SCRIPT_VISUALIZE_SDF | MD_SCRIPT_VISUALIZE_ATOMS)) fail("md_script_vis_eval_payload(SDF | ATOMS)");
        num_reps = md_array_size(vis.sdf.structures);
        for (size_t i = 0; i < num_reps; ++i) {                                                    // :263-269
            rep_model_mats.push_back(vis.sdf.matrices[i]);
            size_t popcount = md_bitfield_popcount(&vis.sdf.structures[i]);
            std::vector<int32_t> idx;
            md_bitfield_iter_t it = md_bitfield_iter_create(&vis.sdf.structures[i]);
            while (md_bitfield_iter_next(&it)) idx.push_back((int32_t)md_bitfield_iter_idx(&it));
            if (idx.size() != popcount) fail("bitfield iteration");
            rep_atom_indices.push_back(idx);
        }
        if (md_bitfield_empty(&vis.atom_mask) || md_bitfield_popcount(&vis.atom_mask) != 9) fail("atom_mask of the reference structures");   // src/viamd.cpp:3205-3207
        md_script_vis_free(&vis);
    }
    if (rep_atom_indices.size() != 3 || !(sdf_extent == 8.0f)) fail("three reference structures, extent = cutoff");

    // ---- export_cube, src/main.cpp:5718-5830: the REFERENCE's own function, cut verbatim into oracle/_ref/viamd_export_slices.inc by
    // oracle/make_ref.py (this block used to be a re-typed copy) - it calls md_trajectory_load_frame, md_script_vis_init,
    // md_script_vis_eval_payload(ATOMS | SDF) through the shim, walks structure 0 with md_bitfield_scan and reads prop_data->dim / ->values
    // (one pair of files per process: the test suite, the sanitizer scripts and the GPU campaign run this program side by side)
    const std::string cube_md_s = "/tmp/viamd_shim_callsites_md." + std::to_string((long)getpid()) + ".cube";
    const std::string cube_vmd_s = "/tmp/viamd_shim_callsites_vmd." + std::to_string((long)getpid()) + ".cube";
    const char* cube_md = cube_md_s.c_str();
    const char* cube_vmd = cube_vmd_s.c_str();
    {
        static ApplicationState data;
        data.mold.sys = sys;
        data.script.eval_ir = const_cast<md_script_ir_t*>(eval_ir);
        if (!export_cube(data, vol_dp->prop_data, vol_dp->vis_payload, str_t{cube_md, strlen(cube_md)})) fail("the reference's export_cube returned false");
        host_frame_reset();
    }

    // ---- the same two evaluations through the ABI directly: the shim must not change a bit
    auto direct = [&](uint32_t fb, uint32_t fe, std::vector<float>* r_values, std::vector<float>* d_values, std::vector<uint8_t>* mask) {
        vmd_script_eval_t* e = vmd_eval_create(F, vir);
        vmd_system_t vsys = vmd_shim::wrap_system(&sys);
        vmd_trajectory_i vt = vmd_shim::wrap_trajectory(&traj_i);
        if (!e || !vmd_eval_frame_range(e, vir, &vsys, &vt, fb, fe)) fail("direct evaluation");
        const vmd_script_property_data_t* r = vmd_eval_property_data(e, "r");
        const vmd_script_property_data_t* d = vmd_eval_property_data(e, "d");
        r_values->assign(r->values, r->values + r->dim[2]);
        d_values->assign(d->values, d->values + (size_t)d->dim[0] * (size_t)d->dim[1]);
        mask->assign(vmd_eval_frame_mask(e), vmd_eval_frame_mask(e) + F);
        if (fb == 0 && fe == F) {
            // the volume, its payload and its cube file through the ABI: what the md_* sequence above must reproduce byte for byte
            const vmd_script_property_data_t* v = vmd_eval_property_data(e, "v");
            const md_script_property_data_t* sv = vol_dp->prop_data;
            if (v->num_values != sv->num_values || memcmp(v->values, sv->values, v->num_values * sizeof(float)) != 0) fail("sdf volume differs from the direct call");
            if (v->max_value != sv->max_value || !(v->max_value > 0.0f)) fail("sdf max_value");
            vmd_sdf_payload_t pl;
            if (!vmd_eval_sdf_payload(e, "v", &vsys, &vt, 0, &pl)) fail("vmd_eval_sdf_payload");
            if (pl.num_structures != rep_model_mats.size() || pl.extent != sdf_extent) fail("payload shape");
            for (size_t k = 0; k < pl.num_structures; ++k) {
                if (memcmp(&rep_model_mats[k], pl.matrices + 16 * k, 16 * sizeof(float)) != 0) fail("vis.sdf.matrices differ from vmd_eval_sdf_payload");
                std::vector<int32_t> want(pl.structures + k * pl.atoms_per_structure, pl.structures + (k + 1) * pl.atoms_per_structure);
                std::sort(want.begin(), want.end());
                if (want != rep_atom_indices[k]) fail("vis.sdf.structures differ from vmd_eval_sdf_payload");
            }
            if (!vmd_export_cube(cube_vmd, e, "v", &vsys, &vt, 0, nullptr)) fail("vmd_export_cube");
        }
        vmd_eval_free(e);
    };
    double hits[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        std::vector<float> rv, dv;
        std::vector<uint8_t> mk;
        direct(which ? beg_frame : 0, which ? end_frame : (uint32_t)F, &rv, &dv, &mk);
        const md_script_property_data_t* r = display_properties[(size_t)which * 3 + 0].prop_data;
        const md_script_property_data_t* d = display_properties[(size_t)which * 3 + 1].prop_data;
        if (r->dim[2] != (int)rv.size() || memcmp(r->values, rv.data(), rv.size() * sizeof(float)) != 0) fail("rdf values differ from the direct call");
        if (memcmp(d->values, dv.data(), dv.size() * sizeof(float)) != 0) fail("distance values differ from the direct call");
        const md_bitfield_t* mask = md_script_eval_frame_mask(evals[which]);
        for (size_t f = 0; f < F; ++f) if (md_bitfield_test_bit(mask, f) != (mk[f] != 0)) fail("frame mask differs from the direct call");
        for (float v : rv) hits[which] += v;
    }
    if (!(hits[0] > 0 && hits[1] > 0 && hits[1] < hits[0])) fail("hit counts");
    {
        auto slurp = [&](const char* path) { std::vector<char> b; FILE* f = fopen(path, "rb"); if (!f) fail(path); int c; while ((c = fgetc(f)) != EOF) b.push_back((char)c); fclose(f); return b; };
        const std::vector<char> a_md = slurp(cube_md), a_vmd = slurp(cube_vmd);
        if (a_md.size() < 1000 || a_md != a_vmd) fail("cube written through md_* names differs from vmd_export_cube");
        remove(cube_md); remove(cube_vmd);
    }

    // ---- a trajectory the host also holds in HBM (VIAMD's frame cache, src/loader.cpp:111-159, as a vmd_devtraj): bound once, the same
    // md_script_eval_frame_range calls evaluate from the device copy - no load_frame, no staging - and return the same bits
    {
        vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
        if (!dt) fail("devtraj");
        vmd_unitcell_t cell{L, L, L, 0, 0, 0, 7u};
        for (size_t f = 0; f < F; ++f) {
            const float* p = mt.xyz.data() + f * 3 * N;
            if (!vmd_devtraj_upload_frame(dt, f, &cell, p, p + N, p + 2 * N)) fail("upload_frame");
        }
        static std::atomic<long> loads{0};
        static bool (*inner)(void*, int64_t, md_trajectory_frame_header_t*, float*, float*, float*) = mock_load_frame;
        md_trajectory_i counted{&mt, mock_get_header, [](void* inst, int64_t idx, md_trajectory_frame_header_t* h, float* x, float* y, float* z) { loads += 1; return inner(inst, idx, h, x, y, z); }};
        md_system_t sys2 = sys;
        sys2.trajectory = &counted;
        vmd_shim_bind_trajectory(&counted, vmd_devtraj_interface(dt));
        md_script_eval_t* dev_eval = md_script_eval_create(F, eval_ir, &persistent);
        if (!dev_eval) fail("md_script_eval_create (bound trajectory)");
        md_script_eval_clear_data(dev_eval);
        loads = 0;
        for (uint32_t fb = 0; fb < F; fb += 4) if (!md_script_eval_frame_range(dev_eval, eval_ir, &sys2, sys2.trajectory, fb, fb + 4 < F ? fb + 4 : (uint32_t)F)) fail("frame_range (bound trajectory)");
        if (loads.load() != 0) fail("a bound trajectory must not be staged through md_trajectory_load_frame");
        for (const char* nm : {"r", "d", "v"}) {
            const md_script_property_data_t* a = md_script_eval_property_data(dev_eval, str_t{nm, 1});
            const md_script_property_data_t* b = md_script_eval_property_data(full_eval, str_t{nm, 1});
            if (a->num_values != b->num_values || memcmp(a->values, b->values, a->num_values * sizeof(float)) != 0) fail("bound trajectory: results differ from the staged evaluation");
        }
        md_script_eval_free(dev_eval);
        vmd_shim_bind_trajectory(&counted, nullptr);
        vmd_devtraj_free(dt);
    }

    // :952-953 interrupt while a task runs, then :960-964 free
    std::thread late([&] { md_script_eval_clear_data(full_eval); pool_task(full_eval, 0, (uint32_t)F, 2); });
    md_script_eval_interrupt(full_eval);
    late.join();
    md_script_eval_free(full_eval);
    md_script_eval_free(filt_eval);
    vmd_shim_bind_ir(eval_ir, nullptr);
    vmd_ir_free(vir);
    std::printf("OK frames=%zu hits_full=%.0f hits_filt=%.0f properties=%zu\n", F, hits[0], hits[1], display_properties.size());
    return 0;
}
