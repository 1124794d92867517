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
//
// Prints "OK ..." and exits 0 when the shimmed sequence returns, bit for bit, what direct vmd_* calls return.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "md_mock.h"
#define VMD_SHIM_PREFIX(name) name          // emit the md_script_eval_* names themselves
#include "vmd_md_script_shim.h"

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
    if (refreshed != 4) fail("every property's fingerprint must have moved");

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
        vmd_eval_free(e);
    };
    double hits[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        std::vector<float> rv, dv;
        std::vector<uint8_t> mk;
        direct(which ? beg_frame : 0, which ? end_frame : (uint32_t)F, &rv, &dv, &mk);
        const md_script_property_data_t* r = display_properties[(size_t)which * 2 + 0].prop_data;
        const md_script_property_data_t* d = display_properties[(size_t)which * 2 + 1].prop_data;
        if (r->dim[2] != (int)rv.size() || memcmp(r->values, rv.data(), rv.size() * sizeof(float)) != 0) fail("rdf values differ from the direct call");
        if (memcmp(d->values, dv.data(), dv.size() * sizeof(float)) != 0) fail("distance values differ from the direct call");
        const md_bitfield_t* mask = md_script_eval_frame_mask(evals[which]);
        for (size_t f = 0; f < F; ++f) if (md_bitfield_test_bit(mask, f) != (mk[f] != 0)) fail("frame mask differs from the direct call");
        for (float v : rv) hits[which] += v;
    }
    if (!(hits[0] > 0 && hits[1] > 0 && hits[1] < hits[0])) fail("hit counts");

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
