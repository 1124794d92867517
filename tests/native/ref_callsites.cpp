// tests/native/ref_callsites.cpp - VIAMD's OWN evaluation call sites, VERBATIM, as a host of the drop-in boundary.
//
// Nothing of VIAMD is re-typed here.  oracle/make_ref.py cuts these out of /root/reference/src where they lie (oracle/_ref/*.inc,
// generated, git-ignored) and this file includes them:
//     struct DisplayProperty                       src/viamd.h:272-370
//     namespace task_system (declarations)         src/task_system.h:9-64
//     compute_histogram[_masked], downsample_histogram, free_histogram      src/main.cpp:132-261
//     display_property_copy_param_from_old, init_display_properties         src/main.cpp:1237-1500
//     update_display_properties                                             src/main.cpp:1502-1529
//     export_xvg, export_csv, export_cube, sample_range                     src/main.cpp:5640-5841
//     the evaluation block of the main loop                                 src/main.cpp:950-1040
// Around them: tests/native/md_mock.h + md_mock_eval.h (a test double of mdlib: declarations, a script "compiler", a CPU evaluator that
// stays BEHIND the shim), include/vmd_md_script_shim.h (the boundary under test, emitting the md_script_eval_* names those slices
// call) and tests/native/viamd_host_double.h (files, logging, allocators, the touched fields of ApplicationState, a thread pool).
// The script is the literal of src/main.cpp:528.  The program drives VIAMD's main-loop block until both pool tasks are done, lets
// update_display_properties build the display histograms, runs the three exporters, and compares with direct vmd_* calls:
//   * every DisplayProperty the reference builds reads the shim's records: temporal rows, rdf values + weights, the sdf volume - bit for bit
//   * dp.hist of `r` = vmd_downsample_histogram, dp.hist of `d1` = vmd_compute_histogram_masked over the frame mask - bit for bit
//   * export_cube's file  == vmd_export_cube's, byte for byte (atoms with atomic numbers, matrices through md_script_vis_eval_payload)
//   * export_csv / export_xvg of the tables draw_property_export_window assembles (src/main.cpp:5953-6020, the assembly is ~10 lines
//     re-stated below since the window itself is ImGui code) == vmd_export_property_table, byte for byte (xvg: but the time stamp)
//   * a script edit (eval_init again with a new ir) walks the block's free -> create path; interrupting a running task is forwarded
// Prints "OK ..." and exits 0.  Needs /root/reference to COMPILE (ImGui headers + slices): built by oracle/make_ref.py in this
// container, runs prebuilt on the GPU box.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <imgui.h>        // /root/reference/ext/imgui   - types only (ImVec4, ImU32)
#include <implot.h>       // /root/reference/ext/implot  - types only (ImPlotPoint, ImPlotGetter, ImPlotColormap, ImPlotMarker)

#include "md_mock.h"
#include "md_mock_eval.h"
#define VMD_SHIM_FALLBACK(name) mockmd_##name
#define VMD_SHIM_FALLBACK_DECLARED
#define VMD_SHIM_PREFIX(name) name
#include "vmd_md_script_shim.h"
#include "viamd_host_double.h"

// ======================================================================= the reference's code
#include "_ref/viamd_callsite_slices.inc"
#include "_ref/viamd_export_slices.inc"
static void viamd_main_loop_evaluation_block(ApplicationState& state, const size_t num_frames) {
#include "_ref/viamd_eval_block.inc"
}
// ===========================================================================================

// how long the deferred-settle helper may take before the run counts as hung: seconds on the product and the plain emulator build, far longer
// under ThreadSanitizer (the emulator is ~20 x slower there and scripts/tsan_emu.sh runs ten programs side by side)
#if defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)
static const std::chrono::seconds kSettleLimit(1800);
#else
static const std::chrono::seconds kSettleLimit(60);
#endif

static void fail(const char* what) {
    std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error());
    std::exit(1);
}

struct MockTraj { size_t F, N; float L; std::vector<float> xyz; };
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

// the literal of /root/reference/src/main.cpp:528
static const char* kDefaultScript =
    "s1 = resname(\"ALA\")[2:8];\nd1 = distance(10,30);\na1 = angle(2,1,3) in resname(\"ALA\");\nr = rdf(element('C'), element('H'), 10.0);\nv = sdf(s1, element('H'), 10.0);\n{lin,plan,iso} = shape_weights(all);";

static std::vector<char> slurp(const std::string& path) {
    std::vector<char> b;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "%s\n", path.c_str()); fail("cannot read a file that was just exported"); }
    int c;
    while ((c = fgetc(f)) != EOF) b.push_back((char)c);
    fclose(f);
    return b;
}
// an XVG file without its first line ("# This file was created <asctime>")
static std::vector<char> but_first_line(const std::vector<char>& b) {
    auto nl = std::find(b.begin(), b.end(), '\n');
    if (nl == b.end() || std::string(b.begin(), b.begin() + 24) != "# This file was created ") fail("xvg: the first line is the creation stamp");
    return std::vector<char>(nl + 1, b.end());
}
static bool same_floats(const float* a, const float* b, size_t n) { return memcmp(a, b, n * sizeof(float)) == 0; }

int main(int argc, char** argv) {
    const size_t F = argc > 1 ? (size_t)std::atoi(argv[1]) : 16;
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
    const size_t n_res = 20, n_blob = n_res * 10, N = n_blob + 933 * 3;
    const float L = 40.0f;
    if (vmd_device_count() <= 0) fail("no HIP device");
    vmd_shim_set_min_work(0);                      // a test-sized system: below the default work threshold, send what is bound to the GPU anyway

    MockTraj mt{F, N, L, std::vector<float>(F * 3 * N)};
    {
        vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
        if (!dt || !vmd_devtraj_synth(dt, 21, L, 0.05f, 0, 0, F)) fail("synth");
        vmd_trajectory_i* ti = vmd_devtraj_interface(dt);
        for (size_t f = 0; f < F; ++f) { float* p = mt.xyz.data() + f * 3 * N; if (!ti->load_frame(ti->inst, (int64_t)f, nullptr, p, p + N, p + 2 * N)) fail("download"); }
        vmd_devtraj_free(dt);
    }
    md_trajectory_i traj_i{&mt, mock_get_header, mock_load_frame};
    std::vector<float> sx(N), sy(N), sz(N), mass(N, 1.0f);

    // the molecule's topology: 20 ALA residues of 10 atoms (N C C O C H H H C H), then waters
    static const char* ala[10] = {"N", "C", "C", "O", "C", "H", "H", "H", "C", "H"};
    std::vector<const char*> elements(N), resnames(N);
    std::vector<int32_t> residue_index(N);
    std::vector<uint8_t> atomic_numbers(N);
    for (size_t i = 0; i < N; ++i) {
        if (i < n_blob) { elements[i] = ala[i % 10]; resnames[i] = "ALA"; residue_index[i] = (int32_t)(i / 10); }
        else { const size_t w = i - n_blob; elements[i] = w % 3 == 0 ? "O" : "H"; resnames[i] = "HOH"; residue_index[i] = (int32_t)(n_res + w / 3); }
        atomic_numbers[i] = elements[i][0] == 'H' ? 1 : elements[i][0] == 'C' ? 6 : elements[i][0] == 'N' ? 7 : 8;
    }
    host_atomic_numbers = atomic_numbers.data();
    vmd_topology_t topo{N, elements.data(), nullptr, resnames.data(), residue_index.data(), nullptr};
    auto residues_of = [&](const std::string& resname) {
        std::vector<std::vector<int32_t>> out;
        for (size_t i = 0; i < N; ++i) {
            if (resname != resnames[i]) continue;
            if (out.empty() || residue_index[(size_t)out.back().back()] != residue_index[i]) out.emplace_back();
            out.back().push_back((int32_t)i);
        }
        return out;
    };

    // ---- the application state the slices work on
    static ApplicationState state;                 // the pool lambdas of the block capture it by reference: it outlives every task
    state.allocator.frame = frame_alloc;
    state.allocator.persistent = persistent_alloc;
    state.mold.sys.atom.count = N; state.mold.sys.atom.x = sx.data(); state.mold.sys.atom.y = sy.data(); state.mold.sys.atom.z = sz.data();
    state.mold.sys.atom.mass = mass.data();
    state.mold.sys.unitcell = md_unitcell_t{L, L, L, 0, 0, 0, 7u};
    state.mold.sys.trajectory = &traj_i;
    md_array_resize(state.timeline.x_values, F, persistent_alloc);
    for (size_t f = 0; f < F; ++f) state.timeline.x_values[f] = (float)f;
    state.timeline.filter.enabled = true;
    state.timeline.filter.beg_frame = (double)(F / 4);
    state.timeline.filter.end_frame = (double)(F - F / 4 - 1);          // the block evaluates [beg, end + 1)
    const uint32_t filt_beg = (uint32_t)(F / 4), filt_end = (uint32_t)(F - F / 4);
    task_system::initialize(4);

    // ---- "md_script_ir_compile_from_source" (src/main.cpp:878): mdlib compiles the whole script; the backend takes d1, r, v
    auto compile_and_bind = [&](const char* text, vmd_script_ir_t** vir_out) {
        md_script_ir_t* ir = mock_ir_compile(text, residues_of);
        if (!ir) fail("mock mdlib: compile");
        vmd_script_ir_t* vir = vmd_ir_create();
        vmd_script_report_t* report = nullptr;
        if (!vmd_ir_compile_from_source_partial(vir, text, &topo, &report)) fail("vmd_ir_compile_from_source_partial");
        vmd_script_report_free(report);
        vmd_shim_bind_ir(ir, vir);
        *vir_out = vir;
        return ir;
    };
    vmd_script_ir_t* vir = nullptr;
    state.script.ir = compile_and_bind(kDefaultScript, &vir);
    if (md_script_ir_property_count(state.script.ir) != 7 || vmd_ir_property_count(vir) != 3) fail("seven properties, three of them on the GPU");
    state.script.eval_init = true;                                      // what src/main.cpp:928 sets after a successful compile

    // ---- VIAMD's main loop, as far as evaluation goes: the block, then update_display_properties (src/main.cpp:1064), once per GUI frame
    auto gui_frames_until_idle = [&](const char* what) {
        const auto t_begin = std::chrono::steady_clock::now();
        for (int frame = 0; std::chrono::steady_clock::now() - t_begin < std::chrono::seconds(1800); ++frame) {      // (ThreadSanitizer builds are 20 x slower)
            viamd_main_loop_evaluation_block(state, F);
            update_display_properties(&state);
            host_frame_reset();
            const bool busy = task_system::task_is_running(state.tasks.evaluate_full) || task_system::task_is_running(state.tasks.evaluate_filt) ||
                              state.script.eval_init || state.script.evaluate_full || state.script.evaluate_filt;
            if (!busy) return frame + 1;
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        fail(what);
        return 0;
    };
    const int gui_frames = gui_frames_until_idle("the evaluation tasks never finished");
#ifdef VMD_SHIM_DEFERRED_SETTLE
    // Results trail the last call by the quiet period, and nobody calls into the shim any more: VIAMD just keeps drawing frames, and
    // update_display_properties looks at prop_data->fingerprint in each (the records must follow the helper's settle by
    // themselves, or the GUI keeps a pre-settle histogram for good).  So: GUI frames until the backend has every requested frame.
    {
        const auto t0 = std::chrono::steady_clock::now();
        while (vmd_eval_frames_done(state.script.full_eval->eval) != F || vmd_eval_frames_done(state.script.filt_eval->eval) != F - 2 * (F / 4)) {
            update_display_properties(&state);
            host_frame_reset();
            if (std::chrono::steady_clock::now() - t0 > kSettleLimit) fail("deferred settle: the helper thread never settled");
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        // the helper may still be inside its callback into the shim (the fingerprints move once more there): wait for it, then compare
        if (!vmd_eval_wait_settled(state.script.full_eval->eval) || !vmd_eval_wait_settled(state.script.filt_eval->eval)) fail("vmd_eval_wait_settled");
    }
#endif
    update_display_properties(&state);             // one more GUI frame: fingerprints moved with the last frames
    if (!state.script.full_eval || !state.script.filt_eval || state.script.eval_ir != state.script.ir) fail("the block must have created both evals from the ir");
    if (task_system::host_task(state.tasks.evaluate_full)->calls.load() < 1 || task_system::host_task(state.tasks.evaluate_filt)->calls.load() < 1) fail("both pool tasks ran");

    // ---- the same script through the ABI directly
    vmd_system_t vsys = vmd_shim::wrap_system(&state.mold.sys);
    vmd_trajectory_i vt = vmd_shim::wrap_trajectory(&traj_i);
    vmd_script_eval_t* direct[2] = {vmd_eval_create(F, vir), vmd_eval_create(F, vir)};
    if (!direct[0] || !direct[1] || !vmd_eval_frame_range(direct[0], vir, &vsys, &vt, 0, (uint32_t)F) || !vmd_eval_frame_range(direct[1], vir, &vsys, &vt, filt_beg, filt_end))
        fail("direct evaluation");
    if (!vmd_eval_wait_settled(direct[0]) || !vmd_eval_wait_settled(direct[1])) fail("direct evaluation: settle");

    // ---- what init_display_properties built (src/main.cpp:1259-1500) and what update_display_properties made of it (:1502-1529)
    const size_t num_dp = md_array_size(state.display_properties);
    // per eval: d1 -> dist + temporal; a1 (20 residues) -> dist + agg + temporal [+ mean / var / ext when the record has an aggregate];
    // r -> dist; v -> volume; lin, plan, iso -> dist + temporal each.  The filtered eval gets no temporal items (:1369)
    size_t n_type[3] = {0, 0, 0}, checked_hist = 0, checked_rows = 0;
    const DisplayProperty *dp_v = nullptr, *dp_r = nullptr, *dp_d1 = nullptr, *dp_d1_dist = nullptr;
    for (size_t i = 0; i < num_dp; ++i) {
        const DisplayProperty& dp = state.display_properties[i];
        n_type[dp.type] += 1;
        const int which = dp.partial_evaluation ? 1 : 0;
        if (dp.eval != (which ? state.script.filt_eval : state.script.full_eval)) fail("DisplayProperty::eval");
        std::string name(dp.label);
        name = name.substr(0, name.find(' '));
        const vmd_script_property_data_t* want = vmd_eval_property_data(direct[which], name.c_str());      // NULL for a1 / lin / plan / iso
        if (want) {
            if (dp.prop_data->num_values != want->num_values) fail("num_values of a bound property");
            if (dp.type != DisplayProperty::Type_Temporal || which == 0) {
                // the full eval's arrays whole; of the filtered eval the distribution / volume (a temporal's rows outside the range are unspecified)
                if (!(want->weights || want->dim[3]) && which == 1) {
                    const size_t w = (size_t)want->dim[1];
                    if (!same_floats(dp.prop_data->values + filt_beg * w, want->values + filt_beg * w, (filt_end - filt_beg) * w)) fail("filtered temporal rows differ from the direct call");
                } else if (!same_floats(dp.prop_data->values, want->values, want->num_values)) {
                    std::fprintf(stderr, "%s\n", dp.label); fail("values read through VIAMD's DisplayProperty differ from the direct call");
                }
                checked_rows += 1;
            }
            if (want->weights && !same_floats(dp.prop_data->weights, want->weights, (size_t)want->dim[2])) fail("rdf weights");
            if (dp.prop_data->fingerprint != dp.prop_fingerprint && dp.type == DisplayProperty::Type_Distribution) fail("update_display_properties must have caught up with the fingerprint");
        }
        if (dp.type == DisplayProperty::Type_Distribution) {
            if (dp.hist.num_bins != dp.num_bins || !dp.hist.bins) fail("update_display_properties did not build a display histogram");
            if (want && want->weights) {
                // :1516-1524 downsample_histogram(hist.bins, num_bins, values, weights, dim[2]) == the product's, on the product's arrays
                std::vector<float> g((size_t)dp.num_bins);
                vmd_downsample_histogram(g.data(), dp.num_bins, want->values, want->weights, want->dim[2]);
                if (!same_floats(g.data(), dp.hist.bins, g.size())) fail("display histogram of the rdf: reference downsample_histogram != vmd_downsample_histogram");
                if (dp.hist.x_min != (double)want->min_range[0] || dp.hist.x_max != (double)want->max_range[0]) fail("hist x range of the rdf");
                checked_hist += 1;
                if (!which) dp_r = &dp;
            } else if (want && !dp.aggregate_histogram) {
                // :1512-1513 compute_histogram_masked(..., md_script_eval_frame_mask(dp.eval), aggregate) over the mask the shim hands out
                std::vector<uint8_t> mask(vmd_eval_frame_mask(direct[which]), vmd_eval_frame_mask(direct[which]) + F);
                std::vector<float> h((size_t)dp.num_bins * (size_t)want->dim[1]);
                vmd_compute_histogram_masked(h.data(), dp.num_bins, want->min_range[0], want->max_range[0], want->values, want->dim[1], mask.data(), (int)F, false);
                if (dp.hist.dim != want->dim[1] || !same_floats(h.data(), dp.hist.bins, h.size())) {
                    std::fprintf(stderr, "%s: hist.dim %d want %d, range [%g, %g] vs [%g, %g], fp %llu / %llu\n", dp.label, dp.hist.dim, want->dim[1], dp.prop_data->min_range[0], dp.prop_data->max_range[0],
                                 want->min_range[0], want->max_range[0], (unsigned long long)dp.prop_fingerprint, (unsigned long long)dp.prop_data->fingerprint);
                    for (size_t k = 0; k < h.size(); ++k) if (h[k] != dp.hist.bins[k]) { std::fprintf(stderr, "  bin %zu: ref %g vmd %g\n", k, dp.hist.bins[k], h[k]); break; }
                }
                if (dp.hist.dim != want->dim[1] || !same_floats(h.data(), dp.hist.bins, h.size())) fail("display histogram of a temporal: reference compute_histogram_masked != vmd_compute_histogram_masked");
                checked_hist += 1;
                if (!which && name == "d1") dp_d1_dist = &dp;
            }
        }
        if (dp.type == DisplayProperty::Type_Volume && !which) dp_v = &dp;
        if (dp.type == DisplayProperty::Type_Temporal && name == "d1" && !strchr(dp.label, '(')) dp_d1 = &dp;
        if (!dp.vis_payload) fail("every display property carries a vis payload (src/main.cpp:1304)");
    }
    if (n_type[DisplayProperty::Type_Volume] != 2 || n_type[DisplayProperty::Type_Temporal] < 5 || n_type[DisplayProperty::Type_Distribution] < 12) fail("display property census");
    if (checked_hist != 4 || checked_rows < 7) fail("histogram / row checks did not all run");           // r and d1, full + filtered
    if (!dp_v || !dp_r || !dp_d1 || !dp_d1_dist) fail("v, r, d1 among the display properties");
    if (strcmp(dp_r->unit_str[0], "\xC3\x85") != 0 || strcmp(dp_d1->unit_str[1], "\xC3\x85") != 0 || dp_v->unit_str[0][0]) fail("unit strings (src/main.cpp:1314-1315)");
    if (dp_d1->num_samples != (int)F || dp_d1->y_values != dp_d1->prop_data->values || dp_d1->dim != 1) fail("temporal item of d1 (src/main.cpp:1371-1376)");
    // the masks: the reference iterates them inside compute_histogram_masked; here by hand against the direct evals
    for (int which = 0; which < 2; ++which) {
        const md_bitfield_t* mask = md_script_eval_frame_mask(which ? state.script.filt_eval : state.script.full_eval);
        for (size_t f = 0; f < F; ++f) if (md_bitfield_test_bit(mask, f) != (vmd_eval_frame_mask(direct[which])[f] != 0)) fail("frame mask differs from the direct call");
    }

    // ---- export_cube (src/main.cpp:5718-5830) against vmd_export_cube
    const std::string cube_ref = tmp + "/viamd_ref_callsites_ref.cube", cube_vmd = tmp + "/viamd_ref_callsites_vmd.cube";
    if (!export_cube(state, dp_v->prop_data, dp_v->vis_payload, str_t{cube_ref.data(), cube_ref.size()})) fail("the reference's export_cube returned false");
    if (!vmd_export_cube(cube_vmd.c_str(), direct[0], "v", &vsys, &vt, 0, atomic_numbers.data())) fail("vmd_export_cube");
    const std::vector<char> cube_a = slurp(cube_ref), cube_b = slurp(cube_vmd);
    if (cube_a.size() < 100000 || cube_a != cube_b) fail("cube file: the reference's export_cube and vmd_export_cube differ");
    {
        // the atoms block is not empty and carries real atomic numbers: line 7 starts with the atomic number of the first atom of s1's first structure
        size_t line = 0, pos = 0;
        for (; pos < cube_a.size() && line < 6; ++pos) line += cube_a[pos] == '\n';
        if (std::atoi(&cube_a[pos]) < 1) fail("cube file: atoms carry atomic numbers");
    }

    // ---- export_csv / export_xvg (src/main.cpp:5640-5716) on the tables draw_property_export_window assembles, against vmd_export_property_table
    {
        // distribution (:5998-6020): x = sample_range(hist.x_min, hist.x_max, hist.num_bins), label unit_str[0]; y = hist.bins, label `label` or "label (unit)"
        const DisplayProperty& dp = *dp_r;
        md_array(float) x_values = sample_range((float)dp.hist.x_min, (float)dp.hist.x_max, dp.hist.num_bins, frame_alloc);
        char y_label[128];
        if (strlen(dp.unit_str[1]) > 0) snprintf(y_label, sizeof(y_label), "%s (%s)", dp.label, (const char*)dp.unit_str); else snprintf(y_label, sizeof(y_label), "%s", dp.label);
        const float* column_data[2] = {x_values, dp.hist.bins};
        const char* column_labels[2] = {dp.unit_str[0], y_label};
        for (const char* fmt : {"csv", "xvg"}) {
            const std::string a = tmp + "/viamd_ref_callsites_r_ref." + fmt, b = tmp + "/viamd_ref_callsites_r_vmd." + fmt;
            const bool ok = fmt[0] == 'c' ? export_csv(column_data, column_labels, 2, (size_t)dp.hist.num_bins, str_t{a.data(), a.size()})
                                          : export_xvg(column_data, column_labels, 2, (size_t)dp.hist.num_bins, str_t{a.data(), a.size()});
            if (!ok) fail("the reference's table exporter returned false");
            if (!vmd_export_property_table(b.c_str(), direct[0], "r", fmt, nullptr, nullptr, dp.num_bins)) fail("vmd_export_property_table (distribution)");
            std::vector<char> fa = slurp(a), fb = slurp(b);
            if (fmt[0] == 'x') { fa = but_first_line(fa); fb = but_first_line(fb); }
            if (fa.size() < 1000 || fa != fb) { std::fprintf(stderr, "%s vs %s\n", a.c_str(), b.c_str()); fail("distribution table: the reference's exporter and vmd_export_property_table differ"); }
            remove(a.c_str()); remove(b.c_str());
        }
        md_array_free(x_values, frame_alloc);
    }
    {
        // temporal (:5953-5990): time column ("Frame" - the mock trajectory has no time unit), then y_values with "label (unit)" when unit[1] is set
        const DisplayProperty& dp = *dp_d1;
        std::vector<float> time(F);
        std::vector<double> frame_times(F);
        for (size_t f = 0; f < F; ++f) { frame_times[f] = 0.5 * (double)f; time[f] = (float)frame_times[f]; }
        char y_label[128];
        if (!md_unit_is_none(dp.unit[1])) snprintf(y_label, sizeof(y_label), "%s (%s)", dp.label, (const char*)dp.unit_str); else snprintf(y_label, sizeof(y_label), "%s", dp.label);
        const float* column_data[2] = {time.data(), dp.y_values};
        const char* column_labels[2] = {"Frame", y_label};
        for (const char* fmt : {"csv", "xvg"}) {
            const std::string a = tmp + "/viamd_ref_callsites_d1_ref." + fmt, b = tmp + "/viamd_ref_callsites_d1_vmd." + fmt;
            const bool ok = fmt[0] == 'c' ? export_csv(column_data, column_labels, 2, F, str_t{a.data(), a.size()}) : export_xvg(column_data, column_labels, 2, F, str_t{a.data(), a.size()});
            if (!ok) fail("the reference's table exporter returned false");
            if (!vmd_export_property_table(b.c_str(), direct[0], "d1", fmt, frame_times.data(), nullptr, 0)) fail("vmd_export_property_table (temporal)");
            std::vector<char> fa = slurp(a), fb = slurp(b);
            if (fmt[0] == 'x') { fa = but_first_line(fa); fb = but_first_line(fb); }
            if (fa.size() < 10 * F || fa != fb) { std::fprintf(stderr, "%s vs %s\n", a.c_str(), b.c_str()); fail("temporal table: the reference's exporter and vmd_export_property_table differ"); }
            remove(a.c_str()); remove(b.c_str());
        }
    }
    remove(cube_ref.c_str()); remove(cube_vmd.c_str());

    // ---- a script edit: a new ir, eval_init again (src/main.cpp:928) -> the block interrupts what runs, frees both evals, frees the old ir,
    // creates two new evals and re-builds the display properties; then an interrupt while the new tasks run (:983-984 on the next request)
    md_script_ir_t* old_ir = state.script.ir;
    vmd_script_ir_t* old_vir = vir;
    vmd_shim_bind_ir(old_ir, nullptr);             // the host's md_script_ir_free wrapper unbinds; the block frees old_ir itself
    static const char* kEdited = "d1 = distance(10,40);\nr = rdf(element('O'), element('O'), 8.0);\n{lin,plan,iso} = shape_weights(all);";
    state.script.ir = compile_and_bind(kEdited, &vir);
    state.script.eval_init = true;
    const long live_before = g_mock_live_evals.load();
    const int gui_frames2 = gui_frames_until_idle("the evaluation tasks of the edited script never finished");
    if (live_before != 2 || g_mock_live_evals.load() != 2) fail("the block frees the two old evals (fallback included) and creates two new ones");
    if (state.script.eval_ir != state.script.ir) fail("eval_ir follows ir");
    update_display_properties(&state);
    {
        vmd_script_eval_t* e = vmd_eval_create(F, vir);
        if (!e || !vmd_eval_frame_range(e, vir, &vsys, &vt, 0, (uint32_t)F) || !vmd_eval_wait_settled(e)) fail("direct evaluation (edited script)");
#ifdef VMD_SHIM_DEFERRED_SETTLE
        for (const auto t0 = std::chrono::steady_clock::now(); vmd_eval_frames_done(state.script.full_eval->eval) != F;) {
            if (std::chrono::steady_clock::now() - t0 > kSettleLimit) fail("deferred settle (edited script): the helper thread never settled");
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        if (!vmd_eval_wait_settled(state.script.full_eval->eval)) fail("vmd_eval_wait_settled");
#endif
        size_t seen = 0;
        for (size_t i = 0; i < md_array_size(state.display_properties); ++i) {
            const DisplayProperty& dp = state.display_properties[i];
            if (dp.partial_evaluation) continue;
            const vmd_script_property_data_t* want = vmd_eval_property_data(e, dp.label);
            if (!want) continue;
            if (!same_floats(dp.prop_data->values, want->values, want->num_values)) fail("edited script: values differ from the direct call");
            seen += 1;
        }
        if (seen < 2) fail("edited script: d1 and r among the display properties");
        vmd_eval_free(e);
    }
    // request a re-evaluation while nothing runs, then again at once: the second request finds the task running and interrupts it (:983-984)
    state.script.evaluate_full = true;
    viamd_main_loop_evaluation_block(state, F);
    state.script.evaluate_full = true;
    viamd_main_loop_evaluation_block(state, F);
    const int gui_frames3 = gui_frames_until_idle("the re-evaluation after an interrupt never finished");
#ifdef VMD_SHIM_DEFERRED_SETTLE
    if (!vmd_eval_wait_settled(state.script.full_eval->eval)) fail("vmd_eval_wait_settled");       // (the mask trails the last call by the quiet period)
#endif
    if (md_bitfield_popcount(md_script_eval_frame_mask(state.script.full_eval)) != F) fail("after interrupt + re-evaluation every frame is there");

    // ---- teardown in VIAMD's order (src/main.cpp:959-964 via a last eval_init without a valid ir is not reachable here: free by hand)
    task_system::shutdown();
    md_script_eval_free(state.script.full_eval);
    md_script_eval_free(state.script.filt_eval);
    if (g_mock_live_evals.load() != 0) fail("md_script_eval_free must free the fallback evals too");
    vmd_eval_free(direct[0]); vmd_eval_free(direct[1]);
    vmd_shim_bind_ir(state.script.ir, nullptr);
    vmd_ir_free(vir); vmd_ir_free(old_vir);
    md_script_ir_free(state.script.ir);
    std::printf("OK frames=%zu display_properties=%zu (temporal %zu, distribution %zu, volume %zu) gui_frames=%d+%d+%d cube_bytes=%zu log_errors=%ld\n", F, num_dp,
                n_type[0], n_type[1], n_type[2], gui_frames, gui_frames2, gui_frames3, cube_a.size(), host_log_errors.load());
    return host_log_errors.load() == 0 ? 0 : 1;
}
