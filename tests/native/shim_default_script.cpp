// tests/native/shim_default_script.cpp - VIAMD's OWN default script behind the drop-in (VERDICT r04 missing #1).
//
// The literal string of /root/reference/src/main.cpp:528 mixes three hot-path properties (d1 = distance, r = rdf, v = sdf) with two
// statements mdlib alone evaluates (a1 = angle(...) in resname("ALA"); {lin,plan,iso} = shape_weights(all)).  VIAMD evaluates every
// property of the IR in ONE md_script_eval_frame_range (:993-997) and asks md_script_eval_property_data for every name of
// md_script_ir_property_names (:1277-1291); a property that comes back NULL silently disappears from the timeline (:1288-1291).
// include/vmd_md_script_shim.h therefore decorates mdlib's evaluator instead of replacing it: here mdlib is the CPU mock of
// tests/native/md_mock_eval.h behind VMD_SHIM_FALLBACK(name) = mockmd_##name, and the program checks, through the md_* names only:
//   * all of d1, a1, r, v, lin, plan, iso come back through md_script_eval_property_data; s1 (a selection, no property) gets mdlib's NULL
//   * d1 / r / v are bit-identical to direct vmd_* calls and are NOT the fallback's CPU copies; a1 / lin / plan / iso are the mock's values
//   * md_script_eval_ir_fingerprint == md_script_ir_fingerprint(ir) (:987), perturbed once the GPU binding of the ir changes
//   * frame_mask = AND of the two evaluators' masks; interrupt, clear_data and free reach both
//   * with the reduced script (vmd_script_report_fallback_source) bound as the fallback's IR nothing is evaluated twice
//   * a script without any hot-path statement runs on the fallback alone; mdlib's own vis payloads are forwarded
// Prints "OK ..." and exits 0.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "md_mock.h"
#include "md_mock_eval.h"
#define VMD_SHIM_FALLBACK(name) mockmd_##name
#define VMD_SHIM_FALLBACK_DECLARED
#define VMD_SHIM_PREFIX(name) name
#include "vmd_md_script_shim.h"

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

int main(int argc, char** argv) {
    const size_t F = argc > 1 ? (size_t)std::atoi(argv[1]) : 16;
    const size_t n_res = 20, n_blob = n_res * 10, N = n_blob + 933 * 3;
    const float L = 40.0f;
    if (vmd_device_count() <= 0) fail("no HIP device");
    if (vmd_shim_min_work() != VMD_SHIM_MIN_WORK_DEFAULT) fail("default work threshold");
    vmd_shim_set_min_work(0);                     // this test system is far below the default threshold: send what is bound to the GPU (both sides: below)

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
    md_system_t sys{};
    sys.atom.count = N; sys.atom.x = sx.data(); sys.atom.y = sy.data(); sys.atom.z = sz.data(); sys.atom.mass = mass.data();
    sys.unitcell = md_unitcell_t{L, L, L, 0, 0, 0, 7u};
    sys.trajectory = &traj_i;

    // the molecule's topology: 20 ALA residues of 10 atoms (N C C O C H H H C H), then waters - what selections resolve against
    static const char* ala[10] = {"N", "C", "C", "O", "C", "H", "H", "H", "C", "H"};
    std::vector<const char*> elements(N), resnames(N);
    std::vector<int32_t> residue_index(N);
    for (size_t i = 0; i < N; ++i) {
        if (i < n_blob) { elements[i] = ala[i % 10]; resnames[i] = "ALA"; residue_index[i] = (int32_t)(i / 10); }
        else { const size_t w = i - n_blob; elements[i] = w % 3 == 0 ? "O" : "H"; resnames[i] = "HOH"; residue_index[i] = (int32_t)(n_res + w / 3); }
    }
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

    // ---- "md_script_ir_compile_from_source" (src/main.cpp:878): mdlib compiles the whole script ...
    md_script_ir_t* eval_ir = mock_ir_compile(kDefaultScript, residues_of);
    if (!eval_ir || md_script_ir_property_count(eval_ir) != 7) fail("mock mdlib: the default script has seven properties");
    // ... and the backend takes what it evaluates: d1, r, v.  The rest is reported, not refused (vmd_ir_compile_from_source_partial)
    vmd_script_ir_t* vir = vmd_ir_create();
    vmd_script_report_t* report = nullptr;
    if (!vmd_ir_compile_from_source_partial(vir, kDefaultScript, &topo, &report)) fail("vmd_ir_compile_from_source_partial");
    if (vmd_ir_property_count(vir) != 3 || vmd_script_report_skipped_count(report) != 2) fail("d1, r, v compiled; a1 and {lin,plan,iso} reported");
    if (strcmp(vmd_script_report_skipped(report)[0].names, "a1") != 0 || strcmp(vmd_script_report_skipped(report)[1].names, "lin,plan,iso") != 0) fail("skipped names");
    vmd_shim_bind_ir(eval_ir, vir);
    md_allocator_i persistent{nullptr};

    // ---- src/main.cpp:966-972, 1275-1316
    md_script_eval_t* full_eval = md_script_eval_create(F, eval_ir, &persistent);
    md_script_eval_t* filt_eval = md_script_eval_create(F, eval_ir, &persistent);
    if (!full_eval || !filt_eval) fail("md_script_eval_create");
    struct DisplayProperty { std::string label; md_script_property_flags_t flags; const md_script_property_data_t* prop_data; const md_script_vis_payload_o* vis_payload; const md_script_eval_t* eval; };
    std::vector<DisplayProperty> display_properties;
    const md_script_eval_t* evals[2] = {full_eval, filt_eval};
    for (size_t eval_idx = 0; eval_idx < 2; ++eval_idx) {
        const size_t num_props = md_script_ir_property_count(eval_ir);
        const str_t* prop_names = md_script_ir_property_names(eval_ir);
        for (size_t i = 0; i < num_props; ++i) {
            const md_script_property_data_t* prop_data = md_script_eval_property_data(evals[eval_idx], prop_names[i]);
            if (!prop_data) { std::fprintf(stderr, "property %.*s\n", (int)prop_names[i].len, prop_names[i].ptr); fail("a property of the default script disappeared behind the drop-in (src/main.cpp:1288-1291)"); }
            const md_script_vis_payload_o* payload = md_script_ir_property_vis_payload(eval_ir, prop_names[i]);
            if (!payload) fail("md_script_ir_property_vis_payload");
            display_properties.push_back({std::string(prop_names[i].ptr, prop_names[i].len), md_script_ir_property_flags(eval_ir, prop_names[i]), prop_data, payload, evals[eval_idx]});
        }
    }
    if (display_properties.size() != 14) fail("seven display properties per eval");
    if (md_script_eval_property_data(full_eval, STR_LIT("s1")) != nullptr) fail("s1 is a selection: mdlib has no property record for it");

    // ---- :982-1008 "Eval Full"
    auto pool_task = [&](md_script_eval_t* eval, uint32_t range_beg, uint32_t range_end, int nthreads) {
        std::atomic<uint32_t> next{range_beg};
        std::atomic<int> failed{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const uint32_t frame_beg = next.fetch_add(2);
                    if (frame_beg >= range_end) break;
                    const uint32_t frame_end = frame_beg + 2 < range_end ? frame_beg + 2 : range_end;
                    if (!md_script_eval_frame_range(eval, eval_ir, &sys, sys.trajectory, frame_beg, frame_end)) failed += 1;
                }
            });
        for (auto& t : pool) t.join();
        return failed.load() == 0;
    };
    if (!md_script_ir_valid(eval_ir) || md_script_eval_ir_fingerprint(full_eval) != md_script_ir_fingerprint(eval_ir)) fail("src/main.cpp:986-987: eval and ir fingerprints must match");
    // built with -DVMD_SHIM_DEFERRED_SETTLE the GPU part's results trail the last call by the quiet period (include/vmd_eval.h): VIAMD polls, this
    // program compares at once - so it waits where VIAMD would simply look again a frame later
    auto settled = [&](md_script_eval_t* e, size_t frames_expected = 0) {
#ifdef VMD_SHIM_DEFERRED_SETTLE
        // nobody calls into the shim any more - the helper thread settles on its own, and the records VIAMD polls
        // (fingerprint, max_value: src/main.cpp:1508-1509, density_volume.cpp:281) must follow by themselves (vmd_eval_set_settled_callback)
        if (e->eval && frames_expected) {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                bool follows = vmd_eval_frames_done(e->eval) == frames_expected;
                for (auto& p : e->props)
                    if (p->src) follows = follows && vmd_shim::peek(p->dst.fingerprint) == md_script_eval_t::mix(vmd_shim::peek(p->src->fingerprint), e->epoch.load()) &&
                                          vmd_shim::peek(p->dst.max_value) == vmd_shim::peek(p->src->max_value);
                if (follows) break;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) fail("deferred settle: the records handed to VIAMD did not follow the helper's settle");
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
        if (e->eval && !vmd_eval_wait_settled(e->eval)) fail("vmd_eval_wait_settled");
#else
        (void)e; (void)frames_expected;
#endif
    };
    md_script_eval_clear_data(full_eval);
    if (!pool_task(full_eval, 0, (uint32_t)F, 4)) fail("Eval Full");
    settled(full_eval, F);
    const uint32_t beg_frame = (uint32_t)(F / 4), end_frame = (uint32_t)(F - F / 4);
    md_script_eval_clear_data(filt_eval);
    if (!pool_task(filt_eval, beg_frame, end_frame, 3)) fail("Eval Filt");
    settled(filt_eval, end_frame - beg_frame);

    // ---- what VIAMD then reads: the hot-path properties = direct vmd_* calls, bit for bit; the others = mdlib's (the mock's) own values
    auto prop = [&](const md_script_eval_t* e, const char* nm) { return md_script_eval_property_data(e, str_t{nm, strlen(nm)}); };
    {
        vmd_script_eval_t* e = vmd_eval_create(F, vir);
        vmd_system_t vsys = vmd_shim::wrap_system(&sys);
        vmd_trajectory_i vt = vmd_shim::wrap_trajectory(&traj_i);
        if (!e || !vmd_eval_frame_range(e, vir, &vsys, &vt, 0, (uint32_t)F)) fail("direct evaluation");
        if (!vmd_eval_wait_settled(e)) fail("direct evaluation: settle");       // (a no-op unless the process runs in deferred-settle mode)
        for (const char* nm : {"d1", "r", "v"}) {
            const vmd_script_property_data_t* want = vmd_eval_property_data(e, nm);
            const md_script_property_data_t* got = prop(full_eval, nm);
            if (!want || got->num_values != want->num_values || memcmp(got->values, want->values, want->num_values * sizeof(float)) != 0) fail("d1 / r / v through the shim differ from direct vmd_* calls");
            for (size_t i = 0; i < got->num_values; ++i) if (got->values[i] == MOCK_CPU_COPY) fail("the shim handed out the fallback's CPU copy of a bound property");
            if (got->weights && (!want->weights || memcmp(got->weights, want->weights, (size_t)want->dim[2] * sizeof(float)) != 0)) fail("rdf weights");
        }
        vmd_eval_free(e);
    }
    double a1_sum = 0.0;
    {
        std::vector<float> x(N), y(N), z(N), row(64);
        for (size_t f = 0; f < F; ++f) {
            mock_load_frame(&mt, (int64_t)f, nullptr, x.data(), y.data(), z.data());
            for (const MockProp& p : eval_ir->props) {
                if (p.kind == MockProp::CPU_COPY) continue;
                mock_eval_row(p, x.data(), y.data(), z.data(), N, row.data());
                for (int which = 0; which < 2; ++which) {
                    if (which == 1 && (f < beg_frame || f >= end_frame)) continue;
                    const md_script_property_data_t* got = prop(evals[which], p.name.c_str());
                    if (got->dim[0] != (int32_t)F || got->dim[1] != (int32_t)p.width()) fail("temporal layout of a fallback property (src/main.cpp:1353-1378)");
                    if (memcmp(got->values + f * p.width(), row.data(), p.width() * sizeof(float)) != 0) fail("a1 / lin / plan / iso are not the fallback evaluator's values");
                }
                if (p.kind == MockProp::ANGLE) for (size_t c = 0; c < p.width(); ++c) a1_sum += row[c];
            }
        }
        if (prop(full_eval, "a1")->dim[1] != (int32_t)n_res) fail("angle(...) in resname(\"ALA\"): one value per residue");
    }
    // both evaluators have every frame of their range: the AND of the masks is the range
    for (int which = 0; which < 2; ++which) {
        const md_bitfield_t* mask = md_script_eval_frame_mask(evals[which]);
        for (size_t f = 0; f < F; ++f) if (md_bitfield_test_bit(mask, f) != (which == 0 || (f >= beg_frame && f < end_frame))) fail("frame mask after the evaluation");
    }

    // ---- frame_mask = AND: a frame one evaluator is ahead with does not count yet
    {
        md_script_eval_t* e = md_script_eval_create(F, eval_ir, &persistent);
        md_script_eval_clear_data(e);
        const uint32_t half = (uint32_t)(F / 2);
        if (!md_script_eval_frame_range(e, eval_ir, &sys, sys.trajectory, 0, half)) fail("frame_range (first half)");
        settled(e);
        if (!mockmd_md_script_eval_frame_range(e->fb, eval_ir, &sys, sys.trajectory, half, (uint32_t)F)) fail("fallback ahead");
        const md_bitfield_t* mask = md_script_eval_frame_mask(e);
        for (size_t f = 0; f < F; ++f) if (md_bitfield_test_bit(mask, f) != (f < half)) fail("frame mask must be the AND of the GPU's and the fallback's");
        // interrupt reaches both evaluators (src/main.cpp:952-953, 984); clear_data re-arms both (:990)
        md_script_eval_interrupt(e);
        if (e->fb->interrupts.load() != 1) fail("interrupt was not forwarded to the fallback");
        md_script_eval_clear_data(e);
        if (e->fb->clears.load() != 2 || e->fb->interrupt.load()) fail("clear_data was not forwarded to the fallback");
        if (!md_script_eval_frame_range(e, eval_ir, &sys, sys.trajectory, 0, 2)) fail("frame_range after interrupt + clear_data");
        md_script_eval_free(e);
    }

    // ---- no double work: mdlib compiles the script WITHOUT the statements the GPU took and evaluates only that
    {
        const char* reduced_text = vmd_script_report_fallback_source(report);
        if (strlen(reduced_text) != strlen(kDefaultScript)) fail("fallback source keeps the offsets of the editor's text");
        md_script_ir_t* reduced = mock_ir_compile(reduced_text, residues_of);
        if (!reduced || md_script_ir_property_count(reduced) != 4) fail("the reduced script has a1, lin, plan, iso");
        vmd_shim_bind_fallback_ir(eval_ir, reduced);
        md_script_eval_t* e = md_script_eval_create(F, eval_ir, &persistent);
        if (!e || e->fb->ir != reduced) fail("the fallback eval must be created from the reduced ir");
        if (md_script_eval_ir_fingerprint(e) != md_script_ir_fingerprint(eval_ir)) fail("fingerprint with a reduced fallback ir: still the editor's script (src/main.cpp:987)");
        md_script_eval_clear_data(e);
        for (uint32_t f = 0; f < F; f += 4) if (!md_script_eval_frame_range(e, eval_ir, &sys, sys.trajectory, f, std::min<uint32_t>(f + 4, (uint32_t)F))) fail("frame_range (reduced)");
        settled(e);
        for (size_t i = 0; i < md_script_ir_property_count(eval_ir); ++i) {
            const str_t nm = md_script_ir_property_names(eval_ir)[i];
            const md_script_property_data_t* a = md_script_eval_property_data(e, nm);
            const md_script_property_data_t* b = md_script_eval_property_data(full_eval, nm);
            if (!a || a->num_values != b->num_values || memcmp(a->values, b->values, a->num_values * sizeof(float)) != 0) fail("reduced fallback ir: a property differs");
        }
        for (const MockProp& p : e->fb->ir->props) if (p.kind == MockProp::CPU_COPY) fail("the reduced ir still carries a hot-path property");
        md_script_eval_free(e);
        vmd_shim_bind_fallback_ir(eval_ir, nullptr);
        md_script_ir_free(reduced);
    }

    // ---- mdlib's own vis payload (the atoms an angle is measured on) is forwarded; the sdf payload is still the backend's
    {
        md_allocator_i frame_alloc{nullptr};
        md_script_vis_ctx_t ctx = {eval_ir, &sys, sys.trajectory};
        md_script_vis_t vis = {};
        md_script_vis_init(&vis, &frame_alloc);
        if (!md_script_vis_eval_payload(&vis, display_properties[1].vis_payload, -1, &ctx, MD_SCRIPT_VISUALIZE_ATOMS)) fail("vis payload of a1");
        if (display_properties[1].label != "a1" || md_bitfield_popcount(&vis.atom_mask) != 3 * n_res) fail("a1 highlights three atoms per residue");
        md_script_vis_free(&vis);
        md_script_vis_init(&vis, &frame_alloc);
        if (!md_script_vis_eval_payload(&vis, display_properties[3].vis_payload, -1, &ctx, MD_SCRIPT_VISUALIZE_SDF)) fail("vis payload of v");
        if (display_properties[3].label != "v" || md_array_size(vis.sdf.structures) != 7 || vis.sdf.extent != 10.0f) fail("v: seven reference structures (resname(\"ALA\")[2:8]), extent 10");
        md_script_vis_free(&vis);
    }

    // ---- the GPU binding of the ir changes under a live eval: VIAMD must see a fingerprint mismatch and re-create (src/main.cpp:986-987)
    {
        vmd_script_ir_t* other = vmd_ir_create();
        const int32_t a = 0, b = 5;
        if (!vmd_ir_add_distance(other, "d1", VMD_DISTANCE_COM, &a, 1, &b, 1)) fail("add_distance");
        vmd_shim_bind_ir(eval_ir, other);
        if (md_script_eval_ir_fingerprint(full_eval) == md_script_ir_fingerprint(eval_ir)) fail("a stale GPU binding must show as a fingerprint mismatch");
        if (md_script_eval_frame_range(full_eval, eval_ir, &sys, sys.trajectory, 0, 1)) fail("frame_range with a stale binding must refuse");
        vmd_shim_bind_ir(eval_ir, vir);
        if (md_script_eval_ir_fingerprint(full_eval) != md_script_ir_fingerprint(eval_ir)) fail("fingerprint after re-binding");
        vmd_ir_free(other);
    }

    // ---- the work threshold (include/vmd_md_script_shim.h, vmd_shim_set_min_work): an evaluation below it stays with the
    // evaluator behind the shim - whole script, no GPU eval - and one at or above it goes to the GPU as before
    {
        const uint64_t work = vmd_ir_work_per_frame(vir) * (uint64_t)F;
        // d1: 1 pair; r: |C| x |H|; v: 7 structures (residues 2..8 of ALA, 10 atoms each) x (|H| + 10)
        size_t nC = 0, nH = 0;
        for (size_t i = 0; i < N; ++i) { nC += elements[i][0] == 'C'; nH += elements[i][0] == 'H'; }
        if (work != (1 + (uint64_t)nC * nH + 7 * ((uint64_t)nH + 10)) * F) fail("vmd_ir_work_per_frame of the default script");
        vmd_shim_set_min_work(work + 1);                                   // just too small
        md_script_eval_t* small = md_script_eval_create(F, eval_ir, &persistent);
        if (!small || small->eval || !small->fb || small->fb->ir != eval_ir) fail("below the threshold: no GPU eval, the fallback gets the WHOLE script");
        if (md_script_eval_ir_fingerprint(small) != md_script_ir_fingerprint(eval_ir)) fail("below the threshold: fingerprint (src/main.cpp:987) must still match");
        md_script_eval_clear_data(small);
        if (!md_script_eval_frame_range(small, eval_ir, &sys, sys.trajectory, 0, (uint32_t)F)) fail("below the threshold: frame_range");
        for (const char* nm : {"d1", "r", "v"}) {
            const md_script_property_data_t* rec = prop(small, nm);
            if (!rec || rec->values[0] != MOCK_CPU_COPY) fail("below the threshold: d1 / r / v are the fallback evaluator's own");
        }
        if (memcmp(prop(small, "a1")->values, prop(full_eval, "a1")->values, prop(small, "a1")->num_values * sizeof(float)) != 0) fail("below the threshold: a1");
        if (md_bitfield_popcount(md_script_eval_frame_mask(small)) != F) fail("below the threshold: frame mask");
        md_script_eval_free(small);
        vmd_shim_set_min_work(work);                                       // exactly enough
        md_script_eval_t* big = md_script_eval_create(F, eval_ir, &persistent);
        if (!big || !big->eval) fail("at the threshold: the GPU evaluates the bound properties");
        md_script_eval_free(big);
        vmd_shim_set_min_work(0);
    }

    // ---- a script without any hot-path statement: nothing is bound, the fallback evaluates it alone
    {
        md_script_ir_t* only = mock_ir_compile("a1 = angle(2,1,3) in resname(\"ALA\");", residues_of);
        md_script_eval_t* e = md_script_eval_create(F, only, &persistent);
        if (!e || e->eval) fail("an ir without bound properties runs on the fallback alone");
        if (md_script_eval_ir_fingerprint(e) != md_script_ir_fingerprint(only)) fail("fingerprint (fallback alone)");
        md_script_eval_clear_data(e);
        if (!md_script_eval_frame_range(e, only, &sys, sys.trajectory, 0, (uint32_t)F)) fail("frame_range (fallback alone)");
        const md_script_property_data_t* a = md_script_eval_property_data(e, STR_LIT("a1"));
        if (!a || memcmp(a->values, prop(full_eval, "a1")->values, a->num_values * sizeof(float)) != 0) fail("a1 (fallback alone)");
        if (md_bitfield_popcount(md_script_eval_frame_mask(e)) != F) fail("frame mask (fallback alone)");
        md_script_eval_free(e);
        md_script_ir_free(only);
    }

    // ---- :952-953 interrupt while a task runs, :960-964 free
    std::thread late([&] { md_script_eval_clear_data(full_eval); (void)pool_task(full_eval, 0, (uint32_t)F, 2); });
    md_script_eval_interrupt(full_eval);
    late.join();
    md_script_eval_free(full_eval);
    md_script_eval_free(filt_eval);
    if (g_mock_live_evals.load() != 0) fail("md_script_eval_free must free the fallback evals too");
    vmd_shim_bind_ir(eval_ir, nullptr);
    vmd_script_report_free(report);
    vmd_ir_free(vir);
    md_script_ir_free(eval_ir);
    std::printf("OK frames=%zu properties=7 (3 on the GPU, 4 on the fallback) a1_sum=%.3f\n", F, a1_sum);
    return 0;
}
