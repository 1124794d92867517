// Several INDEPENDENT evaluations at the same time, one per thread, each with its own eval, its own script and its own kind of trajectory
// (HBM-resident, pinned host memory, an XTC file, the same XTC kept compressed in HBM): what VIAMD does when "Eval Full" and "Eval Filt" run
// side by side (src/main.cpp:982-1039), and what a host with several open documents does.  The evals share nothing the caller can see - but
// the library keeps process-wide state (launch parameters set per launch, the resource cache evals hand their device blocks to, the
// checkpoint cache of compressed files, the staging pools), and this program is there to run exactly that under ThreadSanitizer
// (scripts/tsan_emu.sh) and as a plain test (tests/test_native.py).  Every evaluation must reproduce, bit for bit, the integers the same
// script gave when it ran alone.
// usage: concurrent_evals [iterations] [frames] [atoms] [scratch_dir]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "vmd_eval.h"

static void fail(const char* what) { std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error()); std::exit(1); }

struct Script {
    vmd_script_ir_t* ir = nullptr;
    std::vector<uint64_t> g, v;          // the answer: RDF bins and voxels of a lone evaluation
    std::vector<float> d;                // temporal rows
};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 3;
    const size_t F = argc > 2 ? (size_t)std::atol(argv[2]) : 10, N = argc > 3 ? (size_t)std::atol(argv[3]) : 600;
    const std::string dir = argc > 4 ? argv[4] : "/tmp";
    if (vmd_device_count() <= 0) fail("no device");
    const float L = 30.0f;
    vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
    if (!dt || !vmd_devtraj_synth(dt, 11, L, 0.05f, 0, 0, F)) fail("synth");
    vmd_hosttraj_t* ht = vmd_hosttraj_create(F, N);
    if (!ht || !vmd_hosttraj_copy_from_device(ht, dt, 0, F)) fail("host copy");
    // the same frames as an XTC file (0.001 A precision: the file's own rounding, so file-backed scripts get their own answers below)
    const std::string xtc = dir + "/concurrent_evals.xtc";
    {
        vmd_xdrwriter_t* w = vmd_xdrwriter_open(xtc.c_str(), 0, N, 10000.0f);
        if (!w) fail("xtc writer");
        const vmd_unitcell_t cell = {L, L, L, 0.0f, 0.0f, 0.0f, VMD_UNITCELL_PBC_ALL};
        for (size_t f = 0; f < F; ++f) {
            size_t rs = 0;
            const float* x = vmd_hosttraj_frame_ptr(ht, f, &rs);
            vmd_hosttraj_set_cell(ht, f, &cell);
            if (!vmd_xdrwriter_write_frame(w, (int64_t)f, (float)f, &cell, x, x + rs, x + 2 * rs)) fail("xtc frame");
        }
        if (!vmd_xdrwriter_close(w)) fail("xtc close");
    }
    vmd_xdrtraj_t* xt = vmd_xdrtraj_open(xtc.c_str());
    if (!xt) fail("xtc open");
    vmd_rawtraj_t* rt = vmd_rawtraj_create(vmd_xdrtraj_interface(xt));
    if (!rt) fail("compressed copy in HBM");
    vmd_trajectory_i* trajs[4] = {vmd_devtraj_interface(dt), vmd_hosttraj_interface(ht), vmd_xdrtraj_interface(xt), vmd_rawtraj_interface(rt)};
    const char* traj_name[4] = {"HBM", "pinned host", "XTC file", "XTC in HBM"};

    std::vector<int32_t> oxy, hyd, st, tgt;
    for (size_t i = 0; i < N; ++i) (i % 3 == 0 ? oxy : hyd).push_back((int32_t)i);
    for (int32_t i = 0; i < 18; ++i) st.push_back(i);
    tgt.assign(oxy.begin() + 6, oxy.end());
    vmd_system_t sys = {};
    sys.atom_count = N;
    // four scripts with different grids / kernels: a wide O-O RDF (pencil walk), a short O-H RDF on two sets, an RDF whose cutoff
    // exceeds half the cell (all-pairs kernel) next to an SDF, and distances next to an SDF
    const int T = 4;
    std::vector<Script> scripts((size_t)T);
    for (int k = 0; k < T; ++k) {
        vmd_script_ir_t* ir = vmd_ir_create();
        bool ok = true;
        if (k == 0) ok = vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 11.0f);
        if (k == 1) ok = vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), hyd.data(), hyd.size(), 0.5f, 4.0f);
        if (k == 2) ok = vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 16.0f) &&
                         vmd_ir_add_sdf(ir, "v", st.data(), 2, 9, tgt.data(), tgt.size(), 7.0f);
        if (k == 3) ok = vmd_ir_add_rdf(ir, "g", hyd.data(), hyd.size(), hyd.data(), hyd.size(), 0.0f, 6.0f) &&
                         vmd_ir_add_sdf(ir, "v", st.data(), 2, 9, tgt.data(), tgt.size(), 9.0f) &&
                         vmd_ir_add_distance(ir, "d", VMD_DISTANCE_MIN, oxy.data(), 20, hyd.data(), 30);
        if (!ok) fail("script");
        scripts[(size_t)k].ir = ir;
    }
    // one evaluation of script k over trajectory kind t, in two calls; returns false on any difference from `want` (or fills it)
    auto evaluate = [&](int k, int t, Script* fill) -> bool {
        Script& s = scripts[(size_t)k];
        vmd_script_eval_t* e = vmd_eval_create(F, s.ir);
        if (!e) return false;
        bool ok = vmd_eval_frame_range(e, s.ir, &sys, trajs[t], 0, (uint32_t)(F / 2)) && vmd_eval_frame_range(e, s.ir, &sys, trajs[t], (uint32_t)(F / 2), (uint32_t)F);
        const vmd_script_property_data_t* g = ok ? vmd_eval_property_data(e, "g") : nullptr;
        const vmd_script_property_data_t* v = ok ? vmd_eval_property_data(e, "v") : nullptr;
        const vmd_script_property_data_t* d = ok ? vmd_eval_property_data(e, "d") : nullptr;
        if (ok && v) ok = vmd_eval_refresh_counts(e, "v");
        if (ok) {
            const size_t ng = (size_t)g->dim[2], nv = v ? (size_t)v->dim[1] * (size_t)v->dim[2] * (size_t)v->dim[3] : 0, nd = d ? (size_t)d->dim[0] * (size_t)d->dim[1] : 0;
            if (fill) {
                fill->g.assign(g->counts, g->counts + ng);
                if (v) fill->v.assign(v->counts, v->counts + nv);
                if (d) fill->d.assign(d->values, d->values + nd);
            } else {
                const Script& want = scripts[(size_t)(k + T * (t >= 2 ? 1 : 0))];
                ok = std::memcmp(want.g.data(), g->counts, ng * sizeof(uint64_t)) == 0 && (!v || std::memcmp(want.v.data(), v->counts, nv * sizeof(uint64_t)) == 0) &&
                     (!d || std::memcmp(want.d.data(), d->values, nd * sizeof(float)) == 0);
                if (!ok) std::fprintf(stderr, "script %d over %s: differs from its lone evaluation\n", k, traj_name[t]);
            }
        }
        vmd_eval_free(e);
        return ok;
    };
    // the answers, evaluated alone: scripts[k] from the float frames, scripts[T + k] from the XTC's rounded frames
    scripts.resize((size_t)(2 * T));
    for (int k = 0; k < T; ++k) {
        scripts[(size_t)(T + k)].ir = scripts[(size_t)k].ir;
        if (!evaluate(k, 0, &scripts[(size_t)k]) || !evaluate(k, 2, &scripts[(size_t)(T + k)])) fail("lone evaluation");
    }
    std::atomic<int> bad{0};
    uint64_t evals = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<std::thread> pool;
        for (int th = 0; th < T; ++th)
            pool.emplace_back([&, th] {
                // every thread walks all scripts and all trajectory kinds in its own order: at any time the four threads are in
                // different scripts over different kinds of storage
                for (int step = 0; step < T; ++step) {
                    const int k = (th + step + it) % T, t = (th + 2 * step + it) % 4;
                    if (!evaluate(k, t, nullptr)) bad += 1;
                }
            });
        for (auto& t : pool) t.join();
        evals += (uint64_t)(T * T);
        if (bad.load()) fail("a concurrent evaluation differs from the lone one");
    }
    std::printf("OK iterations=%d concurrent evaluations=%llu (4 threads x 4 scripts x {HBM, pinned host, XTC file, XTC in HBM})\n", iters, (unsigned long long)evals);
    for (int k = 0; k < T; ++k) vmd_ir_free(scripts[(size_t)k].ir);
    vmd_rawtraj_free(rt); vmd_xdrtraj_close(xt); vmd_hosttraj_free(ht); vmd_devtraj_free(dt);
    std::remove(xtc.c_str());
    return 0;
}
