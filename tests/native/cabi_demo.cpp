// tests/native/cabi_demo.cpp — a C++ host driving the C ABI exactly the way VIAMD drives mdlib
// (/root/reference/src/main.cpp:951-1039): create the eval from an IR, clear it, hand frame sub-ranges to a pool of worker
// threads that all call frame_range on the SAME eval, poll the property fingerprint from the "GUI" thread meanwhile,
// interrupt and restart once.  No Python, no torch: only include/vmd_eval.h and libviamd_amd.so.
//
// Prints one line "OK frames=<F> hits=<sum of counts> polls=<n> fingerprint_changes=<n>" and exits 0 on success.
#include <atomic>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "vmd_eval.h"

static void fail(const char* what) {
    std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error());
    std::exit(1);
}

int main(int argc, char** argv) {
    const size_t F = argc > 1 ? (size_t)std::atoi(argv[1]) : 64;
    const size_t N = 30000;
    const float L = 80.0f;
    if (vmd_device_count() <= 0) fail("no HIP device");

    vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
    if (!dt || !vmd_devtraj_synth(dt, 7, L, 0.05f, 0, 0, F)) fail("synthetic trajectory");
    vmd_trajectory_i* traj = vmd_devtraj_interface(dt);

    std::vector<int32_t> oxy;
    for (size_t i = 0; i < N; i += 3) oxy.push_back((int32_t)i);
    vmd_script_ir_t* ir = vmd_ir_create();
    if (!vmd_ir_add_rdf(ir, "r", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 12.0f)) fail("add_rdf");
    const int32_t a = 0, b = 300;
    if (!vmd_ir_add_distance(ir, "d", VMD_DISTANCE_COM, &a, 1, &b, 1)) fail("add_distance");

    vmd_script_eval_t* eval = vmd_eval_create(F, ir);
    if (!eval) fail("eval_create");
    if (vmd_eval_ir_fingerprint(eval) != vmd_ir_fingerprint(ir)) fail("fingerprint mismatch");
    const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, "r");      // fetched once, like init_display_properties
    const vmd_script_property_data_t* pdd = vmd_eval_property_data(eval, "d");
    if (!pd || !pdd || pd->dim[2] != VMD_RDF_NUM_BINS || pdd->dim[0] != (int)F) fail("property_data");
    // unit[0] / unit[1] as printed by VIAMD (src/main.cpp:1300-1315): a distribution over a length, a length over frames
    if (strcmp(pd->unit_str[0], "\xC3\x85") || strcmp(pd->unit_str[1], "") || strcmp(pdd->unit_str[0], "") || strcmp(pdd->unit_str[1], "\xC3\x85")) fail("unit strings");

    vmd_system_t sys = {};
    sys.atom_count = N;

    auto run_pool = [&](int nthreads, bool interrupt_midway) {
        vmd_eval_clear_data(eval);
        std::atomic<uint32_t> next{0};
        std::atomic<int> running{nthreads};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&] {
                for (;;) {                                   // enkiTS-style: contiguous sub-ranges pulled dynamically
                    const uint32_t beg = next.fetch_add(4);
                    if (beg >= F) break;
                    const uint32_t end = beg + 4 < F ? beg + 4 : (uint32_t)F;
                    // the interrupt arrives while the task is half way through its ranges (VIAMD: a script edit, src/main.cpp:984).  Raised from
                    // the thread that pulls the middle range, not from a poll of frames_done: with read-ahead the frames of a short trajectory are
                    // evaluated in one region and committed together, so progress is not a clock any more
                    if (interrupt_midway && beg == (uint32_t)(F / 2) / 4 * 4) vmd_eval_interrupt(eval);
                    if (!vmd_eval_frame_range(eval, ir, &sys, traj, beg, end)) break;      // false: interrupted
                }
                running -= 1;
            });
        int polls = 0, changes = 0;
        uint64_t fp = pd->fingerprint;
        while (running.load() > 0) {                         // the GUI thread: poll fingerprints, read (possibly torn) values
            if (pd->fingerprint != fp) { fp = pd->fingerprint; changes += 1; volatile float v = pd->values[100]; (void)v; }
            polls += 1;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        for (auto& th : pool) th.join();
        return std::make_pair(polls, changes);
    };

    run_pool(8, true);                                       // interrupted run: some frames only
    const size_t partial = vmd_eval_frames_done(eval);
    if (partial >= F && F >= 32) fail("interrupt had no effect");
    const auto t0 = std::chrono::steady_clock::now();
    auto pc = run_pool(8, false);                            // restart from scratch (src/main.cpp:990)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (vmd_eval_frames_done(eval) != F) fail("not all frames evaluated");
    const uint8_t* mask = vmd_eval_frame_mask(eval);
    for (size_t f = 0; f < F; ++f) if (!mask[f]) fail("frame mask hole");

    unsigned long long hits = 0;
    for (int k = 0; k < pd->dim[2]; ++k) {
        if (pd->counts[k] % 2) fail("same-set histogram must be even");
        if (pd->values[k] != (float)pd->counts[k]) fail("values != (float)counts");
        hits += pd->counts[k];
    }
    // ideal-gas-like O-O box: rho * (4/3) pi rc^3 ordered neighbours per atom, within 2 %
    const double expect = (double)F * oxy.size() * (oxy.size() / ((double)L * L * L)) * 4.18879 * 12.0 * 12.0 * 12.0;
    if (hits < 0.98 * expect || hits > 1.02 * expect) fail("hit count off the analytic expectation");
    float g[128];
    vmd_downsample_histogram(g, 128, pd->values, pd->weights, pd->dim[2]);
    if (g[100] < 0.9f || g[100] > 1.1f) fail("g(r) not ~1 at large r");
    for (size_t f = 0; f < F; ++f) if (!(pdd->values[f] > 0.0f && pdd->values[f] < 0.87f * L)) fail("distance row");

    std::printf("OK frames=%zu hits=%llu polls=%d fingerprint_changes=%d interrupted_at=%zu eval_ms=%.2f frames_per_s=%.0f\n", F, hits,
                pc.first, pc.second, partial, ms, F / ms * 1e3);
    vmd_eval_free(eval);
    vmd_ir_free(ir);
    vmd_devtraj_free(dt);
    return 0;
}
