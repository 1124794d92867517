// tests/native/stress_readahead.cpp - read-ahead under real thread concurrency: C++ pool threads with random thread counts,
// grains, sub-ranges, hand-out orders, block sizes and region sizes, an occasional large direct call in the middle of the small ones, an
// occasional interrupt + restart.  After every evaluation the eval must hold - bit for bit - what ONE call over the same frames leaves on a
// fresh eval with read-ahead switched off: the RDF bins (u64), the normalisation weights, the temporal rows of the evaluated frames, the
// frame mask and frames_done.  Python threads (tests/cases.py: readahead_case) take turns on the GIL; these do not.
// With "sdf" as fifth argument the script also holds an sdf() volume (block partials of 16.8 MB each); every third iteration a second eval with
// the first as its SOURCE walks a sub-range from the pool (VIAMD's filtered evaluation): its regions adopt the source's finished blocks.
// Every fourth iteration runs in DEFERRED-SETTLE mode (vmd_set_option("readahead_lone", 1), round 5): half of those with ONE caller thread (the
// lone sequential walker the mode is for), the settle owed to the eval's helper thread - awaited with vmd_eval_wait_settled, sometimes after a
// pause in which the helper may have run on its own - and interrupts / clear_data / free racing that thread.
// usage: stress_readahead [iterations = 60] [frames = 96] [atoms = 1500] [seed = 1] [sdf];  prints "OK iterations=<n> ..." and exits 0.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "vmd_eval.h"

static int g_iter = 0;
static void fail(const char* what) {
    std::fprintf(stderr, "FAIL in iteration %d: %s (%s)\n", g_iter, what, vmd_last_error());
    std::exit(1);
}

static bool g_sdf = false;
struct Snapshot {
    std::vector<float> volume;
    float volume_max = 0.0f;
    std::vector<uint64_t> counts;
    std::vector<double> weights;
    std::vector<float> temporal;
    std::vector<uint8_t> mask;
    size_t done = 0;
};

static Snapshot snapshot(vmd_script_eval_t* e, size_t F) {
    Snapshot s;
    const vmd_script_property_data_t* g = vmd_eval_property_data(e, "g");
    const vmd_script_property_data_t* d = vmd_eval_property_data(e, "d");
    s.counts.assign(g->counts, g->counts + g->dim[2]);
    s.weights.assign(g->weights64, g->weights64 + g->dim[2]);
    s.mask.assign(vmd_eval_frame_mask(e), vmd_eval_frame_mask(e) + F);
    s.temporal.assign(d->values, d->values + (size_t)d->dim[0] * (size_t)d->dim[1]);
    for (size_t f = 0; f < F; ++f) if (!s.mask[f]) for (int i = 0; i < d->dim[1]; ++i) s.temporal[f * (size_t)d->dim[1] + (size_t)i] = 0.0f;   // rows nobody asked for: whatever
    s.done = vmd_eval_frames_done(e);
    if (g_sdf) {
        const vmd_script_property_data_t* v = vmd_eval_property_data(e, "v");
        s.volume.assign(v->values, v->values + v->num_values);
        s.volume_max = v->max_value;
    }
    // the float views follow from the integers: check them where they are cheap to predict
    for (int b = 0; b < g->dim[2]; ++b) if (g->values[b] != (float)g->counts[b]) fail("values[] is not (float)counts[]");
    return s;
}

int main(int argc, char** argv) {
    const int iterations = argc > 1 ? std::atoi(argv[1]) : 60;
    const size_t F = argc > 2 ? (size_t)std::atoi(argv[2]) : 96;
    const size_t N = argc > 3 ? (size_t)std::atoi(argv[3]) : 1500;
    std::mt19937 rng(argc > 4 ? (unsigned)std::atoi(argv[4]) : 1u);
    g_sdf = argc > 5 && !strcmp(argv[5], "sdf");
    const float L = 30.0f * std::cbrt((float)N / 1500.0f);
    if (vmd_device_count() <= 0) fail("no HIP device");
    vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
    if (!dt || !vmd_devtraj_synth(dt, 3, L, 0.05f, 0, 0, F)) fail("synthetic trajectory");
    vmd_trajectory_i* traj = vmd_devtraj_interface(dt);
    std::vector<int32_t> oxy;
    for (size_t i = 0; i < N; i += 3) oxy.push_back((int32_t)i);
    vmd_script_ir_t* ir = vmd_ir_create();
    const int32_t a[3] = {0, 1, 2}, b[3] = {30, 31, 32};
    if (!vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 9.0f)) fail("add_rdf");
    if (!vmd_ir_add_distance(ir, "d", VMD_DISTANCE_MIN, a, 3, b, 3)) fail("add_distance");
    if (g_sdf) {
        std::vector<int32_t> st;                                  // 3 reference structures of 3 atoms: the first three waters
        for (int32_t i = 0; i < 9; ++i) st.push_back(i);
        std::vector<int32_t> tgt(oxy.begin() + 3, oxy.end());
        if (!vmd_ir_add_sdf(ir, "v", st.data(), 3, 3, tgt.data(), tgt.size(), 7.0f)) fail("add_sdf");
    }
    vmd_system_t sys = {};
    sys.atom_count = N;
    vmd_set_option("readahead_company_us", 20000);

    auto reference = [&](uint32_t lo, uint32_t hi) {
        const int old = vmd_set_option("readahead", 0);
        vmd_script_eval_t* e = vmd_eval_create(F, ir);
        if (!e || !vmd_eval_frame_range(e, ir, &sys, traj, lo, hi)) fail("reference evaluation");
        Snapshot s = snapshot(e, F);
        vmd_eval_free(e);
        vmd_set_option("readahead", old);
        return s;
    };
    auto same = [&](const Snapshot& got, const Snapshot& want, const char* what) {
        if (got.done != want.done || got.mask != want.mask) { std::fprintf(stderr, "frames_done %zu, wanted %zu\n", got.done, want.done); fail(what); }
        if (got.counts != want.counts) { std::fprintf(stderr, "RDF bins differ\n"); fail(what); }
        if (got.volume != want.volume || got.volume_max != want.volume_max) { std::fprintf(stderr, "voxels differ\n"); fail(what); }
        if (got.temporal != want.temporal) { std::fprintf(stderr, "temporal rows differ\n"); fail(what); }
        for (size_t k = 0; k < got.weights.size(); ++k) {
            const double d = got.weights[k] - want.weights[k];
            if (std::abs(d) > 1e-12 * std::abs(want.weights[k])) fail(what);
        }
    };
    // calls of `grain` frames over [lo, hi) handed to nthreads threads in the order of `starts`; stop_after >= 0: interrupt once that many calls were handed out
    auto pooled = [&](vmd_script_eval_t* e, const std::vector<uint32_t>& starts, uint32_t grain, uint32_t hi, int nthreads, long stop_after) {
        std::atomic<size_t> next{0};
        std::atomic<int> failures{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= starts.size()) break;
                    if (stop_after >= 0 && (long)k == stop_after) vmd_eval_interrupt(e);
                    const uint32_t fb = starts[k], fe = std::min(hi, fb + grain);
                    if (!vmd_eval_frame_range(e, ir, &sys, traj, fb, fe)) { if (vmd_last_error()[0]) failures += 1; if (stop_after < 0) failures += 1; }
                }
            });
        for (auto& th : pool) th.join();
        return failures.load();
    };

    uint64_t regions = 0, direct = 0, blocks = 0, settles = 0, adopted = 0;
    vmd_script_eval_t* eval = vmd_eval_create(F, ir);       // reused across iterations like VIAMD reuses an eval across re-evaluations
    uint64_t lone_iterations = 0;
    for (g_iter = 0; g_iter < iterations; ++g_iter) {
        const bool lone = g_iter % 4 == 3;
        vmd_set_option("readahead_lone", lone ? 1 : 0);
        vmd_set_option("readahead_lone_settle_us", lone ? (int[]){50, 300, 2000}[rng() % 3] : 300);
        lone_iterations += lone ? 1 : 0;
        // deferred mode: what the calls asked for is final after wait_settled (now and then the helper thread gets a head start)
        auto settled = [&](vmd_script_eval_t* e) {
            if (!lone) return;
            if (rng() % 2) std::this_thread::sleep_for(std::chrono::microseconds((int[]){30, 400, 3000}[rng() % 3]));
            if (!vmd_eval_wait_settled(e)) fail("vmd_eval_wait_settled");
        };
        const int nthreads = lone && rng() % 2 ? 1 : 2 + (int)(rng() % 15);
        const uint32_t grain = (uint32_t[]){1, 1, 1, 2, 3, 5}[rng() % 6];
        uint32_t lo = 0, hi = (uint32_t)F;
        if (rng() % 3 == 0) { lo = (uint32_t)(rng() % (F / 2)); hi = lo + 1 + (uint32_t)(rng() % (F - lo)); }
        const int order = (int)(rng() % 3);
        const bool fresh = rng() % 4 == 0;                   // a new eval (new block size) now and then
        if (fresh) {
            vmd_eval_free(eval);
            vmd_set_option("readahead_block", (int[]){0, 4, 8, 16}[rng() % 4]);
            vmd_set_option("readahead_frames", (int[]){8, 16, 128}[rng() % 3]);
            vmd_set_option("readahead_growth", (int[]){1, 2, 4}[rng() % 3]);
            eval = vmd_eval_create(F, ir);
            if (!eval) fail("vmd_eval_create");
        }
        std::vector<uint32_t> starts;
        for (uint32_t f = lo; f < hi; f += grain) starts.push_back(f);
        if (order == 1) std::reverse(starts.begin(), starts.end());
        if (order == 2) std::shuffle(starts.begin(), starts.end(), rng);
        vmd_eval_clear_data(eval);
        if (rng() % 5 == 0) {                                 // interrupted somewhere, then restarted from scratch (src/main.cpp:984-990)
            if (pooled(eval, starts, grain, hi, nthreads, (long)(rng() % starts.size())) != 0) fail("an interrupted evaluation reported an error");
            if (vmd_eval_frames_done(eval) > (size_t)(hi - lo)) fail("an interrupted evaluation counted frames nobody asked for");
            vmd_eval_clear_data(eval);
        }
        const bool split = rng() % 4 == 0 && hi - lo > 8;     // the second half of the range arrives as ONE large call while the pool works on the first
        const uint32_t mid = split ? lo + (hi - lo) / 2 / grain * grain : hi;
        if (split) starts.erase(std::remove_if(starts.begin(), starts.end(), [&](uint32_t f) { return f >= mid; }), starts.end());
        std::thread big;
        std::atomic<int> big_ok{1};
        if (split) big = std::thread([&] { if (!vmd_eval_frame_range(eval, ir, &sys, traj, mid, hi)) big_ok = 0; });
        // "Eval Full" and "Eval Filt" side by side (src/main.cpp:982-1039 enqueues them as two pool tasks): a second eval with the full one as
        // its source walks a sub-range WHILE the full one is being evaluated - it may adopt whatever blocks the source has finished by then
        // and evaluates the rest itself
        const bool beside = g_iter % 3 == 1 && hi - lo > 6;
        vmd_script_eval_t* side = nullptr;
        std::thread side_pool;
        std::atomic<int> side_failures{0};
        bool side_interrupted = false;
        uint32_t slo = 0, shi = 0;
        std::vector<uint32_t> sstarts;
        if (beside) {
            slo = lo + (uint32_t)(rng() % ((hi - lo) / 2)); shi = slo + 1 + (uint32_t)(rng() % (hi - slo));
            side = vmd_eval_create(F, ir);
            if (!side || !vmd_eval_set_source(side, eval)) fail("side-by-side filtered eval");
            for (uint32_t f = slo; f < shi; f += grain) sstarts.push_back(f);
            const int sthreads = 1 + (int)(rng() % 4);
            // now and then the slider moves on before the filtered evaluation is through: interrupted midway, restarted once the full one is done
            const long side_stop = rng() % 3 == 0 ? (long)(rng() % sstarts.size()) : -1;
            side_pool = std::thread([&, sthreads, side_stop] { side_failures = pooled(side, sstarts, grain, shi, sthreads, side_stop); });
            side_interrupted = side_stop >= 0;
        }
        // ... or the script is edited while both run: the full evaluation is interrupted and started over, the filtered one carries on beside it
        const bool full_restart = beside && !split && rng() % 3 == 0;
        if (full_restart) {
            if (pooled(eval, starts, grain, mid, nthreads, (long)(rng() % starts.size())) != 0) fail("an interrupted evaluation (beside a filtered one) reported an error");
            // (VIAMD waits for the interrupted task before it clears; the filtered eval must cope with its source's blocks vanishing)
            vmd_eval_clear_data(eval);
        }
        if (pooled(eval, starts, grain, mid, nthreads, -1) != 0) fail("a call failed");
        if (split) { big.join(); if (!big_ok) fail("the large call failed"); }
        if (beside) {
            side_pool.join();
            if (side_failures.load()) fail("a call of the side-by-side filtered evaluation failed");
            if (side_interrupted) {
                if (vmd_eval_frames_done(side) > (size_t)(shi - slo)) fail("an interrupted filtered evaluation counted frames nobody asked for");
                vmd_eval_clear_data(side);
                if (pooled(side, sstarts, grain, shi, 1 + (int)(rng() % 4), -1) != 0) fail("a call of the restarted filtered evaluation failed");
            }
            settled(side);
            same(snapshot(side, F), reference(slo, shi), "filtered evaluation running beside its source differs from one call over its range");
            size_t computed = 0, reused = 0;
            vmd_eval_frame_stats(side, &computed, &reused);
            adopted += reused;
            vmd_eval_free(side);
        }
        settled(eval);
        same(snapshot(eval, F), reference(lo, hi), "pooled evaluation differs from one call over the same range");
        if (g_iter % 3 == 2 && hi - lo > 6) {
            // the timeline slider: a second eval over a sub-range, the full one (just evaluated over [lo, hi)) as its source
            const uint32_t flo = lo + (uint32_t)(rng() % ((hi - lo) / 2)), fhi = flo + 1 + (uint32_t)(rng() % (hi - flo));
            vmd_script_eval_t* filt = vmd_eval_create(F, ir);
            if (!filt || !vmd_eval_set_source(filt, eval)) fail("filtered eval");
            std::vector<uint32_t> fstarts;
            for (uint32_t f = flo; f < fhi; f += grain) fstarts.push_back(f);
            if (rng() % 2) std::reverse(fstarts.begin(), fstarts.end());
            if (pooled(filt, fstarts, grain, fhi, nthreads, -1) != 0) fail("a call of the filtered evaluation failed");
            settled(filt);
            same(snapshot(filt, F), reference(flo, fhi), "filtered evaluation (source = the full eval) differs from one call over its range");
            size_t computed = 0, reused = 0;
            vmd_eval_frame_stats(filt, &computed, &reused);
            adopted += reused;
            vmd_eval_free(filt);
        }
        vmd_readahead_stats_t st;
        vmd_eval_readahead_stats(eval, &st);
        regions = st.regions; direct = st.direct_frames; blocks = st.committed_blocks; settles = st.settles;
    }
    vmd_set_option("readahead_lone", 0);
    vmd_eval_free(eval); vmd_ir_free(ir); vmd_devtraj_free(dt);
    std::printf("OK iterations=%d (%llu in deferred-settle mode) frames=%zu%s (last eval: %llu regions, %llu blocks committed, %llu frames evaluated directly, %llu settles; filtered evals adopted %llu frames from their source)\n",
                iterations, (unsigned long long)lone_iterations, F, g_sdf ? " +sdf" : "", (unsigned long long)regions, (unsigned long long)blocks, (unsigned long long)direct, (unsigned long long)settles, (unsigned long long)adopted);
    return 0;
}
