// How md_script_eval_frame_range is really called (src/main.cpp:993-997: a task over [0, num_frames) split by enkiTS into small ranges, one
// call per range from every pool thread, all on the SAME eval) against one call over the whole range: the cost of the combining queue.
// usage: exp_threads [atoms] [frames]      (build line: tests/test_native.py / scripts/gpu_r03am.sh)
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "vmd_eval.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t N = argc > 1 ? (size_t)std::atol(argv[1]) : 100002, F = argc > 2 ? (size_t)std::atol(argv[2]) : 1000;
    const float L = 100.0f * std::cbrt((float)N / 100002.0f);
    if (vmd_device_count() <= 0) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    if (const char* o = std::getenv("VMD_OPTS")) {            // "key=value key=value": library options for an A/B
        std::string s(o);
        size_t p = 0;
        while (p < s.size()) {
            const size_t q = s.find(' ', p), e = s.find('=', p);
            const std::string kv = s.substr(p, q == std::string::npos ? std::string::npos : q - p);
            if (e != std::string::npos && e < p + kv.size()) vmd_set_option(s.substr(p, e - p).c_str(), std::atoi(s.c_str() + e + 1));
            if (q == std::string::npos) break;
            p = q + 1;
        }
    }
    vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
    if (!dt || !vmd_devtraj_synth(dt, 2, L, 0.05f, 0, 0, F)) { std::fprintf(stderr, "synth: %s\n", vmd_last_error()); return 1; }
    vmd_trajectory_i* traj = vmd_devtraj_interface(dt);
    std::vector<int32_t> oxy;
    for (size_t i = 0; i < N; i += 3) oxy.push_back((int32_t)i);
    vmd_script_ir_t* ir = vmd_ir_create();
    const bool sdf = std::getenv("VMD_SDF") != nullptr;         // the SDF of BASELINE config 4 instead of the O-O RDF
    if (sdf) {
        std::vector<int32_t> st;                                  // 7 reference structures of 9 atoms: the first 21 waters, three at a time
        for (int32_t i = 0; i < 63; ++i) st.push_back(i);
        std::vector<int32_t> tgt(oxy.begin() + 21, oxy.end());
        if (!vmd_ir_add_sdf(ir, "v", st.data(), 7, 9, tgt.data(), tgt.size(), 10.0f)) { std::fprintf(stderr, "%s\n", vmd_last_error()); return 1; }
    } else
    if (!vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 12.0f)) return 1;
    vmd_script_eval_t* eval = vmd_eval_create(F, ir);
    vmd_system_t sys = {};
    sys.atom_count = N;
    std::vector<float> ref_values;
    auto one = [&]() {
        vmd_eval_clear_data(eval);
        const double t = now_ms();
        if (!vmd_eval_frame_range(eval, ir, &sys, traj, 0, (uint32_t)F)) { std::fprintf(stderr, "%s\n", vmd_last_error()); std::exit(1); }
        return now_ms() - t;
    };
    auto pooled = [&](int nthreads, uint32_t grain) {
        vmd_eval_clear_data(eval);
        std::atomic<uint32_t> next{0};
        std::vector<std::thread> pool;
        const double t = now_ms();
        for (int k = 0; k < nthreads; ++k)
            pool.emplace_back([&] {
                for (;;) {
                    const uint32_t beg = next.fetch_add(grain);
                    if (beg >= F) break;
                    if (!vmd_eval_frame_range(eval, ir, &sys, traj, beg, beg + grain < F ? beg + grain : (uint32_t)F)) break;
                }
            });
        for (auto& th : pool) th.join();
        const double ms = now_ms() - t;
        if (vmd_eval_frames_done(eval) != F) { std::fprintf(stderr, "frames missing\n"); std::exit(1); }
        // the integers of the pooled evaluation are the integers of the one call (RDF bins; an SDF volume is compared through its float view)
        const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, sdf ? "v" : "g");
        if (ref_values.size() == pd->num_values && memcmp(ref_values.data(), pd->values, pd->num_values * sizeof(float)) != 0) {
            std::fprintf(stderr, "pooled evaluation (%d threads, grain %u) differs from the one call\n", nthreads, grain); std::exit(1);
        }
        return ms;
    };
    // the order enkiTS really produces (TaskScheduler::AddTaskSetToPipe): the set is cut into threads x (threads - 1) partitions of at least one
    // grain; the thread that added the set pops partitions off the END of its pipe, every other thread steals from the FRONT
    auto pooled_enki = [&](int nthreads) {
        vmd_eval_clear_data(eval);
        const uint32_t parts = (uint32_t)std::max(1, nthreads * (nthreads - 1));
        const uint32_t per = std::max<uint32_t>(1, (uint32_t)F / parts);
        int64_t front = 0, back = (int64_t)((F + per - 1) / per) - 1;
        std::mutex pipe;
        std::vector<std::thread> pool;
        const double t = now_ms();
        for (int k = 0; k < nthreads; ++k)
            pool.emplace_back([&, k] {
                for (;;) {
                    int64_t i;
                    { std::lock_guard<std::mutex> l(pipe); if (front > back) break; i = k == 0 ? back-- : front++; }
                    const uint32_t beg = (uint32_t)i * per, end = std::min<uint32_t>((uint32_t)F, beg + per);
                    if (!vmd_eval_frame_range(eval, ir, &sys, traj, beg, end)) break;
                }
            });
        for (auto& th : pool) th.join();
        const double ms = now_ms() - t;
        const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, sdf ? "v" : "g");
        const bool same = ref_values.size() == pd->num_values && memcmp(ref_values.data(), pd->values, pd->num_values * sizeof(float)) == 0;
        return std::make_pair(ms, same && vmd_eval_frames_done(eval) == F);
    };
    one();
    { const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, sdf ? "v" : "g"); ref_values.assign(pd->values, pd->values + pd->num_values); }
    double best = 1e30;
    for (int r = 0; r < 3; ++r) best = std::min(best, one());
    std::printf("%s atoms %zu frames %zu: one call %.2f ms", sdf ? "sdf" : "rdf", N, F, best);
    const int cfg[][2] = {{16, 1}, {16, 4}, {16, 16}, {16, 64}, {128, 1}, {4, 1}, {1, 1}};
    for (auto& c : cfg) {
        double b = 1e30;
        for (int r = 0; r < 3; ++r) b = std::min(b, pooled(c[0], (uint32_t)c[1]));
        std::printf(" | %d threads grain %d: %.2f ms", c[0], c[1], b);
    }
    {
        double b = 1e30; int clean = 0;
        for (int r = 0; r < 5; ++r) { auto pr = pooled_enki(16); if (pr.second) { b = std::min(b, pr.first); ++clean; } }
        std::printf(" | 16 threads, enkiTS order (threads x (threads - 1) partitions, owner from the end, thieves from the front; %d of 5 runs identical to the one call): %.2f ms", clean, b);
    }
    {
        // the lone caller once more, with the deferred settle switched on (vmd_set_option("readahead_lone", 1)): the time includes
        // vmd_eval_wait_settled, i.e. it ends when totals and views are final, as the other columns' do
        vmd_set_option("readahead_lone", 1);
        double b = 1e30;
        for (int r = 0; r < 3; ++r) {
            vmd_eval_clear_data(eval);
            const double t = now_ms();
            for (uint32_t f = 0; f < F; ++f) if (!vmd_eval_frame_range(eval, ir, &sys, traj, f, f + 1)) { std::fprintf(stderr, "%s\n", vmd_last_error()); std::exit(1); }
            const double t_calls = now_ms() - t;
            if (!vmd_eval_wait_settled(eval)) { std::fprintf(stderr, "%s\n", vmd_last_error()); std::exit(1); }
            const double ms = now_ms() - t;
            const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, sdf ? "v" : "g");
            if (vmd_eval_frames_done(eval) != F || ref_values.size() != pd->num_values || memcmp(ref_values.data(), pd->values, pd->num_values * sizeof(float)) != 0) {
                std::fprintf(stderr, "lone caller with deferred settle differs from the one call\n"); std::exit(1);
            }
            if (ms < b) { b = ms; std::printf(" | 1 thread grain 1, readahead_lone (calls %.2f ms + settle): %.2f ms", t_calls, ms); }
        }
        vmd_set_option("readahead_lone", 0);
    }
    std::printf("\n");
    vmd_readahead_stats_t st;
    vmd_eval_readahead_stats(eval, &st);
    std::printf("  read-ahead over all pooled runs: engaged %u, blocks of %u frames, %llu regions (%llu frames), %llu calls led or waited (the others only marked), %llu settles, %llu frames evaluated directly, %llu blocks committed\n",
                st.engaged, st.block_frames, (unsigned long long)st.regions, (unsigned long long)st.region_frames, (unsigned long long)st.slow_calls,
                (unsigned long long)st.settles, (unsigned long long)st.direct_frames, (unsigned long long)st.committed_blocks);
    vmd_eval_free(eval); vmd_ir_free(ir); vmd_devtraj_free(dt);
    return 0;
}
