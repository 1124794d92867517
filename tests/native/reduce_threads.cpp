// Several ranks driven from THREADS of one process, merged with vmd_eval_reduce through a host-supplied collective (vmd_collective_i):
// the set-up ADVICE r03 named - with one staging buffer per device behind one mutex, rank A sat in the collective waiting for a rank B
// that waited for A's buffer.  Here every rank is a thread with its own eval over its own block of frames; the collective is an
// in-process rendezvous that sums the ranks' buffers.  After the merge every rank must hold, bit for bit, what one eval over all frames holds.
// EMULATOR ONLY (tests/test_native.py, scripts/tsan_emu.sh): the rendezvous adds "device" buffers with host loops, which is what device
// memory is on the emulator; on hardware the multi-rank path is covered by cabi_reduce_demo.cpp (one process per rank, RCCL).
// usage: reduce_threads [ranks] [frames] [atoms]
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "vmd_eval.h"

static void fail(const char* what) { std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error()); std::exit(1); }

// all ranks arrive with a pointer; the last one adds them up and writes the sum back to every rank; all leave together
struct Hub {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0;
    long generation = 0;
    std::vector<void*> bufs;
    template <class T> bool allreduce(int rank, T* buf, size_t count) {
        std::unique_lock<std::mutex> l(m);
        bufs[(size_t)rank] = buf;
        const long gen = generation;
        if (++arrived == n) {
            std::vector<T> sum(count, T(0));
            for (int r = 0; r < n; ++r) { const T* b = (const T*)bufs[(size_t)r]; for (size_t i = 0; i < count; ++i) sum[i] += b[i]; }
            for (int r = 0; r < n; ++r) std::memcpy(bufs[(size_t)r], sum.data(), count * sizeof(T));
            arrived = 0;
            generation += 1;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != gen; });
        }
        return true;
    }
};
struct RankComm { Hub* hub; int rank; };
static int c_rank(void* i) { return ((RankComm*)i)->rank; }
static int c_size(void* i) { return ((RankComm*)i)->hub->n; }
static bool c_u64(void* i, uint64_t* b, size_t n, void*) { return ((RankComm*)i)->hub->allreduce(((RankComm*)i)->rank, b, n); }
static bool c_f64(void* i, double* b, size_t n, void*) { return ((RankComm*)i)->hub->allreduce(((RankComm*)i)->rank, b, n); }
static bool c_u32(void* i, uint32_t* b, size_t n, void*) { return ((RankComm*)i)->hub->allreduce(((RankComm*)i)->rank, b, n); }

int main(int argc, char** argv) {
    const int R = argc > 1 ? std::atoi(argv[1]) : 3;
    const size_t F = argc > 2 ? (size_t)std::atol(argv[2]) : 12, N = argc > 3 ? (size_t)std::atol(argv[3]) : 600;
    if (vmd_device_count() <= 0) fail("no device");
    const float L = 28.0f;
    vmd_devtraj_t* dt = vmd_devtraj_create(F, N);
    if (!dt || !vmd_devtraj_synth(dt, 5, L, 0.05f, 0, 0, F)) fail("synth");
    vmd_trajectory_i* traj = vmd_devtraj_interface(dt);
    std::vector<int32_t> oxy, st, tgt;
    for (size_t i = 0; i < N; i += 3) oxy.push_back((int32_t)i);
    for (int32_t i = 0; i < 18; ++i) st.push_back(i);                     // 2 reference structures of 9 atoms
    tgt.assign(oxy.begin() + 6, oxy.end());
    vmd_script_ir_t* ir = vmd_ir_create();
    if (!vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 9.0f)) fail("rdf");
    if (!vmd_ir_add_sdf(ir, "v", st.data(), 2, 9, tgt.data(), tgt.size(), 8.0f)) fail("sdf");
    vmd_system_t sys = {};
    sys.atom_count = N;

    // the answer: one eval, one call
    vmd_script_eval_t* whole = vmd_eval_create(F, ir);
    if (!whole || !vmd_eval_frame_range(whole, ir, &sys, traj, 0, (uint32_t)F)) fail("whole");
    if (!vmd_eval_refresh_counts(whole, "v")) fail("refresh");
    const vmd_script_property_data_t* wg = vmd_eval_property_data(whole, "g");
    const vmd_script_property_data_t* wv = vmd_eval_property_data(whole, "v");
    const size_t nv = (size_t)wv->dim[1] * (size_t)wv->dim[2] * (size_t)wv->dim[3];

    Hub hub;
    hub.n = R;
    hub.bufs.assign((size_t)R, nullptr);
    std::vector<RankComm> comm((size_t)R);
    std::vector<vmd_script_eval_t*> evals((size_t)R, nullptr);
    std::atomic<int> bad{0};
    for (int mode = 0; mode < 2; ++mode) {                               // 0: static u32 narrowing, 1: VIAMD_AMD_REDUCE_MEASURE=1 (two collective phases)
        setenv("VIAMD_AMD_REDUCE_MEASURE", mode ? "1" : "0", 1);
        std::vector<std::thread> ranks;
        for (int r = 0; r < R; ++r)
            ranks.emplace_back([&, r] {
                comm[(size_t)r] = RankComm{&hub, r};
                vmd_collective_i coll;
                std::memset(&coll, 0, sizeof(coll));
                coll.inst = &comm[(size_t)r];
                coll.rank = c_rank; coll.size = c_size;
                coll.allreduce_sum_u64 = c_u64; coll.allreduce_sum_f64 = c_f64; coll.allreduce_sum_u32 = c_u32;
                vmd_script_eval_t* e = vmd_eval_create(F, ir);
                evals[(size_t)r] = e;
                const uint32_t beg = (uint32_t)(F * (size_t)r / (size_t)R), end = (uint32_t)(F * (size_t)(r + 1) / (size_t)R);
                bool ok = e != nullptr;
                if (ok && end > beg) ok = vmd_eval_frame_range(e, ir, &sys, traj, beg, end);
                // (a rank that failed must still enter the collective, or the others wait for ever: it merges what it has)
                if (e && !vmd_eval_reduce(e, &coll, nullptr)) ok = false;
                if (ok) ok = vmd_eval_refresh_counts(e, "v");
                if (ok) {
                    const vmd_script_property_data_t* g = vmd_eval_property_data(e, "g");
                    const vmd_script_property_data_t* v = vmd_eval_property_data(e, "v");
                    const char* what = nullptr;
                    if (std::memcmp(g->counts, wg->counts, (size_t)wg->dim[2] * sizeof(uint64_t)) != 0) what = "rdf counts";
                    else if (std::memcmp(v->counts, wv->counts, nv * sizeof(uint64_t)) != 0) what = "voxel counts";
                    else if (std::memcmp(v->values, wv->values, nv * sizeof(float)) != 0) what = "voxel float view";
                    else if (vmd_eval_frames_done(e) != F) what = "frames_done";
                    // the fp64 normalisation weights are sums of per-frame terms: another association across ranks, equal to rounding
                    for (int b = 0; b < wg->dim[2] && !what; ++b)
                        if (std::fabs(g->weights64[b] - wg->weights64[b]) > 1e-12 * std::fabs(wg->weights64[b])) what = "rdf weights";
                    if (what) { std::fprintf(stderr, "rank %d (mode %d): %s differ\n", r, mode, what); ok = false; }
                }
                if (!ok) { std::fprintf(stderr, "rank %d (mode %d): %s\n", r, mode, vmd_last_error()); bad += 1; }
            });
        for (auto& t : ranks) t.join();
        for (auto& e : evals) { vmd_eval_free(e); e = nullptr; }
        if (bad.load()) fail("a rank's merged result differs from the one-eval answer");
    }
    uint64_t hits = 0, vox = 0;
    for (int b = 0; b < wg->dim[2]; ++b) hits += wg->counts[b];
    for (size_t i = 0; i < nv; ++i) vox += wv->counts[i];
    std::printf("OK ranks=%d frames=%zu rdf_hits=%llu voxel_hits=%llu (ranks as threads of one process, in-process collective, both narrowing paths)\n", R, F,
                (unsigned long long)hits, (unsigned long long)vox);
    vmd_eval_free(whole); vmd_ir_free(ir); vmd_devtraj_free(dt);
    return 0;
}
