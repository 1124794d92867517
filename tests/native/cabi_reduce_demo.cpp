// tests/native/cabi_reduce_demo.cpp — a C++ host (no Python, no torch) doing the multi-GPU evaluation of SURVEY.md 8e through the
// C ABI only: every rank = one process = one GPU evaluates its block of frames of a frame-sharded device trajectory with
// vmd_eval_frame_range, then ONE vmd_eval_reduce (RCCL all-reduce, in place on the device accumulators) merges.  The merged
// result must equal an evaluation of the whole trajectory on one GPU bit for bit (integer parts) - each rank checks that itself.
//
//   cabi_reduce_demo <nranks> <rank> <id-file> [frames]
// rank 0 writes the 128-byte RCCL id to <id-file>, the other ranks wait for it (a stand-in for whatever the host program uses
// to rendezvous: MPI, a socket, ...).  Prints "OK ranks=<n> rank=<r> hits=<sum of merged RDF counts> voxels=<...>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "vmd_eval.h"

static void fail(const char* what) {
    std::fprintf(stderr, "FAIL: %s (%s)\n", what, vmd_last_error());
    std::exit(1);
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s nranks rank id-file [frames]\n", argv[0]); return 2; }
    const int nranks = std::atoi(argv[1]), rank = std::atoi(argv[2]);
    const char* idfile = argv[3];
    const size_t F = argc > 4 ? (size_t)std::atoi(argv[4]) : 24;
    const size_t N = 30000;
    const float L = 70.0f;
    const int ndev = vmd_device_count();
    if (ndev <= 0) fail("no HIP device");
    if (!vmd_set_device(rank % ndev)) fail("set_device");

    // rendezvous of the communicator id
    uint8_t id[VMD_COMM_ID_BYTES];
    if (rank == 0) {
        if (!vmd_comm_unique_id(id)) fail("vmd_comm_unique_id");
        std::string tmp = std::string(idfile) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) fail("write id file");
        std::fclose(f);
        std::rename(tmp.c_str(), idfile);
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 3000 && !f; ++tries) { f = std::fopen(idfile, "rb"); if (!f) std::this_thread::sleep_for(std::chrono::milliseconds(10)); }
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) fail("read id file");
        std::fclose(f);
    }
    vmd_comm_t* comm = vmd_comm_create(nranks, rank, id);
    if (!comm) fail("vmd_comm_create");
    if (vmd_comm_size(comm) != nranks || vmd_comm_rank(comm) != rank) fail("communicator rank / size");

    // the script: an RDF, an SDF around 3 reference structures, a distance row
    std::vector<int32_t> oxy, structs;
    for (size_t i = 30; i < N; i += 3) oxy.push_back((int32_t)i);
    for (int32_t i = 0; i < 30; ++i) structs.push_back(i);             // 3 structures x 10 atoms
    vmd_script_ir_t* ir = vmd_ir_create();
    if (!vmd_ir_add_rdf(ir, "g", oxy.data(), oxy.size(), oxy.data(), oxy.size(), 0.0f, 10.0f)) fail("add_rdf");
    if (!vmd_ir_add_sdf(ir, "v", structs.data(), 3, 10, oxy.data(), oxy.size(), 8.0f)) fail("add_sdf");
    const int32_t a = 3, b = 600;
    if (!vmd_ir_add_distance(ir, "d", VMD_DISTANCE_COM, &a, 1, &b, 1)) fail("add_distance");
    vmd_system_t sys = {};
    sys.atom_count = N;

    // this rank's shard (contiguous block of ceil(F / nranks) frames) ...
    const size_t per = (F + nranks - 1) / nranks;
    const size_t beg = std::min(F, rank * per), end = std::min(F, beg + per);
    vmd_devtraj_t* shard = vmd_devtraj_create_shard(F, beg, end, N);
    if (!shard || !vmd_devtraj_synth(shard, 11, L, 0.05f, 0, beg, end)) fail("sharded trajectory");
    vmd_script_eval_t* eval = vmd_eval_create(F, ir);
    if (!eval) fail("eval_create");
    vmd_eval_clear_data(eval);
    if (beg < end && !vmd_eval_frame_range(eval, ir, &sys, vmd_devtraj_interface(shard), (uint32_t)beg, (uint32_t)end)) fail("frame_range (shard)");
    if (vmd_eval_frames_done(eval) != end - beg) fail("frames_done before the merge");
    if (!vmd_eval_reduce(eval, vmd_comm_collective(comm), nullptr)) fail("vmd_eval_reduce");
    if (vmd_eval_frames_done(eval) != F) fail("frames_done after the merge");

    // ... against the whole trajectory on this GPU alone
    vmd_devtraj_t* whole = vmd_devtraj_create(F, N);
    if (!whole || !vmd_devtraj_synth(whole, 11, L, 0.05f, 0, 0, F)) fail("whole trajectory");
    vmd_script_eval_t* ref = vmd_eval_create(F, ir);
    vmd_eval_clear_data(ref);
    if (!vmd_eval_frame_range(ref, ir, &sys, vmd_devtraj_interface(whole), 0, (uint32_t)F)) fail("frame_range (whole)");

    const vmd_script_property_data_t* g = vmd_eval_property_data(eval, "g");
    const vmd_script_property_data_t* g0 = vmd_eval_property_data(ref, "g");
    unsigned long long hits = 0, voxels = 0;
    for (int k = 0; k < g->dim[2]; ++k) {
        if (g->counts[k] != g0->counts[k] || g->values[k] != g0->values[k]) fail("merged RDF counts differ from the single-GPU evaluation");
        const double w = g->weights64[k], w0 = g0->weights64[k];
        if (!(std::abs(w - w0) <= 1e-12 * std::abs(w0))) fail("merged RDF weights differ");
        hits += g->counts[k];
    }
    if (!vmd_eval_refresh_counts(eval, "v") || !vmd_eval_refresh_counts(ref, "v")) fail("refresh_counts");
    const vmd_script_property_data_t* v = vmd_eval_property_data(eval, "v");
    const vmd_script_property_data_t* v0 = vmd_eval_property_data(ref, "v");
    const size_t nvox = (size_t)v->dim[1] * v->dim[2] * v->dim[3];
    for (size_t i = 0; i < nvox; ++i) {
        if (v->counts[i] != v0->counts[i] || v->values[i] != v0->values[i]) fail("merged SDF volume differs");
        voxels += v->counts[i];
    }
    if (v->max_value != v0->max_value) fail("merged SDF max_value differs");
    const vmd_script_property_data_t* d = vmd_eval_property_data(eval, "d");
    const vmd_script_property_data_t* d0 = vmd_eval_property_data(ref, "d");
    for (size_t f = 0; f < F; ++f) if (d->values[f] != d0->values[f] || !(d->values[f] > 0.0f)) fail("merged distance rows differ");
    if (hits == 0 || voxels == 0) fail("empty result");

    std::printf("OK ranks=%d rank=%d frames=[%zu,%zu) hits=%llu voxels=%llu\n", nranks, rank, beg, end, hits, voxels);
    vmd_eval_free(eval); vmd_eval_free(ref);
    vmd_devtraj_free(shard); vmd_devtraj_free(whole);
    vmd_ir_free(ir);
    vmd_comm_destroy(comm);
    return 0;
}
