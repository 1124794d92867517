// A C++ host that goes from files and script text to results through the C ABI alone: DCD / XTC / TRR trajectory chosen by
// the file extension like VIAMD's loader does (vmd_dcdtraj_open, vmd_xdrtraj_open, vmd_texttraj_open; src/loader.cpp:40-56, 79-84),
// script front-end (vmd_ir_compile_from_source), evaluator (vmd_eval_*), consumer post-processing (vmd_downsample_histogram).
// usage: cabi_script_demo <trajectory.dcd|.xtc|.trr|.pdb|.xyz|.lammpstrj> <n_blob_atoms> "<script>"
// Topology: the synthetic system of viamd_amd/synth.py ([ALA-like residues of 10 atoms][O,H,H waters]).
// Prints one line per property: name, flags, dim, sum of the integer accumulators (or of the temporal values).
// tests/test_native.py links it against the SIMT-emulator build (CPU) and compares with the Python host.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vmd_eval.h"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s traj.dcd n_blob script\n", argv[0]); return 2; }
    const char* ext = strrchr(argv[1], '.');
    vmd_dcdtraj_t* dcd = nullptr;
    vmd_xdrtraj_t* xdr = nullptr;
    vmd_texttraj_t* txt = nullptr;
    if (ext && !strcmp(ext, ".dcd")) dcd = vmd_dcdtraj_open(argv[1]);
    else if (ext && (!strcmp(ext, ".xtc") || !strcmp(ext, ".trr"))) xdr = vmd_xdrtraj_open(argv[1]);
    else if (ext && (!strcmp(ext, ".pdb") || !strcmp(ext, ".xyz") || !strcmp(ext, ".xmol") || !strcmp(ext, ".lammpstrj"))) txt = vmd_texttraj_open(argv[1], nullptr);
    else { fprintf(stderr, "could not determine loader type from file extension\n"); return 2; }
    if (!dcd && !xdr && !txt) { fprintf(stderr, "open failed: %s\n", vmd_last_error()); return 1; }
    vmd_trajectory_i* traj = dcd ? vmd_dcdtraj_interface(dcd) : (xdr ? vmd_xdrtraj_interface(xdr) : vmd_texttraj_interface(txt));
    const size_t n = traj->num_atoms(traj->inst), frames = traj->num_frames(traj->inst);
    const size_t n_blob = (size_t)atol(argv[2]);

    // topology (what md_system_t carries in VIAMD)
    static const char* blob_elems[10] = {"N", "C", "C", "O", "C", "H", "H", "H", "C", "H"};
    std::vector<const char*> elements(n), resnames(n);
    std::vector<int32_t> resid(n);
    std::vector<float> mass(n);
    const size_t n_blob_res = (n_blob + 9) / 10;
    for (size_t i = 0; i < n; ++i) {
        if (i < n_blob) { elements[i] = blob_elems[i % 10]; resnames[i] = "ALA"; resid[i] = (int32_t)(i / 10); }
        else { const size_t w = i - n_blob; elements[i] = w % 3 == 0 ? "O" : "H"; resnames[i] = "HOH"; resid[i] = (int32_t)(n_blob_res + w / 3); }
        const char e = elements[i][0];
        mass[i] = e == 'C' ? 12.011f : e == 'N' ? 14.007f : e == 'O' ? 15.999f : 1.008f;
    }
    vmd_topology_t topo = {n, elements.data(), nullptr, resnames.data(), resid.data()};

    vmd_script_ir_t* ir = vmd_ir_create();
    if (!vmd_ir_compile_from_source(ir, argv[3], &topo)) { fprintf(stderr, "script error: %s\n", vmd_last_error()); return 1; }

    vmd_frame_header_t hdr;
    std::vector<float> xyz(3 * n);
    if (!traj->load_frame(traj->inst, 0, &hdr, xyz.data(), xyz.data() + n, xyz.data() + 2 * n)) { fprintf(stderr, "%s\n", vmd_last_error()); return 1; }
    vmd_system_t sys = {n, xyz.data(), xyz.data() + n, xyz.data() + 2 * n, mass.data(), hdr.unitcell};

    vmd_script_eval_t* ev = vmd_eval_create(frames, ir);
    if (!ev) { fprintf(stderr, "eval: %s\n", vmd_last_error()); return 1; }
    if (!vmd_eval_frame_range(ev, ir, &sys, traj, 0, (uint32_t)frames)) { fprintf(stderr, "frame_range: %s\n", vmd_last_error()); return 1; }

    const char* const* names = vmd_ir_property_names(ir);
    for (size_t p = 0; p < vmd_ir_property_count(ir); ++p) {
        const vmd_script_property_data_t* d = vmd_eval_property_data(ev, names[p]);
        const vmd_property_flags_t fl = vmd_ir_property_flags(ir, names[p]);
        double sum = 0.0;
        if (fl & VMD_PROPERTY_FLAG_TEMPORAL) {
            for (size_t i = 0; i < d->num_values; ++i) sum += d->values[i];
        } else {
            vmd_eval_refresh_counts(ev, names[p]);
            const size_t nc = (fl & VMD_PROPERTY_FLAG_VOLUME) ? (size_t)d->dim[1] * d->dim[2] * d->dim[3] : (size_t)d->dim[2];
            for (size_t i = 0; i < nc; ++i) sum += (double)d->counts[i];
        }
        printf("%s flags=%u dim=%d,%d,%d,%d sum=%.9g", names[p], fl, d->dim[0], d->dim[1], d->dim[2], d->dim[3], sum);
        if (fl & VMD_PROPERTY_FLAG_DISTRIBUTION) {
            float g[8];
            vmd_downsample_histogram(g, 8, d->values, d->weights, d->dim[2]);      // what VIAMD plots (src/main.cpp:232-250)
            printf(" g8=%.6g,%.6g", g[6], g[7]);
        }
        printf("\n");
    }
    vmd_eval_free(ev);
    vmd_ir_free(ir);
    vmd_dcdtraj_close(dcd);
    vmd_xdrtraj_close(xdr);
    vmd_texttraj_close(txt);
    return 0;
}
