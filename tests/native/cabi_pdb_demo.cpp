// tests/native/cabi_pdb_demo.cpp - BASELINE configs[0] the way a host without mdlib and without Python runs it: a multi-MODEL PDB file and a
// script string, nothing else.  System (elements, residues, masses, cell) and trajectory both come out of the file through the C ABI
// (vmd_textsys_open / vmd_texttraj_open; VIAMD: md_pdb_system_init_from_file + the PDB trajectory loader, src/loader.cpp:113-128), the
// script is compiled by vmd_ir_compile_from_source (src/main.cpp:878) and evaluated over all frames.
// usage: cabi_pdb_demo <file.pdb> "<script>"     prints one line per property: name, dim, the sum of its integer accumulators / values
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vmd_eval.h"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s file.pdb script\n", argv[0]); return 2; }
    vmd_textsys_t* sysf = vmd_textsys_open(argv[1]);
    vmd_texttraj_t* trj = vmd_texttraj_open(argv[1], nullptr);
    if (!sysf || !trj) { fprintf(stderr, "open failed: %s\n", vmd_last_error()); return 1; }
    const vmd_topology_t* topo = vmd_textsys_topology(sysf);
    vmd_trajectory_i* traj = vmd_texttraj_interface(trj);
    const size_t n = topo->num_atoms, frames = traj->num_frames(traj->inst);
    if (traj->num_atoms(traj->inst) != n) { fprintf(stderr, "system and trajectory disagree on the atom count\n"); return 1; }
    vmd_script_ir_t* ir = vmd_ir_create();
    if (!vmd_ir_compile_from_source(ir, argv[2], topo)) { fprintf(stderr, "script error: %s\n", vmd_last_error()); return 1; }
    vmd_unitcell_t cell;
    const float* xyz = vmd_textsys_coords(sysf, &cell);
    vmd_system_t sys = {n, xyz, xyz + n, xyz + 2 * n, vmd_textsys_mass(sysf), cell};
    vmd_script_eval_t* ev = vmd_eval_create(frames, ir);
    if (!ev || !vmd_eval_frame_range(ev, ir, &sys, traj, 0, (uint32_t)frames)) { fprintf(stderr, "evaluation: %s\n", vmd_last_error()); return 1; }
    printf("atoms=%zu frames=%zu residues=%d first=%s/%s/%s last_mass=%.3f cell=%.3f,%.3f,%.3f\n", n, frames, topo->residue_index[n - 1] + 1, topo->elements[0], topo->names[0],
           topo->resnames[0], vmd_textsys_mass(sysf)[n - 1], cell.x, cell.y, cell.z);
    const char* const* names = vmd_ir_property_names(ir);
    for (size_t p = 0; p < vmd_ir_property_count(ir); ++p) {
        const vmd_script_property_data_t* d = vmd_eval_property_data(ev, names[p]);
        const vmd_property_flags_t fl = vmd_ir_property_flags(ir, names[p]);
        double sum = 0.0;
        if (fl & VMD_PROPERTY_FLAG_TEMPORAL) for (size_t i = 0; i < d->num_values; ++i) sum += d->values[i];
        else {
            vmd_eval_refresh_counts(ev, names[p]);
            const size_t nc = (fl & VMD_PROPERTY_FLAG_VOLUME) ? (size_t)d->dim[1] * d->dim[2] * d->dim[3] : (size_t)d->dim[2];
            for (size_t i = 0; i < nc; ++i) sum += (double)d->counts[i];
        }
        printf("%s dim=%d,%d,%d,%d sum=%.9g\n", names[p], d->dim[0], d->dim[1], d->dim[2], d->dim[3], sum);
    }
    vmd_eval_free(ev); vmd_ir_free(ir); vmd_texttraj_close(trj); vmd_textsys_close(sysf);
    return 0;
}
