// viamd_amd/csrc/vmd_export.cpp — what VIAMD writes to disk from evaluated properties, behind the C ABI:
// XVG / CSV tables (/root/reference/src/main.cpp:5640-5716: export_xvg, export_csv; assembled per property type as in
// draw_property_export_window, :5953-6040) and the Gaussian cube file of a volume with the atoms of reference structure 0
// (export_cube, :5718-5830).  Only the public API of the evaluator is used; format strings follow the reference so that files
// can be diffed against ones exported by a real VIAMD.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

static bool exp_fail(const std::string& msg) {
    vmd_set_last_error(msg.c_str());
    fprintf(stderr, "[viamd_amd] error: %s\n", msg.c_str());
    return false;
}

// src/main.cpp:5640-5683
extern "C" bool vmd_export_xvg(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows) {
    if (!path || !columns || !labels) return exp_fail("vmd_export_xvg: NULL argument");
    FILE* f = fopen(path, "w");
    if (!f) return exp_fail(std::string("Failed to open file '") + path + "' to write data.");
    time_t t;
    time(&t);
    struct tm tmv;
    char tbuf[64];
    localtime_r(&t, &tmv);
    fprintf(f, "# This file was created %s", asctime_r(&tmv, tbuf));
    fprintf(f, "# Created by:\n");
    fprintf(f, "# VIAMD \n");
    fprintf(f, "@    title \"VIAMD Properties\"\n");
    fprintf(f, "@    xaxis  label \"Time\"\n");
    fprintf(f, "@ TYPE xy\n");
    fprintf(f, "@ view 0.15, 0.15, 0.75, 0.85\n");
    fprintf(f, "@ legend on\n");
    fprintf(f, "@ legend box on\n");
    fprintf(f, "@ legend loctype view\n");
    fprintf(f, "@ legend 0.78, 0.8\n");
    fprintf(f, "@ legend length %i\n", (int)num_columns);
    for (size_t j = 0; j < num_columns; ++j) fprintf(f, "@ s%zu legend \"%s\"\n", j, labels[j]);
    for (size_t i = 0; i < num_rows; ++i) {
        for (size_t j = 0; j < num_columns; ++j) fprintf(f, "%12.6f ", columns[j][i]);
        fprintf(f, "\n");
    }
    return fclose(f) == 0 ? true : exp_fail(std::string("writing '") + path + "' failed");
}

// src/main.cpp:5685-5716
extern "C" bool vmd_export_csv(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows) {
    if (!path || !columns || !labels) return exp_fail("vmd_export_csv: NULL argument");
    FILE* f = fopen(path, "w");
    if (!f) return exp_fail(std::string("Failed to open file '") + path + "' to write data.");
    for (size_t i = 0; i < num_columns; ++i) fprintf(f, "%s,", labels[i]);
    fprintf(f, "\n");
    for (size_t i = 0; i < num_rows; ++i) {
        for (size_t j = 0; j < num_columns; ++j) fprintf(f, "%.6g,", columns[j][i]);
        fprintf(f, "\n");
    }
    return fclose(f) == 0 ? true : exp_fail(std::string("writing '") + path + "' failed");
}

// the table draw_property_export_window builds for one property (src/main.cpp:5953-6040): temporal -> time column + one column
// per population member ("label[i]", 1-based); distribution -> sample_range(x_min, x_max, num_bins) + the display histogram
extern "C" bool vmd_export_property_table(const char* path, vmd_script_eval_t* eval, const char* name, const char* format,
                                          const double* frame_times, int num_bins) {
    if (!path || !eval || !name || !format) return exp_fail("vmd_export_property_table: NULL argument");
    const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, name);
    if (!pd) return exp_fail(std::string("Export: the property '") + name + "' does not exist");
    const bool xvg = !strcmp(format, "xvg");
    if (!xvg && strcmp(format, "csv")) return exp_fail("vmd_export_property_table: format must be \"xvg\" or \"csv\"");
    std::vector<std::vector<float>> cols;
    std::vector<std::string> labels;
    size_t rows = 0;
    if (pd->weights) {                                   // distribution
        const int nb = num_bins > 0 ? num_bins : 128;    // display bins (src/viamd.h:341)
        rows = (size_t)nb;
        std::vector<float> x(nb), g(nb);
        const double beg = pd->min_range[0], end = pd->max_range[0];
        const double step = nb > 1 ? (end - beg) / (double)(nb - 1) : 0.0;       // sample_range, src/main.cpp:5832-5840
        for (int i = 0; i < nb; ++i) x[i] = (float)(beg + step * (double)i);
        vmd_downsample_histogram(g.data(), nb, pd->values, pd->weights, pd->dim[2]);
        cols.push_back(std::move(x)); labels.push_back("");
        cols.push_back(std::move(g)); labels.push_back(name);
    } else if (pd->dim[3] == 0 || pd->dim[2] == 0) {    // temporal: values[frame * dim[1] + i]
        const size_t F = (size_t)pd->dim[0], D = (size_t)std::max(1, pd->dim[1]);
        rows = F;
        std::vector<float> t(F);
        for (size_t f = 0; f < F; ++f) t[f] = frame_times ? (float)frame_times[f] : (float)f;
        cols.push_back(std::move(t)); labels.push_back(frame_times ? "Time" : "Frame");
        for (size_t i = 0; i < D; ++i) {
            std::vector<float> c(F);
            for (size_t f = 0; f < F; ++f) c[f] = pd->values[f * D + i];
            cols.push_back(std::move(c));
            labels.push_back(D > 1 ? std::string(name) + "[" + std::to_string(i + 1) + "]" : std::string(name));
        }
    } else {
        return exp_fail(std::string("Export: '") + name + "' is a volume; use vmd_export_cube");
    }
    std::vector<const float*> cp;
    std::vector<const char*> lp;
    for (auto& c : cols) cp.push_back(c.data());
    for (auto& l : labels) lp.push_back(l.c_str());
    return xvg ? vmd_export_xvg(path, cp.data(), lp.data(), cp.size(), rows) : vmd_export_csv(path, cp.data(), lp.data(), cp.size(), rows);
}

// md_script_vis_eval_payload(..., MD_SCRIPT_VISUALIZE_ATOMS | MD_SCRIPT_VISUALIZE_SDF) as VIAMD consumes it
// (density_volume.cpp:183-204, 263-269; src/main.cpp:5751-5803): world->reference matrices of `frame`, the half extent and the
// atoms of every reference structure
extern "C" bool vmd_eval_sdf_payload(vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame,
                                     vmd_sdf_payload_t* out) {
    if (!out) return exp_fail("vmd_eval_sdf_payload: NULL argument");
    memset(out, 0, sizeof(*out));
    size_t K = 0, m = 0;
    const int32_t* st = vmd_eval_sdf_structures(eval, name, &K, &m);
    if (!st) return false;
    static thread_local std::vector<float> mats;
    mats.assign(K * 16, 0.0f);
    float ext = 0.0f;
    if (!vmd_eval_sdf_matrices(eval, name, sys, traj, frame, mats.data(), &K, &ext)) return false;
    out->num_structures = K; out->atoms_per_structure = m; out->structures = st; out->matrices = mats.data(); out->extent = ext;
    return true;
}

// export_cube, src/main.cpp:5718-5830
extern "C" bool vmd_export_cube(const char* path, vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys, vmd_trajectory_i* traj,
                                uint32_t frame, const uint8_t* atomic_numbers) {
    if (!path || !eval || !name || !traj) return exp_fail("vmd_export_cube: NULL argument");
    const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, name);
    if (!pd) return exp_fail("Export Cube: The property to be exported did not exist");
    vmd_sdf_payload_t vis;
    if (!vmd_eval_sdf_payload(eval, name, sys, traj, frame, &vis)) return exp_fail(std::string("Failed to visualize volume for export. ") + vmd_last_error());
    if (!vmd_eval_finalize(eval)) return false;         // the float view VIAMD reads (prop_data->values) is current
    // the atoms block is written from the coordinates of trajectory frame 0 (src/main.cpp:5741-5748)
    const size_t N = traj->num_atoms(traj->inst);
    std::vector<float> xyz(3 * N);
    vmd_frame_header_t hdr;
    memset(&hdr, 0, sizeof(hdr));
    if (!traj->load_frame(traj->inst, 0, &hdr, xyz.data(), xyz.data() + N, xyz.data() + 2 * N)) return exp_fail("Export Cube: loading frame 0 failed");
    FILE* f = fopen(path, "w");
    if (!f) return exp_fail(std::string("Failed to open file '") + path + "' in order to write to it.");
    fprintf(f, "EXPORTED DENSITY VOLUME FROM VIAMD, UNITS IN BOHR\n");
    fprintf(f, "OUTER LOOP: X, MIDDLE LOOP: Y, INNER LOOP: Z\n");
    if (vis.num_structures > 0) {
        const float angstrom_to_bohr = (float)(1.0 / 0.529177210903);
        // a structure is a bitfield in VIAMD: its atoms come out in ascending index order
        std::vector<int32_t> atoms(vis.structures, vis.structures + vis.atoms_per_structure);
        std::sort(atoms.begin(), atoms.end());
        atoms.erase(std::unique(atoms.begin(), atoms.end()), atoms.end());
        if (!atoms.empty() && (atoms.front() < 0 || (size_t)atoms.back() >= N)) {
            fclose(f);
            return exp_fail("Export Cube: a reference structure refers to atoms the trajectory does not have");
        }
        const int num_atoms = (int)atoms.size();
        const int vol_dim[3] = {pd->dim[1], pd->dim[2], pd->dim[3]};
        const double extent = vis.extent * 2.0 * angstrom_to_bohr;
        const double voxel_ext[3] = {extent / (double)vol_dim[0], extent / (double)vol_dim[1], extent / (double)vol_dim[2]};
        const double half_ext = extent * 0.5;
        fprintf(f, "%5i %12.6f %12.6f %12.6f\n", -num_atoms, -half_ext, -half_ext, -half_ext);
        fprintf(f, "%5i %12.6f %12.6f %12.6f\n", vol_dim[0], voxel_ext[0], 0.0, 0.0);
        fprintf(f, "%5i %12.6f %12.6f %12.6f\n", vol_dim[1], 0.0, voxel_ext[1], 0.0);
        fprintf(f, "%5i %12.6f %12.6f %12.6f\n", vol_dim[2], 0.0, 0.0, voxel_ext[2]);
        // M = scale(angstrom_to_bohr) * matrices[0]; column-major mat4
        const float* M = vis.matrices;
        for (int32_t i : atoms) {
            const float x = xyz[i], y = xyz[N + i], z = xyz[2 * N + i];
            float c[3];
            for (int r = 0; r < 3; ++r)
                c[r] = (angstrom_to_bohr * M[0 + r]) * x + (angstrom_to_bohr * M[4 + r]) * y + (angstrom_to_bohr * M[8 + r]) * z + (angstrom_to_bohr * M[12 + r]);
            const int anum = atomic_numbers ? (int)atomic_numbers[i] : 0;
            fprintf(f, "%5i %12.6f %12.6f %12.6f %12.6f\n", anum, (float)anum, c[0], c[1], c[2]);
        }
        fprintf(f, "%5i %5i\n", 1, 1);
        int count = 0;
        for (int x = 0; x < vol_dim[0]; ++x)
            for (int y = 0; y < vol_dim[1]; ++y)
                for (int z = 0; z < vol_dim[2]; ++z) {
                    const size_t idx = (size_t)z * vol_dim[0] * vol_dim[1] + (size_t)y * vol_dim[0] + x;
                    fprintf(f, " %12.6E", pd->values[idx]);
                    if (++count % 6 == 0) fprintf(f, "\n");
                }
    }
    return fclose(f) == 0 ? true : exp_fail(std::string("writing '") + path + "' failed");
}
