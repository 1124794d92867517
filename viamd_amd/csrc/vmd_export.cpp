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

// ---- text tables.  XVG and CSV differ in a preamble, a cell format and nothing else: one emitter, two dialects.  The byte layout of
// each dialect is the reference's (src/main.cpp:5640-5683 xvg, :5685-5716 csv) so that exported files diff clean against VIAMD's.
namespace {
struct OutFile {
    FILE* f = nullptr;
    std::string path;
    explicit OutFile(const char* p) : f(fopen(p, "w")), path(p) {}
    ~OutFile() { if (f) fclose(f); }
    bool finish() { const bool ok = f && fclose(f) == 0; f = nullptr; return ok ? true : exp_fail("writing '" + path + "' failed"); }
};
struct TableDialect {
    const char* cell;                 // one value
    const char* label_cell;           // one label of the heading row (NULL: labels go into the preamble instead)
    const char* const* preamble;      // fixed lines in front of the labels, NULL-terminated
    const char* legend_count;         // "... %i" line announcing the number of columns (NULL: none)
    const char* legend_entry;         // per column: index + label (NULL: none)
    bool stamp;                       // first line carries the creation time
};
const char* const kXvgPreamble[] = {"# Created by:\n", "# VIAMD \n", "@    title \"VIAMD Properties\"\n", "@    xaxis  label \"Time\"\n", "@ TYPE xy\n",
                                    "@ view 0.15, 0.15, 0.75, 0.85\n", "@ legend on\n", "@ legend box on\n", "@ legend loctype view\n", "@ legend 0.78, 0.8\n", nullptr};
const char* const kNoPreamble[] = {nullptr};
const TableDialect kXvg = {"%12.6f ", nullptr, kXvgPreamble, "@ legend length %i\n", "@ s%zu legend \"%s\"\n", true};
const TableDialect kCsv = {"%.6g,", "%s,", kNoPreamble, nullptr, nullptr, false};

bool write_table(const char* who, const TableDialect& d, const char* path, const float* const* columns, const char* const* labels, size_t ncol, size_t nrow) {
    if (!path || !columns || !labels) return exp_fail(std::string(who) + ": NULL argument");
    OutFile out(path);
    if (!out.f) return exp_fail(std::string("Failed to open file '") + path + "' to write data.");
    if (d.stamp) {
        time_t now;
        time(&now);
        struct tm parts;
        char text[64];
        localtime_r(&now, &parts);
        fprintf(out.f, "# This file was created %s", asctime_r(&parts, text));
    }
    for (const char* const* line = d.preamble; *line; ++line) fputs(*line, out.f);
    if (d.legend_count) fprintf(out.f, d.legend_count, (int)ncol);
    for (size_t c = 0; c < ncol; ++c) {
        if (d.legend_entry) fprintf(out.f, d.legend_entry, c, labels[c]);
        if (d.label_cell) fprintf(out.f, d.label_cell, labels[c]);
    }
    if (d.label_cell) fputc('\n', out.f);
    for (size_t r = 0; r < nrow; ++r) {
        for (size_t c = 0; c < ncol; ++c) fprintf(out.f, d.cell, columns[c][r]);
        fputc('\n', out.f);
    }
    return out.finish();
}
}  // namespace

extern "C" bool vmd_export_xvg(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows) {
    return write_table("vmd_export_xvg", kXvg, path, columns, labels, num_columns, num_rows);
}
extern "C" bool vmd_export_csv(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows) {
    return write_table("vmd_export_csv", kCsv, path, columns, labels, num_columns, num_rows);
}

// the table draw_property_export_window builds for one property (src/main.cpp:5953-6040), labels included:
//   temporal (:5953-5990)      time column - trajectory times (frame_times) or, without them, the frame index - labelled "Frame", or
//                              "Time (<unit>)" when the trajectory has a time unit (:5969-5974); then y_values, labelled `name`, or
//                              "name (<unit>)" when unit[1] is set (:5965-5967; a temporal item's unit_str[0] holds the y unit, :1327); a
//                              population gets one column per member, "name[i]" 1-based (:5979-5988)
//   distribution (:5998-6020)  x = sample_range(hist.x_min, hist.x_max, num_bins) labelled with the x unit string (unit_str[0]); y = the
//                              display histogram labelled `name`, or "name (<unit_str[0]>)" when unit_str[1] is not empty - the reference
//                              prints `dp.unit_str`, which decays to unit_str[0] (:6003); kept as it is
// pinned to the reference's own export_csv / export_xvg / sample_range by tests/native/ref_callsites.cpp
extern "C" bool vmd_export_property_table(const char* path, vmd_script_eval_t* eval, const char* name, const char* format,
                                          const double* frame_times, const char* time_unit, int num_bins) {
    if (!path || !eval || !name || !format) return exp_fail("vmd_export_property_table: NULL argument");
    if (!vmd_eval_wait_settled(eval)) return false;
    const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, name);
    if (!pd) return exp_fail(std::string("Export: the property '") + name + "' does not exist");
    const bool xvg = !strcmp(format, "xvg");
    if (!xvg && strcmp(format, "csv")) return exp_fail("vmd_export_property_table: format must be \"xvg\" or \"csv\"");
    std::vector<std::vector<float>> cols;
    std::vector<std::string> labels;
    size_t rows = 0;
    const char* const unit_x = pd->unit_str[0] ? pd->unit_str[0] : "";
    const char* const unit_y = pd->unit_str[1] ? pd->unit_str[1] : "";
    if (pd->weights) {                                   // distribution
        const int nb = num_bins > 0 ? num_bins : 128;    // display bins (src/viamd.h:341)
        rows = (size_t)nb;
        std::vector<float> x(nb), g(nb);
        const float beg = (float)(double)pd->min_range[0], end = (float)(double)pd->max_range[0];      // Histogram::x_min / x_max are doubles handed to float parameters
        const double step = (end - beg) / (double)(nb - 1);                      // sample_range, src/main.cpp:5834-5841 (one bin: inf, like the reference)
        for (int i = 0; i < nb; ++i) x[i] = (float)((double)beg + step * (double)i);
        vmd_downsample_histogram(g.data(), nb, pd->values, pd->weights, pd->dim[2]);
        cols.push_back(std::move(x)); labels.push_back(unit_x);
        cols.push_back(std::move(g)); labels.push_back(unit_y[0] ? std::string(name) + " (" + unit_x + ")" : std::string(name));
    } else if (pd->dim[3] == 0 || pd->dim[2] == 0) {    // temporal: values[frame * dim[1] + i]
        const size_t F = (size_t)pd->dim[0], D = (size_t)std::max(1, pd->dim[1]);
        rows = F;
        std::vector<float> t(F);
        for (size_t f = 0; f < F; ++f) t[f] = frame_times ? (float)frame_times[f] : (float)f;
        cols.push_back(std::move(t));
        labels.push_back(time_unit && time_unit[0] ? std::string("Time (") + time_unit + ")" : std::string("Frame"));
        for (size_t i = 0; i < D; ++i) {
            std::vector<float> c(F);
            for (size_t f = 0; f < F; ++f) c[f] = pd->values[f * D + i];
            cols.push_back(std::move(c));
            labels.push_back(D > 1 ? std::string(name) + "[" + std::to_string(i + 1) + "]" : unit_y[0] ? std::string(name) + " (" + unit_y + ")" : std::string(name));
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

// Gaussian cube file of a volume (the format of export_cube, src/main.cpp:5718-5830: two comment lines, an atom count and origin line, one
// line per axis, the atoms of reference structure 0 in the volume's frame, a density-count line, then the voxels x-outermost / z-innermost,
// six per line).  Everything spatial is in Bohr.  Built as: a geometry record -> header rows from a table -> atoms -> a strided walk.
namespace {
constexpr double kBohrPerAngstrom = 1.0 / 0.529177210903;
struct CubeGeometry {
    int n[3];                 // voxels per axis
    double step[3];           // voxel edge per axis, Bohr
    double corner;            // -half edge of the cube, Bohr (the same on all axes)
};
CubeGeometry cube_geometry(const vmd_script_property_data_t* pd, float half_extent_angstrom) {
    CubeGeometry g;
    const float to_bohr = (float)kBohrPerAngstrom;                          // the reference scales in fp32 first (vis.sdf.extent * 2.0 * angstrom_to_bohr)
    const double edge = half_extent_angstrom * 2.0 * to_bohr;
    for (int a = 0; a < 3; ++a) { g.n[a] = pd->dim[1 + a]; g.step[a] = edge / (double)g.n[a]; }
    g.corner = -(edge * 0.5);
    return g;
}
}  // namespace

extern "C" bool vmd_export_cube(const char* path, vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys, vmd_trajectory_i* traj,
                                uint32_t frame, const uint8_t* atomic_numbers) {
    if (!path || !eval || !name || !traj) return exp_fail("vmd_export_cube: NULL argument");
    if (!vmd_eval_wait_settled(eval)) return false;
    const vmd_script_property_data_t* pd = vmd_eval_property_data(eval, name);
    if (!pd) return exp_fail("Export Cube: The property to be exported did not exist");
    vmd_sdf_payload_t vis;
    if (!vmd_eval_sdf_payload(eval, name, sys, traj, frame, &vis)) return exp_fail(std::string("Failed to visualize volume for export. ") + vmd_last_error());
    if (!vmd_eval_finalize(eval)) return false;         // the float view VIAMD reads (prop_data->values) is current
    // the atoms are placed from the coordinates of trajectory frame 0 (src/main.cpp:5741-5748)
    const size_t N = traj->num_atoms(traj->inst);
    std::vector<float> xyz(3 * N);
    vmd_frame_header_t hdr;
    memset(&hdr, 0, sizeof(hdr));
    if (!traj->load_frame(traj->inst, 0, &hdr, xyz.data(), xyz.data() + N, xyz.data() + 2 * N)) return exp_fail("Export Cube: loading frame 0 failed");
    // a structure is a bitfield in VIAMD: its atoms come out once each, in ascending index order
    std::vector<int32_t> members;
    if (vis.num_structures > 0) {
        members.assign(vis.structures, vis.structures + vis.atoms_per_structure);
        std::sort(members.begin(), members.end());
        members.erase(std::unique(members.begin(), members.end()), members.end());
        if (!members.empty() && (members.front() < 0 || (size_t)members.back() >= N))
            return exp_fail("Export Cube: a reference structure refers to atoms the trajectory does not have");
    }
    OutFile out(path);
    if (!out.f) return exp_fail(std::string("Failed to open file '") + path + "' in order to write to it.");
    fputs("EXPORTED DENSITY VOLUME FROM VIAMD, UNITS IN BOHR\n", out.f);
    fputs("OUTER LOOP: X, MIDDLE LOOP: Y, INNER LOOP: Z\n", out.f);
    if (vis.num_structures == 0) return out.finish();
    const CubeGeometry g = cube_geometry(pd, vis.extent);
    // header: {count, three reals} per row - the origin row (minus the atom count: "the file holds one density per voxel"), then one row
    // per axis with that axis' voxel edge on the diagonal
    const struct { int count; double v[3]; } rows[4] = {
        {-(int)members.size(), {g.corner, g.corner, g.corner}},
        {g.n[0], {g.step[0], 0.0, 0.0}},
        {g.n[1], {0.0, g.step[1], 0.0}},
        {g.n[2], {0.0, 0.0, g.step[2]}},
    };
    for (const auto& r : rows) fprintf(out.f, "%5i %12.6f %12.6f %12.6f\n", r.count, r.v[0], r.v[1], r.v[2]);
    // atoms: world -> reference frame of structure 0 (column-major mat4), every matrix element scaled to Bohr in fp32 before it is used
    const float s = (float)kBohrPerAngstrom;
    const float* M = vis.matrices;
    for (int32_t atom : members) {
        const float p[3] = {xyz[atom], xyz[N + atom], xyz[2 * N + atom]};
        float q[3];
        for (int r = 0; r < 3; ++r) q[r] = (s * M[0 + r]) * p[0] + (s * M[4 + r]) * p[1] + (s * M[8 + r]) * p[2] + (s * M[12 + r]);
        const int z = atomic_numbers ? (int)atomic_numbers[atom] : 0;
        fprintf(out.f, "%5i %12.6f %12.6f %12.6f %12.6f\n", z, (float)z, q[0], q[1], q[2]);
    }
    fprintf(out.f, "%5i %5i\n", 1, 1);
    // voxels: the array is x-fastest, the file z-fastest: walk the array with strides (1, nx, nx * ny) in file order
    const size_t stride[3] = {1, (size_t)g.n[0], (size_t)g.n[0] * (size_t)g.n[1]};
    size_t written = 0;
    for (int ix = 0; ix < g.n[0]; ++ix)
        for (int iy = 0; iy < g.n[1]; ++iy) {
            const float* row = pd->values + ix * stride[0] + iy * stride[1];
            for (int iz = 0; iz < g.n[2]; ++iz) {
                fprintf(out.f, " %12.6E", row[iz * stride[2]]);
                if (++written % 6 == 0) fputc('\n', out.f);
            }
        }
    return out.finish();
}
