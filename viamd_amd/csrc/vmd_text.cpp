// Text trajectory readers behind vmd_trajectory_i ("real decoders, PDB multi-MODEL first"; BASELINE configs[0] is one).
//
// VIAMD attaches these through mdlib: multi-MODEL PDB files (`md_pdb_trajectory_...`, LoaderFlag_Trajectory for "pdb"), XYZ / XMOL
// (`md_xyz_...`) and LAMMPS dump files (`md_lammps_trajectory_attach_from_file`) - /root/reference/src/loader.cpp:22-77 (the table of
// types, extensions and flags), :111-159 (load).  The evaluator pulls frames with md_trajectory_load_frame (src/viamd.cpp:465-467); this
// file is the same role for the MI355X evaluator: the file is mapped read-only, ONE pass builds the frame index (byte range, atom count,
// unit cell per frame), and load_frame(f) parses frame f straight into the pinned staging rows it is handed - re-entrant, so the
// evaluator decodes the frames of a staged batch on its load threads (vmd_set_option("load_threads")).
//
// Why no parse kernel: text costs 6 - 7 times the bytes of the floats it encodes (81 bytes per PDB atom record against 12), so shipping
// the text to the GPU moves 7 x the PCIe traffic of shipping parsed floats; sixteen host threads parse at the rate PCIe would deliver
// the text.  The bytes that cross the link are the 12 per atom the kernels read.
//
// Numbers: a field is converted exactly like Python's float32(float(text)) / C's (float)strtod(text) - correctly rounded to double, then
// to float - but without the locale: up to 15 significant digits and a decimal exponent within +-22 are exact in double arithmetic (one
// correctly rounded multiply or divide by an exact power of ten); anything longer goes through strtod on a copy.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

namespace {

enum Format { FMT_PDB, FMT_XYZ, FMT_LAMMPS };

struct Frame {
    size_t beg = 0, end = 0;        // byte range of the frame's atom lines (PDB: of the whole MODEL block)
    vmd_unitcell_t cell;
    double timestamp = 0.0;
    // LAMMPS: column layout of this frame's ATOMS section and the cell parameters the scaled coordinates need
    int col_id = -1, col[3] = {-1, -1, -1}, ncols = 0;
    bool scaled = false;
    double lo[3] = {0, 0, 0}, ext[3] = {0, 0, 0}, tilt[3] = {0, 0, 0};
};

struct TextTraj {
    Format fmt = FMT_PDB;
    const char* data = nullptr;
    size_t bytes = 0;
    int fd = -1;
    size_t num_atoms = 0;
    std::vector<Frame> frames;
    vmd_trajectory_i iface;
    std::string path;
    bool tinker = false;      // FMT_XYZ, Tinker flavour (.arc / Tinker .xyz): no comment line, atom lines `index symbol x y z type bonds...`
};

bool fail(const std::string& msg) { vmd_set_last_error(msg.c_str()); return false; }

const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// [b, e): optional blanks, sign, digits[.digits][(e|E)[sign]digits]; *ok = false when no number is there
double parse_double(const char* b, const char* e, bool* ok) {
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    const char* p = b;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    uint64_t m = 0;
    int sig = 0, exp10 = 0;
    bool any = false, exact = true;
    for (; p < e && *p >= '0' && *p <= '9'; ++p) {
        any = true;
        if (sig < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) ++sig; } else { exact = false; ++exp10; }
    }
    if (p < e && *p == '.') {
        ++p;
        for (; p < e && *p >= '0' && *p <= '9'; ++p) {
            any = true;
            if (sig < 19) { m = m * 10 + (uint64_t)(*p - '0'); if (m) ++sig; --exp10; } else exact = false;
        }
    }
    if (!any) { if (ok) *ok = false; return 0.0; }
    if (p < e && (*p == 'e' || *p == 'E' || *p == 'd' || *p == 'D')) {
        const char* q = p + 1;
        bool eneg = false;
        if (q < e && (*q == '+' || *q == '-')) { eneg = *q == '-'; ++q; }
        if (q < e && *q >= '0' && *q <= '9') {
            int x = 0;
            for (; q < e && *q >= '0' && *q <= '9'; ++q) if (x < 10000) x = x * 10 + (*q - '0');
            exp10 += eneg ? -x : x;
            p = q;
        }
    }
    if (p != e) { if (ok) *ok = false; return 0.0; }
    if (ok) *ok = true;
    double v;
    if (exact && sig <= 15 && exp10 >= -22 && exp10 <= 22) {
        v = (double)m;                                   // < 10^15 < 2^53: exact
        v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];       // one correctly rounded operation on exact operands = strtod
    } else {
        char buf[96];
        const size_t n = std::min<size_t>((size_t)(e - b), sizeof(buf) - 1);
        memcpy(buf, b, n);
        buf[n] = 0;
        for (size_t i = 0; i < n; ++i) if (buf[i] == 'd' || buf[i] == 'D') buf[i] = 'e';
        v = strtod(buf, nullptr);
        return v;                                         // the sign is in the text
    }
    return neg ? -v : v;
}

struct Line { const char* b; const char* e; };          // [b, e) without the line terminator

// next line of [p, end); returns false at the end of the data
bool next_line(const char*& p, const char* end, Line* out) {
    if (p >= end) return false;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* e = nl ? nl : end;
    out->b = p;
    out->e = (e > p && e[-1] == '\r') ? e - 1 : e;
    p = nl ? nl + 1 : end;
    return true;
}

bool starts(const Line& l, const char* s) { const size_t n = strlen(s); return (size_t)(l.e - l.b) >= n && memcmp(l.b, s, n) == 0; }

// whitespace-separated tokens of a line
int split(const Line& l, Line* tok, int cap) {
    int n = 0;
    const char* p = l.b;
    while (p < l.e) {
        while (p < l.e && (*p == ' ' || *p == '\t')) ++p;
        if (p >= l.e) break;
        const char* q = p;
        while (q < l.e && *q != ' ' && *q != '\t') ++q;
        if (n < cap) tok[n] = {p, q};
        ++n;
        p = q;
    }
    return n;
}

vmd_unitcell_t no_cell() { vmd_unitcell_t c; memset(&c, 0, sizeof(c)); return c; }

// (a, b, c, alpha, beta, gamma) -> a = (x,0,0), b = (xy,y,0), c = (xz,yz,z); the arithmetic of viamd_amd/pdb.py (fp64), tilts below 1e-6 -> 0
vmd_unitcell_t cell_from_parameters(double a, double b, double c, double al, double be, double ga) {
    const double d2r = M_PI / 180.0;
    const double xy = b * std::cos(ga * d2r), xz = c * std::cos(be * d2r);
    const double ly = std::sqrt(b * b - xy * xy);
    const double yz = (b * c * std::cos(al * d2r) - xy * xz) / ly;
    const double lz = std::sqrt(c * c - xz * xz - yz * yz);
    vmd_unitcell_t u = no_cell();
    u.x = (float)a; u.y = (float)ly; u.z = (float)lz;
    u.xy = std::fabs(xy) < 1e-6 ? 0.0f : (float)xy; u.xz = std::fabs(xz) < 1e-6 ? 0.0f : (float)xz; u.yz = std::fabs(yz) < 1e-6 ? 0.0f : (float)yz;
    u.flags = VMD_UNITCELL_PBC_ALL;
    return u;
}

bool is_atom_record(const Line& l) { return starts(l, "ATOM  ") || starts(l, "HETATM"); }

// ---- PDB: MODEL ... ENDMDL blocks (or one block up to END / the end of the file); CRYST1 applies to the frames after it
bool index_pdb(TextTraj* t) {
    const char* p = t->data;
    const char* end = t->data + t->bytes;
    vmd_unitcell_t cell = no_cell();
    Line l;
    size_t blk_beg = 0, natoms = 0;
    bool open = false;
    auto close_frame = [&](size_t at) -> bool {
        if (natoms) {
            if (t->frames.empty()) t->num_atoms = natoms;
            else if (natoms != t->num_atoms) return fail(t->path + ": MODEL " + std::to_string(t->frames.size() + 1) + " has " + std::to_string(natoms) + " atoms, the first " + std::to_string(t->num_atoms));
            Frame f;
            f.beg = blk_beg; f.end = at; f.cell = cell; f.timestamp = (double)t->frames.size();
            t->frames.push_back(f);
        }
        natoms = 0; open = false;
        return true;
    };
    while (true) {
        const char* at = p;
        if (!next_line(p, end, &l)) break;
        if (is_atom_record(l)) {
            if (!open) { open = true; blk_beg = (size_t)(at - t->data); }
            ++natoms;
        } else if (starts(l, "CRYST1")) {
            if ((size_t)(l.e - l.b) >= 54) {
                bool ok[6];
                const double a = parse_double(l.b + 6, l.b + 15, &ok[0]), b = parse_double(l.b + 15, l.b + 24, &ok[1]), c = parse_double(l.b + 24, l.b + 33, &ok[2]);
                const double al = parse_double(l.b + 33, l.b + 40, &ok[3]), be = parse_double(l.b + 40, l.b + 47, &ok[4]), ga = parse_double(l.b + 47, l.b + 54, &ok[5]);
                if (ok[0] && ok[1] && ok[2] && ok[3] && ok[4] && ok[5] && a > 0 && b > 0 && c > 0) cell = cell_from_parameters(a, b, c, al, be, ga);
            }
        } else if (starts(l, "ENDMDL") || (starts(l, "END") && natoms)) {
            if (!close_frame((size_t)(at - t->data))) return false;
        }
    }
    return close_frame(t->bytes);
}

bool load_pdb(const TextTraj* t, const Frame& f, float* x, float* y, float* z) {
    const char* p = t->data + f.beg;
    const char* end = t->data + f.end;
    Line l;
    size_t i = 0;
    while (next_line(p, end, &l)) {
        if (!is_atom_record(l)) continue;
        if (i >= t->num_atoms) return fail(t->path + ": more atom records than the first MODEL has");
        if ((size_t)(l.e - l.b) < 54) return fail(t->path + ": atom record shorter than 54 columns");
        bool okx, oky, okz;
        const float vx = (float)parse_double(l.b + 30, l.b + 38, &okx), vy = (float)parse_double(l.b + 38, l.b + 46, &oky), vz = (float)parse_double(l.b + 46, l.b + 54, &okz);
        if (!okx || !oky || !okz) return fail(t->path + ": unreadable coordinate field in atom record " + std::to_string(i + 1));
        if (x) x[i] = vx;
        if (y) y[i] = vy;
        if (z) z[i] = vz;
        ++i;
    }
    return i == t->num_atoms ? true : fail(t->path + ": atom records missing");
}

// ---- XYZ / XMOL: [natoms] [comment] natoms x "element x y z ..."; extended-XYZ Lattice="ax ay az bx by bz cx cy cz" (lower triangular)
bool lattice_cell(const Line& comment, vmd_unitcell_t* out, const std::string& path) {
    *out = no_cell();
    static const char key[] = "Lattice=\"";
    const char* p = comment.b;
    const size_t klen = sizeof(key) - 1;
    const char* hit = nullptr;
    for (; p + klen <= comment.e; ++p) if (memcmp(p, key, klen) == 0) { hit = p + klen; break; }
    if (!hit) return true;
    const char* q = (const char*)memchr(hit, '"', (size_t)(comment.e - hit));
    if (!q) return fail(path + ": unterminated Lattice=\"...\"");
    Line tok[10];
    const int n = split({hit, q}, tok, 10);
    double v[9];
    bool ok = n == 9;
    for (int i = 0; i < 9 && ok; ++i) v[i] = parse_double(tok[i].b, tok[i].e, &ok);
    if (!ok || std::fabs(v[1]) > 1e-6 || std::fabs(v[2]) > 1e-6 || std::fabs(v[5]) > 1e-6)
        return fail(path + ": extended XYZ lattice must be lower triangular: a=(x,0,0), b=(xy,y,0), c=(xz,yz,z)");
    out->x = (float)v[0]; out->y = (float)v[4]; out->z = (float)v[8];
    out->xy = (float)v[3]; out->xz = (float)v[6]; out->yz = (float)v[7];
    out->flags = VMD_UNITCELL_PBC_ALL;
    return true;
}

// a Tinker atom line: `<index> <symbol> x y z <type> [bonded atoms]` - first token an integer, second one not a number
bool tinker_atom_line(const Line& l, long expect_index) {
    Line tok[5];
    if (split(l, tok, 5) < 5) return false;
    bool ok;
    const double idx = parse_double(tok[0].b, tok[0].e, &ok);
    if (!ok || idx != (double)expect_index) return false;
    (void)parse_double(tok[1].b, tok[1].e, &ok);
    if (ok) return false;                                         // XYZ's own "element x y z" never has a number in the second column... of an index
    bool okx, oky, okz;
    (void)parse_double(tok[2].b, tok[2].e, &okx); (void)parse_double(tok[3].b, tok[3].e, &oky); (void)parse_double(tok[4].b, tok[4].e, &okz);
    return okx && oky && okz;
}

// XYZ / XMOL: [natoms] [comment, optionally Lattice="..."] natoms x "element x y z ...".
// Tinker (.arc archives, Tinker's own .xyz): [natoms title] [optional box line: a b c alpha beta gamma] natoms x
// "index symbol x y z type bonds..." - there is NO comment line; the flavour is recognised on the first frame by its first atom line
// (index 1, a symbol, three numbers) and must then hold for every frame.
bool index_xyz(TextTraj* t) {
    const char* p = t->data;
    const char* end = t->data + t->bytes;
    Line l;
    while (next_line(p, end, &l)) {
        Line tok[2];
        if (split(l, tok, 2) == 0) continue;                      // blank lines between frames
        const std::string where = t->path + ": frame " + std::to_string(t->frames.size());
        bool ok;
        const double nd = parse_double(tok[0].b, tok[0].e, &ok);
        // bounded BEFORE the cast: an atom line has at least two characters, so no frame of this file holds more than bytes / 2 atoms
        if (!ok || nd < 1 || nd != std::floor(nd) || nd > (double)(t->bytes / 2 + 1)) return fail(where + ": the atom count line is not a number");
        const size_t n = (size_t)nd;
        Frame f;
        f.cell = no_cell();
        const char* after_count = p;
        Line second;
        if (!next_line(p, end, &second)) return fail(where + " ends after its atom count");
        bool tinker = false;
        if (tinker_atom_line(second, 1)) { tinker = true; p = after_count; }                  // no comment line at all
        else {
            Line b6[7];
            Line third;
            const char* after_second = p;
            if (split(second, b6, 7) == 6 && next_line(p, end, &third) && tinker_atom_line(third, 1)) {
                double v[6];
                bool okb = true;
                for (int i = 0; i < 6 && okb; ++i) v[i] = parse_double(b6[i].b, b6[i].e, &okb);
                if (okb && v[0] > 0 && v[1] > 0 && v[2] > 0) { tinker = true; f.cell = cell_from_parameters(v[0], v[1], v[2], v[3], v[4], v[5]); }
            }
            p = after_second;
        }
        if (t->frames.empty()) t->tinker = tinker;
        else if (tinker != t->tinker) return fail(where + ": atom lines change their layout inside the file (Tinker `index symbol x y z` / plain `symbol x y z`)");
        if (!tinker && !lattice_cell(second, &f.cell, t->path)) return false;
        f.beg = (size_t)(p - t->data);
        for (size_t i = 0; i < n; ++i) if (!next_line(p, end, &l)) return fail(where + ": atom line " + std::to_string(i) + " is missing");
        f.end = (size_t)(p - t->data);
        f.timestamp = (double)t->frames.size();
        if (t->frames.empty()) t->num_atoms = n;
        else if (n != t->num_atoms) return fail(where + " has " + std::to_string(n) + " atoms, the first frame " + std::to_string(t->num_atoms));
        t->frames.push_back(f);
    }
    return true;
}

bool load_xyz(const TextTraj* t, const Frame& f, float* x, float* y, float* z) {
    const char* p = t->data + f.beg;
    const char* end = t->data + f.end;
    Line l, tok[5];
    const int c0 = t->tinker ? 2 : 1;                              // first coordinate column
    for (size_t i = 0; i < t->num_atoms; ++i) {
        if (!next_line(p, end, &l) || split(l, tok, 5) < c0 + 3) return fail(t->path + ": atom line " + std::to_string(i) + " is incomplete");
        bool okx, oky, okz;
        const float vx = (float)parse_double(tok[c0].b, tok[c0].e, &okx), vy = (float)parse_double(tok[c0 + 1].b, tok[c0 + 1].e, &oky), vz = (float)parse_double(tok[c0 + 2].b, tok[c0 + 2].e, &okz);
        if (!okx || !oky || !okz) return fail(t->path + ": unreadable coordinate in atom line " + std::to_string(i));
        if (x) x[i] = vx;
        if (y) y[i] = vy;
        if (z) z[i] = vz;
    }
    return true;
}

// ---- LAMMPS dump (`dump custom`): ITEM: TIMESTEP / NUMBER OF ATOMS / BOX BOUNDS [xy xz yz] b b b / ATOMS <columns>
bool index_lammps(TextTraj* t) {
    const char* p = t->data;
    const char* end = t->data + t->bytes;
    Line l, tok[64];
    while (next_line(p, end, &l)) {
        if (!starts(l, "ITEM: TIMESTEP")) continue;
        Frame f;
        const std::string where = t->path + ": frame " + std::to_string(t->frames.size());
        bool ok;
        if (!next_line(p, end, &l)) return fail(where + ": no time step");
        f.timestamp = parse_double(l.b, l.e, &ok);
        if (!ok) return fail(where + ": unreadable time step");
        if (!next_line(p, end, &l) || !starts(l, "ITEM: NUMBER OF ATOMS") || !next_line(p, end, &l)) return fail(where + ": ITEM: NUMBER OF ATOMS expected");
        const double nd = parse_double(l.b, l.e, &ok);
        if (!ok || nd < 1 || nd != std::floor(nd) || nd > (double)(t->bytes / 2 + 1)) return fail(where + ": unreadable atom count");     // integral, and bounded before the cast
        const size_t n = (size_t)nd;
        if (!next_line(p, end, &l) || !starts(l, "ITEM: BOX BOUNDS")) return fail(where + ": ITEM: BOX BOUNDS expected");
        int nt = split(l, tok, 64);
        bool tri = false;
        int nper = 0;
        uint32_t flags = 0;
        for (int i = 3; i < nt && i < 64; ++i) {
            const size_t len = (size_t)(tok[i].e - tok[i].b);
            if (len == 2 && tok[i].b[0] == 'x' && tok[i].b[1] == 'y') tri = true;
            if (len == 2 && strchr("pfsm", tok[i].b[0]) && strchr("pfsm", tok[i].b[1])) { if (nper < 3 && tok[i].b[0] == 'p' && tok[i].b[1] == 'p') flags |= 1u << nper; ++nper; }
        }
        if (nper == 0) flags = VMD_UNITCELL_PBC_ALL;
        double rows[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int a = 0; a < 3; ++a) {
            if (!next_line(p, end, &l)) return fail(where + ": box bounds missing");
            nt = split(l, tok, 64);
            if (nt < (tri ? 3 : 2)) return fail(where + ": box bounds incomplete");
            for (int k = 0; k < (tri ? 3 : 2); ++k) { rows[a][k] = parse_double(tok[k].b, tok[k].e, &ok); if (!ok) return fail(where + ": unreadable box bound"); }
        }
        const double xy = tri ? rows[0][2] : 0.0, xz = tri ? rows[1][2] : 0.0, yz = tri ? rows[2][2] : 0.0;
        // the bounds of a triclinic box are those of its bounding box: undo (LAMMPS manual, "triclinic"); the order of operations is
        // the Python reader's (viamd_amd/textio.py)
        const double xlo = rows[0][0] - std::min(std::min(0.0, xy), std::min(xz, xy + xz)), xhi = rows[0][1] - std::max(std::max(0.0, xy), std::max(xz, xy + xz));
        const double ylo = rows[1][0] - std::min(0.0, yz), yhi = rows[1][1] - std::max(0.0, yz);
        const double zlo = rows[2][0], zhi = rows[2][1];
        f.lo[0] = xlo; f.lo[1] = ylo; f.lo[2] = zlo;
        f.ext[0] = xhi - xlo; f.ext[1] = yhi - ylo; f.ext[2] = zhi - zlo;
        f.tilt[0] = xy; f.tilt[1] = xz; f.tilt[2] = yz;
        f.cell = no_cell();
        f.cell.x = (float)f.ext[0]; f.cell.y = (float)f.ext[1]; f.cell.z = (float)f.ext[2];
        f.cell.xy = (float)xy; f.cell.xz = (float)xz; f.cell.yz = (float)yz;
        f.cell.flags = flags;
        if (!next_line(p, end, &l) || !starts(l, "ITEM: ATOMS")) return fail(where + ": ITEM: ATOMS expected");
        nt = split(l, tok, 64);
        f.ncols = std::min(nt - 2, 62);
        auto find = [&](const char* name) {
            for (int i = 0; i < f.ncols; ++i) { const Line& c = tok[i + 2]; if ((size_t)(c.e - c.b) == strlen(name) && memcmp(c.b, name, strlen(name)) == 0) return i; }
            return -1;
        };
        f.col_id = find("id");
        const char* sets[3][3] = {{"x", "y", "z"}, {"xu", "yu", "zu"}, {"xs", "ys", "zs"}};
        bool found = false;
        for (int s = 0; s < 3 && !found; ++s) {
            const int c0 = find(sets[s][0]), c1 = find(sets[s][1]), c2 = find(sets[s][2]);
            if (c0 >= 0 && c1 >= 0 && c2 >= 0) { f.col[0] = c0; f.col[1] = c1; f.col[2] = c2; f.scaled = s == 2; found = true; }
        }
        if (!found) return fail(where + ": no coordinate columns (x y z | xu yu zu | xs ys zs)");
        f.beg = (size_t)(p - t->data);
        for (size_t i = 0; i < n; ++i) if (!next_line(p, end, &l)) return fail(where + ": atom line " + std::to_string(i) + " is missing");
        f.end = (size_t)(p - t->data);
        if (t->frames.empty()) t->num_atoms = n;
        else if (n != t->num_atoms) return fail(t->path + ": the atom count changes between frames");
        t->frames.push_back(f);
    }
    return true;
}

bool load_lammps(const TextTraj* t, const Frame& f, float* x, float* y, float* z) {
    const size_t n = t->num_atoms;
    const char* p = t->data + f.beg;
    const char* end = t->data + f.end;
    Line l, tok[64];
    std::vector<double> v(3 * n), id(f.col_id >= 0 ? n : 0);
    for (size_t i = 0; i < n; ++i) {
        if (!next_line(p, end, &l) || split(l, tok, 64) < f.ncols) return fail(t->path + ": atom line " + std::to_string(i) + " is incomplete");
        bool ok = true, o;
        for (int c = 0; c < 3; ++c) { v[3 * i + c] = parse_double(tok[f.col[c]].b, tok[f.col[c]].e, &o); ok = ok && o; }
        if (f.col_id >= 0) { id[i] = parse_double(tok[f.col_id].b, tok[f.col_id].e, &o); ok = ok && o; }
        if (!ok) return fail(t->path + ": unreadable number in atom line " + std::to_string(i));
    }
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    if (f.col_id >= 0) std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return id[a] < id[b]; });      // frames are sorted by id
    for (size_t k = 0; k < n; ++k) {
        const double* s = &v[3 * order[k]];
        double c[3] = {s[0], s[1], s[2]};
        if (f.scaled) {                                   // scaled coordinates span the cell (the Python reader's expression order)
            c[0] = f.lo[0] + s[0] * f.ext[0] + s[1] * f.tilt[0] + s[2] * f.tilt[1];
            c[1] = f.lo[1] + s[1] * f.ext[1] + s[2] * f.tilt[2];
            c[2] = f.lo[2] + s[2] * f.ext[2];
        }
        if (x) x[k] = (float)c[0];
        if (y) y[k] = (float)c[1];
        if (z) z[k] = (float)c[2];
    }
    return true;
}

size_t tt_num_frames(void* inst) { return ((TextTraj*)inst)->frames.size(); }
size_t tt_num_atoms(void* inst) { return ((TextTraj*)inst)->num_atoms; }
bool tt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    const TextTraj* t = (const TextTraj*)inst;
    if (idx < 0 || (size_t)idx >= t->frames.size()) return fail(t->path + ": frame index out of range");
    const Frame& f = t->frames[(size_t)idx];
    if (hdr) { hdr->num_atoms = t->num_atoms; hdr->index = idx; hdr->timestamp = f.timestamp; hdr->unitcell = f.cell; }
    if (!x && !y && !z) return true;
    switch (t->fmt) {
    case FMT_PDB: return load_pdb(t, f, x, y, z);
    case FMT_XYZ: return load_xyz(t, f, x, y, z);
    default: return load_lammps(t, f, x, y, z);
    }
}

}  // namespace

struct vmd_texttraj_t { TextTraj t; };

extern "C" void vmd_texttraj_close(vmd_texttraj_t* h) {
    if (!h) return;
    if (h->t.data) munmap((void*)h->t.data, h->t.bytes);
    if (h->t.fd >= 0) close(h->t.fd);
    delete h;
}

extern "C" vmd_texttraj_t* vmd_texttraj_open(const char* path, const char* format) {
    if (!path) { fail("vmd_texttraj_open: NULL path"); return nullptr; }
    std::string ext = format ? format : "";
    if (ext.empty()) {
        const char* dot = strrchr(path, '.');
        ext = dot ? dot + 1 : "";
    }
    for (auto& ch : ext) ch = (char)tolower((unsigned char)ch);
    vmd_texttraj_t* h = new vmd_texttraj_t();
    TextTraj* t = &h->t;
    t->path = path;
    if (ext == "pdb") t->fmt = FMT_PDB;
    else if (ext == "xyz" || ext == "xmol" || ext == "arc") t->fmt = FMT_XYZ;
    else if (ext == "lammpstrj") t->fmt = FMT_LAMMPS;
    else { fail(std::string("vmd_texttraj_open: '") + ext + "' is not one of pdb, xyz, xmol, arc, lammpstrj"); delete h; return nullptr; }
    t->fd = open(path, O_RDONLY);
    struct stat st;
    if (t->fd < 0 || fstat(t->fd, &st) != 0) { fail(std::string("cannot open '") + path + "'"); vmd_texttraj_close(h); return nullptr; }
    t->bytes = (size_t)st.st_size;
    if (t->bytes == 0) { fail(t->path + ": no frames"); vmd_texttraj_close(h); return nullptr; }
    void* m = mmap(nullptr, t->bytes, PROT_READ, MAP_PRIVATE, t->fd, 0);
    if (m == MAP_FAILED) { t->data = nullptr; fail(std::string("cannot map '") + path + "'"); vmd_texttraj_close(h); return nullptr; }
    t->data = (const char*)m;
    (void)madvise(m, t->bytes, MADV_SEQUENTIAL);
    const bool ok = t->fmt == FMT_PDB ? index_pdb(t) : (t->fmt == FMT_XYZ ? index_xyz(t) : index_lammps(t));
    if (!ok) { vmd_texttraj_close(h); return nullptr; }
    if (t->frames.empty()) { fail(t->path + ": no frames"); vmd_texttraj_close(h); return nullptr; }
    (void)madvise(m, t->bytes, MADV_RANDOM);
    memset(&t->iface, 0, sizeof(t->iface));
    t->iface.inst = t;
    t->iface.num_frames = tt_num_frames;
    t->iface.num_atoms = tt_num_atoms;
    t->iface.load_frame = tt_load_frame;
    return h;
}

extern "C" vmd_trajectory_i* vmd_texttraj_interface(vmd_texttraj_t* h) { return h ? &h->t.iface : nullptr; }

// ---- the system (md_system_t as far as the script front-end and the evaluator read it) out of a PDB file: what
// md_pdb_system_init_from_file gives VIAMD for LoaderFlag_System (src/loader.cpp:113-128).  Columns as in viamd_amd/pdb.py: element = 77-78,
// else the first letter of the atom name; name 13-16; residue name 18-20; chain 22; resSeq 23-26.  A new residue starts whenever
// (chain, resSeq) changes.  Only the atoms of the FIRST model are read.
namespace {
struct ElementMass { const char* sym; float mass; };
const ElementMass kMass[] = {{"H", 1.008f}, {"He", 4.0026f}, {"Li", 6.94f}, {"Be", 9.0122f}, {"B", 10.81f}, {"C", 12.011f}, {"N", 14.007f}, {"O", 15.999f},
    {"F", 18.998f}, {"Ne", 20.180f}, {"Na", 22.990f}, {"Mg", 24.305f}, {"Al", 26.982f}, {"Si", 28.085f}, {"P", 30.974f}, {"S", 32.06f}, {"Cl", 35.45f},
    {"Ar", 39.948f}, {"K", 39.098f}, {"Ca", 40.078f}, {"Ti", 47.867f}, {"Cr", 51.996f}, {"Mn", 54.938f}, {"Fe", 55.845f}, {"Co", 58.933f}, {"Ni", 58.693f},
    {"Cu", 63.546f}, {"Zn", 65.38f}, {"Se", 78.971f}, {"Br", 79.904f}, {"Rb", 85.468f}, {"Sr", 87.62f}, {"Mo", 95.95f}, {"Ag", 107.87f}, {"Cd", 112.41f},
    {"I", 126.90f}, {"Cs", 132.91f}, {"Ba", 137.33f}, {"Pt", 195.08f}, {"Au", 196.97f}, {"Hg", 200.59f}, {"Pb", 207.2f}};
float element_mass(const std::string& e) {
    for (const ElementMass& m : kMass) if (e == m.sym) return m.mass;
    return 12.0f;
}
std::string strip(const char* b, const char* e) {
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t')) --e;
    return std::string(b, e);
}
}  // namespace

struct vmd_textsys_t {
    std::vector<std::string> elements, names, resnames;
    std::vector<const char*> p_elements, p_names, p_resnames;
    std::vector<int32_t> residue_index, residue_seq_id;
    std::vector<float> mass, xyz;
    vmd_unitcell_t cell;
    vmd_topology_t topo;
};

extern "C" void vmd_textsys_close(vmd_textsys_t* s) { delete s; }

extern "C" vmd_textsys_t* vmd_textsys_open(const char* path) {
    vmd_texttraj_t* h = vmd_texttraj_open(path, "pdb");
    if (!h) return nullptr;
    const TextTraj* t = &h->t;
    vmd_textsys_t* s = new vmd_textsys_t();
    const Frame& f = t->frames[0];
    const size_t n = t->num_atoms;
    s->cell = f.cell;
    s->xyz.resize(3 * n);
    bool ok = load_pdb(t, f, s->xyz.data(), s->xyz.data() + n, s->xyz.data() + 2 * n);
    const char* p = t->data + f.beg;
    const char* end = t->data + f.end;
    Line l;
    char last_chain = 0;
    long last_seq = 0;
    int32_t res = -1;
    while (ok && next_line(p, end, &l)) {
        if (!is_atom_record(l)) continue;
        const size_t len = (size_t)(l.e - l.b);
        std::string name = strip(l.b + 12, l.b + std::min<size_t>(16, len));
        std::string el = len >= 78 ? strip(l.b + 76, l.b + 78) : std::string();
        if (el.empty()) el = name.substr(0, 1);
        if (!el.empty()) { el[0] = (char)toupper((unsigned char)el[0]); for (size_t k = 1; k < el.size(); ++k) el[k] = (char)tolower((unsigned char)el[k]); }
        const char chain = len > 21 ? l.b[21] : ' ';
        bool okn = false;
        const long seq = len >= 26 ? (long)parse_double(l.b + 22, l.b + 26, &okn) : 0;
        if (!okn) { ok = fail(t->path + ": unreadable residue number in atom record " + std::to_string(s->elements.size() + 1)); break; }
        if (res < 0 || chain != last_chain || seq != last_seq) { ++res; last_chain = chain; last_seq = seq; }
        s->elements.push_back(el); s->names.push_back(name); s->resnames.push_back(strip(l.b + 17, l.b + std::min<size_t>(20, len)));
        s->residue_index.push_back(res); s->residue_seq_id.push_back((int32_t)seq);
        s->mass.push_back(element_mass(el));
    }
    vmd_texttraj_close(h);
    if (!ok) { delete s; return nullptr; }
    for (size_t i = 0; i < n; ++i) { s->p_elements.push_back(s->elements[i].c_str()); s->p_names.push_back(s->names[i].c_str()); s->p_resnames.push_back(s->resnames[i].c_str()); }
    s->topo.num_atoms = n;
    s->topo.elements = s->p_elements.data(); s->topo.names = s->p_names.data(); s->topo.resnames = s->p_resnames.data();
    s->topo.residue_index = s->residue_index.data(); s->topo.residue_seq_id = s->residue_seq_id.data();
    return s;
}

extern "C" const vmd_topology_t* vmd_textsys_topology(const vmd_textsys_t* s) { return s ? &s->topo : nullptr; }
extern "C" const float* vmd_textsys_mass(const vmd_textsys_t* s) { return s ? s->mass.data() : nullptr; }
extern "C" const float* vmd_textsys_coords(const vmd_textsys_t* s, vmd_unitcell_t* cell) {
    if (!s) return nullptr;
    if (cell) *cell = s->cell;
    return s->xyz.data();
}
