// DCD trajectory reader behind vmd_trajectory_i (the step before the hot path).
//
// VIAMD attaches DCD files through mdlib (`md_dcd_attach_from_file`, /root/reference/src/loader.cpp:151-152) and the
// evaluator then pulls frames with md_trajectory_load_frame (src/viamd.cpp:465-467).  This is the same role for the
// MI355X evaluator: random access to frame f (fixed-size Fortran records -> one pread per frame), straight into the
// pinned staging buffer vmd_eval_frame_range hands to load_frame, so decode of batch k+1 overlaps the kernels of batch k.
//
// Format (CHARMM / NAMD / X-PLOR "CORD" files): Fortran unformatted records [int32 n][n bytes][int32 n].
//   record 1: "CORD" + int32 icntrl[20]  (icntrl[0] frames, [8] fixed atoms, [10] unit-cell block present, [11] 4th dimension)
//   record 2: int32 ntitle + ntitle * 80 chars;  record 3: int32 natoms
//   per frame: [6 x float64 unit cell: A, gamma, B, beta, alpha, C (angles in degrees, or their cosines)] X Y Z [W]
// Either byte order is accepted.  Files with fixed atoms (frames after the first hold only the free atoms) are rejected.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

namespace {

uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }

struct Dcd {
    int fd = -1;
    size_t num_frames = 0, num_atoms = 0;
    bool swap = false, has_cell = false, has_4d = false;
    size_t header_bytes = 0, frame_bytes = 0;
    vmd_trajectory_i iface;
    std::string path;
    // the file mapped read-only on first request (raw_mapped_view): the evaluator DMAs the coordinate records straight out of it
    std::mutex map_mtx;
    std::atomic<const unsigned char*> map{nullptr};
    size_t map_bytes = 0;
    bool map_failed = false;
    std::vector<uint64_t> stream_off;
};

bool fail(const char* fmt, const std::string& path) {
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, path.c_str());
    vmd_set_last_error(buf);
    return false;
}

bool read_at(const Dcd* d, void* dst, size_t n, size_t off) {
    char* p = (char*)dst;
    while (n) {
        const ssize_t r = pread(d->fd, p, n, (off_t)off);
        if (r <= 0) return false;
        p += r; off += (size_t)r; n -= (size_t)r;
    }
    return true;
}

uint32_t u32(const Dcd* d, const void* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return d->swap ? bswap32(v) : v;
}

// unit cell block -> the lattice VIAMD's md_unitcell_t describes: a = (x,0,0), b = (xy,y,0), c = (xz,yz,z)
vmd_unitcell_t decode_cell(const Dcd* d, const unsigned char* rec) {
    double v[6];
    for (int i = 0; i < 6; ++i) {
        uint64_t u;
        memcpy(&u, rec + 8 * i, 8);
        if (d->swap) u = bswap64(u);
        memcpy(&v[i], &u, 8);
    }
    vmd_unitcell_t c;
    memset(&c, 0, sizeof(c));
    const double A = v[0], B = v[2], C = v[5];
    if (!(A > 0.0 && B > 0.0 && C > 0.0)) return c;          // no periodic cell
    double cg = v[1], cb = v[3], ca = v[4];                   // gamma (a,b), beta (a,c), alpha (b,c)
    const bool cosines = std::fabs(cg) <= 1.0 && std::fabs(cb) <= 1.0 && std::fabs(ca) <= 1.0;
    if (!cosines) {
        const double rad = M_PI / 180.0;
        // exact zeros for right angles: cos(90 deg) in floating point is 6e-17, which would make the cell "triclinic"
        cg = v[1] == 90.0 ? 0.0 : std::cos(v[1] * rad);
        cb = v[3] == 90.0 ? 0.0 : std::cos(v[3] * rad);
        ca = v[4] == 90.0 ? 0.0 : std::cos(v[4] * rad);
    }
    const double xy = B * cg;
    const double ly = std::sqrt(std::max(0.0, B * B - xy * xy));
    const double xz = C * cb;
    const double yz = ly > 0.0 ? (B * C * ca - xy * xz) / ly : 0.0;
    const double lz = std::sqrt(std::max(0.0, C * C - xz * xz - yz * yz));
    c.x = (float)A; c.y = (float)ly; c.z = (float)lz;
    c.xy = (float)xy; c.xz = (float)xz; c.yz = (float)yz;
    // tilts below float resolution of the edge are rounding noise of the angle representation
    if (std::fabs(c.xy) < 1.0e-6f * c.x) c.xy = 0.0f;
    if (std::fabs(c.xz) < 1.0e-6f * c.x) c.xz = 0.0f;
    if (std::fabs(c.yz) < 1.0e-6f * c.y) c.yz = 0.0f;
    c.flags = VMD_UNITCELL_PBC_ALL;
    return c;
}

size_t dcd_num_frames(void* inst) { return ((Dcd*)inst)->num_frames; }
size_t dcd_num_atoms(void* inst) { return ((Dcd*)inst)->num_atoms; }

bool dcd_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    Dcd* d = (Dcd*)inst;
    if (idx < 0 || (size_t)idx >= d->num_frames) return fail("DCD '%s': frame index out of range", d->path);
    size_t off = d->header_bytes + (size_t)idx * d->frame_bytes;
    vmd_unitcell_t cell;
    memset(&cell, 0, sizeof(cell));
    if (d->has_cell) {
        unsigned char rec[56];
        if (!read_at(d, rec, sizeof(rec), off) || u32(d, rec) != 48 || u32(d, rec + 52) != 48) return fail("DCD '%s': bad unit-cell record", d->path);
        cell = decode_cell(d, rec + 4);
        off += sizeof(rec);
    }
    const size_t nbytes = 4 * d->num_atoms;
    float* dst[3] = {x, y, z};
    for (int a = 0; a < 3; ++a) {
        unsigned char mark[4];
        if (!read_at(d, mark, 4, off) || u32(d, mark) != nbytes) return fail("DCD '%s': bad coordinate record", d->path);
        if (dst[a]) {
            if (!read_at(d, dst[a], nbytes, off + 4)) return fail("DCD '%s': truncated frame", d->path);
            if (d->swap) {
                uint32_t* w = (uint32_t*)dst[a];
                for (size_t i = 0; i < d->num_atoms; ++i) w[i] = bswap32(w[i]);
            }
        }
        off += nbytes + 8;
    }
    if (hdr) {
        memset(hdr, 0, sizeof(*hdr));
        hdr->num_atoms = d->num_atoms;
        hdr->index = idx;
        hdr->timestamp = (double)idx;
        hdr->unitcell = cell;
    }
    return true;
}

// The frame as the file stores it, for the device path (vmd_trajectory_i::load_raw + raw_mapped_view): three blocks of floats
// behind their record markers.  Only `info` (and the header) - the payload is never copied, the copy engine reads the mapping.
bool dcd_load_raw(void* inst, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst, size_t) {
    Dcd* d = (Dcd*)inst;
    if (idx < 0 || (size_t)idx >= d->num_frames || !info || dst) return false;
    const size_t off = d->header_bytes + (size_t)idx * d->frame_bytes;
    const size_t lead = d->has_cell ? 56 : 0;
    const size_t nbytes = 4 * d->num_atoms;
    unsigned char rec[56];
    vmd_unitcell_t cell;
    memset(&cell, 0, sizeof(cell));
    const unsigned char* mb = d->map.load(std::memory_order_acquire);
    const unsigned char* m = mb ? mb + off : nullptr;             // headers come from the mapping once there is one: no system call
    if (d->has_cell) {
        if (m) memcpy(rec, m, sizeof(rec));
        else if (!read_at(d, rec, sizeof(rec), off)) return false;
        if (u32(d, rec) != 48 || u32(d, rec + 52) != 48) return false;
        cell = decode_cell(d, rec + 4);
    }
    for (int a = 0; a < 3; ++a) {                                   // the three record markers: a frame that fails here goes through load_frame
        unsigned char mark[4];
        const size_t at = lead + (size_t)a * (nbytes + 8);
        if (m) memcpy(mark, m + at, 4);
        else if (!read_at(d, mark, 4, off + at)) return false;
        if (u32(d, mark) != nbytes) return false;
    }
    memset(info, 0, sizeof(*info));
    info->codec = VMD_RAW_CODEC_F32;
    for (int a = 0; a < 3; ++a) info->f32_offset[a] = lead + (uint64_t)a * (nbytes + 8) + 4;
    info->f32_stride = 1;
    info->f32_flags = d->swap ? VMD_RAW_F32_BIG_ENDIAN : 0u;      // "the other byte order": the kernel swaps, whatever the host's is called
    info->f32_scale = 1.0f;
    info->nbytes = lead + 3 * (nbytes + 8) - 4;
    if (hdr) {
        memset(hdr, 0, sizeof(*hdr));
        hdr->num_atoms = d->num_atoms;
        hdr->index = idx;
        hdr->timestamp = (double)idx;
        hdr->unitcell = cell;
    }
    return true;
}

bool dcd_raw_mapped_view(void* inst, vmd_raw_mapped_view_t* out) {
    Dcd* d = (Dcd*)inst;
    if (!out || d->num_frames == 0) return false;
    std::lock_guard<std::mutex> lk(d->map_mtx);
    if (d->map_failed) return false;
    if (!d->map.load()) {
        struct stat sb;
        const size_t need = d->header_bytes + d->num_frames * d->frame_bytes;
        if (fstat(d->fd, &sb) != 0 || (size_t)sb.st_size < need) { d->map_failed = true; return false; }
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, d->fd, 0);
        if (m == MAP_FAILED) { d->map_failed = true; return false; }
        (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        d->stream_off.resize(d->num_frames);
        for (size_t i = 0; i < d->num_frames; ++i) d->stream_off[i] = d->header_bytes + i * d->frame_bytes;
        d->map_bytes = (size_t)sb.st_size;
        d->map.store((const unsigned char*)m, std::memory_order_release);
    }
    out->base = d->map.load();
    out->bytes = d->map_bytes;
    out->stream_offset = d->stream_off.data();
    out->codec = VMD_RAW_CODEC_F32;
    return true;
}

}  // namespace

struct vmd_dcdtraj_t { Dcd d; };

extern "C" vmd_dcdtraj_t* vmd_dcdtraj_open(const char* path) {
    if (!path) { vmd_set_last_error("vmd_dcdtraj_open: path is NULL"); return nullptr; }
    vmd_dcdtraj_t* t = new vmd_dcdtraj_t();
    Dcd& d = t->d;
    d.path = path;
    auto bail = [&](const char* fmt) -> vmd_dcdtraj_t* {
        fail(fmt, d.path);
        if (d.fd >= 0) close(d.fd);
        delete t;
        return nullptr;
    };
    d.fd = open(path, O_RDONLY);
    if (d.fd < 0) return bail("DCD '%s': cannot open");
    struct stat sb;
    if (fstat(d.fd, &sb) != 0) return bail("DCD '%s': cannot stat");
    const size_t file_bytes = (size_t)sb.st_size;
    unsigned char h[92];
    if (!read_at(&d, h, sizeof(h), 0)) return bail("DCD '%s': shorter than a header");
    uint32_t first;
    memcpy(&first, h, 4);
    if (first == 84) d.swap = false;
    else if (bswap32(first) == 84) d.swap = true;
    else return bail("DCD '%s': not a DCD file (first record is not 84 bytes)");
    if (memcmp(h + 4, "CORD", 4) != 0) return bail("DCD '%s': not a coordinate DCD (no CORD tag)");
    uint32_t ic[20];
    for (int i = 0; i < 20; ++i) ic[i] = u32(&d, h + 8 + 4 * i);
    if (u32(&d, h + 88) != 84) return bail("DCD '%s': corrupt header record");
    if (ic[8] != 0) return bail("DCD '%s': fixed atoms are not supported");
    d.has_cell = ic[10] != 0;
    d.has_4d = ic[11] != 0;
    // title record
    size_t off = 92;
    unsigned char m[4];
    if (!read_at(&d, m, 4, off)) return bail("DCD '%s': truncated title");
    const size_t title_bytes = u32(&d, m);
    off += 4 + title_bytes;
    if (!read_at(&d, m, 4, off) || u32(&d, m) != title_bytes) return bail("DCD '%s': corrupt title record");
    off += 4;
    unsigned char na[12];
    if (!read_at(&d, na, 12, off) || u32(&d, na) != 4 || u32(&d, na + 8) != 4) return bail("DCD '%s': corrupt atom-count record");
    d.num_atoms = u32(&d, na + 4);
    off += 12;
    if (d.num_atoms == 0) return bail("DCD '%s': no atoms");
    d.header_bytes = off;
    d.frame_bytes = (d.has_cell ? 56 : 0) + (size_t)(d.has_4d ? 4 : 3) * (4 * d.num_atoms + 8);
    // the frame count of the header is not always kept up to date by writers: trust the file size
    const size_t by_size = (file_bytes - d.header_bytes) / d.frame_bytes;
    d.num_frames = ic[0] ? std::min<size_t>(ic[0], by_size) : by_size;
    if (d.num_frames == 0) return bail("DCD '%s': no complete frame");
    d.iface.inst = &t->d;
    d.iface.num_frames = dcd_num_frames;
    d.iface.num_atoms = dcd_num_atoms;
    d.iface.load_frame = dcd_load_frame;
    d.iface.device_view = nullptr;
    d.iface.host_view = nullptr;
    d.iface.load_raw = dcd_load_raw;
    d.iface.raw_device_view = nullptr;
    d.iface.raw_mapped_view = dcd_raw_mapped_view;
    return t;
}

extern "C" void vmd_mapreg_drop(const void* base);       // vmd_eval_traj.cpp: the pinned windows of this mapping
extern "C" void vmd_dcdtraj_close(vmd_dcdtraj_t* t) {
    if (!t) return;
    if (const unsigned char* m = t->d.map.load()) {
        vmd_mapreg_drop(m);
        munmap((void*)m, t->d.map_bytes);
    }
    if (t->d.fd >= 0) close(t->d.fd);
    delete t;
}

extern "C" vmd_trajectory_i* vmd_dcdtraj_interface(vmd_dcdtraj_t* t) { return t ? &t->d.iface : nullptr; }
