// viamd_amd/csrc/vmd_eval_traj.cpp - trajectories the library itself holds: resident in HBM (vmd_devtraj), compressed-resident
// XTC streams (vmd_rawtraj), pinned host frames (vmd_hosttraj); plus two process-wide caches the staging code uses - decoder
// checkpoints of XTC streams (with their sidecar file) and pinned windows of mapped trajectory files.
#include "vmd_eval_internal.h"

size_t record_stride_for(size_t frames, size_t atoms, int level) {
    if (g_opt.xtc_records.load() < level || g_opt.xtc_device_decode.load() != 3) return 0;
    const size_t stride = (atoms + 63) & ~(size_t)63;
    const size_t budget = (size_t)std::max(0, g_opt.xtc_record_mb.load()) << 20;
    return (frames && stride * 2 <= budget / frames) ? stride : 0;
}

std::mutex g_ck_mtx;

std::map<CkKey, std::shared_ptr<CkCache>> g_ck_store;

std::shared_ptr<CkCache> ckcache_for(const void* inst_, size_t frames, size_t atoms, int device) {
    std::lock_guard<std::mutex> l(g_ck_mtx);
    const CkKey inst(inst_, device);
    std::shared_ptr<CkCache>& c = g_ck_store[inst];
    if (!c || c->frames != frames || c->atoms != atoms || c->device != device) {
        c = std::make_shared<CkCache>();
        c->frames = frames; c->atoms = atoms; c->device = device;
        c->have.assign(frames, 0);
        c->sig.assign(frames, 0);
        if (!c->ck.ensure(std::max<size_t>(frames, 1) * VMD_XTC_CK_MAX) || !c->nck.ensure(std::max<size_t>(frames,
                1))) { g_ck_store.erase(inst); return nullptr; }
        c->rec_stride = record_stride_for(frames, atoms);
        if (c->rec_stride && (!c->rec.ensure(frames * c->rec_stride) || !c->nrec.ensure(frames))) { (void)hipGetLastError();
                c->rec.release(); c->nrec.release(); c->rec_stride = 0; }
        if (g_ck_store.size() > 16) {                       // a handful of open trajectories at most: forget the others
            for (auto it = g_ck_store.begin(); it != g_ck_store.end();) it = it->first == inst ? std::next(it) : g_ck_store.erase(it);
        }
    }
    return c;
}

extern "C" void vmd_ckcache_drop(const void* inst) {
    std::lock_guard<std::mutex> l(g_ck_mtx);
    for (auto it = g_ck_store.lower_bound(CkKey(inst, INT_MIN)); it != g_ck_store.end() && it->first.first == inst;) it =
            g_ck_store.erase(it);
}

extern "C" bool vmd_ckcache_save(const vmd_trajectory_i* traj, const char* path) {
    g_last_error.clear();
    if (!traj || !path) return vmd_fail("vmd_ckcache_save: NULL argument");
    std::shared_ptr<CkCache> c;
    { std::lock_guard<std::mutex> l(g_ck_mtx);
      auto it = g_ck_store.lower_bound(CkKey(traj->inst, INT_MIN));       // whichever device decoded it: the table describes the file
      if (it != g_ck_store.end() && it->first.first == traj->inst) c = it->second; }
    if (!c || c->frames == 0)
        return vmd_fail("vmd_ckcache_save: no decoder checkpoints exist for this trajectory (nothing of it was decoded on the device yet)");
    int prev = 0;
    HIP_OK(hipGetDevice(&prev));
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipDeviceSynchronize());                       // the tables are written by decode kernels on the evals' streams
    std::vector<uint32_t> nck(c->frames);
    std::vector<vmd_xtc_ck_t> ck(c->frames * VMD_XTC_CK_MAX);
    const bool copied = hipMemcpy(nck.data(), c->nck.p, nck.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess &&
                        hipMemcpy(ck.data(), c->ck.p, ck.size() * sizeof(vmd_xtc_ck_t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipSetDevice(prev);
    if (!copied) return vmd_fail("vmd_ckcache_save: reading the checkpoint tables back failed");
    CkFileHeader h;
    memcpy(h.magic, kCkMagic, 8);
    h.version = 1; h.ck_max = VMD_XTC_CK_MAX; h.frames = c->frames; h.atoms = c->atoms;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return vmd_fail("vmd_ckcache_save: cannot create %s", tmp.c_str());
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(c->have.data(), 1, c->frames, f) == c->frames &&
              fwrite(c->sig.data(), sizeof(uint64_t), c->frames, f) == c->frames && fwrite(nck.data(), sizeof(uint32_t), nck.size(),
                      f) == nck.size() &&
              fwrite(ck.data(), sizeof(vmd_xtc_ck_t), ck.size(), f) == ck.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return vmd_fail("vmd_ckcache_save: writing %s failed", path); }
    return true;
}

// -> number of frames whose checkpoints were installed (0: the file does not describe this trajectory), -1 on error
extern "C" long vmd_ckcache_load(const vmd_trajectory_i* traj, const char* path, int device) {
    g_last_error.clear();
    if (!traj || !path) { vmd_fail("vmd_ckcache_load: NULL argument"); return -1; }
    FILE* f = fopen(path, "rb");
    if (!f) { vmd_fail("vmd_ckcache_load: cannot open %s", path); return -1; }
    CkFileHeader h;
    const size_t frames = traj->num_frames(traj->inst), atoms = traj->num_atoms(traj->inst);
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, kCkMagic, 8) != 0 || h.version != 1) { fclose(f);
            vmd_fail("vmd_ckcache_load: %s is not a checkpoint file", path); return -1; }
    if (h.ck_max != VMD_XTC_CK_MAX || h.frames != frames || h.atoms != atoms || frames == 0) { fclose(f); return 0; }
    std::vector<uint8_t> have(frames);
    std::vector<uint64_t> sig(frames);
    std::vector<uint32_t> nck(frames);
    std::vector<vmd_xtc_ck_t> ck(frames * VMD_XTC_CK_MAX);
    const bool ok = fread(have.data(), 1, frames, f) == frames && fread(sig.data(), sizeof(uint64_t), frames, f) == frames &&
                    fread(nck.data(), sizeof(uint32_t), frames, f) == frames && fread(ck.data(), sizeof(vmd_xtc_ck_t), ck.size(),
                            f) == ck.size();
    fclose(f);
    if (!ok) { vmd_fail("vmd_ckcache_load: %s is truncated", path); return -1; }
    long n = 0;
    for (size_t i = 0; i < frames; ++i) {
        if (have[i] && (nck[i] < 1 || nck[i] > VMD_XTC_CK_MAX)) have[i] = 0;       // nothing the kernels would accept anyway
        n += have[i] ? 1 : 0;
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { vmd_fail("vmd_ckcache_load: no such device"); return -1;
            }
    std::shared_ptr<CkCache> c = ckcache_for(traj->inst, frames, atoms, device);
    bool up = c != nullptr;
    if (up) {
        (void)hipDeviceSynchronize();                     // an eval may be decoding this trajectory from the tables being replaced
        up = hipMemcpy(c->nck.p, nck.data(), nck.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->ck.p, ck.data(), ck.size() * sizeof(vmd_xtc_ck_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    (void)hipSetDevice(prev);
    if (!up) { vmd_fail("vmd_ckcache_load: uploading the checkpoint tables failed"); return -1; }
    std::lock_guard<std::mutex> l(g_ck_mtx);
    std::copy(have.begin(), have.end(), c->have.begin());        // in place: stages of a running eval point into these vectors
    std::copy(sig.begin(), sig.end(), c->sig.begin());
    c->rec_failed = true;                                  // no records came with them: sections from the checkpoints
    return n;
}

std::mutex g_map_mtx;

std::map<const unsigned char*, MapReg> g_map_store;

size_t g_map_pinned = 0;

void mapreg_release(const unsigned char* base, MapReg& m) {
    for (size_t w = 0; w < m.state.size(); ++w) {
        if (m.state[w] != 1) continue;
        (void)hipHostUnregister((void*)(base + w * kMapWindow));
        g_map_pinned -= std::min(kMapWindow, m.bytes - w * kMapWindow);
    }
    m.state.clear();
}

bool mapreg_pin(const unsigned char* base, size_t bytes, size_t lo, size_t hi) {
    std::lock_guard<std::mutex> l(g_map_mtx);
    MapReg& m = g_map_store[base];
    if (m.bytes != bytes) {                  // a new mapping at a recycled address whose owner never dropped the old one
        mapreg_release(base, m);
        m.bytes = bytes;
        m.state.assign((bytes + kMapWindow - 1) / kMapWindow, 0);
    }
    size_t limit = (size_t)std::max(0, g_opt.xtc_map_limit_mb.load()) << 20;
    if (!limit) {
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
        limit = (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz / 2 : ((size_t)8 << 30);
    }
    for (size_t w = lo / kMapWindow; w <= (hi - 1) / kMapWindow; ++w) {
        if (m.state[w] == 1) continue;
        if (m.state[w] == 2) return false;
        const size_t len = std::min(kMapWindow, bytes - w * kMapWindow);
        if (g_map_pinned + len > limit || hipHostRegister((void*)(base + w * kMapWindow), len, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            m.state[w] = 2;
            return false;
        }
        m.state[w] = 1;
        g_map_pinned += len;
    }
    return true;
}

extern "C" void vmd_mapreg_drop(const void* base) {
    std::lock_guard<std::mutex> l(g_map_mtx);
    auto it = g_map_store.find((const unsigned char*)base);
    if (it == g_map_store.end()) return;
    mapreg_release(it->first, it->second);
    g_map_store.erase(it);
}

uint64_t frame_signature(const vmd_xtc_frame_t& fi, const unsigned char* bytes) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ fi.nbytes;
    auto mix = [&](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); };
    uint32_t p; memcpy(&p, &fi.precision, 4);
    mix(p); mix((uint32_t)fi.smallidx);
    for (int k = 0; k < 3; ++k) { mix((uint32_t)fi.minint[k]); mix((uint32_t)fi.maxint[k]); }
    const size_t n = fi.nbytes < 64 ? (size_t)fi.nbytes : 64;
    for (size_t i = 0; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, bytes + i, 8); mix(w); }
    if (fi.nbytes >= 72) { uint64_t w; memcpy(&w, bytes + fi.nbytes - 8, 8); mix(w); }
    return h | 1ull;
}

// ------------------------------------------------------------------------------------------------ device trajectory
uint64_t next_cells_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1) + 1;
}

size_t dt_num_frames(void* inst) { return ((vmd_devtraj_t*)inst)->num_frames; }

size_t dt_num_atoms(void* inst) { return ((vmd_devtraj_t*)inst)->num_atoms; }

bool dt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_devtraj_t* t = (vmd_devtraj_t*)inst;
    if (idx < 0 || !t->has((size_t)idx, (size_t)idx + 1)) return vmd_fail("devtraj: frame %lld is not resident on this rank",
            (long long)idx);
    const float* f = t->frame((size_t)idx);
    if (x) HIP_OK(hipMemcpy(x, f, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (y) HIP_OK(hipMemcpy(y, f + t->npad, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (z) HIP_OK(hipMemcpy(z, f + 2 * t->npad, t->num_atoms * sizeof(float), hipMemcpyDeviceToHost));
    if (hdr) { hdr->num_atoms = t->num_atoms; hdr->index = idx; hdr->timestamp = (double)idx; hdr->unitcell = t->cells[idx]; }
    return true;
}

bool dt_device_view(void* inst, vmd_device_view_t* out) {
    vmd_devtraj_t* t = (vmd_devtraj_t*)inst;
    // frame f sits at base + f * frame_stride: for a shard the base lies `first` frames before the allocation and is only ever
    // used with resident frame indices (the evaluator is handed ranges inside the shard)
    out->base = t->d - t->first * 3 * t->npad; out->frame_stride = 3 * t->npad; out->row_stride = t->npad; out->cells = t->cells.data();
            out->device = t->device;
    out->resident_beg = t->first; out->resident_end = t->first + t->resident;
    out->cells_version = t->cells_version;
    return true;
}

extern "C" vmd_devtraj_t* vmd_devtraj_create_shard(size_t num_frames, size_t frame_beg, size_t frame_end, size_t num_atoms) {
    if (vmd_device_count() <= 0) { vmd_fail("vmd_devtraj_create: no usable HIP device"); return nullptr; }
    if (frame_beg > frame_end || frame_end > num_frames) { vmd_fail("vmd_devtraj_create_shard: bad frame range"); return nullptr; }
    auto t = std::make_unique<vmd_devtraj_t>();
    t->num_frames = num_frames; t->num_atoms = num_atoms; t->npad = (num_atoms + 63) & ~(size_t)63;
    t->first = frame_beg; t->resident = frame_end - frame_beg;
    if (hipGetDevice(&t->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    const size_t bytes = std::max<size_t>(t->resident * 3 * t->npad, 1) * sizeof(float);
    hipError_t err = hipMalloc((void**)&t->d, bytes);
    if (err == hipSuccess && frame_beg > 0) err = hipMalloc((void**)&t->d0, 3 * t->npad * sizeof(float));
    if (err != hipSuccess) { vmd_fail("vmd_devtraj_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err)); return nullptr; }
    vmd_unitcell_t none;
    memset(&none, 0, sizeof(none));
    t->cells.assign(num_frames, none);
    t->iface.inst = t.get();
    t->iface.num_frames = dt_num_frames; t->iface.num_atoms = dt_num_atoms;
    t->iface.load_frame = dt_load_frame; t->iface.device_view = dt_device_view; t->iface.host_view = nullptr;
    t->iface.load_raw = nullptr;
    t->iface.raw_device_view = nullptr;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}

extern "C" vmd_devtraj_t* vmd_devtraj_create(size_t num_frames, size_t num_atoms) { return vmd_devtraj_create_shard(num_frames, 0,
        num_frames, num_atoms); }

extern "C" void vmd_devtraj_free(vmd_devtraj_t* t) {
    if (!t) return;
    if (t->d) (void)hipFree(t->d);
    if (t->d0) (void)hipFree(t->d0);
    delete t;
}

extern "C" vmd_trajectory_i* vmd_devtraj_interface(vmd_devtraj_t* t) { return t ? &t->iface : nullptr; }

extern "C" bool vmd_devtraj_upload_frame(vmd_devtraj_t* t, size_t frame, const vmd_unitcell_t* cell,
                                         const float* x, const float* y, const float* z) {
    if (!t || !t->has(frame, frame + 1)) return vmd_fail("vmd_devtraj_upload_frame: bad frame");
    float* f = t->frame(frame);
    HIP_OK(hipMemcpy(f, x, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f + t->npad, y, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f + 2 * t->npad, z, t->num_atoms * sizeof(float), hipMemcpyHostToDevice));
    if (cell) t->cells[frame] = *cell;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" bool vmd_devtraj_upload_atoms(vmd_devtraj_t* t, size_t frame_beg, size_t frame_count, size_t first_atom, size_t atom_count,
                                        const float* xyz /* [frame_count][3][atom_count] */) {
    if (!t || !t->has(frame_beg, frame_beg + frame_count) || first_atom
            + atom_count > t->num_atoms) return vmd_fail("vmd_devtraj_upload_atoms: bad range");
    for (size_t f = 0; f < frame_count; ++f)
        for (int c = 0; c < 3; ++c)
            HIP_OK(hipMemcpyAsync(t->frame(frame_beg + f) + (size_t)c * t->npad + first_atom,
                                  xyz + (f * 3 + c) * atom_count, atom_count * sizeof(float), hipMemcpyHostToDevice, nullptr));
    HIP_OK(hipDeviceSynchronize());
    t->cells_version = next_cells_version();       // coordinates changed: bounding boxes cached per range (open axes) are stale too
    return true;
}

extern "C" bool vmd_devtraj_synth(vmd_devtraj_t* t, uint64_t seed, float L, float sigma, uint32_t n_blob,
                                  size_t frame_beg, size_t frame_end) {
    if (!t || !t->has(frame_beg, frame_end)) return vmd_fail("vmd_devtraj_synth: bad range");
    vmd_unitcell_t c;
    memset(&c, 0, sizeof(c));
    c.x = c.y = c.z = L; c.flags = VMD_UNITCELL_PBC_ALL;
    for (size_t f0 = frame_beg; f0 < frame_end; f0 += 1024) {
        const size_t nb = std::min<size_t>(1024, frame_end - f0);
        KRN_OK(vmd_hip_synth_frames(nullptr, t->frame(f0), 3 * t->npad, t->npad, (int)nb, (uint32_t)f0, seed,
                                    (uint32_t)t->num_atoms, n_blob, L, sigma));
    }
    if (t->d0) {    // the shard's copy of frame 0
        KRN_OK(vmd_hip_synth_frames(nullptr, t->d0, 3 * t->npad, t->npad, 1, 0u, seed, (uint32_t)t->num_atoms, n_blob, L, sigma));
        t->cells[0] = c;
    }
    HIP_OK(hipDeviceSynchronize());
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = c;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" bool vmd_devtraj_set_cell(vmd_devtraj_t* t, size_t frame_beg, size_t frame_end, const vmd_unitcell_t* cell) {
    if (!t || !cell || frame_end > t->num_frames || frame_beg > frame_end) return vmd_fail("vmd_devtraj_set_cell: bad frame range");
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = *cell;
    t->cells_version = next_cells_version();
    return true;
}

extern "C" float* vmd_devtraj_device_ptr(vmd_devtraj_t* t, size_t* frame_stride, size_t* row_stride) {
    if (!t) return nullptr;
    if (frame_stride) *frame_stride = 3 * t->npad;
    if (row_stride) *row_stride = t->npad;
    return t->d;
}

size_t rt_num_frames(void* inst) { return ((vmd_rawtraj_t*)inst)->num_frames; }

size_t rt_num_atoms(void* inst) { return ((vmd_rawtraj_t*)inst)->num_atoms; }

bool rt_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    return t->src->load_frame(t->src->inst, idx, hdr, x, y, z);
}

bool rt_load_raw(void* inst, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst, size_t cap) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    return t->src->load_raw(t->src->inst, idx, hdr, info, dst, cap);
}

bool rt_raw_device_view(void* inst, vmd_raw_device_view_t* out) {
    vmd_rawtraj_t* t = (vmd_rawtraj_t*)inst;
    out->base = t->d_raw; out->info = t->d_info; out->cells = t->cells.data(); out->codec = VMD_RAW_CODEC_XTC; out->device = t->device;
    out->ck = t->d_ck; out->nck = t->d_nck; out->ck_have = t->d_ck ? t->ck_have.data() : nullptr;
    out->rec = t->d_rec; out->nrec = t->d_nrec; out->rec_stride = t->d_rec ? t->rec_stride : 0; out->rec_failed = &t->rec_failed;
    return true;
}

extern "C" void vmd_rawtraj_free(vmd_rawtraj_t* t) {
    if (!t) return;
    if (t->d_raw) (void)hipFree(t->d_raw);
    if (t->d_info) (void)hipFree(t->d_info);
    if (t->d_ck) (void)hipFree(t->d_ck);
    if (t->d_nck) (void)hipFree(t->d_nck);
    if (t->d_rec) (void)hipFree(t->d_rec);
    if (t->d_nrec) (void)hipFree(t->d_nrec);
    delete t;
}

extern "C" vmd_rawtraj_t* vmd_rawtraj_create(vmd_trajectory_i* src) {
    if (!src || !src->load_raw) { vmd_fail("vmd_rawtraj_create: the trajectory does not offer its frames compressed (load_raw)");
            return nullptr; }
    if (vmd_device_count() <= 0) { vmd_fail("vmd_rawtraj_create: no usable HIP device"); return nullptr; }
    std::unique_ptr<vmd_rawtraj_t, void (*)(vmd_rawtraj_t*)> t(new vmd_rawtraj_t(), vmd_rawtraj_free);
    t->src = src;
    t->num_frames = src->num_frames(src->inst);
    t->num_atoms = src->num_atoms(src->inst);
    if (hipGetDevice(&t->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    const size_t F = t->num_frames;
    std::vector<vmd_xtc_frame_t> info(F);
    t->cells.resize(F);
    size_t total = 0;
    for (size_t f = 0; f < F; ++f) {
        vmd_frame_header_t hdr;
        vmd_raw_frame_t ri;
        if (!src->load_raw(src->inst, (int64_t)f, &hdr, &ri, nullptr, 0) || ri.codec != VMD_RAW_CODEC_XTC
                || hdr.num_atoms != t->num_atoms) {
            vmd_fail("vmd_rawtraj_create: frame %zu is not available compressed", f);
            return nullptr;
        }
        t->cells[f] = hdr.unitcell;
        info[f].precision = ri.precision;
        for (int k = 0; k < 3; ++k) { info[f].minint[k] = ri.minint[k]; info[f].maxint[k] = ri.maxint[k]; }
        info[f].smallidx = ri.smallidx;
        info[f].offset = total;
        info[f].nbytes = ri.nbytes;
        total += ((size_t)ri.nbytes + 32 + 63) & ~(size_t)63;            // the layout the decode kernels expect (vmd_hip.h)
    }
    t->bytes = total;
    hipError_t err = hipMalloc((void**)&t->d_raw, std::max<size_t>(total, 64));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_info, std::max<size_t>(F, 1) * sizeof(vmd_xtc_frame_t));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_ck, std::max<size_t>(F, 1) * VMD_XTC_CK_MAX * sizeof(vmd_xtc_ck_t));
    if (err == hipSuccess) err = hipMalloc((void**)&t->d_nck, std::max<size_t>(F, 1) * sizeof(uint32_t));
    t->ck_have.assign(F, 0);
    if (err != hipSuccess) { vmd_fail("vmd_rawtraj_create: hipMalloc(%zu) failed: %s", total, hipGetErrorString(err)); return nullptr; }
    t->rec_stride = record_stride_for(F, t->num_atoms, 2);
    if (t->rec_stride && (hipMalloc((void**)&t->d_rec, F * t->rec_stride * sizeof(uint16_t)) != hipSuccess || hipMalloc((void**)&t->d_nrec,
            F * sizeof(uint32_t)) != hipSuccess)) {
        (void)hipGetLastError();                   // no room for the records: the sections are walked from their checkpoints as before
        if (t->d_rec) (void)hipFree(t->d_rec);
        t->d_rec = nullptr; t->rec_stride = 0;
    }
    if (F && hipMemcpy(t->d_info, info.data(), F * sizeof(vmd_xtc_frame_t),
            hipMemcpyHostToDevice) != hipSuccess) { vmd_fail("vmd_rawtraj_create: upload failed"); return nullptr; }
    // upload in pinned pieces of <= 256 MB, each filled by the load threads
    const size_t piece_cap = std::min<size_t>(std::max<size_t>(total, 64), (size_t)256 << 20);
    unsigned char* pin = nullptr;
    size_t pin_cap = 0;
    auto grow = [&](size_t need) {
        if (need <= pin_cap) return true;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; pin_cap = 0;
        if (hipHostMalloc((void**)&pin, need, hipHostMallocDefault) != hipSuccess) return false;
        pin_cap = need;
        return true;
    };
    bool good = true;
    for (size_t f0 = 0; f0 < F && good;) {
        size_t f1 = f0, piece = 0;
        while (f1 < F && (f1 == f0 || piece + (info[f1].offset + (((size_t)info[f1].nbytes + 32 + 63) & ~(size_t)63)
                - info[f1].offset) <= piece_cap)) {
            piece = info[f1].offset + (((size_t)info[f1].nbytes + 32 + 63) & ~(size_t)63) - info[f0].offset;
            ++f1;
        }
        if (!grow(piece)) { vmd_fail("hipHostMalloc(%zu bytes) failed", piece); good = false; break; }
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), (f1 - f0) / 4));
        std::atomic<size_t> next{f0};
        std::atomic<bool> ok{true};
        auto work = [&]() {
            for (;;) {
                const size_t f = next.fetch_add(1);
                if (f >= f1 || !ok.load()) break;
                unsigned char* dst = pin + (info[f].offset - info[f0].offset);
                vmd_raw_frame_t ri;
                if (!src->load_raw(src->inst, (int64_t)f, nullptr, &ri, dst, (size_t)info[f].nbytes)
                        || ri.nbytes != info[f].nbytes) { ok = false; break; }
                memset(dst + info[f].nbytes, 0, (((size_t)info[f].nbytes + 32 + 63) & ~(size_t)63) - (size_t)info[f].nbytes);
            }
        };
        if (nthreads == 1) work();
        else {
            std::vector<std::thread> pool;
            for (size_t k = 1; k < nthreads; ++k) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
        }
        if (!ok.load()) { if (g_last_error.empty()) vmd_fail("vmd_rawtraj_create: reading the compressed frames failed"); good = false;
                break; }
        if (hipMemcpy(t->d_raw + info[f0].offset, pin, piece,
                hipMemcpyHostToDevice) != hipSuccess) { vmd_fail("vmd_rawtraj_create: upload failed"); good = false; break; }
        f0 = f1;
    }
    if (pin) (void)hipHostFree(pin);
    if (!good) return nullptr;
    t->iface.inst = t.get();
    t->iface.num_frames = rt_num_frames; t->iface.num_atoms = rt_num_atoms;
    t->iface.load_frame = rt_load_frame; t->iface.device_view = nullptr; t->iface.host_view = nullptr;
    t->iface.load_raw = rt_load_raw;
    t->iface.raw_device_view = rt_raw_device_view;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}

extern "C" vmd_trajectory_i* vmd_rawtraj_interface(vmd_rawtraj_t* t) { return t ? &t->iface : nullptr; }

extern "C" size_t vmd_rawtraj_device_bytes(const vmd_rawtraj_t* t) {
    if (!t) return 0;           // everything the object keeps in HBM: bit streams, frame table, checkpoints, group records
    return t->bytes + t->num_frames * (sizeof(vmd_xtc_frame_t) + VMD_XTC_CK_MAX * sizeof(vmd_xtc_ck_t) + sizeof(uint32_t)) +
           (t->d_rec ? t->num_frames * (t->rec_stride * sizeof(uint16_t) + sizeof(uint32_t)) : 0);
}

size_t ht_num_frames(void* inst) { return ((vmd_hosttraj_t*)inst)->num_frames; }

size_t ht_num_atoms(void* inst) { return ((vmd_hosttraj_t*)inst)->num_atoms; }

bool ht_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    vmd_hosttraj_t* t = (vmd_hosttraj_t*)inst;
    if (idx < 0 || (size_t)idx >= t->num_frames) return vmd_fail("hosttraj: frame %lld out of range", (long long)idx);
    const float* f = t->h + (size_t)idx * 3 * t->npad;
    if (x) memcpy(x, f, t->num_atoms * sizeof(float));
    if (y) memcpy(y, f + t->npad, t->num_atoms * sizeof(float));
    if (z) memcpy(z, f + 2 * t->npad, t->num_atoms * sizeof(float));
    if (hdr) { hdr->num_atoms = t->num_atoms; hdr->index = idx; hdr->timestamp = (double)idx; hdr->unitcell = t->cells[idx]; }
    return true;
}

bool ht_host_view(void* inst, vmd_host_view_t* out) {
    vmd_hosttraj_t* t = (vmd_hosttraj_t*)inst;
    out->base = t->h; out->frame_stride = 3 * t->npad; out->row_stride = t->npad; out->cells = t->cells.data();
    return true;
}

extern "C" vmd_hosttraj_t* vmd_hosttraj_create(size_t num_frames, size_t num_atoms) {
    auto t = std::make_unique<vmd_hosttraj_t>();
    t->num_frames = num_frames; t->num_atoms = num_atoms; t->npad = (num_atoms + 63) & ~(size_t)63;
    const size_t bytes = std::max<size_t>(num_frames * 3 * t->npad, 1) * sizeof(float);
    if (hipHostMalloc((void**)&t->h, bytes,
            hipHostMallocDefault) != hipSuccess) { vmd_fail("vmd_hosttraj_create: hipHostMalloc(%zu) failed", bytes); return nullptr; }
    vmd_unitcell_t none;
    memset(&none, 0, sizeof(none));
    t->cells.assign(num_frames, none);
    t->iface.inst = t.get();
    t->iface.num_frames = ht_num_frames; t->iface.num_atoms = ht_num_atoms; t->iface.load_frame = ht_load_frame;
    t->iface.device_view = nullptr; t->iface.host_view = ht_host_view;
    t->iface.load_raw = nullptr;
    t->iface.raw_device_view = nullptr;
    t->iface.raw_mapped_view = nullptr;
    return t.release();
}

extern "C" void vmd_hosttraj_free(vmd_hosttraj_t* t) { if (!t) return; if (t->h) (void)hipHostFree(t->h); delete t; }

extern "C" vmd_trajectory_i* vmd_hosttraj_interface(vmd_hosttraj_t* t) { return t ? &t->iface : nullptr; }

extern "C" float* vmd_hosttraj_frame_ptr(vmd_hosttraj_t* t, size_t frame, size_t* row_stride) {
    if (!t || frame >= t->num_frames) return nullptr;
    if (row_stride) *row_stride = t->npad;
    return t->h + frame * 3 * t->npad;
}

extern "C" bool vmd_hosttraj_set_cell(vmd_hosttraj_t* t, size_t frame, const vmd_unitcell_t* cell) {
    if (!t || !cell || frame >= t->num_frames) return vmd_fail("vmd_hosttraj_set_cell: bad frame");
    t->cells[frame] = *cell;
    return true;
}

extern "C" bool vmd_hosttraj_copy_from_device(vmd_hosttraj_t* t, vmd_devtraj_t* src, size_t frame_beg, size_t frame_end) {
    if (!t || !src || frame_end > t->num_frames || frame_end > src->num_frames
            || src->num_atoms != t->num_atoms) return vmd_fail("vmd_hosttraj_copy_from_device: shape mismatch");
    if (frame_beg >= frame_end) return true;
    if (!src->has(frame_beg, frame_end)) return vmd_fail("vmd_hosttraj_copy_from_device: frames are not resident on this rank");
    HIP_OK(hipMemcpy(t->h + frame_beg * 3 * t->npad, src->frame(frame_beg), (frame_end - frame_beg) * 3 * t->npad * sizeof(float),
            hipMemcpyDeviceToHost));
    for (size_t f = frame_beg; f < frame_end; ++f) t->cells[f] = src->cells[f];
    return true;
}
