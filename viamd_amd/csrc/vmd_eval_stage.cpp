// viamd_amd/csrc/vmd_eval_stage.cpp - getting a batch of frames to where the kernels read it: the static per-eval uploads (index lists,
// masses, bond trees), frames read in place from HBM (device views), frames staged through pinned memory behind the copy engine, and
// compressed XTC frames uploaded raw and decoded on the device next to the pair kernel (/root/reference/src/loader.cpp:111-159 is the
// CPU-side counterpart: md_trajectory_load_frame behind a frame cache).
#include "vmd_eval_internal.h"

bool check_atoms(vmd_script_eval_t* e, size_t num_atoms) {
    if (e->atoms_checked == num_atoms) return true;          // the index lists never change: one pass per trajectory size
    for (auto& p : e->props) {
        for (int32_t i : p->prop.a)
            if ((size_t)i >= num_atoms)
                return vmd_fail("property '%s' references atom %d but the trajectory has %zu atoms", p->prop.name.c_str(), i, num_atoms);
        for (int32_t i : p->prop.b)
            if ((size_t)i >= num_atoms)
                return vmd_fail("property '%s' references atom %d but the trajectory has %zu atoms", p->prop.name.c_str(), i, num_atoms);
    }
    e->atoms_checked = num_atoms;
    return true;
}

// Device-side decompression of a staged batch (vmd_trajectory_i::load_raw + k_xtc_wave): the compressed bit streams are read
// into pinned memory on the decode threads, cross PCIe as they are (0.4x the float bytes for water) and are decompressed on the
// copy stream, i.e. under the kernels of the previous batch.  Nothing here waits for the device: the status words come back
// with the batch's `ready` event and are looked at when the batch is about to be used (settle_stage).  Returns 1 when the decode
// is queued into st.d, 0 when the batch has to go through load_frame (a frame is not available raw), -1 on error.
int launch_raw_decode(vmd_script_eval_t* e, Stage& st, const unsigned char* d_raw, const vmd_xtc_frame_t* d_info, size_t num_atoms,
                             size_t nb, size_t npad, hipStream_t stream, vmd_xtc_ck_t* ck, uint32_t* nck,
                             uint8_t* ck_have, uint16_t* rec, uint32_t* nrec, size_t rec_stride,
                             bool* rec_failed) {
    st.ck_mark = nullptr;
    st.ck_clear = nullptr;
    st.rec_failed = nullptr;
    st.sectioned = false;
    if (nb > st.h_raw_status_cap) {
        if (st.h_raw_status) pool_give(st.h_raw_status);
        st.h_raw_status = nullptr; st.h_raw_status_cap = 0;
        if (pool_take(kPinned, (void**)&st.h_raw_status, nb * sizeof(uint32_t)) != hipSuccess) { vmd_fail("hipHostMalloc failed");
                return -1; }
        st.h_raw_status_cap = nb;
    }
    if (!st.d.ensure(nb * 3 * npad) || !st.d_raw_status.ensure(nb)) return -1;
    int rc;
    const int mode = g_opt.xtc_device_decode.load();
    e->prof_copy.begin("xtc_decode", stream);
    if (mode == 2) {
        const int chunk = std::max(64, g_opt.xtc_chunk.load());
        if (!st.d_raw_scratch.ensure((vmd_hip_xtc_scratch_bytes((int)nb, (int)num_atoms, chunk) + 7) / 8)) return -1;
        rc = vmd_hip_xtc_decode_chunked(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad,
                                        st.d_raw_status.p, chunk, st.d_raw_scratch.p);
    } else if (mode == 1) {
        rc = vmd_hip_xtc_decode(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p);
    } else if (ck && nck && ck_have && g_opt.xtc_checkpoints.load()) {
        bool all = true;
        for (size_t b = 0; b < nb; ++b) all = all && flag_get(&ck_have[b]) != 0;
        // every frame of the batch has been decoded before: sections from its checkpoints; otherwise decode and leave checkpoints.
        // With group records next to the checkpoints (the first pass writes both) a later pass walks nothing at all.
        const bool recs = rec && nrec && rec_stride >= num_atoms && rec_failed && !flag_get(rec_failed) && g_opt.xtc_records.load();
        if (recs) {
            rc = vmd_hip_xtc_decode_wave_rec(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p, all
                    ? 1 : 0, ck, nck, rec, nrec, rec_stride);
            if (all) st.rec_failed = rec_failed;
        } else
        rc = vmd_hip_xtc_decode_wave_ck(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p, all ? 1
                : 0, ck, nck);
        if (!all) st.ck_mark = ck_have;
        else st.ck_clear = ck_have;
        st.sectioned = all;
    } else {
        rc = vmd_hip_xtc_decode_wave(stream, d_raw, d_info, (int)nb, (int)num_atoms, st.d.p, 3 * npad, npad, st.d_raw_status.p);
    }
    e->prof_copy.end(stream);
    if (rc != 0) { vmd_fail("XTC decode kernel launch failed"); return -1; }
    for (size_t b = 0; b < nb; ++b) st.h_raw_status[b] = 99u;
    if (hipMemcpyAsync(st.h_raw_status, st.d_raw_status.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost,
            stream) != hipSuccess) { vmd_fail("device XTC decode failed"); return -1; }
    st.raw_pending = true;
    return 1;
}

// Frames stored as plain floats: the copy engine takes the batch's span of the mapped file, k_raw_f32 turns it into the frame layout.
// 1 = queued, 0 = not this way (no mapping, not pinnable, option off), -1 error.
int raw_upload_f32(vmd_script_eval_t* e, vmd_script_eval_t::RawSlot& rs, vmd_trajectory_i* traj, const std::vector<vmd_raw_frame_t>& infos,
                          size_t num_atoms, size_t f0, size_t nb) {
    vmd_raw_mapped_view_t mv;
    if (!g_opt.raw_f32_device.load() || !g_opt.xtc_mapped.load() || !traj->raw_mapped_view || !traj->raw_mapped_view(traj->inst, &mv) ||
        mv.codec != VMD_RAW_CODEC_F32 || !mv.base || !mv.stream_offset) return 0;
    uint64_t lo64 = ~(uint64_t)0, hi64 = 0;
    for (size_t b = 0; b < nb; ++b) {
        const uint64_t so = mv.stream_offset[f0 + b];
        const vmd_raw_frame_t& fi = infos[b];
        if (fi.f32_stride == 0 || so + fi.nbytes > mv.bytes) return 0;
        for (int c = 0; c < 3; ++c)
            if (((so + fi.f32_offset[c]) & 3u) != 0 || fi.f32_offset[c] + 4ull * fi.f32_stride * (num_atoms - 1) + 4 > fi.nbytes) return 0;
        lo64 = std::min(lo64, so);
        hi64 = std::max(hi64, so + fi.nbytes);
    }
    const size_t lo = (size_t)(lo64 & ~(uint64_t)7), hi = (size_t)hi64;
    if (hi <= lo || !mapreg_pin(mv.base, mv.bytes, lo, hi)) return 0;
    HostTimer map_timer("host_raw_map");
    rs.f32.resize(nb);
    for (size_t b = 0; b < nb; ++b) {
        vmd_f32_frame_t& o = rs.f32[b];
        memset(&o, 0, sizeof(o));
        for (int c = 0; c < 3; ++c) o.offset[c] = mv.stream_offset[f0 + b] - lo + infos[b].f32_offset[c];
        o.stride = infos[b].f32_stride; o.flags = infos[b].f32_flags; o.scale = infos[b].f32_scale;
    }
    rs.info_bytes = (nb * sizeof(vmd_f32_frame_t) + 255) & ~(size_t)255;
    if (rs.info_bytes > rs.hcap) {
        if (rs.h) pool_give(rs.h);
        rs.h = nullptr; rs.hcap = 0;
        if (pool_take(kPinned, (void**)&rs.h, 2 * rs.info_bytes) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", 2
                * rs.info_bytes); return -1; }
        rs.hcap = 2 * rs.info_bytes;
    }
    memcpy(rs.h, rs.f32.data(), nb * sizeof(vmd_f32_frame_t));
    rs.h_streams = mv.base + lo;
    const size_t span = hi - lo;
    if (!rs.d.ensure(rs.info_bytes + span + span / 8 + 64)) return -1;
    if (hipMemcpyAsync(rs.d.p, rs.h, nb * sizeof(vmd_f32_frame_t), hipMemcpyHostToDevice,
            e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the frame table failed"); return -1; }
    for (size_t a = lo; a < hi;) {                          // one copy per pinned window the span touches
        const size_t stop = std::min(hi, (a / kMapWindow + 1) * kMapWindow);
        if (hipMemcpyAsync(rs.d.p + rs.info_bytes + (a - lo), mv.base + a, stop - a, hipMemcpyHostToDevice,
                e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync from the mapped trajectory file failed"); return -1; }
        a = stop;
    }
    if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
    e->frames_mapped += nb;
    rs.state = 1;
    return 1;
}

// First half of the compressed path: read the bit streams of frames [f0, f0 + nb) into the slot's pinned block (load threads) and queue
// their DMA on copy_stream.  1 = queued (slot.uploaded recorded), 0 = a frame is not available raw, -1 error.
int raw_upload(vmd_script_eval_t* e, RawSlot& rs, vmd_trajectory_i* traj, size_t num_atoms, size_t f0, size_t nb) {
    HostTimer host_timer("host_raw_upload");
    rs.state = 0; rs.f0 = f0; rs.nb = nb;
    rs.info.resize(nb);
    rs.cells.resize(nb);
    std::vector<vmd_raw_frame_t> infos(nb);
    size_t total = 0;
    for (size_t b = 0; b < nb; ++b) {                      // sizes first (no payload), then one pinned block for the batch
        vmd_frame_header_t hdr;
        if (!traj->load_raw(traj->inst, (int64_t)(f0 + b), &hdr, &infos[b], nullptr, 0) || hdr.num_atoms != num_atoms
                || infos[b].codec != infos[0].codec ||
            (infos[b].codec != VMD_RAW_CODEC_XTC && infos[b].codec != VMD_RAW_CODEC_F32)) { rs.state = -1; return 0; }
        rs.cells[b] = hdr.unitcell;
        vmd_xtc_frame_t& fi = rs.info[b];
        fi.precision = infos[b].precision;
        for (int k = 0; k < 3; ++k) { fi.minint[k] = infos[b].minint[k]; fi.maxint[k] = infos[b].maxint[k]; }
        fi.smallidx = infos[b].smallidx;
        fi.offset = total;
        fi.nbytes = infos[b].nbytes;
        total += ((size_t)infos[b].nbytes + 32 + 63) & ~(size_t)63;     // >= 32 readable bytes behind every stream, 64-byte aligned starts
    }
    rs.codec = infos[0].codec;
    if (rs.codec == VMD_RAW_CODEC_F32) {
        // plain floats (TRR, DCD): only out of the mapped file - copying them through a pinned block first is what load_frame does
        const int up = raw_upload_f32(e, rs, traj, infos, num_atoms, f0, nb);
        if (up <= 0) rs.state = -1;
        return up;
    }
    rs.info_bytes = (nb * sizeof(vmd_xtc_frame_t) + 255) & ~(size_t)255;
    // The file is mapped: the copy engine takes the batch's span of it as it lies there (frame headers in between and all), this
    // thread only writes the frame table.  r03m: reading the streams into the pinned block took 9.7 ms of a 15.8 ms c2 step (1 000
    // frames, 0.51 GB, ~53 GB/s whatever the thread count) and sat on the eval thread's critical path.
    vmd_raw_mapped_view_t mv;
    if (g_opt.xtc_mapped.load() && g_opt.xtc_device_decode.load() == 3 && traj->raw_mapped_view && traj->raw_mapped_view(traj->inst, &mv) &&
        mv.codec == VMD_RAW_CODEC_XTC && mv.base && mv.stream_offset) {
        bool usable = true;
        uint64_t prev_end = 0;
        for (size_t b = 0; b < nb && usable; ++b) {
            const uint64_t so = mv.stream_offset[f0 + b];
            usable = (so & 3u) == 0 && so >= prev_end && so + infos[b].nbytes <= mv.bytes;
            prev_end = so + infos[b].nbytes;
        }
        const size_t lo = usable ? (size_t)(mv.stream_offset[f0] & ~(uint64_t)7) : 0;
        const size_t hi = usable ? std::min<size_t>(mv.bytes, (size_t)prev_end + 40) : 0;
        if (usable && hi > lo && mapreg_pin(mv.base, mv.bytes, lo, hi)) {
            HostTimer map_timer("host_raw_map");
            for (size_t b = 0; b < nb; ++b) rs.info[b].offset = mv.stream_offset[f0 + b] - lo;
            if (rs.info_bytes > rs.hcap) {
                if (rs.h) pool_give(rs.h);
                rs.h = nullptr; rs.hcap = 0;
                if (pool_take(kPinned, (void**)&rs.h, 2 * rs.info_bytes) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", 2
                        * rs.info_bytes); return -1; }
                rs.hcap = 2 * rs.info_bytes;
            }
            memcpy(rs.h, rs.info.data(), nb * sizeof(vmd_xtc_frame_t));
            rs.h_streams = mv.base + lo;
            const size_t span = hi - lo;
            // >= 32 readable bytes behind the last stream even at the file's end
            if (!rs.d.ensure(rs.info_bytes + span + span / 8 + 64)) return -1;
            if (hipMemcpyAsync(rs.d.p, rs.h, nb * sizeof(vmd_xtc_frame_t), hipMemcpyHostToDevice,
                    e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the frame table failed"); return -1; }
            for (size_t a = lo; a < hi;) {                  // one copy per pinned window the span touches
                const size_t stop = std::min(hi, (a / kMapWindow + 1) * kMapWindow);
                if (hipMemcpyAsync(rs.d.p + rs.info_bytes + (a - lo), mv.base + a, stop - a, hipMemcpyHostToDevice,
                        e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync from the mapped trajectory file failed"); return -1; }
                a = stop;
            }
            if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
            e->frames_mapped += nb;
            rs.state = 1;
            return 1;
        }
    }
    // the frame table travels at the head of the same pinned block: a second copy from pageable memory would stall this thread
    // behind the DMA already queued on copy_stream (r03m: 11.5 ms of a 17.4 ms c2 step were spent in this function)
    total += rs.info_bytes;
    if (total > rs.hcap) {
        if (rs.h) pool_give(rs.h);
        rs.h = nullptr; rs.hcap = 0;
        const size_t cap = total + total / 8;                            // frames of one trajectory differ by a few per cent
        if (pool_take(kPinned, (void**)&rs.h, cap) != hipSuccess) { vmd_fail("hipHostMalloc(%zu bytes) failed", cap); return -1; }
        rs.hcap = cap;
    }
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), nb / 4));
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nb || !ok.load()) break;
            const vmd_xtc_frame_t& fi = rs.info[b];
            vmd_raw_frame_t info;
            unsigned char* dst = rs.h + rs.info_bytes + fi.offset;
            if (!traj->load_raw(traj->inst, (int64_t)(f0 + b), nullptr, &info, dst, (size_t)fi.nbytes)
                    || info.nbytes != fi.nbytes) { ok = false; break; }
            memset(dst + fi.nbytes, 0, (((size_t)fi.nbytes + 32 + 63) & ~(size_t)63) - (size_t)fi.nbytes);
        }
    };
    {
        HostTimer read_timer("host_raw_read");
        memcpy(rs.h, rs.info.data(), nb * sizeof(vmd_xtc_frame_t));
        if (nthreads == 1) work();
        else {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        }
    }
    if (!ok.load()) { rs.state = -1; return 0; }           // let load_frame produce the real error message
    rs.h_streams = rs.h + rs.info_bytes;
    if (!rs.d.ensure(total + total / 8)) return -1;
    if (hipMemcpyAsync(rs.d.p, rs.h, total, hipMemcpyHostToDevice,
            e->copy_stream) != hipSuccess) { vmd_fail("hipMemcpyAsync of the compressed batch failed"); return -1; }
    if (hipEventRecord(rs.uploaded, e->copy_stream) != hipSuccess) { vmd_fail("hipEventRecord failed"); return -1; }
    rs.state = 1;
    return 1;
}

// bring frames [f0, f0+nb) to the device (or alias them in place) through stage `st`: fills st.cells / st.h_boxes, queues
// the copies on copy_stream and records st.ready
bool fetch_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, bool force_host, RawSlot* pre) {
    st.raw_pending = false;
    hipStream_t ss = e->copy_stream;         // the stream this stage's `ready` is recorded on
    vmd_host_view_t hv;
    const vmd_host_view_t* hview = (!view && traj->host_view && traj->host_view(traj->inst, &hv)) ? &hv : nullptr;
    st.f0 = f0; st.nb = nb;
    if (view && view->cells_version != 0 && st.boxes_version == view->cells_version && st.boxes_cells == view->cells && st.boxes_f0 == f0 &&
        st.boxes_nb == nb) {
        // the same frames of an unchanged resident trajectory as last time (VIAMD re-evaluates after every script edit; a 10 000-frame
        // SDF step spent 0.1 ms here): cells, boxes (also the bounding-box ones of open axes) and their device copy are still valid
        st.base = view->base + f0 * view->frame_stride;
        st.frame_stride = view->frame_stride;
        st.row_stride = view->row_stride;
        HIP_OK(hipEventRecord(st.ready, e->copy_stream));
        return true;
    }
    st.boxes_version = 0;
    st.gboxes_ready = false;
    st.cells.resize(nb);
    st.h_boxes.resize(nb * 9);
    if (view) {
        st.base = view->base + f0 * view->frame_stride;
        st.frame_stride = view->frame_stride;
        st.row_stride = view->row_stride;
        for (size_t b = 0; b < nb; ++b) st.cells[b] = view->cells[f0 + b];
    } else if (hview) {
        // frames already sit in host memory in our layout: DMA them as one block, no load_frame copies
        const size_t need = nb * hview->frame_stride;
        if (!st.d.ensure(need)) return false;
        HIP_OK(hipMemcpyAsync(st.d.p, hview->base + f0 * hview->frame_stride, need * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
        st.base = st.d.p;
        st.frame_stride = hview->frame_stride;
        st.row_stride = hview->row_stride;
        for (size_t b = 0; b < nb; ++b) st.cells[b] = hview->cells[f0 + b];
    } else {
        const size_t npad = (num_atoms + 63) & ~(size_t)63;
        const size_t need = nb * 3 * npad;
        int raw = 0;
        vmd_raw_device_view_t rv;
        memset(&rv, 0, sizeof(rv));
        if (!force_host && traj->raw_device_view && traj->raw_device_view(traj->inst, &rv) && rv.codec == VMD_RAW_CODEC_XTC
                && rv.device == e->device) {
            // the compressed trajectory is resident in HBM: no host work, no PCIe - decode the batch where it lies
            for (size_t b = 0; b < nb; ++b) st.cells[b] = rv.cells[f0 + b];
            raw = launch_raw_decode(e, st, rv.base, (const vmd_xtc_frame_t*)rv.info + f0, num_atoms, nb, npad, ss, rv.ck
                    ? (vmd_xtc_ck_t*)rv.ck + f0 * VMD_XTC_CK_MAX : nullptr, rv.nck ? rv.nck + f0 : nullptr, rv.ck_have ? rv.ck_have
                    + f0 : nullptr, (rv.rec && rv.rec_stride) ? rv.rec + f0 * rv.rec_stride : nullptr, (rv.rec && rv.rec_stride) ? rv.nrec
                    + f0 : nullptr, rv.rec_stride, rv.rec_failed);
            if (raw < 0) return false;
        } else if (!force_host && g_opt.xtc_device_decode.load() && traj->load_raw && !(e->raw_skip && !pre)) {
            // the bit streams were (or are now) sent ahead through a slot of the ring; decompression runs on its own stream
            RawSlot* rs = (pre && pre->f0 == f0 && pre->nb == nb && pre->state != 0) ? pre : &e->raw_slots[0];
            if (rs != pre && (raw = raw_upload(e, *rs, traj, num_atoms, f0, nb)) < 0) return false;
            if (rs->state == 1) {
                ss = e->decode_streams[pre ? (size_t)(pre - e->raw_slots) % vmd_script_eval_t::kDecodeStreams : 0];
                st.cells = rs->cells;
                HIP_OK(hipStreamWaitEvent(ss, rs->uploaded, 0));
                if (rs->codec == VMD_RAW_CODEC_F32) {
                    if (!st.d.ensure(nb * 3 * npad)) return false;
                    e->prof_copy.begin("raw_f32", ss);
                    KRN_OK(vmd_hip_raw_f32_decode(ss, rs->d_streams(), (const vmd_f32_frame_t*)rs->d.p, (int)nb, (int)num_atoms, st.d.p, 3
                            * npad, npad));
                    e->prof_copy.end(ss);
                    e->frames_device_decoded += nb;
                    raw = 1;
                } else {
                std::shared_ptr<CkCache> cc = ckcache_for(traj->inst, traj->num_frames(traj->inst), num_atoms, e->device);
                e->ck_cache = cc;
                st.ck_hold = cc;
                if (cc) {
                    // a frame's checkpoints count only for the very bytes they were written for
                    for (size_t b = 0; b < nb; ++b) {
                        const uint64_t sg = frame_signature(rs->info[b], rs->h_streams + rs->info[b].offset);
                        if (flag_get(&cc->sig[f0 + b]) != sg) { flag_set(&cc->sig[f0 + b], sg); flag_set(&cc->have[f0 + b], (uint8_t)0); }
                    }
                    raw = launch_raw_decode(e, st, rs->d_streams(), rs->d_info(), num_atoms, nb, npad, ss, cc->ck.p + f0 * VMD_XTC_CK_MAX,
                            cc->nck.p + f0, cc->have.data() + f0, cc->rec_stride ? cc->rec.p + f0 * cc->rec_stride : nullptr,
                            cc->rec_stride ? cc->nrec.p + f0 : nullptr, cc->rec_stride, &cc->rec_failed);
                } else {
                    raw = launch_raw_decode(e, st, rs->d_streams(), rs->d_info(), num_atoms, nb, npad, ss);
                }
                if (raw < 0) return false;
                }
            } else {
                raw = 0;
            }
        }
        if (raw == 1) {
            st.base = st.d.p;
            st.frame_stride = 3 * npad;
            st.row_stride = npad;
        } else {
            if (need > st.hcap) {
                if (st.h) pool_give(st.h);
                st.h = nullptr; st.hcap = 0;
                HIP_OK(pool_take(kPinned, (void**)&st.h, need * sizeof(float)));
                st.hcap = need;
            }
            if (!st.d.ensure(need)) return false;
            // md_trajectory_load_frame is called from all of VIAMD's pool threads at once (src/main.cpp:995-996 inside the
            // enkiTS range tasks), so the decoder behind it is re-entrant: decode the batch on a few threads
            const size_t nthreads = std::max<size_t>(1, std::min<size_t>(load_threads(), nb / 4));
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            std::mutex err_mtx;
            std::string err;
            auto work = [&]() {
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb || !ok.load()) break;
                    vmd_frame_header_t hdr;
                    memset(&hdr, 0, sizeof(hdr));
                    float* x = st.h + b * 3 * npad;
                    if (!traj->load_frame(traj->inst, (int64_t)(f0 + b), &hdr, x, x + npad, x + 2 * npad)) {
                        std::lock_guard<std::mutex> l(err_mtx);
                        if (ok.exchange(false)) {
                            char buf[96];
                            snprintf(buf, sizeof(buf), "trajectory load_frame(%zu) failed", f0 + b);
                            err = buf;
                            if (!g_last_error.empty()) err += ": " + g_last_error;     // the decoder's own message (this thread's)
                        }
                        break;
                    }
                    st.cells[b] = hdr.unitcell;
                }
            };
            if (nthreads == 1) work();
            else {
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
                work();
                for (auto& t : pool) t.join();
            }
            if (!ok.load()) return vmd_fail("%s", err.c_str());
            HIP_OK(hipMemcpyAsync(st.d.p, st.h, need * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
            st.base = st.d.p;
            st.frame_stride = 3 * npad;
            st.row_stride = npad;
        }
    }
    for (size_t b = 0; b < nb; ++b) {
        const vmd_unitcell_t& c = st.cells[b];
        const bool tri = c.xy != 0.0f || c.xz != 0.0f || c.yz != 0.0f;
        if (tri && ((c.flags & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL || !(c.x > 0.0f && c.y > 0.0f && c.z > 0.0f)))
            return vmd_fail("frame %zu: a triclinic unit cell must be periodic along all three axes (SPEC S3t)", f0 + b);
        const vmd_unitcell_t& c0 = st.cells[0];
        if (c.flags != c0.flags || tri != (c0.xy != 0.0f || c0.xz != 0.0f || c0.yz != 0.0f))
            return vmd_fail("frame %zu: periodicity / cell type changes inside the trajectory", f0 + b);
        float* hb = &st.h_boxes[9 * b];
        hb[0] = c.x; hb[1] = c.y; hb[2] = c.z;
        hb[3] = 1.0f / c.x; hb[4] = 1.0f / c.y; hb[5] = 1.0f / c.z;      // SPEC S2: invL = fl(1.0f / L)
        hb[6] = c.xy; hb[7] = c.xz; hb[8] = c.yz;
    }
    if (!st.d_boxes.upload(st.h_boxes.data(), nb * 9, ss)) return false;
    HIP_OK(hipEventRecord(st.ready, ss));
    if (view && view->cells_version != 0) { st.boxes_cells = view->cells; st.boxes_f0 = f0; st.boxes_nb = nb;
            st.boxes_version = view->cells_version; }
    return true;
}

// A stage whose frames were decompressed on the device: wait for its `ready` event (the decode ran under the previous batch's
// kernels, so this rarely waits) and look at the status words.  A stream the device rejects - damaged, or a packed number above
// 2^64 - sends the whole batch through the host reader, which decides and reports.
bool settle_stage(vmd_script_eval_t* e, Stage& st, vmd_trajectory_i* traj, size_t num_atoms) {
    if (!st.raw_pending) return true;
    HIP_OK(hipEventSynchronize(st.ready));
    e->prof_copy.resolve();
    st.raw_pending = false;
    std::shared_ptr<CkCache> hold = std::move(st.ck_hold);      // released when this function is done with ck_mark / ck_clear / rec_failed
    bool good = true;
    for (size_t b = 0; b < st.nb; ++b) if (st.h_raw_status[b] != 0) good = false;
    if (good) {
        if (st.ck_mark) for (size_t b = 0; b < st.nb; ++b) flag_set(&st.ck_mark[b], (uint8_t)1);
        st.ck_mark = nullptr;
        st.ck_clear = nullptr;
        st.rec_failed = nullptr;
        e->frames_device_decoded += st.nb;
        if (st.sectioned) e->frames_section_decoded += st.nb;
        return true;
    }
    st.ck_mark = nullptr;
    // checkpoints that did not describe these streams (a sidecar table that passed the signature test and still lies): the frames
    // walk from bit 0 again next time
    if (st.ck_clear) for (size_t b = 0; b < st.nb; ++b) flag_set(&st.ck_clear[b], (uint8_t)0);
    st.ck_clear = nullptr;
    // the records did not describe these streams: never again for this trajectory
    if (st.rec_failed) flag_set(st.rec_failed, true);
    st.rec_failed = nullptr;
    return fetch_stage(e, st, traj, nullptr, num_atoms, st.f0, st.nb, true);
}

// synchronous variant used for single frames (reference pose, vis payload)
bool fetch_batch(vmd_script_eval_t* e, vmd_trajectory_i* traj, const vmd_device_view_t* view, size_t num_atoms,
                        size_t f0, size_t nb, BatchSrc* src) {
    Stage& st = e->stages[0];
    if (!fetch_stage(e, st, traj, view, num_atoms, f0, nb)) return false;
    if (!settle_stage(e, st, traj, num_atoms)) return false;
    HIP_OK(hipEventSynchronize(st.ready));
    src->base = st.base; src->frame_stride = st.frame_stride; src->row_stride = st.row_stride;
    return true;
}

uint32_t batch_pbc(const Stage& st) {
    const vmd_unitcell_t& c = st.cells[0];
    uint32_t f = c.flags & VMD_UNITCELL_PBC_ALL;
    if (!(c.x > 0.0f)) f &= ~VMD_UNITCELL_PBC_X;
    if (!(c.y > 0.0f)) f &= ~VMD_UNITCELL_PBC_Y;
    if (!(c.z > 0.0f)) f &= ~VMD_UNITCELL_PBC_Z;
    if (c.xy != 0.0f || c.xz != 0.0f || c.yz != 0.0f) f |= 8u;         // triclinic (kernels: VMD_PBC_TRICLINIC)
    return f;
}

// Batches with open axes: the grid spans the bounding box of the batch's atoms.  Fills st.h_gboxes / st.d_gboxes with
// {extent or L, inverse, origin or 0} per frame (one bbox kernel + one small readback per batch).
bool prepare_open_boxes(vmd_script_eval_t* e, Stage& st, size_t nb, uint32_t pbc, size_t num_atoms) {
    if (st.gboxes_ready) return true;
    if (!st.d_bbox.ensure(nb * 6) || !st.d_gboxes.ensure(nb * 9)) return false;
    st.h_bbox.resize(nb * 6);
    KRN_OK(vmd_hip_bbox(e->stream, st.base, st.frame_stride, st.row_stride, (int)nb, (int)num_atoms, st.d_bbox.p));
    HIP_OK(hipMemcpyAsync(st.h_bbox.data(), st.d_bbox.p, nb * 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    st.h_gboxes = st.h_boxes;
    for (size_t b = 0; b < nb; ++b) {
        float* g = &st.h_gboxes[9 * b];
        for (int a = 0; a < 3; ++a) {
            g[6 + a] = 0.0f;
            if (pbc & (1u << a)) continue;
            const float lo = st.h_bbox[6 * b + a], hi = st.h_bbox[6 * b + 3 + a];
            const float pad = std::max(1.0e-2f, 1.0e-3f * (hi - lo));
            g[6 + a] = lo - pad;                    // origin
            g[a] = (hi - lo) + 2.0f * pad;          // extent
            g[3 + a] = 1.0f / g[a];
        }
    }
    if (!st.d_gboxes.upload(st.h_gboxes.data(), nb * 9, e->stream)) return false;
    st.gboxes_ready = true;
    return true;
}

// pencil grid for a batch and cutoff; false when the batch cannot use the grid kernel.  `boxes` = st.h_boxes, or
// st.h_gboxes when some axes are open (pbc bits clear): those carry the bounding-box extent instead of a cell edge.
bool choose_grid(const std::vector<float>& boxes, uint32_t pbc, size_t nb, float rmax, vmd_grid_t* g, bool dense_lanes) {
    if (g_opt.force_brute) return false;
    const bool tri = (pbc & 8u) != 0;
    if (tri && (pbc & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL) return false;
    // smallest extent per axis over the batch, measured perpendicular to the cell faces (SPEC S3t: a triclinic cell's
    // pencils are sheared, what has to be >= rmax is their width w_k = 1 / |reciprocal vector k|)
    float wmin[3] = {3.4e38f, 3.4e38f, 3.4e38f}, Lxmin = 3.4e38f;
    for (size_t b = 0; b < nb; ++b) {
        const float* q = &boxes[9 * b];
        const double Lx = q[0], Ly = q[1], Lz = q[2];
        const double xy = tri ? q[6] : 0.0, xz = tri ? q[7] : 0.0, yz = tri ? q[8] : 0.0;
        const double wx = Lx / std::sqrt(1.0 + (xy / Ly) * (xy / Ly) + ((xy * yz - Ly * xz) / (Ly * Lz)) * ((xy * yz - Ly * xz) / (Ly
                * Lz)));
        const double wy = Ly / std::sqrt(1.0 + (yz / Lz) * (yz / Lz));
        wmin[0] = std::min(wmin[0], (float)wx); wmin[1] = std::min(wmin[1], (float)wy); wmin[2] = std::min(wmin[2], (float)Lz);
        Lxmin = std::min(Lxmin, q[0]);
    }
    // periodic axes: the minimum image must be unique for every hit (rmax < w/2 with margin); open axes: no restriction
    for (int a = 0; a < 3; ++a) if ((pbc & (1u << a)) && !(rmax * 2.0f * 1.001f < wmin[a])) return false;
    int n[3];
    const int sy = g_opt.pencil_split_y.load();
    const int split[3] = {1, sy <= 0 ? (dense_lanes ? 2 : 1) : std::min(4, sy), std::max(1, std::min(4, g_opt.pencil_split_z.load()))};
    vmd_hip_set_pencil_reach(split[1], split[2]);
    for (int a = 1; a < 3; ++a) {
        const float redge = rmax / (float)split[a];
        int k = (int)std::floor(wmin[a] / redge);
        // head room between the pencil edge and rmax: wrapped coordinates are exact to ~1e-6 of the edge; on an open axis
        // coordinates keep their raw magnitude (possibly far from the origin), so leave ten times more
        const float edge_margin = (pbc & (1u << a)) ? 0.9999f : 0.999f;
        while (k > 1 && ((float)k / wmin[a]) * redge > edge_margin) k -= 1;
        if (pbc & (1u << a)) { if (k < 2) return false; }
        else k = std::max(k, 1);
        n[a] = std::min(k, 1024);
    }
    const float cx = rmax / (float)std::max(1, g_opt.nxf_divisor.load());
    int nxf = (int)std::floor(Lxmin / cx);
    nxf = std::max(1, std::min(nxf, 4096));
    // keep the cell table small enough for the LDS-resident build (24576 counters) as long as the fine cells stay <= rmax/3
    // (the single-level builds only: the two-level build keeps a table of pencils, not of cells)
    const int nxf_lds = 24575 / (n[1] * n[2]);
    vmd_grid_t probe{nxf, n[1], n[2], 0};
    if (!vmd_hip_cells_pencil_ok(probe) && nxf > nxf_lds && nxf_lds >= (int)std::ceil(3.0f * Lxmin / rmax)) nxf = nxf_lds;
    g->nxf = nxf; g->ny = n[1]; g->nz = n[2];
    const long long ncell = (long long)nxf * n[1] * n[2];
    if (ncell > (1ll << 26)) return false;
    g->ncell = (int32_t)ncell;
    return true;
}
