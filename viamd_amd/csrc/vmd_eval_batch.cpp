// viamd_amd/csrc/vmd_eval_batch.cpp - one call's worth of evaluation: the pencil grid of a batch, the two-level cell build per
// selection (md_spatial_hash's role), batch planning, reuse of another eval's block partials (filtered evaluation,
// /root/reference/src/main.cpp:1014-1039) and process_range - the loop that queues the kernels of batch k + 1 before it waits for
// batch k, repeats a batch whose buckets overflowed and books frames, temporal rows and the frame mask.
#include "vmd_eval_internal.h"

// Bucket capacities of the two-level build for selection `s` on the pencils of grid `g`: per-pencil maximum over the first and
// last (up to) 4 frames of the batch x margin + a few standard deviations.  One small readback, then kept for the eval's
// lifetime (frames of one trajectory look alike; a bucket that overflows later is caught by the device flag and re-measured).
bool ensure_pencil_caps(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb,
        const vmd_grid_t& g) {
    if (!s->pen_off.empty() && s->pen_ny == g.ny && s->pen_nz == g.nz) return true;
    if (!s->pen_off.empty()) {                 // keep what was measured for the layout we are leaving
        bool known = false;
        for (auto& c : s->caps_cache) known = known || (c.ny == s->pen_ny && c.nz == s->pen_nz);
        if (!known) {
            if (s->caps_cache.size() >= 4) s->caps_cache.erase(s->caps_cache.begin());
            s->caps_cache.push_back({s->pen_ny, s->pen_nz, s->cap_max, s->total_cap, s->pen_off});
        }
    }
    for (auto& c : s->caps_cache) {
        if (c.ny != g.ny || c.nz != g.nz) continue;
        s->pen_off = c.pen_off; s->cap_max = c.cap_max; s->total_cap = c.total_cap; s->pen_ny = c.ny; s->pen_nz = c.nz;
        // pageable source: the copy is staged before the call returns
        return s->d_pen_off.upload(s->pen_off.data(), s->pen_off.size(), e->stream);
    }
    const int npen = g.ny * g.nz, nsel = (int)s->idx.size();
    // after an overflow: every frame of the batch (exact populations), otherwise 4 frames each from its beginning, middle and end (round 6:
    // the middle was not looked at, and a solute that wanders through the batch - config 5 - overflowed the solvent's buckets there)
    // (cells_cap_sample = 2: beginning and end only, as before - the overflow tests ask for it)
    const bool exhaustive = s->overflows > 0 || nb <= 8;
    const size_t nsamp = exhaustive ? 1 : (g_opt.cells_cap_sample.load() >= 3 ? 3 : 2);
    const size_t S = exhaustive ? nb : 4, rows = exhaustive ? nb : 4 * nsamp;
    if (!e->d_pen_sample.ensure(rows * (size_t)npen)) return false;
    std::vector<uint32_t> h(rows * (size_t)npen);
    const size_t starts[3] = {0, nsamp == 3 ? (nb - S) / 2 : nb - S, nb - S};
    for (size_t k = 0; k < nsamp; ++k)
        KRN_OK(vmd_hip_cells_pencil_count(e->stream, src.base + starts[k] * src.frame_stride, src.frame_stride, src.row_stride,
                d_boxes + 9 * starts[k], pbc, (int)S, s->d_idx.p, nsel, g, e->d_pen_sample.p + k * S * (size_t)npen));
    HIP_OK(hipMemcpyAsync(h.data(), e->d_pen_sample.p, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    s->pen_off.assign((size_t)npen + 1, 0);
    s->cap_max = 0;
    uint64_t total = 0;
    const double mean_pop = g_opt.cells_cap_floor.load() ? (double)nsel / (double)npen : 0.0;
    for (int p = 0; p < npen; ++p) {
        uint32_t m = 0;
        for (size_t k = 0; k < rows; ++k) m = std::max(m, h[k * npen + p]);
        uint32_t cap = (uint32_t)std::ceil((double)m * s->cap_margin + 6.0 * std::sqrt((double)m)) + 16;
        // ... but never below what a pencil holds at the selection's MEAN density (round 6).  A solute that sits in a pencil for the whole
        // of the batch the capacities were measured on (config 5's wandering blob) leaves that pencil's bucket too small for the batches in
        // which it has moved on: two overflows, i.e. two repeated batches, per evaluation of c5.  Costs at most 1.15 x nsel records.
        cap = std::max(cap, (uint32_t)std::ceil(mean_pop * 1.15 + 6.0 * std::sqrt(mean_pop)) + 16);
        cap = (cap + 3u) & ~3u;
        s->pen_off[p] = (uint32_t)total;
        total += cap;
        s->cap_max = std::max<int>(s->cap_max, (int)cap);
    }
    if (total > 0x7fffffffull) { s->pen_off.clear(); s->overflows = 99; return true; }     // not a job for the buckets
    s->pen_off[npen] = (uint32_t)total;
    s->total_cap = (int)total;
    s->pen_ny = g.ny; s->pen_nz = g.nz;
    return s->d_pen_off.upload(s->pen_off.data(), s->pen_off.size(), e->stream);
}

// the kernels of this selection's builds compute a periodic index list instead of reading it: set for the calls below, cleared on every way
// out
struct SelPatternScope {
    explicit SelPatternScope(const Selection* s) {
        if (s->pat_m) vmd_hip_set_cells_sel_pattern(s->pat_m, s->pat_first, s->pat_period, s->pat_off);
    }
    ~SelPatternScope() { vmd_hip_set_cells_sel_pattern(0, 0, 0, nullptr); }
};

bool build_selection(vmd_script_eval_t* e, Selection* s, const Stage& src, const float* d_boxes, uint32_t pbc, size_t nb,
        const vmd_grid_t& g) {
    if (s->built && s->built_grid.nxf == g.nxf && s->built_grid.ny == g.ny && s->built_grid.nz == g.nz) return true;
    const SelPatternScope pattern(s);
    const int nsel = (int)s->idx.size();
    s->nsel_pad = (nsel + 63) & ~63;
    // +64: the pair kernel prefetches past a segment
    if (!s->cell_start.ensure(nb * (size_t)(g.ncell + 1)) || !s->sorted.ensure(nb * 3 * (size_t)s->nsel_pad + 64)) return false;
    s->used_pencil = false;
    // two-level build through per-pencil buckets (one read of the frame, coalesced sorted rows); single-level builds otherwise A small
    // selection (a solute: the 2 000-atom blob of config 5) is not spread evenly over the pencils and wanders through them as the
    // trajectory goes on: capacities measured on one batch overflow in the next, and every overflow repeats the batch's pair passes.  It is
    // sorted by ONE block per frame in LDS instead (k_cells_fused: no buckets, nothing to overflow), which costs such a selection nothing.
    // (round 6: whether or not the grid's cell table fits the fused kernel's LDS - c5's 30 000 cells do not, and its 1 200-atom solute
    // class went through the buckets after all: one overflow, i.e. one repeated batch, in each of an evaluation's first two passes.  The
    // single-level builds behind vmd_hip_cells_build pick the fused, the split or the atomic three-kernel variant themselves.)
    const bool small = nsel <= g_opt.cells_small.load();
    if (vmd_hip_cells_pencil_ok(g) && s->overflows < 3 && !small) {
        if (!ensure_pencil_caps(e, s, src, d_boxes, pbc, nb, g)) return false;
        if (!s->pen_off.empty() && s->cap_max <= vmd_hip_cells_pencil_cap_max()) {
            const size_t npen = (size_t)g.ny * g.nz;
            if (!s->pen_count.ensure(nb * npen) || !s->pen_start.ensure(nb * (npen + 1)) || !s->bucket.ensure(nb * (size_t)s->total_cap
                    * 4)) return false;
            e->prof.begin("cells_build", e->stream);
            vmd_hip_set_cells_overflow_bit(s->overflow_bit);
            KRN_OK(vmd_hip_cells_build_pencil(e->stream, src.base, src.frame_stride, src.row_stride, d_boxes, pbc, (int)nb, s->d_idx.p,
                    nsel, s->nsel_pad, g, s->d_pen_off.p, s->total_cap, s->cap_max, s->pen_count.p, s->pen_start.p, s->bucket.p,
                    e->d_overflow.p, s->cell_start.p, s->sorted.p));
            e->prof.end(e->stream);
            s->built = true; s->built_grid = g; s->used_pencil = true;
            return true;
        }
    }
    if (!s->cell_count.ensure(nb * (size_t)(g.ncell + 1)) || !s->rank.ensure(nb * vmd_hip_cells_scratch_words(g, nsel))) return false;
    const bool use_aos = g_opt.cells_aos != 0;
    if (use_aos && !s->aos.ensure(nb * 4 * (size_t)s->nsel_pad)) return false;
    e->prof.begin("cells_build", e->stream);
    KRN_OK(vmd_hip_cells_build(e->stream, src.base, src.frame_stride, src.row_stride, d_boxes, pbc, (int)nb, s->d_idx.p, nsel,
                               s->nsel_pad, g, s->cell_count.p, s->rank.p, s->cell_start.p, s->sorted.p, use_aos ? s->aos.p : nullptr));
    e->prof.end(e->stream);
    s->built = true;
    s->built_grid = g;
    return true;
}

size_t auto_batch(const vmd_script_eval_t* e, size_t num_atoms, bool staged) {
    const int forced = g_opt.batch_frames;
    if (forced > 0) return (size_t)forced;
    // scratch per frame: a selection that takes part in a pair pass holds ~40 B per atom (sorted rows, bucket records, tables);
    // host trajectories add the staged frame itself; SDF / distance properties need a few hundred bytes
    size_t per_frame = staged ? 12 * num_atoms : 0;
    std::vector<char> used(e->sels.size(), 0);
    for (auto& g : e->rdf_groups) for (auto& ps : g.passes) { used[ps.sel_a] = 1; used[ps.sel_b] = 1; }
    for (size_t i = 0; i < e->sels.size(); ++i) if (used[i]) per_frame += 40 * e->sels[i]->idx.size();
    for (auto& p : e->props) per_frame += p->prop.kind == PROP_SDF ? 64 * p->prop.K : (p->prop.kind == PROP_DIST ? 4 * p->dim1 : 0);
    // 288 GB of HBM: a 16 GB scratch budget holds the 1 000 frames of the 1M-atom RDF (333k selected atoms) in ONE batch
    // (every batch boundary costs ~1 ms of host round trips against ~37 ms of kernels per 500 frames)
    size_t B = (size_t)(16ull << 30) / std::max<size_t>(per_frame, 1);
    // pair passes are long (a 1 024-frame batch of the 1M-atom RDF runs ~90 ms: interrupts are polled between batches); scripts
    // without them stream whole frames at HBM speed and take much larger batches, so that launches, the alignment kernel's
    // latency and the per-batch synchronisation stay small against the stream (grid.y = frames of the batch <= 65535)
    const size_t cap = e->rdf_groups.empty() ? 16384 : 1024;
    B = std::max<size_t>(1, std::min<size_t>(B, cap));
    return B;
}

void plan_batches(const vmd_script_eval_t* e, size_t beg, size_t end, size_t Bmax, std::vector<Batch>* out) {
    auto even = [&](size_t a, size_t b) {
        const size_t total = b - a;
        if (!total) return;
        const size_t nbatch = (total + Bmax - 1) / Bmax;
        const size_t B = (total + nbatch - 1) / nbatch;
        for (size_t f = a; f < b; f += B) out->push_back({f, std::min(B, b - f), -1, 0});
    };
    const size_t S = e->block_frames;
    if (S == 0) { even(beg, end); return; }
    const bool super = g_opt.block_superbatch.load() != 0;
    // whole blocks that fit one batch become a batch of blocks (their partials are kept), everything else is a plain piece
    size_t run = beg;                          // start of the pending plain piece
    for (size_t f = beg; f < end;) {
        const size_t blk = f / S;
        const size_t bend = std::min((blk + 1) * S, e->num_frames);
        if (f == blk * S && bend <= end && bend - f <= Bmax) {
            even(run, f);
            Batch* last = out->empty() ? nullptr : &out->back();
            if (super && last && last->blk >= 0 && last->f0 + last->nb == f && last->nb + (bend - f) <= Bmax) { last->nb += bend - f;
                    last->nblk += 1; }
            else out->push_back({f, bend - f, (long)blk, 1});
            f = bend; run = f;
        } else {
            f = std::min(bend, end);
        }
    }
    even(run, end);
}

const float* block_rows(const vmd_script_eval_t* src, const PropState* q, size_t blk) {
    return src->block_ready[blk].load() == BLOCK_ROWS_AHEAD && q->ahead_values.size() == q->values.size()
            ? q->ahead_values.data() : q->values.data();
}

bool reuse_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t beg, size_t end, std::vector<std::pair<size_t, size_t>>* todo) {
    vmd_script_eval_t* src = e->source;
    if (!src) { todo->push_back({beg, end}); return true; }
    std::lock_guard<std::mutex> lock(src->mtx);   // order: own mutex, then the source's (a source never locks its users)
    // (looked up under its mutex: read-ahead may be giving it blocks right now)
    if (src->block_frames == 0 || src->blocks_inst != traj_inst) { todo->push_back({beg, end}); return true; }
    const size_t S = src->block_frames;
    size_t run = beg, reused = 0;
    for (size_t f = beg; f < end;) {
        const size_t blk = f / S;
        const size_t bend = std::min((blk + 1) * S, e->num_frames);
        if (f == blk * S && bend <= end && blk < src->num_blocks && src->block_ready[blk]) {
            if (run < f) todo->push_back({run, f});
            for (size_t i = 0; i < e->props.size(); ++i) {
                PropState* p = e->props[i].get();
                const PropState* q = src->props[i].get();
                if (p->ncounts) {
                    KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, q->d_blocks.p + blk * p->ncounts, p->ncounts));
                    if (p->prop.kind == PROP_RDF)
                        for (size_t k = 0; k < p->ncounts; ++k) p->weights64[k] += q->block_weights64[blk * p->ncounts + k];
                } else {
                    memcpy(&p->values[f * p->dim1], block_rows(src, q, blk) + f * p->dim1, (bend - f) * p->dim1 * sizeof(float));
                }
                p->dirty = true;
            }
            for (size_t g = f; g < bend; ++g) mask_set(e->frame_mask, g);
            reused += bend - f;
            f = bend; run = f;
        } else {
            f = std::min(bend, end);
        }
    }
    if (run < end) todo->push_back({run, end});
    if (reused) {
        HIP_OK(hipStreamSynchronize(e->stream));   // the source's partials are read before its mutex is released
        e->frames_done += reused;
        e->frames_reused += reused;
        { HostTimer host_timer("host_refresh");
          for (auto& p : e->props) if (p->prop.kind == PROP_RDF) { if (!refresh_distribution(e, p.get())) return false; } }
    }
    return true;
}

// a sharded device trajectory keeps only its block of frames behind the view; other frames (frame 0 for the SDF reference
// pose) come through load_frame
// (resident_beg, resident_end) = (0, 0) means "every frame"; any other pair is a shard, and beg == end != 0 is an EMPTY shard (a rank
// that owns no frame: 4 ranks on 5 frames) - nothing is resident then, not everything
bool view_sharded(const vmd_device_view_t& view) { return view.resident_beg != 0 || view.resident_end != 0; }

bool view_holds(bool have_view, const vmd_device_view_t& view, size_t frame) {
    return have_view && (!view_sharded(view) || (frame >= view.resident_beg && frame < view.resident_end));
}

// evaluates frames [frame_beg, frame_end) in large batches; returns false on interrupt (empty error) or failure
// views: bring the host views (values / weights / volume / aggregates) up to date before returning; false = the caller does it later
// (refresh_views), the device accumulators and the frame mask are complete either way
// spec (read-ahead): [frame_beg, frame_end) is a run of whole frame blocks; every block is evaluated into its own partial and
// NOTHING else changes - no add into the totals, no frame mask, no frames_done, no normalisation weights outside the block's own, no view
// (temporal rows are written: a frame's row is the same whenever it is computed, and nobody reads it before its mask bit is set)
bool process_range_locked(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end,
        bool views, bool spec) {
    HIP_OK(hipSetDevice(eval->device));
    vmd_script_eval_t* e = eval;
    const size_t num_atoms = traj->num_atoms(traj->inst);
    if (traj->num_frames(traj->inst) < frame_end) return vmd_fail("trajectory has fewer frames than the requested range");
    if (!check_atoms(e, num_atoms)) return false;
    if (!upload_static(e, sys, num_atoms)) return false;

    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    if (have_view && view_sharded(view) && frame_beg < frame_end && (frame_beg < view.resident_beg || frame_end > view.resident_end))
        return vmd_fail("frames [%u, %u) are not resident on this rank (its shard holds [%zu, %zu))", frame_beg, frame_end,
                view.resident_beg, view.resident_end);

    // SDF reference pose: structure 0 at trajectory frame 0 (SPEC S5)
    for (auto& p : e->props) {
        if (p->prop.kind != PROP_SDF || p->ref_pose_ready) continue;
        BatchSrc src;
        if (!fetch_batch(e, traj, view_holds(have_view, view, 0) ? &view : nullptr, num_atoms, 0, 1, &src)) return false;
        KRN_OK(vmd_hip_sdf_ref_pose(e->stream, src.base, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), p->d_structs.p,
                p->d_mass.p, (int)p->prop.m, p->d_ref_pose.p, p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree
                ? p->d_tree_parent.p : nullptr));
        HIP_OK(hipStreamSynchronize(e->stream));
        p->ref_pose_ready = true;
    }

    // frames served from the block partials of the source eval (filtered evaluation), the rest is computed
    std::vector<std::pair<size_t, size_t>> segments;
    // a region's blocks are adopted from the source by the region leader, or evaluated here
    if (spec) segments.push_back({frame_beg, frame_end});
    else if (!reuse_blocks(e, traj_id(traj), frame_beg, frame_end, &segments)) return false;
    if (e->block_frames) {
        // a source evaluated over trajectory A and then, WITHOUT clear_data, over another trajectory of the same length must
        // not hand A's block partials to users that evaluate B - the blocks kept from now on belong to B, the ones kept so far are
        // forgotten
        const TrajId now = traj_id(traj);
        if (e->blocks_inst.inst && e->blocks_inst != now)
            for (size_t b = 0; b < e->num_blocks; ++b) e->block_ready[b] = 0;
        e->blocks_inst = now;
    }

    // compressed frames for the device decoder travel two batches ahead through a ring of three slots (RawSlot)
    vmd_host_view_t hv_probe;
    vmd_raw_device_view_t rv_probe;
    bool raw_ring = !have_view && traj->load_raw && g_opt.xtc_device_decode.load() != 0 &&
                    !(traj->host_view && traj->host_view(traj->inst, &hv_probe)) &&
                    !(traj->raw_device_view && traj->raw_device_view(traj->inst, &rv_probe));
    vmd_raw_mapped_view_t mv_probe;
    memset(&mv_probe, 0, sizeof(mv_probe));
    const bool have_map = raw_ring && g_opt.xtc_mapped.load() && traj->raw_mapped_view && traj->raw_mapped_view(traj->inst, &mv_probe);
    // plain-float files (TRR, DCD) take the ring only out of a mapping: without one their frames go through load_frame as before
    bool f32_ring = false;
    if (raw_ring && frame_beg < frame_end) {
        vmd_raw_frame_t probe;
        memset(&probe, 0, sizeof(probe));
        if (!traj->load_raw(traj->inst, (int64_t)frame_beg, nullptr, &probe, nullptr, 0)) raw_ring = false;
        else if (probe.codec == VMD_RAW_CODEC_F32) { f32_ring = have_map && mv_probe.codec == VMD_RAW_CODEC_F32
                && g_opt.raw_f32_device.load(); raw_ring = f32_ring; }
    }
    e->raw_skip = traj->load_raw && !raw_ring;              // fetch_stage: do not ask this trajectory for raw frames batch by batch
    // how many batches the bit streams run ahead of the kernels (one more than the decoder, which runs two ahead).  Copied through pinned
    // blocks (host threads read them, this thread waits): 3.  Taken out of the mapped file by the copy engine alone: as many as the ring
    // holds minus the one being decoded - the DMAs then queue back to back and PCIe never waits for this thread (r03n: 12.4 ms per c2 step
    // against 9.4 ms of transfers). always > stage_ahead
    const size_t raw_ahead = (raw_ring && have_map && (f32_ring || g_opt.xtc_device_decode.load() == 3)) ? vmd_script_eval_t::kRawSlots
            - 1 : 3;
    auto slot_of = [&](size_t bi) -> RawSlot* { return raw_ring ? &e->raw_slots[bi % vmd_script_eval_t::kRawSlots] : nullptr; };
    // batches decompressed on the device while the previous batch is in the pair kernel: the persistent pair grid leaves room for them
    const bool device_decode = raw_ring || (!have_view && traj->raw_device_view && traj->raw_device_view(traj->inst, &rv_probe));

    bool cold_walk = false;
    // frames per launch: as many as the scratch budget allows, split evenly so that no small tail batch is left
    size_t Bmax = auto_batch(e, num_atoms, !have_view);
    const vmd_device_view_t* vw = have_view ? &view : nullptr;
    // host trajectories are staged in smaller batches so that load_frame of batch k+1 overlaps the kernels of batch k
    // ... except when the batches are decompressed on the device (profiles/r03_xtc_device_decode.txt).  The FIRST decode of a frame
    // walks its whole bit stream - a latency-bound chain, 7 ms per batch whether it holds 64 or 1 000 synthetic frames - so first passes
    // use large batches (4 x stage_frames).  It leaves checkpoints; every later pass decodes in sections at 2 - 3 us per frame, and
    // then small batches win from a file (the PCIe trip of batch k + 1 hides under decode + pair kernel of batch k: 64.6k frames/s
    // with batches of 128 against 51.3k with 512) and one large batch from HBM (103.8k against 98.8k).
    if (!have_view && g_opt.batch_frames <= 0) {
        // stage_frames is quoted for a 100 000-atom system (a 154 MB float stage); larger systems get proportionally fewer frames per
        // batch - r03u: 1M atoms in batches of 128 frames (0.64 GB of bit streams each) spent 15 of 34 ms waiting for the first batch
        const size_t npad_s = (num_atoms + 63) & ~(size_t)63;
        const size_t S0 = (size_t)std::max(1, g_opt.stage_frames.load());
        const size_t S = std::max<size_t>(1, std::min<size_t>(S0, S0 * 100032 / std::max<size_t>(npad_s, 1)));
        // does the first frame of the range have checkpoints already?  (plain floats need none)
        bool warm = f32_ring;
        if (!f32_ring && device_decode && g_opt.xtc_checkpoints.load() && frame_beg < frame_end) {
            if (raw_ring) {
                std::lock_guard<std::mutex> l(g_ck_mtx);
                auto it = g_ck_store.find(CkKey(traj->inst, e->device));
                warm = it != g_ck_store.end() && it->second && it->second->frames == traj->num_frames(traj->inst)
                        && it->second->atoms == num_atoms &&
                       it->second->device == e->device && it->second->have.size() > frame_beg && flag_get(&it->second->have[frame_beg]);
            }
            else warm = rv_probe.ck_have && flag_get(&rv_probe.ck_have[frame_beg]);
        }
        // a first pass out of a mapped file keeps four walks in flight (stage_ahead below): half-size batches, twice as many
        cold_walk = raw_ring && !f32_ring && !warm && have_map && g_opt.xtc_device_decode.load() == 3 && g_opt.xtc_cold_streams.load() != 0;
        // (a walk takes as long for one frame as for a thousand - 7 ms for a c2 frame, 70 ms for 1M atoms -: never more launches than
        // decode streams for a short range)
        const size_t cold_b = std::max<size_t>(2 * S, (frame_end - frame_beg + vmd_script_eval_t::kDecodeStreams
                - 1) / vmd_script_eval_t::kDecodeStreams);
        Bmax = std::min<size_t>(Bmax, !device_decode ? S : (raw_ring ? (warm ? S : (cold_walk ? cold_b : 4 * S)) : (warm ? 8 * S : 4 * S)));
    }
    std::vector<Batch> batches;
    for (auto& sg : segments) plan_batches(e, sg.first, sg.second, Bmax, &batches);
    if (spec) for (auto& b : batches) if (b.blk < 0) return vmd_fail("read-ahead: region [%u, %u) is not made of whole frame blocks",
            frame_beg, frame_end);
    // From a file the first batch has to cross PCIe and be decompressed before any kernel can start, and nothing overlaps the last
    // batch's kernels (r03p timeline: 1.7 ms of a 12.3 ms c2 step before the first pair kernel, one DMA = 1.13 ms per 128 frames).
    // Option xtc_ramp: the run starts with an eighth and a quarter of a batch and ends with a quarter.  Measured (r03o): the shorter
    // fill is paid back by the pair kernel's lower efficiency on small launches - 81.5k frames/s either way, so it is off.
    if (raw_ring && g_opt.xtc_ramp.load() && batches.size() >= 3) {
        std::vector<Batch> ramped;
        auto carve_front = [&](Batch& b, size_t n) { ramped.push_back({b.f0, n, -1, 0}); b.f0 += n; b.nb -= n; };
        Batch first = batches.front(), last = batches.back();
        if (first.blk < 0 && first.nb >= 64) { carve_front(first, first.nb / 8); carve_front(first, first.nb / 3); }
        ramped.push_back(first);
        for (size_t i = 1; i + 1 < batches.size(); ++i) ramped.push_back(batches[i]);
        if (last.blk < 0 && last.nb >= 64) { const size_t tail = last.nb / 4; ramped.push_back({last.f0, last.nb - tail, -1, 0});
                ramped.push_back({last.f0 + last.nb - tail, tail, -1, 0}); }
        else ramped.push_back(last);
        batches.swap(ramped);
    }

    bool completed = true;
    // Batches staged ahead of the one being evaluated: one; optionally two when they are decompressed on the device (the decoder runs
    // UNDER the pair kernel, in the wave slots that kernel leaves).  r03n/r03p: the wait in settle_stage is the pipeline filling at the
    // start of a range, not a late decoder - two ahead measures the same, so one is the default.
    // A FIRST pass (no checkpoints yet) out of a mapped file: the walks of up to four batches run side by side (decode_streams).
    size_t stage_ahead = (raw_ring && g_opt.xtc_device_decode.load() == 3 && g_opt.xtc_decode_ahead.load() >= 2) ? 2 : 1;
    if (cold_walk) stage_ahead = std::min<size_t>(vmd_script_eval_t::kDecodeStreams, raw_ahead - 1);
    auto stage_of = [&](size_t bi) -> Stage& { return e->stages[bi % (stage_ahead + 1)]; };
    struct BlocksGuard {
        int old = -1;
        ~BlocksGuard() { if (old > 0) vmd_hip_set_rdf_blocks(old); }
    } blocks_guard;
    if (device_decode && batches.size() > 1 && g_opt.rdf_blocks_decode.load() >= 8) {
        blocks_guard.old = vmd_hip_set_rdf_blocks(g_opt.rdf_blocks_decode.load());
        // never raise a smaller setting
        if (blocks_guard.old < g_opt.rdf_blocks_decode.load()) vmd_hip_set_rdf_blocks(blocks_guard.old);
    }
    if (raw_ring) {
        for (auto& rs : e->raw_slots) rs.state = 0;
        for (size_t bi = 0; bi < std::min<size_t>(raw_ahead, batches.size()); ++bi)
            if (raw_upload(e, *slot_of(bi), traj, num_atoms, batches[bi].f0, batches[bi].nb) < 0) return false;
    }
    for (size_t bi = 0; bi < std::min(stage_ahead, batches.size()); ++bi)
        if (!fetch_stage(e, stage_of(bi), traj, vw, num_atoms, batches[bi].f0, batches[bi].nb, false, slot_of(bi))) return false;
    // ---- one batch in flight, one being queued.  The kernels of batch k + 1 are queued BEFORE the host waits for batch k (on an event,
    // not on the stream): the device never idles across the host's per-batch work - the wait itself, the bookkeeping, the ~15 launches
    // of the next batch (~0.15 ms per boundary, a tenth of a step when batches are the 128 frames a file-backed pass stages).  Everything
    // a batch hands to the host has two slots (overflow flag, temporal rows, a snapshot of the RDF counts behind its commits); a batch
    // whose cell build overflowed still voids itself AND whatever was queued behind it (the flag is sticky): the later batch is marked
    // and repeats its RDF part when its turn comes.  Evals that keep block partials complete every batch before the next is queued.
    struct Sub { size_t off, nb; long blk; };
    struct BatchCtx {
        Batch bt{0, 0, -1, 0};
        Stage* src = nullptr;
        size_t f0 = 0, nb = 0;
        uint32_t pbc = 0;
        std::vector<Sub> subs;
        bool two_streams = false;
        int slot = 0;
        bool active = false;        // queued, not completed
        bool poisoned = false;      // queued behind a batch that overflowed: its RDF part saw the flag and did nothing
        bool snapshot = false;      // h_snap[slot] holds the RDF counts behind this batch's commits (+ w_snap: the weights)
    };
    BatchCtx ctx[2];
    const bool defer = g_opt.defer_sync.load() != 0 && e->block_frames == 0 && batches.size() > 1;
    size_t rdf_counts = 0;
    for (auto& p : e->props) if (p->prop.kind == PROP_RDF) rdf_counts += p->ncounts;
    if (defer && rdf_counts) {
        if (e->h_snap_cap < 2 * rdf_counts) {
            if (e->h_snap) pool_give(e->h_snap);
            e->h_snap = nullptr; e->h_snap_cap = 0;
            HIP_OK(pool_take(kPinned, (void**)&e->h_snap, 2 * rdf_counts * sizeof(uint64_t)));
            e->h_snap_cap = 2 * rdf_counts;
        }
        e->w_snap.resize(2 * rdf_counts);
    }
    auto acc_of = [&](PropState* p, const Sub& sb) -> uint64_t* {
        return (sb.blk >= 0 && p->ncounts) ? p->d_blocks.p + (size_t)sb.blk * p->ncounts : p->d_counts.p;
    };
    // ---- RDF: one pair pass per (group, pass); launch_rdf may run again for this batch when a cell-build bucket overflowed
    // Every pass accumulates into its own scratch row and the rows are committed to the properties' accumulators by ONE
    // group of k_axpy_u64 launches at the very end, behind the overflow flag: by then every cell build of the batch has run,
    // so the flag is final and the batch's RDF part is all-or-nothing (a bucket of a LATER build may overflow after earlier
    // passes have long finished; nothing of them may stay behind when the batch is repeated).
    auto launch_rdf = [&](BatchCtx& c) -> bool {
        VMD_STAGE("batch: cell build + pair kernels");
        vmd_hip_set_rdf_closed(e->spec.rdf_closed ? 1 : 0);
        vmd_hip_set_rdf_raw(e->spec.rdf_raw ? 1 : 0);
        size_t scratch_rows = 0;
        for (auto& g : e->rdf_groups) scratch_rows += std::max(g.passes.size(), g.props.size());
        scratch_rows *= c.subs.size();
        if (!e->d_pass.ensure(std::max<size_t>(scratch_rows, 1) * VMD_RDF_NUM_BINS)) return false;
        HIP_OK(hipMemsetAsync(e->d_pass.p, 0, scratch_rows * VMD_RDF_NUM_BINS * sizeof(uint64_t), e->stream));
        struct Commit { uint64_t* dst; const uint64_t* src; uint64_t mult; };
        std::vector<Commit> commits;
        size_t row = 0;
        bool forked = false;
        for (auto& g : e->rdf_groups) {
            vmd_grid_t grid;
            // fully periodic cells use the frame boxes; open axes (non-periodic systems, slabs) span the batch's bounding box
            const bool open_axes = (c.pbc & 8u) == 0 && (c.pbc & VMD_UNITCELL_PBC_ALL) != VMD_UNITCELL_PBC_ALL;
            if (open_axes && !g_opt.force_brute && !prepare_open_boxes(e, *c.src, c.nb, c.pbc, num_atoms)) return false;
            const std::vector<float>& gb = (open_axes && c.src->gboxes_ready) ? c.src->h_gboxes : c.src->h_boxes;
            const float* d_gb = (open_axes && c.src->gboxes_ready) ? c.src->d_gboxes.p : c.src->d_boxes.p;
            // density of the sparsest selection any pass of this group puts in the lanes (the denser of its two), against the first frame's
            // cell
            bool dense_lanes = !open_axes && !g.passes.empty();
            if (dense_lanes) {
                const float* q = gb.data();
                const double vol = (double)q[0] * q[1] * q[2];
                for (auto& ps : g.passes) {
                    const size_t lanes = std::max(e->sels[ps.sel_a]->idx.size(), e->sels[ps.sel_b]->idx.size());
                    dense_lanes = dense_lanes && vol > 0.0 && (double)lanes / vol >= 0.08;
                }
            }
            if (e->spec.rdf_raw || !choose_grid(gb, c.pbc, c.nb, g.rmax, &grid, dense_lanes)) {
                // no grid for this batch (cutoff >= half the cell width, ...): all pairs, per property
                for (int pi : g.props) {
                    PropState* p = e->props[pi].get();
                    Selection* sa = e->sels[p->sel_a].get();
                    Selection* sb = e->sels[p->sel_b].get();
                    for (auto& su : c.subs) {
                        uint64_t* dst = e->d_pass.p + (row++) * VMD_RDF_NUM_BINS;
                        e->prof.begin("rdf_brute", e->stream);
                        KRN_OK(vmd_hip_rdf_brute(e->stream, c.src->base + su.off * c.src->frame_stride, c.src->frame_stride,
                                c.src->row_stride, c.src->d_boxes.p + 9 * su.off, c.pbc, (int)su.nb, sa->d_idx.p, (int)sa->idx.size(),
                                sb->d_idx.p, (int)sb->idx.size(), g.rmin, g.rmax, VMD_RDF_NUM_BINS, dst));
                        e->prof.end(e->stream);
                        commits.push_back({acc_of(p, su), dst, 1});
                    }
                }
                continue;
            }
            if (!e->d_partial.ensure(vmd_hip_rdf_partial_words())) return false;
            if (c.two_streams && !e->d_partial2.ensure(vmd_hip_rdf_partial_words())) return false;
            for (auto& ps : g.passes) {
                Selection* sa = e->sels[ps.sel_a].get();
                Selection* sb = e->sels[ps.sel_b].get();
                // passes with the same cutoff share the sorted copies; build_selection re-sorts when the grid differs
                if (forked) {     // the second stream still reads the sorted copies of the previous pass
                    HIP_OK(hipEventRecord(e->pair_join, e->pair_stream));
                    HIP_OK(hipStreamWaitEvent(e->stream, e->pair_join, 0));
                    forked = false;
                }
                if (!build_selection(e, sa, *c.src, d_gb, c.pbc, c.nb, grid)) return false;
                if (sb != sa && !build_selection(e, sb, *c.src, d_gb, c.pbc, c.nb, grid)) return false;
                // the pair set is symmetric in (ref, target): put the denser selection in the lanes - 64 of its atoms span a
                // shorter stretch of the pencil, so the x window of every segment carries less padding
                if (sb->idx.size() > sa->idx.size()) std::swap(sa, sb);
                if (c.two_streams) {
                    HIP_OK(hipEventRecord(e->pair_fork, e->stream));
                    HIP_OK(hipStreamWaitEvent(e->pair_stream, e->pair_fork, 0));
                    forked = true;
                }
                size_t si = 0;
                for (auto& su : c.subs) {
                    uint64_t* dst = e->d_pass.p + (row++) * VMD_RDF_NUM_BINS;
                    const bool second = c.two_streams && (si++ & 1);
                    hipStream_t ks = second ? e->pair_stream : e->stream;
                    if (!second) e->prof.begin("rdf_pencil", ks);
                    KRN_OK(vmd_hip_rdf_pencil(ks, sa->sorted.p + su.off * 3 * (size_t)sa->nsel_pad, sa->cell_start.p + su.off
                            * (size_t)(grid.ncell + 1), (int)sa->idx.size(), sa->nsel_pad, sb->sorted.p + su.off * 3
                            * (size_t)sb->nsel_pad, sb->cell_start.p + su.off * (size_t)(grid.ncell + 1), (int)sb->idx.size(),
                            sb->nsel_pad, d_gb + 9 * su.off, (int)su.nb, grid, g.rmin, g.rmax, VMD_RDF_NUM_BINS, ps.same ? 1 : 0,
                            g_opt.rdf_variant, c.pbc, second ? e->d_partial2.p : e->d_partial.p, dst, e->d_overflow.p));
                    if (!second) e->prof.end(ks);
                    if (e->spec.rdf_closed && ps.same && g.rmin <= 0.0f && 0.0f <= g.rmax) {
                        // closed interval: d = 0 is a hit, but a same-set pass walks the half shell (j > i, every hit twice) and never
                        // meets the pairs (i, i) - one per list entry and frame, all in the bin of d = 0 (SPEC S4 binning of 0)
                        int bin0 = (int)(((0.0f - g.rmin) * (1.0f / (g.rmax - g.rmin))) * (float)VMD_RDF_NUM_BINS);
                        bin0 = std::min(std::max(bin0, 0), VMD_RDF_NUM_BINS - 1);
                        KRN_OK(vmd_hip_bump_u64(ks, dst + bin0, (uint64_t)su.nb * (uint64_t)sa->idx.size()));
                    }
                    for (auto& tg : ps.targets) commits.push_back({acc_of(e->props[tg.first].get(), su), dst, tg.second});
                }
            }
        }
        if (forked) {
            HIP_OK(hipEventRecord(e->pair_join, e->pair_stream));
            HIP_OK(hipStreamWaitEvent(e->stream, e->pair_join, 0));
        }
        for (auto& cm : commits) KRN_OK(vmd_hip_axpy_u64(e->stream, cm.dst, cm.src, VMD_RDF_NUM_BINS, cm.mult, e->d_overflow.p));
        HIP_OK(hipMemcpyAsync(&e->h_overflow[c.slot], e->d_overflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        return true;
    };

    // waits for a queued batch (`later`: the batch already queued behind it, if any), repeats its RDF part when a bucket overflowed, books
    // its frames
    auto complete_batch = [&](BatchCtx& c, BatchCtx* later) -> bool {
        if (!c.active) return true;
        c.active = false;
        const bool behind = later && later->active;
        VMD_STAGE("batch: waiting for its kernels");
        { HostTimer host_timer("host_sync_wait");
          if (behind) HIP_OK(hipEventSynchronize(e->batch_done[c.slot]));
          else HIP_OK(hipStreamSynchronize(e->stream)); }
        VMD_STAGE("batch: host bookkeeping");
        // a bucket of the two-level cell build was too small: nothing reached the histograms (every consumer saw the flag).
        // Re-measure the selections that used buckets with more head room and evaluate the RDF part of this batch again.
        bool repeated = false;
        for (int attempt = 0; e->h_overflow[c.slot] != 0; ++attempt) {
            if (attempt >= 4) return vmd_fail("cell build: pencil buckets keep overflowing");
            // the batch behind this one saw the flag too: let it drain, it repeats its RDF part at its own completion
            if (behind) {
                HIP_OK(hipStreamSynchronize(e->stream));
                later->poisoned = true;
            }
            const bool own = !(c.poisoned && attempt == 0);      // a poisoned batch did not overflow itself (as far as anyone knows)
            // one bit per selection (Selection::overflow_bit; selections beyond 32 share)
            const uint32_t who = e->h_overflow[c.slot];
            for (size_t si = 0; si < e->sels.size(); ++si) {
                Selection* sl = e->sels[si].get();
                sl->built = false;
                // only the selection whose buckets were too small gets wider ones.  (Its bit, not used_pencil, says so: a selection can
                // be sorted through buckets on one group's grid and by the single-block build on another's within ONE batch - co-evaluated
                // RDFs with different cutoffs - and used_pencil only remembers the last of them; fuzz seed 8941, round 4.)
                if (!own) continue;
                if (!(who & sl->overflow_bit)) {
                    // round 6: the others are re-measured with more head room too, without a strike against them - what crowded one
                    // selection's pencils crowds the next one's a batch later, and every overflow repeats a batch
                    if (sl->used_pencil && sl->cap_margin < 2.0f) { sl->pen_off.clear(); sl->caps_cache.clear(); sl->cap_margin = 2.0f; }
                    continue;
                }
                // a rare event worth a line in the host's log: it costs the batch a second cell build, and after three of them the
                // selection leaves the two-level build for good
                char msg[256];
                snprintf(msg, sizeof(msg),
                        "cell build: a pencil bucket of selection %zu (%zu atoms, %d x %d pencils, capacity margin %.2f, "
                         "largest bucket %d) overflowed in frames [%zu, %zu): re-measured on every frame of the batch, margin x 1.6",
                         si, sl->idx.size(), sl->pen_ny, sl->pen_nz, (double)sl->cap_margin, sl->cap_max, c.f0, c.f0 + c.nb);
                vmd_log(VMD_LOG_INFO, msg);
                sl->pen_off.clear();
                sl->caps_cache.clear();
                sl->cap_margin *= 1.6f;
                sl->overflows += 1;
            }
            e->h_overflow[c.slot] = 0;
            HIP_OK(hipMemsetAsync(e->d_overflow.p, 0, sizeof(uint32_t), e->stream));
            if (!launch_rdf(c)) return false;
            HIP_OK(hipStreamSynchronize(e->stream));
            repeated = true;
        }
        if (c.bt.blk >= 0 && !spec)
            for (auto& su : c.subs)
                for (auto& p : e->props) if (p->ncounts) KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, acc_of(p.get(), su), p->ncounts));
        e->prof.resolve();
        if (g_prof_on) { std::lock_guard<std::mutex> l(g_prof_mtx); g_prof["batches"].launches += 1; }
        size_t toff = 0;
        for (auto& p : e->props) {
            if (p->prop.kind != PROP_DIST) continue;
            // evaluated ahead: the rows wait beside the view until their block is committed (a reader of the values array never sees a
            // frame nobody asked for)
            if (spec && p->ahead_values.size() != p->values.size()) p->ahead_values.assign(p->values.size(), 0.0f);
            memcpy(spec ? &p->ahead_values[c.f0 * p->dim1] : &p->values[c.f0 * p->dim1], e->h_temporal_slot[c.slot].data() + toff, c.nb
                    * p->dim1 * sizeof(float));
            toff += c.nb * p->dim1;
        }
        e->frames_computed += c.nb;
        if (c.bt.blk >= 0) for (auto& su : c.subs) e->block_ready[su.blk] = spec ? BLOCK_ROWS_AHEAD : BLOCK_ROWS_IN_PLACE;
        if (spec) return true;
        for (size_t b = 0; b < c.nb; ++b) mask_set(e->frame_mask, c.f0 + b);
        e->frames_done += c.nb;
        // cheap views are refreshed every batch so a polling GUI sees progress (src/main.cpp:1508-1524): from the device when nothing
        // is queued behind this batch, from the snapshot taken behind its commits otherwise
        size_t soff = (size_t)c.slot * rdf_counts;
        for (auto& p : e->props) {
            if (p->prop.kind != PROP_RDF) continue;
            if (behind && c.snapshot && !repeated && !(later && later->poisoned)) refresh_distribution_from(p.get(), e->h_snap + soff,
                    e->w_snap.data() + soff);
            else if (!behind && views) { if (!refresh_distribution(e, p.get())) return false; }
            soff += p->ncounts;
        }
        return true;
    };

    for (size_t bi = 0; bi < batches.size(); ++bi) {
        if (e->interrupt) { completed = false; break; }
        BatchCtx& c = ctx[bi & 1];
        BatchCtx& prev = ctx[(bi & 1) ^ 1];
        c = BatchCtx{};
        c.bt = batches[bi]; c.f0 = c.bt.f0; c.nb = c.bt.nb; c.slot = (int)(bi & 1);
        c.src = &stage_of(bi);
        { HostTimer host_timer("host_settle"); if (!settle_stage(e, *c.src, traj, num_atoms)) return false; }
        VMD_STAGE("batch: kernels queued");
        HostTimer queue_timer("host_queue_to_sync");
        HIP_OK(hipStreamWaitEvent(e->stream, c.src->ready, 0));
        c.pbc = batch_pbc(*c.src);
        for (auto& s : e->sels) s->built = false;

        size_t temporal_floats = 0;
        for (auto& p : e->props) if (p->prop.kind == PROP_DIST) temporal_floats += c.nb * p->dim1;
        e->h_temporal_slot[c.slot].resize(temporal_floats);
        size_t toff = 0;

        // a whole frame block accumulates into its own partial first and is merged into the totals afterwards.  A batch of blocks
        // (filtered evaluation) is evaluated block by block - `subs` - behind one cell build and in front of one synchronisation.
        if (c.bt.blk >= 0 && c.bt.nblk > 1) {
            const size_t S = e->block_frames;
            for (size_t j = 0; j < c.bt.nblk; ++j) c.subs.push_back({j * S, std::min(S, c.nb - j * S), c.bt.blk + (long)j});
        } else c.subs.push_back({0, c.nb, c.bt.blk});
        if (c.bt.blk >= 0)
            for (auto& sb : c.subs)
                for (auto& p : e->props) if (p->ncounts) HIP_OK(hipMemsetAsync(acc_of(p.get(), sb), 0, p->ncounts * sizeof(uint64_t),
                        e->stream));
        // the blocks' pair launches alternate between the eval's stream and a second one (own partial rows): a 50-frame launch of a
        // 100k-atom system is ~3 work items per resident wave, and the tail of one launch then runs under the head of the next
        c.two_streams = c.subs.size() > 1 && g_opt.block_two_streams.load() != 0;

        e->h_overflow[c.slot] = 0;
        if (!e->rdf_groups.empty() && !launch_rdf(c)) return false;

        for (auto& p : e->props) {
            const Property& d = p->prop;
            if (d.kind == PROP_RDF) {
                // SPEC S4 normalisation, fp64 on the host (needs only the box)
                for (auto& su : c.subs) {
                    double* bw = su.blk >= 0 ? &p->block_weights64[(size_t)su.blk * p->ncounts] : nullptr;
                    if (bw) std::fill(bw, bw + p->ncounts, 0.0);
                    for (size_t b = su.off; b < su.off + su.nb; ++b) {
                        const float* L = &c.src->h_boxes[9 * b];
                        double V;
                        // also the triclinic volume
                        if ((c.pbc & VMD_UNITCELL_PBC_ALL) == VMD_UNITCELL_PBC_ALL && e->spec.rdf_norm != 1) V = (double)L[0]
                                * (double)L[1] * (double)L[2];
                        else V = (4.0 / 3.0) * M_PI * (double)d.rmax * (double)d.rmax * (double)d.rmax;
                        const double rho = (e->spec.rdf_norm == 2 ? 1.0 : (double)d.a.size()) * (double)d.b.size() / V;
                        const double w = ((double)d.rmax - (double)d.rmin) / (double)p->ncounts;
                        for (size_t k = 0; k < p->ncounts; ++k) {
                            const double r0 = (double)d.rmin + w * (double)k;
                            const double r1 = (double)d.rmin + w * (double)(k + 1);
                            const double wk = rho * (4.0 / 3.0) * M_PI * (r1 * r1 * r1 - r0 * r0 * r0);
                            if (!spec) p->weights64[k] += wk;
                            if (bw) bw[k] += wk;
                        }
                    }
                }
                p->dirty = p->dirty || !spec;
            } else if (d.kind == PROP_SDF) {
                if (!p->d_R32.ensure(c.nb * d.K * 9) || !p->d_c32.ensure(c.nb * d.K * 3) || !p->d_group.ensure(c.nb * 4)) return false;
                VMD_STAGE("batch: sdf align + scatter");
                e->prof.begin("sdf_align", e->stream);
                if (p->have_tree && !p->d_tree_pos.ensure(c.nb * d.K * d.m * 3)) return false;
                KRN_OK(vmd_hip_sdf_align(e->stream, c.src->base, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p, c.pbc,
                        (int)c.nb, p->d_structs.p, p->d_mass.p, (int)d.K, (int)d.m, p->d_ref_pose.p, p->d_R32.p, p->d_c32.p, nullptr,
                        p->d_group.p, p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr, p->have_tree
                        ? p->d_tree_pos.p : nullptr));
                e->prof.end(e->stream);
                e->prof.begin("sdf_scatter", e->stream);
                for (auto& su : c.subs) KRN_OK(vmd_hip_sdf_scatter(e->stream, c.src->base + su.off * c.src->frame_stride,
                        c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p + 9 * su.off, c.pbc, (int)su.nb, p->d_structs.p, (int)d.K,
                        (int)d.m, p->d_R32.p + su.off * d.K * 9, p->d_c32.p + su.off * d.K * 3, p->d_tgt.p, (p->have_owner
                        && !e->spec.sdf_include_self) ? p->d_owner.p : nullptr, (int)d.b.size(), d.rmax, VMD_VOLUME_DIM, acc_of(p.get(),
                        su), p->d_group.p + 4 * su.off, (p->have_tag && p->tag_len == c.src->row_stride && !e->spec.sdf_include_self)
                        ? p->d_tag.p : nullptr, p->tgt_first, p->tgt_stride, (p->unowned || e->spec.sdf_include_self) ? 1 : 0));
                e->prof.end(e->stream);
                p->dirty = p->dirty || !spec;
            } else {
                if (!p->d_out.ensure(c.nb * p->dim1)) return false;
                e->prof.begin("distance", e->stream);
                KRN_OK(vmd_hip_distance(e->stream, c.src->base, c.src->frame_stride, c.src->row_stride, c.src->d_boxes.p, c.pbc, (int)c.nb,
                        d.dist_kind, (int)p->dist_P, (int)p->dist_per, p->d_a.p, p->d_ma.p, p->d_aoff.p, p->d_b.p, p->d_mb.p, p->d_boff.p,
                        p->d_out.p));
                e->prof.end(e->stream);
                HIP_OK(hipMemcpyAsync(e->h_temporal_slot[c.slot].data() + toff, p->d_out.p, c.nb * p->dim1 * sizeof(float),
                        hipMemcpyDeviceToHost, e->stream));
                toff += c.nb * p->dim1;
                p->dirty = p->dirty || !spec;
            }
        }
        if (defer) {
            // what the host will want from this batch once a later one is queued behind it: the RDF counts as they stand behind its
            // commits (the weights as they stand now), and an event to wait on
            size_t soff = (size_t)c.slot * rdf_counts;
            for (auto& p : e->props) {
                if (p->prop.kind != PROP_RDF) continue;
                HIP_OK(hipMemcpyAsync(e->h_snap + soff, p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
                memcpy(e->w_snap.data() + soff, p->weights64.data(), p->ncounts * sizeof(double));
                soff += p->ncounts;
            }
            c.snapshot = true;
            HIP_OK(hipEventRecord(e->batch_done[c.slot], e->stream));
        }
        c.active = true;
        // deferred: the batch in front of this one is completed now that the device has this one to go on with (its stage is free
        // for the staging below only then)
        if (defer && !complete_batch(prev, &c)) return false;
        VMD_STAGE("batch: staging the next batch (fetch_stage)");
        // the kernels of this batch are queued: load the next batch on the host while they run
        if (bi + 1 < batches.size() && !e->interrupt) {
            if (bi + stage_ahead < batches.size()) {
                HostTimer host_timer("host_fetch_stage");
                const size_t nx = bi + stage_ahead;
                if (!fetch_stage(e, stage_of(nx), traj, vw, num_atoms, batches[nx].f0, batches[nx].nb, false, slot_of(nx))) return false;
            }
            // ... and send the bit streams of the batch after that on their way (its slot held batch bi - 1: decoded long ago)
            if (raw_ring && bi + raw_ahead < batches.size() &&
                raw_upload(e, *slot_of(bi + raw_ahead), traj, num_atoms, batches[bi + raw_ahead].f0, batches[bi
                        + raw_ahead].nb) < 0) return false;
        }
        if (!defer && !complete_batch(c, nullptr)) return false;
    }
    // whatever is still in flight (deferred: the last batch queued; after an interrupt: the one before the break)
    { BatchCtx& a = ctx[0].active && ctx[1].active ? (ctx[0].f0 < ctx[1].f0 ? ctx[0] : ctx[1]) : ctx[0];
      BatchCtx& b = &a == &ctx[0] ? ctx[1] : ctx[0];
      if (!complete_batch(a, b.active ? &b : nullptr)) return false;
      if (!complete_batch(b, nullptr)) return false; }
    if (views) {
        for (auto& p : e->props) {
            if (!p->dirty) continue;
            if (p->prop.kind == PROP_SDF) { if (!refresh_volume(e, p.get())) return false; }
            else if (p->prop.kind == PROP_DIST) refresh_temporal_stats(e, p.get());
        }
        e->views_at = std::chrono::steady_clock::now();
    }
    return completed;
}

bool process_range(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end,
        bool views) {
    g_last_error.clear();
    if (eval->interrupt) return false;
    std::lock_guard<std::mutex> lock(eval->mtx);
    return process_range_locked(eval, sys, traj, frame_beg, frame_end, views, false);
}

// the host views of every property whose accumulators changed since its last refresh (the combining queue calls this when no call
// is waiting, process_range(views = true) does the same at its end)
bool refresh_views_locked(vmd_script_eval_t* e) {
    HIP_OK(hipSetDevice(e->device));
    for (auto& p : e->props) {
        if (!p->dirty) continue;
        if (p->prop.kind == PROP_RDF) { if (!refresh_distribution(e, p.get())) return false; }
        else if (p->prop.kind == PROP_SDF) {
            // vmd_eval_defer_volume_views: a rank of a multi-GPU evaluation does not materialise ITS partial volume's float view (8.4 MB
            // over PCIe after every range) - the merge re-derives the view of the merged counts (vmd_eval_reduce -> vmd_eval_finalize); the
            // volume stays dirty until then
            if (e->defer_volume_views.load(std::memory_order_relaxed)) continue;
            if (!refresh_volume(e, p.get())) return false;
        }
        else refresh_temporal_stats(e, p.get());
    }
    e->views_at = std::chrono::steady_clock::now();
    return true;
}

bool refresh_views(vmd_script_eval_t* e) {
    std::lock_guard<std::mutex> lock(e->mtx);
    return refresh_views_locked(e);
}

// The hot call.  VIAMD invokes it from N pool threads with small disjoint ranges (grain 1, src/main.cpp:993-997,
// src/task_system.cpp:73-81).  Launching kernels per call would drown the GPU in tiny batches, so calls COMBINE: the first
// caller becomes the leader, later callers queue their range and sleep; the leader repeatedly takes everything queued so far,
// merges adjacent ranges into long runs and evaluates those in large frame batches, then wakes the owners.
bool combine_call(vmd_script_eval_t* eval, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    RangeRequest me;
    me.beg = frame_beg; me.end = frame_end; me.sys = sys; me.traj = traj;
    std::unique_lock<std::mutex> ql(eval->queue_mtx);
    eval->queue.push_back(&me);
    if (eval->leader_active) {
        eval->queue_cv.wait(ql, [&] { return me.done; });
        if (!me.ok) g_last_error = me.error;
        return me.ok;
    }
    eval->leader_active = true;
    while (!eval->queue.empty()) {
        // VIAMD's pool threads pull ranges of a few frames each (enkiTS: num_frames / (threads x (threads - 1)), at least 1) and every
        // one of them blocks in here, so a round can never hold more than threads x grain frames - and far fewer if the leader runs
        // off with whatever is queued the instant it looks: the threads it has just released are back with their next ranges within
        // microseconds.  It waits for them (gather_us at most, only while requests keep arriving) - a batch of 16 x 4 frames costs the
        // same ~0.15 ms of launches and round trips as a batch of 4.
        // Waiting is only worth a fraction of what a round costs: the slowest of the released threads needs 50 - 100 us to come back,
        // which a round of the 10 000-frame SDF (0.08 ms for 16 frames) cannot afford and a round of the 100k-atom RDF (0.3 ms) can:
        // at most half the previous round's time.  A large pool brings enough frames per round by itself, and on an oversubscribed
        // host waiting for 128 threads costs more than it gathers: pools of up to 32 callers only.
        const int gather = (int)std::min<long>(g_opt.gather_us.load(), eval->last_round_us / 2);
        // Scripts without pair passes (SDF / distance only: 0.7 us of kernels per frame) never gain from it - measured r03an: 129 ms
        // without, 195 ms with, for the 10 000 frames of config 4 from 16 threads - so only evals with RDF groups wait.
        if (gather >= 20 && !eval->rdf_groups.empty() && eval->queue.size() < eval->last_round && eval->last_round <= 32) {
            const auto t0 = std::chrono::steady_clock::now();
            const auto deadline = t0 + std::chrono::microseconds(gather);
            auto last_arrival = t0;
            size_t seen = eval->queue.size();
            while (eval->queue.size() < eval->last_round && !eval->interrupt) {
                ql.unlock();
                std::this_thread::yield();
                ql.lock();
                const auto now = std::chrono::steady_clock::now();
                if (eval->queue.size() != seen) { seen = eval->queue.size(); last_arrival = now; }
                // nobody new for a third of the window: the task is running out of ranges (its tail), or the pool is busy elsewhere
                if (now >= deadline || now - last_arrival > std::chrono::microseconds(gather / 3 + 1)) break;
            }
        }
        std::vector<RangeRequest*> taken;
        taken.swap(eval->queue);
        eval->last_round = taken.size();
        ql.unlock();
        // the views are for readers, and a reader only needs them final when the LAST call returns: while other calls are waiting they
        // are brought up to date at most every lazy_views_ms (a polling GUI still sees progress), and always before a round whose end
        // finds the queue empty hands its callers back
        const bool lazy = g_opt.lazy_views.load() != 0;
        bool all_ok = true;
        const auto round_t0 = std::chrono::steady_clock::now();
        // requests for the same trajectory, sorted by first frame; touching ranges fuse into one run
        // "the same trajectory" = the same instance behind the same callbacks, not the same interface STRUCT: a host that wraps its own
        // trajectory type per call (include/vmd_md_script_shim.h did, from every pool thread) presents a different address each time
        // (such ranges never fused)
        auto same_traj = [](const vmd_trajectory_i* a, const vmd_trajectory_i* b) {
            return a == b || (a->inst == b->inst && a->load_frame == b->load_frame && a->device_view == b->device_view
                    && a->load_raw == b->load_raw);
        };
        std::sort(taken.begin(), taken.end(), [](const RangeRequest* a, const RangeRequest* b) {
            return a->traj->inst != b->traj->inst ? a->traj->inst < b->traj->inst : a->beg < b->beg; });
        size_t i = 0;
        while (i < taken.size()) {
            size_t j = i + 1;
            uint32_t run_end = taken[i]->end;
            while (j < taken.size() && same_traj(taken[j]->traj, taken[i]->traj) && taken[j]->beg == run_end) { run_end = taken[j]->end;
                    ++j; }
            const bool ok = process_range(eval, taken[i]->sys, taken[i]->traj, taken[i]->beg, run_end, !lazy);
            const std::string err = ok ? std::string() : g_last_error;
            for (size_t k = i; k < j; ++k) { taken[k]->ok = ok; taken[k]->error = err; }
            all_ok = all_ok && ok;
            i = j;
        }
        const long round_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now()
                - round_t0).count();
        ql.lock();
        eval->last_round_us = round_us;
        if (lazy) {
            const bool overdue = std::chrono::steady_clock::now() - eval->views_at > std::chrono::milliseconds(std::max(1,
                    g_opt.lazy_views_ms.load()));
            if (eval->queue.empty() || overdue) {
                ql.unlock();
                const bool vok = refresh_views(eval);
                if (!vok && all_ok) { const std::string err = g_last_error; for (RangeRequest* r : taken) { r->ok = false; r->error = err;
                        } }
                ql.lock();
            }
        }
        for (RangeRequest* r : taken) r->done = true;
        eval->queue_cv.notify_all();
    }
    eval->leader_active = false;
    ql.unlock();
    if (!me.ok) g_last_error = me.error;
    return me.ok;
}
