// viamd_amd/csrc/vmd_eval_calls.cpp - how calls ARRIVE.  VIAMD does not call md_script_eval_frame_range once: a pool
// task hands single frames to many threads that all block in it (/root/reference/src/main.cpp:993-997, src/task_system.cpp:73-81).
// Combining queue (the first caller leads, later ones queue), read-ahead (regions evaluated ahead into block partials, calls only mark
// their frames, whoever leaves last settles), the opt-in deferred settle on a helper thread, and vmd_eval_frame_range itself.
#include "vmd_eval_internal.h"

void ra_reset(vmd_script_eval_t* e) {          // clear_data (mtx held): a new evaluation starts
    ReadAhead& ra = e->ra;
    if (ra.blk_state) for (size_t b = 0; b < e->num_blocks; ++b) ra.blk_state[b] = vmd_script_eval_t::RA_NONE;
    if (ra.frame_req) for (size_t i = 0; i < 64 * ra.req_stride; ++i) ra.frame_req[i] = 0;
    ra.marks_pending = false; ra.views_dirty = false;
    ra.concurrent = false; ra.lonely = false; ra.disabled = false; ra.strikes = 0; ra.next_region = 0; ra.failed = false; ra.error.clear();
    ra.lone.store(false);
}

size_t ra_block_frames(const vmd_script_eval_t* e, size_t Bmax) {
    size_t G = (size_t)std::max(0, g_opt.readahead_block.load());
    if (!G) {
        // streaming scripts: 16.8 MB of memset + add per volume and block - few, large blocks (r04c, 10 000-frame SDF at grain 1: 10.8 ms
        // with 256, 9.7 with 512, 9.2 with 1 024; one call 7.0)
        if (e->rdf_groups.empty()) G = 1024;
        else {
            // pair passes: one pair launch per block; it needs ~4M selected atoms to fill the chip (50-frame launches of the
            // 100k-atom box cost +12 %, 125-frame launches +5 %), and a block is also the most a ragged range end evaluates directly
            size_t sel = 1;
            for (auto& g : e->rdf_groups) for (auto& ps : g.passes) sel = std::max(sel, std::max(e->sels[ps.sel_a]->idx.size(),
                    e->sels[ps.sel_b]->idx.size()));
            G = 16;
            while (G < 128 && G * sel < 4000000) G *= 2;
        }
    }
    return std::max<size_t>(1, std::min(G, Bmax));
}

// queue_mtx held by the caller (and no combining call in flight): allocate the block partials and the states
bool ra_engage(vmd_script_eval_t* e, vmd_trajectory_i* traj) {
    ReadAhead& ra = e->ra;
    std::lock_guard<std::mutex> lock(e->mtx);
    HIP_OK(hipSetDevice(e->device));
    const size_t num_atoms = traj->num_atoms(traj->inst);
    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    const size_t Bmax = auto_batch(e, num_atoms, !have_view);
    if (e->block_frames == 0) {
        // a filtered eval (src/main.cpp:1014-1039) adopts whole blocks from its source's partials: same blocks as the source (the source
        // may be engaging at this very moment - "Eval Full" and "Eval Filt" side by side: its block size is read under its mutex; order:
        // own mutex, then the source's, as everywhere)
        size_t src_S = 0;
        if (e->source) { std::lock_guard<std::mutex> sl(e->source->mtx); src_S = e->source->block_frames; }
        const size_t S = src_S ? std::min(src_S, std::max<size_t>(Bmax, 1)) : ra_block_frames(e, Bmax);
        const size_t nblocks = (e->num_frames + S - 1) / S;
        size_t bytes = 0;
        for (auto& p : e->props) bytes += nblocks * p->ncounts * sizeof(uint64_t);
        // not worth a ninth of the HBM, nor more than half of what is free right now (a trajectory resident in HBM may have taken most of
        // it): the combining queue serves this eval, as it did before read-ahead existed
        size_t cap_bytes = (size_t)32 << 30, free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_bytes = std::min(cap_bytes, free_b / 2);
        if (bytes > cap_bytes) { ra.disabled = true; return true; }
        for (auto& p : e->props) {
            if (!p->ncounts) continue;
            if (!p->d_blocks.ensure(nblocks * p->ncounts) || g_opt.readahead_fail_alloc.load()) {
                // an allocation that fails here must not fail the evaluation (and every later call with it) - read-ahead is an
                // optimisation; give back what was taken and let the combining queue serve the calls
                for (auto& q : e->props) { q->d_blocks.release(); q->block_weights64.clear(); q->block_weights64.shrink_to_fit(); }
                g_last_error.clear();
                ra.disabled = true;
                return true;
            }
            if (p->prop.kind == PROP_RDF) p->block_weights64.assign(nblocks * p->ncounts, 0.0);
        }
        e->block_ready.reset(new std::atomic<uint8_t>[nblocks]);
        for (size_t b = 0; b < nblocks; ++b) e->block_ready[b] = 0;
        e->num_blocks = nblocks;
        e->block_frames = S;
        ra.own_blocks = true;
    }
    const size_t S = e->block_frames;
    ra.blk_state.reset(new std::atomic<uint8_t>[e->num_blocks]);
    ra.req_stride = std::max<size_t>(64, (e->num_frames + 63) / 64);
    ra.frame_req.reset(new std::atomic<uint8_t>[64 * ra.req_stride]);
    for (size_t i = 0; i < 64 * ra.req_stride; ++i) ra.frame_req[i] = 0;
    // a rank's shard of a device trajectory: only blocks that lie inside it can be evaluated ahead
    size_t lo = 0, hi = e->num_frames;
    if (have_view && view_sharded(view)) { lo = view.resident_beg; hi = view.resident_end; }
    for (size_t b = 0; b < e->num_blocks; ++b) {
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        bool done = false;
        for (size_t f = f0; f < f1; ++f) { ra.req(f) = e->frame_mask[f] ? 1 : 0; done = done || e->frame_mask[f]; }
        ra.blk_state[b] = (done || f0 < lo || f1 > hi || f1 - f0 > Bmax) ? vmd_script_eval_t::RA_DIRECT : vmd_script_eval_t::RA_NONE;
    }
    ra.bmax = Bmax;
    ra.traj_inst = traj_id(traj);
    ra.on.store(true, std::memory_order_release);
    return true;
}

// The filtered evaluation under VIAMD's call pattern: blocks of a region that the source eval has finished are not evaluated again - their
// partials (and temporal rows) are copied from the source into this eval's own block partials, where they wait to be requested like any
// block evaluated ahead.  mtx held, device set.  adopted[b - b0] = 1 for the blocks taken.
bool ra_adopt_blocks(vmd_script_eval_t* e, const TrajId& traj_inst, size_t b0, size_t b1, std::vector<char>* adopted) {
    adopted->assign(b1 - b0, 0);
    vmd_script_eval_t* src = e->source;
    if (!src || src->props.size() != e->props.size()) return true;
    std::lock_guard<std::mutex> lock(src->mtx);       // order: own mutex, then the source's (as reuse_blocks)
    if (src->block_frames != e->block_frames || src->blocks_inst != traj_inst) return true;
    const size_t S = e->block_frames;
    size_t taken = 0;
    for (size_t b = b0; b < b1; ++b) {
        if (b >= src->num_blocks || !src->block_ready[b]) continue;
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        for (size_t i = 0; i < e->props.size(); ++i) {
            PropState* p = e->props[i].get();
            const PropState* q = src->props[i].get();
            if (p->ncounts) {
                HIP_OK(hipMemcpyAsync(p->d_blocks.p + b * p->ncounts, q->d_blocks.p + b * p->ncounts, p->ncounts * sizeof(uint64_t),
                        hipMemcpyDeviceToDevice, e->stream));
                if (p->prop.kind == PROP_RDF) memcpy(&p->block_weights64[b * p->ncounts], &q->block_weights64[b * p->ncounts], p->ncounts
                        * sizeof(double));
            } else {
                if (p->ahead_values.size() != p->values.size()) p->ahead_values.assign(p->values.size(), 0.0f);
                memcpy(&p->ahead_values[f0 * p->dim1], block_rows(src, q, b) + f0 * p->dim1, (f1 - f0) * p->dim1 * sizeof(float));
            }
        }
        e->block_ready[b] = BLOCK_ROWS_AHEAD;           // (the rows went into the side buffer above)
        (*adopted)[b - b0] = 1;
        taken += f1 - f0;
    }
    if (taken) {
        HIP_OK(hipStreamSynchronize(e->stream));          // the source's partials are read before its mutex is released
        e->frames_reused += taken;
    }
    return true;
}

// mtx held, device set: the block's partial joins the totals
bool ra_commit_block(vmd_script_eval_t* e, size_t blk) {
    const size_t S = e->block_frames, f0 = blk * S, f1 = std::min(f0 + S, e->num_frames);
    for (auto& p : e->props) {
        if (p->ncounts) {
            KRN_OK(vmd_hip_add_u64(e->stream, p->d_counts.p, p->d_blocks.p + blk * p->ncounts, p->ncounts));
            if (p->prop.kind == PROP_RDF) for (size_t k = 0; k < p->ncounts; ++k) p->weights64[k] += p->block_weights64[blk * p->ncounts
                    + k];
        } else if (p->ahead_values.size() == p->values.size()) {
            memcpy(&p->values[f0 * p->dim1], &p->ahead_values[f0 * p->dim1], (f1 - f0) * p->dim1 * sizeof(float));
        }
        p->dirty = true;
    }
    for (size_t f = f0; f < f1; ++f) mask_set(e->frame_mask, f);
    e->frames_done += f1 - f0;
    if (e->block_ready[blk]) e->block_ready[blk] = BLOCK_ROWS_IN_PLACE;
    e->ra.blk_state[blk].store(vmd_script_eval_t::RA_COMMITTED, std::memory_order_release);
    e->ra.committed_blocks += 1;
    e->ra.views_dirty = true;
    return true;
}

// Brings the accumulators up to what has been requested.  full = false (a region leader, before its region): whole requested blocks are
// committed, requested frames of direct blocks evaluated.  full = true (a call that leaves alone): also the requested frames of partly
// requested blocks - evaluated directly, the block is direct from then on - and the host views.
bool ra_settle(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, bool full) {
    ReadAhead& ra = e->ra;
    std::lock_guard<std::mutex> sl(ra.settle_mtx);
    std::lock_guard<std::mutex> lock(e->mtx);
    HIP_OK(hipSetDevice(e->device));
    const size_t S = e->block_frames;
    // before the scan: whoever marks after this point sets it again (ra_fast)
    if (full) (void)ra.marks_pending.exchange(false, std::memory_order_seq_cst);
    std::vector<std::pair<uint32_t, uint32_t>> runs;          // frames to evaluate directly
    bool tainted = false;
    for (size_t b = 0; b < e->num_blocks; ++b) {
        const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
        if (st != vmd_script_eval_t::RA_READY && st != vmd_script_eval_t::RA_DIRECT) continue;
        const size_t f0 = b * S, f1 = std::min(f0 + S, e->num_frames);
        if (st == vmd_script_eval_t::RA_READY) {
            size_t req = 0;
            // seq_cst: ordered after the exchange of marks_pending above
            for (size_t f = f0; f < f1; ++f) req += ra.req(f).load(std::memory_order_seq_cst) ? 1 : 0;
            if (req == f1 - f0) { if (!ra_commit_block(e, b)) return false; continue; }
            if (req == 0 || !full) continue;
            // partly requested: its frames are evaluated one by one from now on
            ra.blk_state[b].store(vmd_script_eval_t::RA_DIRECT, std::memory_order_release);
            tainted = true;
        }
        for (size_t f = f0; f < f1; ++f) {
            if (!ra.req(f).load(std::memory_order_seq_cst) || e->frame_mask[f]) continue;
            if (!runs.empty() && runs.back().second == f) runs.back().second = (uint32_t)f + 1;
            else runs.push_back({(uint32_t)f, (uint32_t)f + 1});
        }
    }
    for (auto& r : runs) {
        g_last_error.clear();
        if (e->interrupt) return false;
        if (!process_range_locked(e, sys, traj, r.first, r.second, false, false)) return false;
        ra.direct_frames += r.second - r.first;
        ra.views_dirty = true;
    }
    // (settle_mtx) callers that keep leaving blocks half requested are not a pool walking a range
    if (tainted && ++ra.strikes >= 3) ra.disabled = true;
    const bool overdue = std::chrono::steady_clock::now() - e->views_at > std::chrono::milliseconds(std::max(1,
            g_opt.lazy_views_ms.load()));
    if ((full || overdue) && ra.views_dirty.exchange(false)) { if (!refresh_views_locked(e)) return false; }
    ra.settles += full ? 1 : 0;
    return true;
}

// no lock: 1 = every frame of [beg, end) lies in an evaluated (or direct) block and is now marked requested
bool ra_fast(vmd_script_eval_t* e, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    const size_t S = e->block_frames;
    for (size_t b = beg / S; b <= (size_t)(end - 1) / S; ++b) {
        const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
        if (st != vmd_script_eval_t::RA_READY && st != vmd_script_eval_t::RA_DIRECT) return false;
    }
    // asked for twice: the slow path sorts that out
    for (uint32_t f = beg; f < end; ++f) if (ra.req(f).load(std::memory_order_relaxed)) return false;
    // the marks FIRST, then the flag, both sequentially consistent (the other order lost marks - B sees or sets the flag, the
    // settling A clears it and scans B's block before B's CAS lands, B marks, and B's ra_leave finds the flag clear: requested frames that
    // nobody commits).  A settle clears the flag with a seq_cst exchange and scans after it: a mark that the scan misses is followed by a
    // store of the flag that the exchange did not clear, so the marker's own ra_leave (or a later caller's) settles again.
    // a lost race = another call for the same frame owns it
    for (uint32_t f = beg; f < end; ++f) { uint8_t z = 0; (void)ra.req(f).compare_exchange_strong(z, 1, std::memory_order_seq_cst); }
    ra.marks_pending.store(true, std::memory_order_seq_cst);
    return true;
}

// the eval is not evaluating ahead for this call (a large range, a lone caller, read-ahead given up): the combining queue evaluates it when
// it arrives.  With block states in place the blocks it touches become direct FIRST, so that no partial of theirs is committed later.
bool ra_direct_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    if (!ra.on.load(std::memory_order_acquire)) {
        { std::lock_guard<std::mutex> ql(e->queue_mtx); ra.combining += 1; }
        const bool ok = combine_call(e, sys, traj, beg, end);
        { std::lock_guard<std::mutex> ql(e->queue_mtx); ra.combining -= 1; }
        e->queue_cv.notify_all();
        return ok;
    }
    const size_t S = e->block_frames;
    const size_t b0 = beg / S, b1 = (size_t)(end - 1) / S;
    for (;;) {
        {
            std::unique_lock<std::mutex> ql(e->queue_mtx);
            e->queue_cv.wait(ql, [&] {
                if (e->interrupt) return true;
                for (size_t b = b0; b <= b1; ++b) if (ra.blk_state[b].load() == vmd_script_eval_t::RA_PENDING) return false;
                return true; });
        }
        if (e->interrupt) { g_last_error.clear(); return false; }
        // whatever has been requested in those blocks so far is settled first (committed whole, or evaluated), then they are direct
        bool ready = false;
        for (size_t b = b0; b <= b1; ++b) ready = ready || ra.blk_state[b].load() == vmd_script_eval_t::RA_READY;
        if (ready && !ra_settle(e, sys, traj, true)) return false;
        std::lock_guard<std::mutex> sl(ra.settle_mtx);
        std::lock_guard<std::mutex> ql(e->queue_mtx);
        // a region leader took one of them in the meantime: wait for it, or its partial would count these frames again
        bool pending = false;
        for (size_t b = b0; b <= b1; ++b) pending = pending || ra.blk_state[b].load() == vmd_script_eval_t::RA_PENDING;
        if (pending) continue;
        for (size_t b = b0; b <= b1; ++b) {
            const uint8_t st = ra.blk_state[b].load();
            if (st == vmd_script_eval_t::RA_READY || st == vmd_script_eval_t::RA_NONE) ra.blk_state[b].store(vmd_script_eval_t::RA_DIRECT,
                    std::memory_order_release);
        }
        break;
    }
    const bool ok = combine_call(e, sys, traj, beg, end);
    if (ok) for (uint32_t f = beg; f < end; ++f) ra.req(f).store(1, std::memory_order_release);
    return ok;
}

bool ra_call(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t beg, uint32_t end) {
    ReadAhead& ra = e->ra;
    const bool small = (int)(end - beg) <= g_opt.readahead_small.load();
    if (small && !ra.disabled && ra.on.load(std::memory_order_acquire) && ra.concurrent.load(std::memory_order_relaxed)
            && ra.traj_inst == traj_id(traj) && ra_fast(e, beg, end)) return true;
    std::unique_lock<std::mutex> ql(e->queue_mtx);
    if ((uint32_t)ra.flight.load() >= 2 && !ra.concurrent) { ra.concurrent = true; e->queue_cv.notify_all(); }
    if (small && !ra.disabled) {
        // (also on an eval whose blocks exist from an earlier evaluation: whether THIS evaluation is driven by a pool is found out anew)
        if (!ra.concurrent && !ra.lonely) {
            const int pref = ra.lone_pref.load(std::memory_order_relaxed);
            if (pref < 0 ? g_opt.readahead_lone.load() > 0 : pref > 0) {
                // opted in: every small call is part of a walk, whoever makes it - served like a pool's, settled by the helper thread
                ra.lone.store(true);
                ra.concurrent = true;
            } else {
                // the first call of an evaluation: is this a pool?  Its other threads are microseconds behind
                cv_wait_us(e->queue_cv, ql, std::max(0, g_opt.readahead_company_us.load()), [&] { return ra.concurrent
                        || e->interrupt.load(); });
                if (!ra.concurrent) ra.lonely = true;
            }
        }
        if (ra.concurrent && !ra.on.load()) {
            // calls that went to the combining queue before anyone knew
            e->queue_cv.wait(ql, [&] { return ra.combining == 0 || ra.on.load(); });
            if (!ra.on.load() && !ra_engage(e, traj)) return false;
        }
    }
    if (!(small && !ra.disabled && ra.concurrent && ra.on.load() && ra.traj_inst == traj_id(traj))) {
        ql.unlock();
        return ra_direct_call(e, sys, traj, beg, end);
    }
    ra.slow_calls += 1;
    const size_t S = e->block_frames;
    const size_t b0 = beg / S, b1 = (size_t)(end - 1) / S;
    for (;;) {
        if (e->interrupt) { g_last_error.clear(); return false; }
        if (ra.failed) { g_last_error = ra.error; return false; }
        size_t need = (size_t)-1;
        bool pending = false;
        for (size_t b = b0; b <= b1; ++b) {
            const uint8_t st = ra.blk_state[b].load(std::memory_order_acquire);
            if (st == vmd_script_eval_t::RA_NONE) { need = b; break; }
            pending = pending || st == vmd_script_eval_t::RA_PENDING;
        }
        if (need == (size_t)-1 && !pending) break;
        if (need != (size_t)-1 && !ra.spec_active) {
            // this call leads a region: whole blocks from `need` on, as far as nobody has touched them
            size_t want = ra.next_region ? ra.next_region : (size_t)std::max(1, g_opt.readahead_frames.load());
            want = std::min(std::max(want, S), std::max(ra.bmax, S));
            size_t e1 = need, frames = 0;
            while (e1 < e->num_blocks && ra.blk_state[e1].load() == vmd_script_eval_t::RA_NONE && frames < want) {
                frames += std::min((e1 + 1) * S, e->num_frames) - e1 * S;
                ++e1;
            }
            // a caller that walks the range downwards (enkiTS: the thread that owns the task set pops its partitions from the far end while
            // the others steal from the near end) finds everything above its block taken: the region grows towards lower frames instead
            // (only when the way up is blocked - by evaluated blocks or the end of the trajectory - and never across blocks that are not
            // free)
            while (frames < want && need > 0 && ra.blk_state[need - 1].load() == vmd_script_eval_t::RA_NONE) {
                --need;
                frames += S;
            }
            ra.next_region = std::min(std::max(ra.bmax, S), want * (size_t)std::max(1, g_opt.readahead_growth.load()));
            for (size_t b = need; b < e1; ++b) ra.blk_state[b].store(vmd_script_eval_t::RA_PENDING, std::memory_order_release);
            ra.spec_active = true;
            ql.unlock();
            const uint32_t f_lo = (uint32_t)(need * S), f_hi = (uint32_t)std::min(e1 * S, e->num_frames);
            // what the callers have asked for so far joins the totals: progress for a polling GUI
            bool ok = ra_settle(e, sys, traj, false);
            if (ok) {
                g_last_error.clear();
                std::lock_guard<std::mutex> lock(e->mtx);
                std::vector<char> adopted;
                ok = hipSetDevice(e->device) == hipSuccess && ra_adopt_blocks(e, traj_id(traj), need, e1, &adopted);
                for (size_t b = need; b < e1 && ok;) {           // what the source could not supply: evaluated, in runs of blocks
                    if (adopted[b - need]) { ++b; continue; }
                    size_t r1 = b;
                    while (r1 < e1 && !adopted[r1 - need]) ++r1;
                    ok = !e->interrupt && process_range_locked(e, sys, traj, (uint32_t)(b * S), (uint32_t)std::min(r1 * S, e->num_frames),
                            false, true);
                    b = r1;
                }
            }
            const std::string err = ok ? std::string() : g_last_error;
            ql.lock();
            ra.spec_active = false;
            for (size_t b = need; b < e1; ++b) ra.blk_state[b].store(ok ? vmd_script_eval_t::RA_READY : vmd_script_eval_t::RA_NONE,
                    std::memory_order_release);
            if (ok) { ra.regions += 1; ra.region_frames += f_hi - f_lo; }
            else if (!e->interrupt) { ra.failed = true; ra.error = err; }
            e->queue_cv.notify_all();
            if (!ok) { g_last_error = err; return false; }
            continue;
        }
        e->queue_cv.wait(ql);
    }
    ql.unlock();
    // every block is evaluated (READY), direct or already committed: mark what can be marked, evaluate the rest now (frames asked for
    // twice - the combining queue counts them twice, as it always has)
    std::vector<std::pair<uint32_t, uint32_t>> again;
    for (uint32_t f = beg; f < end; ++f) {
        uint8_t z = 0;
        const bool committed = ra.blk_state[f / S].load(std::memory_order_acquire) == vmd_script_eval_t::RA_COMMITTED;
        if (!committed && ra.req(f).compare_exchange_strong(z, 1, std::memory_order_seq_cst)) continue;
        if (!again.empty() && again.back().second == f) again.back().second = f + 1;
        else again.push_back({f, f + 1});
    }
    ra.marks_pending.store(true, std::memory_order_seq_cst);      // after the marks, as in ra_fast
    for (auto& r : again) if (!combine_call(e, sys, traj, r.first, r.second)) return false;
    return true;
}

// ---- deferred settle (option readahead_lone) ----------------------------------------------------------------------------------
int64_t steady_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The helper thread of an eval in deferred-settle mode: sleeps until a settle is owed (armed) and the eval has been quiet for
// readahead_lone_settle_us since the last call left, then does what a pool's last leaver does - as a call of its own (flight + 1), so every
// hand-over rule of ra_settle / ra_leave holds unchanged.  Marks that arrive while it settles keep it armed.
void lone_helper_main(vmd_script_eval_t* e) {
    ReadAhead& ra = e->ra;
    ReadAhead::Helper& h = ra.helper;
    std::unique_lock<std::mutex> lk(h.mtx);
    for (;;) {
        h.cv.wait(lk, [&] { return h.quit || h.armed.load(); });
        if (h.quit) break;
        for (;;) {
            const int64_t due = h.last_leave_ns.load() + (int64_t)std::max(1, g_opt.readahead_lone_settle_us.load()) * 1000;
            const int64_t now = steady_ns();
            if (h.quit || !h.armed.load() || now >= due) break;
            cv_wait_us(h.cv, lk, (int)((due - now) / 1000 + 1), [&] { return h.quit || !h.armed.load(); });
        }
        if (h.quit) break;
        if (!h.armed.load() || !h.have) continue;        // cancelled (clear_data, wait_settled)
        // an interrupted evaluation is not completed behind the host's back
        if (e->interrupt.load()) { h.armed.store(false); h.idle_cv.notify_all(); continue; }
        h.busy = true;
        const uint64_t seq = h.cancel_seq;
        vmd_system_t sys = h.sys;
        vmd_trajectory_i traj = h.traj;
        lk.unlock();
        bool retry = false;
        {
            const uint64_t w = ra.flight.fetch_add(1, std::memory_order_acq_rel);
            if ((uint32_t)w == 0) {
                g_last_error.clear();
                const bool ok = ra_settle(e, &sys, &traj, true);
                if (!ok && !e->interrupt && !g_last_error.empty()) {
                    std::lock_guard<std::mutex> ql(e->queue_mtx);
                    ra.failed = true; ra.error = g_last_error;                 // the next call reports it
                }
                h.settles += 1;
            } else {
                retry = true;                                                     // a call is inside: it stamps last_leave when it goes
            }
            ra.flight.fetch_sub(1, std::memory_order_acq_rel);
            // the host's records of this eval follow NOW (the shim re-publishes fingerprint / ranges / max_value: ADVICE r05 #1) - after
            // the settle, before `busy` drops: clear_data / interrupt / free wait for the callback too, it never runs on a freed host
            // object
            if (!retry) if (auto cb = h.on_settled.load(std::memory_order_acquire)) cb(h.on_settled_user.load(std::memory_order_acquire));
        }
        lk.lock();
        h.busy = false;
        if (h.cancel_seq != seq) {
            // cancelled while it ran (interrupt, clear_data, wait_settled): whatever is marked from now on belongs to calls that arm afresh
            // -
            // with THEIR system and trajectory (lone_arm copies them only when it arms)
            h.armed.store(false, std::memory_order_seq_cst);
        } else if (retry) {
            h.last_leave_ns.store(std::max(h.last_leave_ns.load(), steady_ns()));
        } else {
            // Disarm, THEN look at the marks (both seq_cst) - the mirror image of a leaving call, which marks and then looks at `armed`
            // (lone_arm): at least one of the two sees the other, so a mark made while this settle ran is never left without an owner
            h.armed.store(false, std::memory_order_seq_cst);
            if (ra.marks_pending.load(std::memory_order_seq_cst) || ra.views_dirty.load(std::memory_order_seq_cst)) h.armed.store(true,
                    std::memory_order_seq_cst);
        }
        h.idle_cv.notify_all();
    }
}

// a call that leaves last in deferred-settle mode: stamp the time, make sure the helper knows a settle is owed
void lone_arm(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj) {
    ReadAhead::Helper& h = e->ra.helper;
    h.last_leave_ns.store(steady_ns(), std::memory_order_relaxed);
    // (the caller's marks are seq_cst stores before this load: see lone_helper_main)
    if (h.armed.load(std::memory_order_seq_cst)) return;
    std::lock_guard<std::mutex> l(h.mtx);
    if (sys) h.sys = *sys; else memset(&h.sys, 0, sizeof(h.sys));
    h.traj = *traj;
    h.have = true;
    if (!h.started) { h.started = true; h.th = std::thread(lone_helper_main, e); }
    h.armed.store(true, std::memory_order_release);
    h.cv.notify_one();
}

// clear_data / wait_settled: no settle may start from now on, and none is running when this returns (call WITHOUT e->mtx held)
void lone_cancel(vmd_script_eval_t* e) {
    ReadAhead::Helper& h = e->ra.helper;
    std::unique_lock<std::mutex> lk(h.mtx);
    if (!h.started) return;
    h.cancel_seq += 1;
    h.armed.store(false);
    h.cv.notify_one();
    h.idle_cv.wait(lk, [&] { return !h.busy; });
}

void lone_stop(vmd_script_eval_t* e) {           // vmd_eval_free
    ReadAhead::Helper& h = e->ra.helper;
    {
        std::lock_guard<std::mutex> l(h.mtx);
        if (!h.started) return;
        // a settle that is running ends at its next batch boundary: nobody will read its results
        e->interrupt = true;
        h.quit = true;
        h.cv.notify_one();
    }
    h.th.join();
}

// the end of every call: whoever leaves last settles (or hands the duty to a call that has arrived since)
bool ra_leave(vmd_script_eval_t* e, const vmd_system_t* sys, vmd_trajectory_i* traj) {
    ReadAhead& ra = e->ra;
    const bool lone = ra.lone.load(std::memory_order_relaxed);
    if (lone) ra.helper.last_leave_ns.store(steady_ns(), std::memory_order_relaxed);
    for (;;) {
        const uint64_t w = ra.flight.fetch_sub(1, std::memory_order_acq_rel);
        if ((uint32_t)w != 1) return true;
        if (!ra.on.load(std::memory_order_acquire) || (!ra.marks_pending.load() && !ra.views_dirty.load())) return true;
        if (e->interrupt) return true;
        if (lone) { lone_arm(e, sys, traj); return true; }      // deferred: the helper settles once the eval has been quiet
        const uint64_t a0 = w >> 32;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(std::max(0, g_opt.readahead_linger_us.load()));
        while (std::chrono::steady_clock::now() < deadline) {
            if ((ra.flight.load(std::memory_order_acquire) >> 32) != a0) return true;
            std::this_thread::yield();
        }
        if ((ra.flight.load(std::memory_order_acquire) >> 32) != a0) return true;
        ra.flight.fetch_add(1, std::memory_order_acq_rel);
        if (!ra_settle(e, sys, traj, true)) { ra.flight.fetch_sub(1, std::memory_order_acq_rel); return false; }
    }
}

extern "C" bool vmd_eval_frame_range(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys,
                                     vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end) {
    g_last_error.clear();   // a false return with an empty message means "interrupted"
    if (!eval || !traj) return vmd_fail("vmd_eval_frame_range: NULL argument");
    if (ir && vmd_ir_fingerprint(ir) != eval->ir_fingerprint) return vmd_fail("vmd_eval_frame_range: eval was created from a different ir");
    if (frame_end > eval->num_frames) frame_end = (uint32_t)eval->num_frames;
    if (frame_beg >= frame_end) return true;
    if (eval->interrupt) return false;
    if (!g_opt.readahead.load()) {
        if (!eval->ra.on.load()) return combine_call(eval, sys, traj, frame_beg, frame_end);
        eval->ra.flight.fetch_add(((uint64_t)1 << 32) | 1, std::memory_order_acq_rel);
        const bool ok = ra_direct_call(eval, sys, traj, frame_beg, frame_end);
        const std::string err = ok ? std::string() : g_last_error;
        const bool lok = ra_leave(eval, sys, traj);
        if (!ok) g_last_error = err;
        return ok && lok;
    }
    eval->ra.flight.fetch_add(((uint64_t)1 << 32) | 1, std::memory_order_acq_rel);
    const bool ok = ra_call(eval, sys, traj, frame_beg, frame_end);
    const std::string err = ok ? std::string() : g_last_error;
    const bool lok = ra_leave(eval, sys, traj);
    if (!ok) g_last_error = err;
    return ok && lok;
}

// VIAMD's call pattern as a utility (src/main.cpp:993-997, src/task_system.cpp:73-81): `num_threads` pool threads pull ranges of `grain`
// frames off [frame_beg, frame_end) and call vmd_eval_frame_range on the ONE eval, each blocking until its frames are evaluated.  What
// bench.py times VIAMD's pattern with (native threads: a Python thread per call costs more than a small call does), and what a host
// without a task system of its own can use as is.  Returns false if any call failed or was interrupted.
extern "C" bool vmd_eval_frame_range_pooled(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys,
        vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end, int num_threads, uint32_t grain) {
    if (!eval || !traj) return vmd_fail("vmd_eval_frame_range_pooled: NULL argument");
    if (num_threads < 1) num_threads = 1;
    if (grain < 1) grain = 1;
    std::atomic<uint32_t> next{frame_beg};
    std::atomic<bool> ok{true};
    std::mutex err_mtx;
    std::string err;
    auto work = [&] {
        for (;;) {
            const uint32_t b = next.fetch_add(grain, std::memory_order_relaxed);
            if (b >= frame_end || b < frame_beg) break;           // (b < frame_beg: the counter wrapped)
            const uint32_t e = frame_end - b < grain ? frame_end : b + grain;
            if (!vmd_eval_frame_range(eval, ir, sys, traj, b, e)) {
                std::lock_guard<std::mutex> l(err_mtx);
                if (err.empty()) err = g_last_error;               // thread-local in the worker: carried to the caller below
                ok.store(false);
                break;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < num_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (!ok.load()) g_last_error = err;
    return ok.load();
}

extern "C" bool vmd_eval_set_settled_callback(vmd_script_eval_t* eval, void (*fn)(void*), void* user) {
    if (!eval) return vmd_fail("eval is NULL");
    eval->ra.helper.on_settled_user.store(user, std::memory_order_release);
    eval->ra.helper.on_settled.store(fn, std::memory_order_release);
    return true;
}

extern "C" bool vmd_eval_set_deferred_settle(vmd_script_eval_t* eval, int mode) {
    if (!eval) return vmd_fail("eval is NULL");
    // takes effect at the next clear_data / first small call of an evaluation
    eval->ra.lone_pref.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed);
    return true;
}

// Deferred-settle mode (option readahead_lone): everything the calls so far have asked for joins the totals and the views NOW, on the
// calling thread, instead of when the helper's quiet period is over.  Call after the last vmd_eval_frame_range has returned; a no-op for
// every other eval (their last call has settled before it returned).
extern "C" bool vmd_eval_wait_settled(vmd_script_eval_t* eval) {
    if (!eval) return vmd_fail("eval is NULL");
    ReadAhead& ra = eval->ra;
    if (!ra.lone.load()) return true;
    lone_cancel(eval);
    ReadAhead::Helper& h = ra.helper;
    vmd_system_t sys; vmd_trajectory_i traj;
    {
        std::lock_guard<std::mutex> l(h.mtx);
        if (!h.have) return true;
        sys = h.sys; traj = h.traj;
    }
    g_last_error.clear();
    bool ok = true;
    ra.flight.fetch_add(1, std::memory_order_acq_rel);
    if (ra.on.load(std::memory_order_acquire) && (ra.marks_pending.load() || ra.views_dirty.load())) ok = ra_settle(eval, &sys, &traj,
            true);
    ra.flight.fetch_sub(1, std::memory_order_acq_rel);
    if (ok) { std::lock_guard<std::mutex> ql(eval->queue_mtx); if (ra.failed) { g_last_error = ra.error; ok = false; } }
    if (auto cb = h.on_settled.load(std::memory_order_acquire)) cb(h.on_settled_user.load(std::memory_order_acquire));
    return ok;
}

extern "C" void vmd_eval_readahead_stats(const vmd_script_eval_t* eval, vmd_readahead_stats_t* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    if (!eval) return;
    const ReadAhead& ra = eval->ra;
    out->engaged = ra.on.load() ? 1 : 0;
    out->block_frames = ra.on.load() ? (uint32_t)eval->block_frames : 0;
    out->regions = ra.regions.load(); out->region_frames = ra.region_frames.load();
    out->slow_calls = ra.slow_calls.load();
    out->settles = ra.settles.load(); out->direct_frames = ra.direct_frames.load(); out->committed_blocks = ra.committed_blocks.load();
}
