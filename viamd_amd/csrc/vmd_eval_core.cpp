// viamd_amd/csrc/vmd_eval_core.cpp - the evaluator object: vmd_eval_create / _free / _clear_data / _interrupt (md_script_eval_create
// ..., /root/reference/src/main.cpp:960-990), the plan of pair passes for co-evaluated RDFs, the host views VIAMD polls
// (md_script_property_data_t: values / weights / aggregate / ranges / fingerprint, :1286-1524), accessors, block / source settings of
// the filtered evaluation, and the sdf() vis payload (reference pose + per-frame matrices, density_volume.cpp:183-204).
#include "vmd_eval_internal.h"

// what `values` of a volume points at between clear_data and the evaluation's first view: zero pages shared by every volume of the process,
// mapped read-only (a write through the pointer is a bug and faults loudly) and never backed by memory of their own (anonymous pages that
// are only ever read all alias the kernel's zero page)
float* zero_volume_view(size_t nfloats) {
    static std::mutex mtx;
    static float* view = nullptr;
    static size_t cap = 0;
    std::lock_guard<std::mutex> l(mtx);
    if (nfloats > cap) {
        void* m = mmap(nullptr, nfloats * sizeof(float), PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return nullptr;
        view = (float*)m; cap = nfloats;        // (an earlier, smaller mapping stays: readers may still hold it)
    }
    return view;
}

TrajId traj_id(const vmd_trajectory_i* t) {
    TrajId id;
    if (t) { id.inst = t->inst; id.fn = t->load_frame ? reinterpret_cast<const void*>(t->load_frame)
            : reinterpret_cast<const void*>(t->device_view); }
    return id;
}

PropState* find_prop(const vmd_script_eval_t* e, const char* name) {
    if (!e || !name) return nullptr;
    for (auto& p : e->props) if (p->prop.name == name) return p.get();
    return nullptr;
}

int intern_selection(vmd_script_eval_t* e, const std::vector<int32_t>& idx) {
    for (size_t i = 0; i < e->sels.size(); ++i) if (e->sels[i]->idx == idx) return (int)i;
    auto s = std::make_unique<Selection>();
    s->idx = idx;
    // periodic? the smallest m <= 4 with idx[t + m] - idx[t] the same for every t ("every third atom", "the two H of every water")
    if (g_opt.cells_sel_pattern.load() && idx.size() >= 8) {
        for (int m = 1; m <= 4 && !s->pat_m; ++m) {
            const int64_t period = (int64_t)idx[(size_t)m] - (int64_t)idx[0];
            if (period <= 0 || period > (1 << 20)) continue;
            bool ok = true;
            for (size_t t = 0; t + (size_t)m < idx.size() && ok; ++t) ok = (int64_t)idx[t + (size_t)m] - (int64_t)idx[t] == period;
            if (!ok) continue;
            s->pat_m = m; s->pat_first = idx[0]; s->pat_period = (int)period;
            for (int k = 0; k < m; ++k) s->pat_off[k] = idx[(size_t)k] - idx[0];
        }
    }
    s->overflow_bit = 1u << (e->sels.size() % 32);
    e->sels.push_back(std::move(s));
    return (int)e->sels.size() - 1;
}

// Groups the RDF properties by range and decides, per group, between one pair pass per property and the class decomposition
// (see PairPass).  Classes are used when every set is duplicate-free, there are at most 8 of them, and the pair work
// (sum over passes of n_a * n_b, halved for same-set passes) drops by at least 10 %.
void build_rdf_plan(vmd_script_eval_t* e) {
    e->rdf_groups.clear();
    for (size_t i = 0; i < e->props.size(); ++i) {
        const Property& d = e->props[i]->prop;
        if (d.kind != PROP_RDF) continue;
        RdfGroup* g = nullptr;
        for (auto& q : e->rdf_groups) if (memcmp(&q.rmin, &d.rmin, sizeof(float)) == 0 && memcmp(&q.rmax, &d.rmax, sizeof(float)) == 0) g =
                &q;
        if (!g) { e->rdf_groups.emplace_back(); g = &e->rdf_groups.back(); g->rmin = d.rmin; g->rmax = d.rmax; }
        g->props.push_back((int)i);
    }
    // every property keeps its own sets as selections (index lists only; sorted copies exist for the selections passes use):
    // the all-pairs kernel, which takes over when a batch cannot use the grid, works per property
    for (auto& p : e->props) {
        if (p->prop.kind != PROP_RDF) continue;
        p->sel_a = intern_selection(e, p->prop.a);
        p->sel_b = p->same_set ? p->sel_a : intern_selection(e, p->prop.b);
    }
    for (auto& g : e->rdf_groups) {
        auto direct = [&]() {
            g.passes.clear(); g.classes = false;
            for (int pi : g.props) {
                PropState* p = e->props[pi].get();
                PairPass ps;
                ps.sel_a = p->sel_a; ps.sel_b = p->sel_b;
                ps.same = p->same_set;
                ps.targets.push_back({pi, 1});
                g.passes.push_back(std::move(ps));
            }
        };
        const size_t np = g.props.size();
        if (np < 2 || np > 30 || !g_opt.rdf_classes) { direct(); continue; }
        // signature of every atom: bit 2k = in the reference set of the group's k-th property, bit 2k + 1 = in its target set
        int32_t amax = 0;
        for (int pi : g.props) { for (int32_t a : e->props[pi]->prop.a) amax = std::max(amax, a); for (int32_t b
                : e->props[pi]->prop.b) amax = std::max(amax, b); }
        std::vector<uint64_t> sig((size_t)amax + 1, 0);
        bool dup = false;
        for (size_t k = 0; k < np && !dup; ++k) {
            const Property& d = e->props[g.props[k]]->prop;
            for (int side = 0; side < 2 && !dup; ++side) {
                const uint64_t bit = 1ull << (2 * k + side);
                for (int32_t a : (side ? d.b : d.a)) { if (sig[a] & bit) { dup = true; break; } sig[a] |= bit; }
            }
        }
        if (dup) { direct(); continue; }      // a set that lists an atom twice counts it twice: only the direct passes reproduce that
        std::vector<uint64_t> csig;
        std::vector<std::vector<int32_t>> cidx;
        bool too_many = false;
        for (int32_t a = 0; a <= amax && !too_many; ++a) {
            if (!sig[a]) continue;
            size_t c = 0;
            while (c < csig.size() && csig[c] != sig[a]) ++c;
            if (c == csig.size()) { if (csig.size() == 8) { too_many = true; break; } csig.push_back(sig[a]); cidx.emplace_back(); }
            cidx[c].push_back(a);
        }
        if (too_many) { direct(); continue; }
        std::vector<PairPass> passes;
        double cost_classes = 0.0, cost_direct = 0.0;
        for (size_t c = 0; c < csig.size(); ++c)
            for (size_t d = c; d < csig.size(); ++d) {
                PairPass ps;
                for (size_t k = 0; k < np; ++k) {
                    const uint64_t X = 1ull << (2 * k), Y = 1ull << (2 * k + 1);
                    uint64_t mult;
                    if (c == d) mult = ((csig[c] & X) && (csig[c] & Y)) ? 1 : 0;
                    else mult = (((csig[c] & X) && (csig[d] & Y)) ? 1 : 0) + (((csig[d] & X) && (csig[c] & Y)) ? 1 : 0);
                    if (mult) ps.targets.push_back({g.props[k], mult});
                }
                if (ps.targets.empty()) continue;
                ps.same = c == d;
                ps.sel_a = (int)c; ps.sel_b = (int)d;          // class indices for now
                cost_classes += (double)cidx[c].size() * (double)cidx[d].size() * (c == d ? 0.5 : 1.0);
                passes.push_back(std::move(ps));
            }
        for (int pi : g.props) {
            const PropState* p = e->props[pi].get();
            cost_direct += (double)p->prop.a.size() * (double)p->prop.b.size() * (p->same_set ? 0.5 : 1.0);
        }
        if (!(cost_classes < 0.9 * cost_direct)) { direct(); continue; }
        std::vector<int> csel(csig.size());
        for (size_t c = 0; c < csig.size(); ++c) csel[c] = intern_selection(e, cidx[c]);
        for (auto& ps : passes) { ps.sel_a = csel[ps.sel_a]; ps.sel_b = csel[ps.sel_b]; }
        g.passes = std::move(passes);
        g.classes = true;
    }
}

extern "C" vmd_script_eval_t* vmd_eval_create(size_t num_frames, const vmd_script_ir_t* ir) {
    if (!ir) { vmd_fail("vmd_eval_create: ir is NULL"); return nullptr; }
    if (vmd_device_count() <= 0) { vmd_fail("vmd_eval_create: no usable HIP device (the evaluator has no CPU path)"); return nullptr; }
    auto e = std::make_unique<vmd_script_eval_t>();
    if (hipGetDevice(&e->device) != hipSuccess) { vmd_fail("hipGetDevice failed"); return nullptr; }
    // streams and events come out of the process-wide cache (an eval has nine streams and ~20 events; VIAMD makes one per script edit)
    if (!(e->stream = pool_stream(false)) || !(e->copy_stream = pool_stream(false)) || !(e->aux_stream = pool_stream(false)) ||
        !(e->pair_stream = pool_stream(false))) { vmd_fail("hipStreamCreate failed"); return nullptr; }
    if (!(e->pair_fork = pool_event(false)) || !(e->pair_join = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    // the decoder's waves should get the wave slots the pair kernel leaves free as soon as a batch has arrived: highest priority
    for (auto& ds : e->decode_streams) if (!(ds = pool_stream(true))) { vmd_fail("hipStreamCreate failed"); return nullptr; }
    e->decode_stream = e->decode_streams[0];
    for (auto& rs : e->raw_slots) if (!(rs.uploaded = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    for (auto& st : e->stages) if (!(st.ready = pool_event(true))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    e->ir_fingerprint = vmd_ir_fingerprint(ir);
    e->num_frames = num_frames;
    e->spec.rdf_closed = g_opt.spec_rdf_closed.load() != 0;
    e->spec.sdf_include_self = g_opt.spec_sdf_include_self.load() != 0;
    e->spec.sdf_density = g_opt.spec_sdf_density.load() != 0;
    e->spec.rdf_raw = g_opt.spec_rdf_raw.load() != 0;
    e->spec.rdf_norm = g_opt.spec_rdf_norm.load();
    e->spec.dist_geometric_com = g_opt.spec_dist_geometric_com.load() != 0;
    e->frame_mask.assign(num_frames, 0);
    for (auto& p : ir->props) {
        auto st = std::make_unique<PropState>();
        st->prop = p;
        memset(&st->data, 0, sizeof(st->data));
        memset(&st->aggregate, 0, sizeof(st->aggregate));
        switch (p.kind) {
        case PROP_RDF:
            st->ncounts = VMD_RDF_NUM_BINS;
            st->values.assign(st->ncounts, 0.0f); st->weights.assign(st->ncounts, 0.0f);
            st->counts.assign(st->ncounts, 0); st->weights64.assign(st->ncounts, 0.0);
            st->data.dim[0] = 1; st->data.dim[1] = 1; st->data.dim[2] = (int32_t)st->ncounts; st->data.dim[3] = 0;
            st->data.weights = st->weights.data();
            st->data.weights64 = st->weights64.data();
            st->data.min_range[0] = p.rmin; st->data.max_range[0] = p.rmax;
            st->data.unit_str[0] = "\xC3\x85"; st->data.unit_str[1] = "";
            st->same_set = (p.a == p.b);         // selections are interned by build_rdf_plan (own sets, or the classes they split into)
            break;
        case PROP_SDF:
            st->ncounts = (size_t)VMD_VOLUME_DIM * VMD_VOLUME_DIM * VMD_VOLUME_DIM;
            // the 8 + 17 MB host views of a volume are pinned: their D2H refresh runs at PCIe speed
            st->values.assign(st->ncounts, 0.0f, true);
            st->counts.assign(st->ncounts, 0, true);
            st->pinned = st->values.pinned && st->counts.pinned;
            st->data.dim[0] = 1; st->data.dim[1] = st->data.dim[2] = st->data.dim[3] = VMD_VOLUME_DIM;
            st->data.min_range[0] = -p.rmax; st->data.max_range[0] = p.rmax;
            st->data.unit_str[0] = ""; st->data.unit_str[1] = "";
            break;
        case PROP_DIST:
            st->dist_P = p.aoff.size() - 1;
            st->dist_per = p.dist_kind == VMD_DISTANCE_PAIR ? (size_t)p.aoff[1] * (size_t)p.boff[1] : 1;
            st->dim1 = st->dist_P * st->dist_per;
            st->values.assign(num_frames * st->dim1, 0.0f);
            st->data.dim[0] = (int32_t)num_frames; st->data.dim[1] = (int32_t)st->dim1;
            st->data.unit_str[0] = ""; st->data.unit_str[1] = "\xC3\x85";
            if (st->dim1 > 1) {
                st->agg_mean.assign(num_frames, 0.0f); st->agg_var.assign(num_frames, 0.0f); st->agg_ext.assign(num_frames * 2, 0.0f);
                st->aggregate.num_values = num_frames;
                st->aggregate.population_mean = st->agg_mean.data();
                st->aggregate.population_var = st->agg_var.data();
                st->aggregate.population_ext = (float(*)[2])st->agg_ext.data();
                st->data.aggregate = &st->aggregate;
            }
            break;
        }
        st->data.values = st->values.data();
        st->data.num_values = st->values.size();
        st->data.counts = st->counts.empty() ? nullptr : st->counts.data();
        st->data.fingerprint = 1;
        if (st->ncounts) {
            if (!st->d_counts.ensure(st->ncounts)) return nullptr;
            if (hipMemsetAsync(st->d_counts.p, 0, st->ncounts * sizeof(uint64_t), e->stream) != hipSuccess) { vmd_fail("hipMemset failed");
                    return nullptr; }
        }
        e->props.push_back(std::move(st));
    }
    build_rdf_plan(e.get());
    if (!e->d_overflow.ensure(1) || hipMemsetAsync(e->d_overflow.p, 0, sizeof(uint32_t), e->stream) != hipSuccess ||
        pool_take(kPinned, (void**)&e->h_overflow, 2 * sizeof(uint32_t)) != hipSuccess) { vmd_fail("allocating the overflow flag failed");
                return nullptr; }
    e->h_overflow[0] = e->h_overflow[1] = 0;
    for (auto& ev : e->batch_done) if (!(ev = pool_event(false))) { vmd_fail("hipEventCreate failed"); return nullptr; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { vmd_fail("hipStreamSynchronize failed"); return nullptr; }
    return e.release();
}

extern "C" void vmd_eval_free(vmd_script_eval_t* eval) {
    if (!eval) return;
    VMD_STAGE("vmd_eval_free");
    lone_stop(eval);                     // the helper thread of a deferred-settle eval finishes what it is doing and ends
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(eval->device);
    {
        std::lock_guard<std::mutex> l(eval->mtx);
        // everything this eval ever queued ran on its own streams: once they are idle its blocks, streams and events can go back to
        // the process-wide cache without another synchronisation (PoolIdle)
        if (eval->stream) { (void)hipStreamSynchronize(eval->stream); }
        if (eval->copy_stream) { (void)hipStreamSynchronize(eval->copy_stream); }
        if (eval->aux_stream) { (void)hipStreamSynchronize(eval->aux_stream); }
        if (eval->pair_stream) { (void)hipStreamSynchronize(eval->pair_stream); }
        for (auto& ds : eval->decode_streams) { if (ds) { (void)hipStreamSynchronize(ds); pool_stream_give(ds, true); } ds = nullptr; }
        eval->decode_stream = nullptr;
        PoolIdle idle;
        for (auto& rs : eval->raw_slots) {
            if (rs.h) pool_give(rs.h);
            rs.h = nullptr;
            rs.d.release();
            pool_event_give(rs.uploaded, false);
            rs.uploaded = nullptr;
        }
        for (auto& st : eval->stages) {
            if (st.h) pool_give(st.h);
            st.h = nullptr;
            if (st.hraw) pool_give(st.hraw);
            st.hraw = nullptr;
            if (st.h_raw_status) pool_give(st.h_raw_status);
            st.h_raw_status = nullptr;
            st.d.release(); st.d_boxes.release(); st.d_bbox.release(); st.d_gboxes.release();
            st.d_raw.release(); st.d_raw_info.release(); st.d_raw_status.release(); st.d_raw_scratch.release();
            pool_event_give(st.ready, true);
            st.ready = nullptr;
        }
        pool_stream_give(eval->copy_stream, false); eval->copy_stream = nullptr;
        pool_stream_give(eval->aux_stream, false); eval->aux_stream = nullptr;
        pool_stream_give(eval->pair_stream, false); eval->pair_stream = nullptr;
        pool_event_give(eval->pair_fork, false); pool_event_give(eval->pair_join, false);
        eval->pair_fork = eval->pair_join = nullptr;
        eval->props.clear();
        eval->sels.clear();
        if (eval->h_overflow) pool_give(eval->h_overflow);
        eval->h_overflow = nullptr;
        if (eval->h_snap) pool_give(eval->h_snap);
        eval->h_snap = nullptr;
        for (auto& ev : eval->batch_done) { pool_event_give(ev, false); ev = nullptr; }
        eval->d_partial.release(); eval->d_partial2.release(); eval->d_pass.release(); eval->d_overflow.release();
                eval->d_pen_sample.release();
        pool_stream_give(eval->stream, false);
        eval->stream = nullptr;
    }
    {
        PoolIdle idle;              // whatever the destructors still hold (the profilers' events, the remaining buffers)
        delete eval;
    }
    (void)hipSetDevice(prev_dev);
}

extern "C" void vmd_eval_clear_data(vmd_script_eval_t* eval) {
    VMD_STAGE("vmd_eval_clear_data");
    if (!eval) return;
    lone_cancel(eval);                   // a deferred settle of the evaluation that ends here must neither start nor be running
    std::lock_guard<std::mutex> l(eval->mtx);
    eval->interrupt = false;
    for (size_t f = 0; f < eval->frame_mask.size(); ++f) mask_set(eval->frame_mask, f, 0);
    eval->frames_done = 0;
    eval->frames_computed = 0; eval->frames_reused = 0; eval->frames_device_decoded = 0; eval->frames_section_decoded = 0;
            eval->frames_mapped = 0;
    for (size_t b = 0; b < eval->num_blocks; ++b) eval->block_ready[b] = 0;
    eval->blocks_inst = TrajId();
    ra_reset(eval);
    for (auto& p : eval->props) {
        // The float view of a volume (8.4 MB, pinned) is NOT zeroed: `data.values` is pointed at a shared, read-only page range of zeros
        // until the next view of this evaluation has been written - k_counts_to_float rewrites every voxel of the real view, then the
        // pointer flips back (refresh_volume).  Round 6: the zeroing was a second 8.4 MB pass over PCIe per
        // evaluation - 0.15 ms that the kernel trace showed IN FRONT of the evaluation's kernels, not under them
        // (profiles/r06a_c4_1250_timeline.txt) - a fifth of a rank's 1 250-frame share of configs[3].  A reader polling `fingerprint`
        // (src/main.cpp:1508; density_volume.cpp:159-163, 279-283) sees zeros under the new fingerprint at once, never the previous run's
        // voxels; VIAMD dereferences prop_data->values when it uploads.
        if (p->prop.kind == PROP_SDF) {
            float* zeros = zero_volume_view(p->ncounts);
            if (zeros) pub(p->data.values, zeros);
            else {      // no address space for the shared zeros: zero the view itself, as before round 6 (never a NULL `values`)
                std::fill(p->values.begin(), p->values.end(), 0.0f);
                pub(p->data.values, p->values.data());
            }
        }
        else std::fill(p->values.begin(), p->values.end(), 0.0f);
        std::fill(p->weights.begin(), p->weights.end(), 0.0f);
        // the 17 MB u64 mirror of a volume is only ever read after vmd_eval_refresh_counts: mark it stale instead of zeroing it
        if (p->prop.kind == PROP_SDF) p->counts_stale = true;
        else std::fill(p->counts.begin(), p->counts.end(), (uint64_t)0);
        std::fill(p->weights64.begin(), p->weights64.end(), 0.0);
        std::fill(p->agg_mean.begin(), p->agg_mean.end(), 0.0f);
        std::fill(p->agg_var.begin(), p->agg_var.end(), 0.0f);
        std::fill(p->agg_ext.begin(), p->agg_ext.end(), 0.0f);
        if (p->ncounts) (void)hipMemsetAsync(p->d_counts.p, 0, p->ncounts * sizeof(uint64_t), eval->stream);
        pub(p->data.max_value, 0.0f); pub(p->data.min_value, 0.0f);
        pub(p->data.max_range[1], 0.0f);
        p->dirty = false;
        if (p->prop.kind != PROP_SDF) p->counts_stale = false;
        pub_touch(p->data.fingerprint);
    }
    (void)hipStreamSynchronize(eval->stream);
}

extern "C" void vmd_eval_interrupt(vmd_script_eval_t* eval) {
    if (!eval) return;
    eval->interrupt = true;
    // deferred-settle mode: a settle that is owed is dropped, one that is running ends at its next batch boundary - and has ended when this
    // returns: VIAMD resets the arena that holds molecule and trajectory right after interrupt_async_tasks (src/viamd.cpp:234-241, 624-630)
    if (eval->ra.lone.load()) lone_cancel(eval);
}

extern "C" uint64_t vmd_eval_ir_fingerprint(const vmd_script_eval_t* eval) { return eval ? eval->ir_fingerprint : 0; }

extern "C" const vmd_script_property_data_t* vmd_eval_property_data(const vmd_script_eval_t* eval, const char* name) {
    PropState* p = find_prop(eval, name);
    return p ? &p->data : nullptr;
}

extern "C" const uint8_t* vmd_eval_frame_mask(const vmd_script_eval_t* eval) { return eval ? eval->frame_mask.data() : nullptr; }

extern "C" size_t vmd_eval_frame_mask_bits(const vmd_script_eval_t* eval, uint64_t* words, size_t cap) {
    if (!eval) return 0;
    const size_t nw = (eval->num_frames + 63) / 64;
    for (size_t w = 0; w < nw && w < cap && words; ++w) {
        uint64_t v = 0;
        const size_t f1 = std::min(eval->num_frames, (w + 1) * 64);
        for (size_t f = w * 64; f < f1; ++f) if (mask_get(eval->frame_mask, f)) v |= 1ull << (f & 63);
        words[w] = v;
    }
    return nw;
}

extern "C" size_t vmd_eval_num_frames(const vmd_script_eval_t* eval) { return eval ? eval->num_frames : 0; }

extern "C" size_t vmd_eval_frames_done(const vmd_script_eval_t* eval) { return eval ? eval->frames_done.load() : 0; }

// ---- host views -------------------------------------------------------------------------------------------------
// the float views of a distribution from integer counts and fp64 weights (the device accumulators, or a snapshot of them)
void refresh_distribution_from(PropState* p, const uint64_t* counts, const double* weights64) {
    float ymax = 0.0f, vmax = 0.0f;
    for (size_t b = 0; b < p->ncounts; ++b) {
        p->counts[b] = counts[b];
        const float v = (float)counts[b];
        const float w = (float)weights64[b];
        p->values[b] = v; p->weights[b] = w;
        vmax = std::max(vmax, v);
        if (w > 0.0f) ymax = std::max(ymax, v / w);
    }
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub(p->data.min_range[1], 0.0f); pub(p->data.max_range[1], ymax);
    pub_touch(p->data.fingerprint);
}

bool refresh_distribution(vmd_script_eval_t* e, PropState* p) {
    HIP_OK(hipMemcpyAsync(p->counts.data(), p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    float ymax = 0.0f, vmax = 0.0f;
    for (size_t b = 0; b < p->ncounts; ++b) {
        const float v = (float)p->counts[b];
        const float w = (float)p->weights64[b];
        p->values[b] = v; p->weights[b] = w;
        vmax = std::max(vmax, v);
        if (w > 0.0f) ymax = std::max(ymax, v / w);
    }
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub(p->data.min_range[1], 0.0f); pub(p->data.max_range[1], ymax);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
    return true;
}

bool refresh_volume(vmd_script_eval_t* e, PropState* p) {
    VMD_STAGE("refresh_volume: counts -> float view, D2H");
    if (!p->d_max.ensure(1)) return false;
    float scale = 1.0f;
    if (e->spec.sdf_density) {
        // DECISION(D-SDF-NORM) flipped: number density per cubic Angstrom, averaged over the frames evaluated so far
        const double edge = 2.0 * (double)p->prop.rmax / (double)VMD_VOLUME_DIM;
        const size_t nf = e->frames_done.load();
        scale = nf ? (float)(1.0 / ((double)nf * edge * edge * edge)) : 0.0f;
    }
    float vmax = 0.0f;
    float* host_view_dev = nullptr;
    if (g_opt.sdf_direct_view.load() && p->values.pinned && hipHostGetDevicePointer((void**)&host_view_dev, p->values.data(),
            0) != hipSuccess) {
        (void)hipGetLastError();
        host_view_dev = nullptr;
    }
    if (host_view_dev) {
        // the conversion kernel writes the float view VIAMD reads straight into its pinned host pages (8.4 MB over PCIe at the
        // DMA's rate): no device-side copy of the view, no separate DMA behind the kernel
        KRN_OK(vmd_hip_counts_to_float(e->stream, p->d_counts.p, p->ncounts, host_view_dev, p->d_max.p, scale));
    } else {
        if (!p->d_values.ensure(p->ncounts)) return false;
        KRN_OK(vmd_hip_counts_to_float(e->stream, p->d_counts.p, p->ncounts, p->d_values.p, p->d_max.p, scale));
        HIP_OK(hipMemcpyAsync(p->values.data(), p->d_values.p, p->ncounts * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    }
    // the 17 MB u64 mirror behind `counts` is an extension VIAMD never reads: it is synchronised on demand
    // (vmd_eval_refresh_counts), only the float view travels after every range
    p->counts_stale = true;
    HIP_OK(hipMemcpyAsync(&vmax, p->d_max.p, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    // every voxel of the real view has just been rewritten: readers leave the shared zeros (clear_data)
    pub(p->data.values, p->values.data());
    pub(p->data.min_value, 0.0f); pub(p->data.max_value, vmax);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
    return true;
}

void refresh_temporal_stats(vmd_script_eval_t* e, PropState* p) {
    float lo = 3.4e38f, hi = -3.4e38f;
    bool any = false;
    for (size_t f = 0; f < e->num_frames; ++f) {
        if (!e->frame_mask[f]) continue;
        const float* row = &p->values[f * p->dim1];
        float rlo = row[0], rhi = row[0];
        double s = 0.0;
        for (size_t i = 0; i < p->dim1; ++i) { rlo = std::min(rlo, row[i]); rhi = std::max(rhi, row[i]); s += row[i]; }
        if (p->dim1 > 1) {
            const double mean = s / (double)p->dim1;
            double v = 0.0;
            for (size_t i = 0; i < p->dim1; ++i) { const double d = row[i] - mean; v += d * d; }
            p->agg_mean[f] = (float)mean;
            p->agg_var[f] = (float)std::sqrt(v / (double)p->dim1);   // VIAMD plots mean +- var as a band (src/main.cpp:1409-1424)
            p->agg_ext[2 * f] = rlo; p->agg_ext[2 * f + 1] = rhi;
        }
        lo = std::min(lo, rlo); hi = std::max(hi, rhi);
        any = true;
    }
    if (!any) { lo = hi = 0.0f; }
    pub(p->data.min_value, lo); pub(p->data.max_value, hi);
    pub(p->data.min_range[0], lo); pub(p->data.max_range[0], hi);
    pub_touch(p->data.fingerprint);
    p->dirty = false;
}

extern "C" bool vmd_eval_defer_volume_views(vmd_script_eval_t* eval, bool defer) {
    if (!eval) return vmd_fail("eval is NULL");
    eval->defer_volume_views.store(defer, std::memory_order_relaxed);
    return true;
}

extern "C" bool vmd_eval_finalize(vmd_script_eval_t* eval) {
    if (!eval) return vmd_fail("eval is NULL");
    if (!vmd_eval_wait_settled(eval)) return false;       // deferred-settle evals: the totals first
    std::lock_guard<std::mutex> l(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    for (auto& p : eval->props) {
        bool ok = true;
        if (p->prop.kind == PROP_RDF) ok = refresh_distribution(eval, p.get());
        else if (p->prop.kind == PROP_SDF) ok = refresh_volume(eval, p.get());
        else refresh_temporal_stats(eval, p.get());
        if (!ok) return false;
    }
    return true;
}

extern "C" bool vmd_eval_refresh_counts(vmd_script_eval_t* eval, const char* name) {
    PropState* p = find_prop(eval, name);
    if (!p) return vmd_fail("vmd_eval_refresh_counts: no property '%s'", name ? name : "(null)");
    std::lock_guard<std::mutex> l(eval->mtx);
    if (!p->counts_stale || !p->ncounts) return true;
    HIP_OK(hipSetDevice(eval->device));
    HIP_OK(hipMemcpyAsync(p->counts.data(), p->d_counts.p, p->ncounts * sizeof(uint64_t), hipMemcpyDeviceToHost, eval->stream));
    HIP_OK(hipStreamSynchronize(eval->stream));
    p->counts_stale = false;
    return true;
}

extern "C" bool vmd_eval_set_block_frames(vmd_script_eval_t* eval, size_t block_frames) {
    if (!eval) return vmd_fail("eval is NULL");
    std::lock_guard<std::mutex> l(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    if (eval->frames_done.load() != 0)
        return vmd_fail("vmd_eval_set_block_frames: call before the first frame_range or right after clear_data");
    // read-ahead re-engages on the new blocks
    eval->ra.on = false; eval->ra.own_blocks = false; eval->ra.blk_state.reset(); eval->ra.frame_req.reset();
    eval->block_frames = 0; eval->num_blocks = 0; eval->block_ready.reset();
    for (auto& p : eval->props) { p->d_blocks.release(); p->block_weights64.clear(); }
    if (block_frames == 0) return true;
    const size_t nblocks = (eval->num_frames + block_frames - 1) / block_frames;
    size_t bytes = 0;
    for (auto& p : eval->props) bytes += nblocks * p->ncounts * sizeof(uint64_t);
    if (bytes > ((size_t)96 << 30))
        return vmd_fail("vmd_eval_set_block_frames: %zu blocks need %.1f GB of block partials; use larger blocks", nblocks,
                (double)bytes / 1073741824.0);
    for (auto& p : eval->props) {
        if (!p->ncounts) continue;
        if (!p->d_blocks.ensure(nblocks * p->ncounts)) return false;
        if (p->prop.kind == PROP_RDF) p->block_weights64.assign(nblocks * p->ncounts, 0.0);
    }
    eval->block_ready.reset(new std::atomic<uint8_t>[nblocks]);
    for (size_t b = 0; b < nblocks; ++b) eval->block_ready[b] = 0;
    eval->num_blocks = nblocks;
    eval->block_frames = block_frames;
    return true;
}

extern "C" bool vmd_eval_set_source(vmd_script_eval_t* eval, vmd_script_eval_t* source) {
    if (!eval) return vmd_fail("eval is NULL");
    std::lock_guard<std::mutex> l(eval->mtx);
    if (!source) { eval->source = nullptr; return true; }
    if (source == eval) return vmd_fail("vmd_eval_set_source: an eval cannot be its own source");
    if (source->ir_fingerprint != eval->ir_fingerprint || source->num_frames != eval->num_frames
            || source->props.size() != eval->props.size())
        return vmd_fail("vmd_eval_set_source: source was created from a different script or frame count");
    if (source->device != eval->device) return vmd_fail("vmd_eval_set_source: source lives on another device");
    // (a source without block partials is accepted since round 4: read-ahead gives an eval driven by pool threads block partials of its own
    // accord, and whether the source has any is looked up, under its mutex, whenever a range is served)
    eval->source = source;
    return true;
}

extern "C" size_t vmd_eval_frames_device_decoded(const vmd_script_eval_t* eval) { return eval ? eval->frames_device_decoded.load() : 0; }

extern "C" size_t vmd_eval_frames_section_decoded(const vmd_script_eval_t* eval) { return eval ? eval->frames_section_decoded.load() : 0; }

extern "C" size_t vmd_eval_frames_mapped(const vmd_script_eval_t* eval) { return eval ? eval->frames_mapped.load() : 0; }

extern "C" void vmd_eval_cell_build_stats(const vmd_script_eval_t* eval, size_t* bucket_overflows, size_t* selections_off_buckets) {
    size_t ov = 0, off = 0;
    if (eval) for (auto& s : eval->sels) { ov += (size_t)std::min(s->overflows, 98); off += s->overflows >= 3 ? 1 : 0; }
    if (bucket_overflows) *bucket_overflows = ov;
    if (selections_off_buckets) *selections_off_buckets = off;
}

extern "C" void vmd_eval_frame_stats(const vmd_script_eval_t* eval, size_t* frames_computed, size_t* frames_reused) {
    if (frames_computed) *frames_computed = eval ? eval->frames_computed.load() : 0;
    if (frames_reused) *frames_reused = eval ? eval->frames_reused.load() : 0;
}

extern "C" void vmd_eval_set_frame_mask(vmd_script_eval_t* eval, const uint8_t* mask, size_t n) {
    if (!eval || !mask) return;
    std::lock_guard<std::mutex> l(eval->mtx);
    size_t done = 0;
    for (size_t f = 0; f < eval->num_frames; ++f) {
        if (f < n) mask_set(eval->frame_mask, f, mask[f] ? 1 : 0);
        done += eval->frame_mask[f] ? 1 : 0;
    }
    eval->frames_done = done;
}

extern "C" size_t vmd_eval_accum_views(vmd_script_eval_t* eval, vmd_accum_view_t* out, size_t cap) {
    if (!eval) return 0;
    size_t n = 0;
    for (auto& p : eval->props) {
        if (n < cap && out) {
            vmd_accum_view_t v;
            memset(&v, 0, sizeof(v));
            v.name = p->prop.name.c_str();
            v.flags = p->prop.flags;
            if (p->ncounts) { v.counts_dev = p->d_counts.p; v.num_counts = p->ncounts; }
            // a voxel receives at most one count per (frame, structure, target atom)
            if (p->prop.kind == PROP_SDF) {
                const long double b = (long double)eval->num_frames * (long double)p->prop.K * (long double)p->prop.b.size();
                v.count_bound = b < 1.8e19L ? (uint64_t)b : 0;
            }
            if (!p->weights64.empty()) { v.weights64 = p->weights64.data(); v.num_weights = p->weights64.size(); }
            if (p->prop.kind == PROP_DIST) { v.temporal = p->values.data(); v.num_temporal = p->values.size(); }
            out[n] = v;
        }
        n += 1;
    }
    return n;
}

// hooks for vmd_reduce.cpp (same library, not part of the public headers)
extern "C" int vmd_eval_internal_device(const vmd_script_eval_t* eval) { return eval ? eval->device : 0; }

extern "C" void vmd_eval_internal_lock(vmd_script_eval_t* eval, int lock) { if (eval) { if (lock) eval->mtx.lock();
        else eval->mtx.unlock(); } }

extern "C" vmd_reduce_stats_t* vmd_eval_internal_reduce_stats(vmd_script_eval_t* eval) { return eval ? &eval->reduce_stats : nullptr; }

extern "C" void vmd_eval_reduce_stats(const vmd_script_eval_t* eval, vmd_reduce_stats_t* out) {
    if (!out) return;
    if (eval) *out = eval->reduce_stats; else memset(out, 0, sizeof(*out));
}

// ---- the hot call -----------------------------------------------------------------------------------------------
bool upload_static(vmd_script_eval_t* e, const vmd_system_t* sys, size_t traj_atoms) {
    for (auto& s : e->sels) {
        if (!s->d_idx.p) { if (!s->d_idx.upload(s->idx.data(), s->idx.size(), e->stream)) return false; }
    }
    for (auto& p : e->props) {
        if (p->uploaded) continue;
        const Property& d = p->prop;
        auto masses = [&](const std::vector<int32_t>& idx, std::vector<float>& out) {
            out.resize(idx.size());
            for (size_t i = 0; i < idx.size(); ++i)
                out[i] = (sys && sys->mass && (size_t)idx[i] < sys->atom_count) ? sys->mass[idx[i]] : 1.0f;
            if (d.kind == PROP_DIST && e->spec.dist_geometric_com) std::fill(out.begin(), out.end(), 1.0f);   // D-DIST-COM flipped
        };
        std::vector<float> tmp;
        if (d.kind == PROP_SDF) {
            if (!p->d_structs.upload(d.a.data(), d.a.size(), e->stream)) return false;
            if (!p->d_tgt.upload(d.b.data(), d.b.size(), e->stream)) return false;
            masses(d.a, tmp);
            if (!p->d_mass.upload(tmp.data(), tmp.size(), e->stream)) return false;
            if (!p->d_ref_pose.ensure(d.m * 3)) return false;
            p->have_tree = false;
            if (sys && sys->bonds && sys->bond_count) {
                // D-SDF-UNWRAP with bonds: breadth-first from local atom 0 over the bonds among the structure's atoms, neighbours in
                // increasing local index; atoms the walk does not reach hang on their index predecessor (oracle: vo_bond_tree)
                std::vector<int32_t> order(d.K * d.m), parent(d.K * d.m);
                std::map<int32_t, std::vector<int32_t>> adj;            // only atoms of reference structures matter
                std::map<int32_t, char> member;
                for (int32_t a : d.a) member[a] = 1;
                for (size_t b = 0; b < sys->bond_count; ++b) {
                    const int32_t i = sys->bonds[b][0], j = sys->bonds[b][1];
                    if (member.count(i) && member.count(j)) { adj[i].push_back(j); adj[j].push_back(i); }
                }
                for (size_t k = 0; k < d.K; ++k) {
                    const int32_t* idx = &d.a[k * d.m];
                    int32_t* ord = &order[k * d.m];
                    int32_t* par = &parent[k * d.m];
                    std::map<int32_t, int32_t> local;
                    for (size_t a = 0; a < d.m; ++a) local.emplace(idx[a], (int32_t)a);
                    std::vector<char> seen(d.m, 0);
                    size_t head = 0, tail = 0;
                    ord[tail++] = 0; seen[0] = 1; par[0] = -1;
                    while (head < tail) {
                        const int32_t a = ord[head++];
                        std::vector<int32_t> nb;
                        auto it = adj.find(idx[a]);
                        if (it != adj.end()) for (int32_t g : it->second) { auto l = local.find(g);
                                if (l != local.end()) nb.push_back(l->second); }
                        std::sort(nb.begin(), nb.end());
                        for (int32_t c : nb) if (!seen[c]) { seen[c] = 1; par[c] = a; ord[tail++] = c; }
                    }
                    for (size_t a = 1; a < d.m; ++a) if (!seen[a]) { par[a] = (int32_t)a - 1; ord[tail++] = (int32_t)a; }
                }
                if (!p->d_tree_order.upload(order.data(), order.size(), e->stream) || !p->d_tree_parent.upload(parent.data(),
                        parent.size(), e->stream)) return false;
                HIP_OK(hipStreamSynchronize(e->stream));                 // the vectors go out of scope
                p->have_tree = true;
            }
            // owner[t]: the structure target t is a member of (exclusion rule); only valid when memberships are unique
            std::vector<int8_t> owner(d.b.size(), (int8_t)-1);
            bool unique = d.K <= 127;
            if (unique) {
                std::map<int32_t, int> where;
                for (size_t k = 0; k < d.K && unique; ++k)
                    for (size_t a = 0; a < d.m; ++a) {
                        auto it = where.find(d.a[k * d.m + a]);
                        if (it != where.end() && it->second != (int)k) { unique = false; break; }
                        where[d.a[k * d.m + a]] = (int)k;
                    }
                if (unique) for (size_t t = 0; t < d.b.size(); ++t) { auto it = where.find(d.b[t]);
                        if (it != where.end()) owner[t] = (int8_t)it->second; }
            }
            p->have_owner = unique;
            if (unique && !p->d_owner.upload(owner.data(), owner.size(), e->stream)) return false;
            p->unowned = unique && std::all_of(owner.begin(), owner.end(), [](int8_t o) { return o < 0; });
            // an arithmetic progression (every water oxygen of a regular solvent box: first + 3 t) needs no index list on the device
            p->tgt_first = d.b[0]; p->tgt_stride = 0;
            if (d.b.size() >= 2 && d.b[1] > d.b[0] && g_opt.sdf_arith != 0) {
                const int64_t st = (int64_t)d.b[1] - d.b[0];
                bool ok = true;
                for (size_t t = 2; t < d.b.size() && ok; ++t) ok = (int64_t)d.b[t] - d.b[t - 1] == st;
                if (ok && st < (1 << 20)) p->tgt_stride = (int)st;
            }
            // dense targets: stream whole frames and select by a per-atom tag instead of gathering through the index list
            // sized from the TRAJECTORY's atom count (the target indices were validated against it, check_atoms), never from
            // sys->atom_count, which a host may leave unset or out of step
            const size_t natoms = traj_atoms;
            p->have_tag = unique && d.K <= 253 && natoms > 0 && d.b.size() * 8 >= natoms && g_opt.sdf_dense != 0;
            if (p->have_tag) {
                p->tag_len = (natoms + 63) & ~(size_t)63;
                std::vector<uint8_t> tag(p->tag_len, (uint8_t)255);
                for (size_t t = 0; t < d.b.size(); ++t) tag[d.b[t]] = owner[t] < 0 ? (uint8_t)254 : (uint8_t)owner[t];
                if (!p->d_tag.upload(tag.data(), tag.size(), e->stream)) return false;
                HIP_OK(hipStreamSynchronize(e->stream));
            }
            HIP_OK(hipStreamSynchronize(e->stream));
        } else if (d.kind == PROP_DIST) {
            if (!p->d_a.upload(d.a.data(), d.a.size(), e->stream)) return false;
            if (!p->d_b.upload(d.b.data(), d.b.size(), e->stream)) return false;
            if (!p->d_aoff.upload(d.aoff.data(), d.aoff.size(), e->stream)) return false;
            if (!p->d_boff.upload(d.boff.data(), d.boff.size(), e->stream)) return false;
            masses(d.a, tmp);
            if (!p->d_ma.upload(tmp.data(), tmp.size(), e->stream)) return false;
            masses(d.b, tmp);
            if (!p->d_mb.upload(tmp.data(), tmp.size(), e->stream)) return false;
        }
        HIP_OK(hipStreamSynchronize(e->stream));   // tmp goes out of scope
        p->uploaded = true;
    }
    return true;
}

extern "C" const int32_t* vmd_eval_sdf_structures(const vmd_script_eval_t* eval, const char* name, size_t* num_structures,
        size_t* atoms_per_structure) {
    PropState* p = find_prop(eval, name);
    if (!p || p->prop.kind != PROP_SDF) { vmd_fail("'%s' is not an sdf property", name ? name : "(null)"); return nullptr; }
    if (num_structures) *num_structures = p->prop.K;
    if (atoms_per_structure) *atoms_per_structure = p->prop.m;
    return p->prop.a.data();
}

extern "C" bool vmd_eval_sdf_matrices(vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys,
                                      vmd_trajectory_i* traj, uint32_t frame, float* matrices, size_t* K_out, float* extent_out) {
    if (!eval || !traj) return vmd_fail("vmd_eval_sdf_matrices: NULL argument");
    PropState* p = find_prop(eval, name);
    if (!p || p->prop.kind != PROP_SDF) return vmd_fail("'%s' is not an sdf property", name ? name : "(null)");
    std::lock_guard<std::mutex> lock(eval->mtx);
    HIP_OK(hipSetDevice(eval->device));
    vmd_script_eval_t* e = eval;
    const size_t num_atoms = traj->num_atoms(traj->inst);
    if (!check_atoms(e, num_atoms) || !upload_static(e, sys, num_atoms)) return false;
    vmd_device_view_t view;
    memset(&view, 0, sizeof(view));
    const bool have_view = traj->device_view && traj->device_view(traj->inst, &view) && view.device == e->device;
    BatchSrc src;
    if (!p->ref_pose_ready) {
        if (!fetch_batch(e, traj, view_holds(have_view, view, 0) ? &view : nullptr, num_atoms, 0, 1, &src)) return false;
        KRN_OK(vmd_hip_sdf_ref_pose(e->stream, src.base, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), p->d_structs.p,
                p->d_mass.p, (int)p->prop.m, p->d_ref_pose.p, p->have_tree ? p->d_tree_order.p : nullptr, p->have_tree
                ? p->d_tree_parent.p : nullptr));
        HIP_OK(hipStreamSynchronize(e->stream));
        p->ref_pose_ready = true;
    }
    if (!fetch_batch(e, traj, view_holds(have_view, view, frame) ? &view : nullptr, num_atoms, frame, 1, &src)) return false;
    const size_t K = p->prop.K;
    DevBuf<double> dM;
    if (!dM.ensure(K * 12) || !p->d_R32.ensure(K * 9) || !p->d_c32.ensure(K * 3)) return false;
    if (p->have_tree && !p->d_tree_pos.ensure(K * p->prop.m * 3)) return false;
    KRN_OK(vmd_hip_sdf_align(e->stream, src.base, src.frame_stride, src.row_stride, e->stages[0].d_boxes.p, batch_pbc(e->stages[0]), 1,
            p->d_structs.p, p->d_mass.p, (int)K, (int)p->prop.m, p->d_ref_pose.p, p->d_R32.p, p->d_c32.p, dM.p, nullptr, p->have_tree
            ? p->d_tree_order.p : nullptr, p->have_tree ? p->d_tree_parent.p : nullptr, p->have_tree ? p->d_tree_pos.p : nullptr));
    std::vector<double> M(K * 12);
    HIP_OK(hipMemcpyAsync(M.data(), dM.p, K * 12 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (matrices) {
        for (size_t k = 0; k < K; ++k) {
            float* o = matrices + 16 * k;   // column-major mat4
            const double* r = &M[12 * k];
            for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 3; ++rr) o[4 * c + rr] = (float)r[4 * rr + c];
            o[3] = 0.0f; o[7] = 0.0f; o[11] = 0.0f; o[15] = 1.0f;
        }
    }
    if (K_out) *K_out = K;
    if (extent_out) *extent_out = p->prop.rmax;
    return true;
}
