"""Shared parity scenarios: the same checks run against the SIMT emulator build (CPU, small sizes, `-m "not gpu"`)
and against the hipcc-built product library on a real MI355X (`-m gpu`).  The checker is always the oracle."""
import time

import ctypes as C

import numpy as np

import viamd_amd as V
from viamd_amd import _lib as L


def water_box(O, seed, n_atoms, box, frames, sigma=0.05):
    """[F,3,N] seeded synthetic O,H,H box (oracle S9)."""
    return np.stack([O.synth_frame(seed, n_atoms, box, sigma, f) for f in range(frames)])


def host_frames(O, seed, n_atoms, box, frames, n_blob=0, sigma=0.05):
    """The synthetic system of viamd_amd.synth on the host (oracle generator for the waters), float32 [F, 3, n_atoms]."""
    from viamd_amd import synth
    out = np.stack([O.synth_frame(seed, n_atoms, box, sigma, f, n_blob=n_blob) for f in range(frames)])
    if n_blob:
        for f0, xyz in synth.blob_trajectory(seed, n_blob, box, frames):
            out[f0:f0 + xyz.shape[0], :, :n_blob] = xyz
    return out


def oxygen(n_atoms):
    return np.arange(0, n_atoms, 3, dtype=np.int32)


def hydrogen(n_atoms):
    return np.array([i for i in range(n_atoms) if i % 3], dtype=np.int32)


def cell_pair(O, box, flags=L.PBC_ALL):
    """(oracle cell, product unitcell) for the same box; box = (x, y, z, xy, xz, yz) is a triclinic cell"""
    tilt = (0.0, 0.0, 0.0)
    if box is not None and not np.isscalar(box) and len(box) == 6:
        box, tilt = tuple(box[:3]), tuple(box[3:])
    return O.make_cell(box, flags if box is not None else 0, tilt), V.make_unitcell(box, flags, tilt)


def oracle_rdf(O, coords, ocell, ref, tgt, rmin, rmax, frames=None, method="cells"):
    counts = np.zeros(1024, np.uint64)
    weights = np.zeros(1024, np.float64)
    frames = range(coords.shape[0]) if frames is None else frames
    for f in frames:
        oc = ocell[f] if isinstance(ocell, list) else ocell
        O.rdf_frame(coords[f, 0], coords[f, 1], coords[f, 2], oc, ref, tgt, rmin, rmax, counts=counts, method=method)
        O.rdf_weights(oc, len(ref), len(tgt), rmin, rmax, weights=weights)
    return counts, weights


def make_traj(lib, coords, vcell, device):
    if device == "pinned":
        t = V.PinnedHostTrajectory(coords.shape[0], coords.shape[2], lib=lib)
        t.upload(coords, vcell)
        return t
    if device:
        t = V.DeviceTrajectory(coords.shape[0], coords.shape[2], lib=lib)
        t.upload(coords, vcell)
        return t
    return V.HostTrajectory(coords, vcell)


def check_rdf(lib, O, coords, box, props, flags=L.PBC_ALL, device=False, ranges=None, variant=None, oracle_method="cells"):
    """props: list of (name, ref, tgt, rmin, rmax).  Bit-exact counts, 1e-12 weights, 1e-5 normalised g(r).
    box may be a list with one box per frame (NPT trajectories)."""
    F, _, N = coords.shape
    if isinstance(box, list):
        pairs = [cell_pair(O, b, flags) for b in box]
        ocell, vcell = [p[0] for p in pairs], [p[1] for p in pairs]
    else:
        ocell, vcell = cell_pair(O, box, flags)
    ir = V.ScriptIR(lib)
    for name, ref, tgt, rmin, rmax in props:
        ir.add_rdf(name, ref, tgt, (rmin, rmax))
    ev = V.ScriptEval(F, ir)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, unitcell=vcell[0] if isinstance(vcell, list) else vcell)
    old = lib.vmd_set_option(b"rdf_variant", variant) if variant is not None else None      # None: whatever the library is set to
    try:
        for beg, end in (ranges or [(0, F)]):
            assert ev.frame_range(sysm, traj, beg, end)
    finally:
        if variant is not None:
            lib.vmd_set_option(b"rdf_variant", old)
    assert ev.frame_mask().all() and ev.frames_done() == F
    for name, ref, tgt, rmin, rmax in props:
        pd = ev.property_data(name)
        counts, weights = oracle_rdf(O, coords, ocell, ref, tgt, rmin, rmax, method=oracle_method)
        assert pd.dim[2] == 1024
        np.testing.assert_array_equal(pd.counts, counts, err_msg=f"{name}: integer histogram differs from the oracle")
        np.testing.assert_allclose(pd.weights64, weights, rtol=1e-12)
        np.testing.assert_array_equal(pd.values, counts.astype(np.float32))
        g_dev = V.downsample_histogram(pd.values, pd.weights, 128, lib=lib)
        g_ref = O.downsample_histogram(counts.astype(np.float32), weights.astype(np.float32), 128)
        np.testing.assert_allclose(g_dev, g_ref, rtol=1e-5, atol=0)   # tolerance stated by BASELINE.json north_star
        assert pd.min_range[0] == np.float32(rmin) and pd.max_range[0] == np.float32(rmax)
    return ev


def rdf_edge_cases(lib, O, device=False):
    """Inputs the reference's spatial hash has to survive: atoms exactly on cell faces / box faces, coincident atoms, far
    outside the cell, a single pair, selections of one atom, atom counts that are not a multiple of the wave size."""
    rng = np.random.default_rng(17)
    box = 50.0
    n = 1237                                              # not a multiple of 64
    c = rng.uniform(0, box, (2, 3, n)).astype(np.float32)
    c[:, :, 0] = 0.0                                      # on the origin
    c[:, 0, 1] = box                                      # exactly on the far face (wraps to 0)
    c[:, :, 2] = c[:, :, 3]                               # two coincident atoms: d = 0 never counts (open interval)
    c[:, :, 4] = np.float32(-0.0)
    c[:, 0, 5] = np.nextafter(np.float32(box), np.float32(0))
    c[:, :, 6] = c[:, :, 7] + np.float32(1e6 * box)      # far outside the cell: same image as atom 7 after wrapping
    c[:, 1, 8] = 12.5 * 2                                 # exactly on a pencil boundary (ny = 4 -> edge 12.5)
    c[:, 2, 9] = 12.5 * 3
    c[0, :, 10:40] = c[0, :, 40:70] + np.float32(12.0)   # pairs at (about) the cutoff along every axis
    everything = np.arange(n)
    check_rdf(lib, O, c, box, [("all", everything, everything, 0.0, 12.0), ("one", [11], everything, 0.0, 12.0),
                               ("to_one", everything, [12], 0.5, 11.0), ("pair", [2], [3], 0.0, 5.0),
                               ("shell", everything[::2], everything[1::2], 11.9, 12.0)], device=device)
    # a selection that leaves most pencils empty, and a target far denser than the reference
    c2 = rng.uniform(0, box, (1, 3, 4000)).astype(np.float32)
    c2[:, :, :50] = rng.uniform(20, 24, (1, 3, 50))
    check_rdf(lib, O, c2, box, [("blob", np.arange(50), np.arange(50), 0.0, 10.0), ("blob_all", np.arange(50), np.arange(4000), 0.0, 12.0)],
              device=device)


def cell_build_cases(lib, O, coords, box, device=False):
    """every build of the cell-sorted copies: two-level (pencil buckets, the default), atomic 3-kernel build, LDS-fused
    single-block build, split build (G blocks per frame, forced with 2048-atom slices)"""
    n = coords.shape[2]
    o, h = oxygen(n), hydrogen(n)
    # rec3: 12- / 16-byte bucket records; rec3 = 2, 3: the same with level 1 writing one scattered record per lane (no block-local sort)
    for pencil, fused, split, rec3 in ((1, 1, 1, 1), (1, 1, 1, 0), (1, 1, 1, 2), (1, 1, 1, 3), (0, 0, 1, 1), (0, 1, 1, 1), (0, 1, 2, 1)):
        old = (lib.vmd_set_option(b"cells_pencil", pencil), lib.vmd_set_option(b"cells_fused", fused), lib.vmd_set_option(b"cells_split", split),
               lib.vmd_set_option(b"cells_rec3", rec3 & 1))
        old_lds = lib.vmd_set_option(b"cells_bin_lds", 0 if rec3 >= 2 else 1)
        old_small = lib.vmd_set_option(b"cells_small", 0)          # every build path on this (small) box, the bucket ones included
        try:
            check_rdf(lib, O, coords[:2], box, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 9.0)], device=device)
        finally:
            lib.vmd_set_option(b"cells_small", old_small)
            lib.vmd_set_option(b"cells_pencil", old[0]); lib.vmd_set_option(b"cells_fused", old[1]); lib.vmd_set_option(b"cells_split", old[2])
            lib.vmd_set_option(b"cells_rec3", old[3]); lib.vmd_set_option(b"cells_bin_lds", old_lds)
    # round 6: a periodic index list is computed by the build kernels, not read (cells_sel_pattern): the O of every water (m = 1, period 3),
    # the two H (m = 2), every atom (period 1), a periodic list that starts late, one that is NOT periodic, one that is periodic but for its
    # last entry - with the switch on and off, through the bucket build and the single-level ones
    rng = np.random.default_rng(11)
    ragged = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
    late = o[37:]
    almost = np.concatenate([o[:-1], [o[-1] + 1]]).astype(np.int32)
    every = np.arange(n, dtype=np.int32)
    old_small = lib.vmd_set_option(b"cells_small", 0)
    try:
        for pattern in (1, 0):
            for pencil in (1, 0):
                old = (lib.vmd_set_option(b"cells_sel_pattern", pattern), lib.vmd_set_option(b"cells_pencil", pencil))
                try:
                    check_rdf(lib, O, coords[:2], box, [("a", every, o, 0.0, 7.0), ("b", late, h, 0.0, 9.0), ("c", ragged, ragged, 0.0, 8.0),
                                                        ("d", almost, almost, 0.0, 8.0)], device=device)
                finally:
                    lib.vmd_set_option(b"cells_sel_pattern", old[0]); lib.vmd_set_option(b"cells_pencil", old[1])
    finally:
        lib.vmd_set_option(b"cells_small", old_small)


def cell_build_overflow_case(lib, O, device=False, n=3000, box=60.0):
    """(small selections are normally sorted by one block per frame, without buckets: the case asks for buckets - cells_small = 0)"""
    # cells_cap_sample = 2: capacities from the batch's beginning and end only - the middle frames of these cases are what overflows (the
    # default also looks at the middle of a batch, round 6)
    old = lib.vmd_set_option(b"cells_small", 0), lib.vmd_set_option(b"cells_cap_sample", 2)
    try:
        _cell_build_overflow_case(lib, O, device, n, box)
    finally:
        lib.vmd_set_option(b"cells_small", old[0]); lib.vmd_set_option(b"cells_cap_sample", old[1])


def _cell_build_overflow_case(lib, O, device=False, n=3000, box=60.0):
    """The two-level build sizes its pencil buckets from the first and last frames of a batch.  Here the middle frames pile
    every oxygen into one corner of the cell: their buckets overflow, nothing may reach the histograms (device flag), and the
    evaluator has to re-measure and repeat the batch - the result still equals the oracle bit for bit."""
    F = 12
    coords = water_box(O, 5, n, box, F)
    o = oxygen(n)
    rng = np.random.default_rng(3)
    for f in (5, 6, 7):
        coords[f][:, o] = rng.uniform(1.0, 11.0, (3, o.size)).astype(np.float32)      # all inside one 12 A pencil
    import ctypes as C
    lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
    try:
        check_rdf(lib, O, coords, box, [("goo", o, o, 0.0, 12.0)], device=device)
    finally:
        lib.vmd_profile_enable(False)
    nb = C.c_uint64(0)
    lib.vmd_profile_ms(b"cells_build", C.byref(nb))
    assert nb.value >= 2, "the overflowing batch was not rebuilt"
    # several batches: the next batch is queued before the host looks at the overflowing one (deferred completion) - it saw the flag
    # as well and has to repeat its RDF part when its turn comes; with and without the deferral, with the overflow in the middle
    # batch and in the last one
    for bf, defer in (((4, 1), (3, 1), (5, 1), (4, 0)) if device else ((4, 1), (4, 0))):     # the emulator takes the two that matter
        old = (lib.vmd_set_option(b"batch_frames", bf), lib.vmd_set_option(b"defer_sync", defer))
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        try:
            check_rdf(lib, O, coords, box, [("goo", o, o, 0.0, 12.0)], device=device)
            if device or defer:
                check_rdf(lib, O, coords[::-1].copy(), box, [("goo", o, o, 0.0, 12.0)], device=device)
        finally:
            lib.vmd_profile_enable(False)
            lib.vmd_set_option(b"batch_frames", old[0]); lib.vmd_set_option(b"defer_sync", old[1])
        lib.vmd_profile_ms(b"cells_build", C.byref(nb))
        assert nb.value > (2 if device or defer else 1) * ((F + bf - 1) // bf), "no batch was rebuilt"
    # the overflow happens in the build of a LATER pass (the hydrogens of the second property, the third range): the passes that
    # ran before it must not survive into the repeated batch (all-or-nothing commit at the end of the batch)
    coords = water_box(O, 6, n, box, F)
    h = hydrogen(n)[: o.size]
    for f in (4, 5, 6):
        coords[f][:, h] = rng.uniform(30.0, 41.0, (3, h.size)).astype(np.float32)
    old = lib.vmd_set_option(b"rdf_classes", 0)
    try:
        check_rdf(lib, O, coords, box, [("goo", o, o, 0.0, 12.0), ("ghh", h, h, 0.0, 12.0), ("short", o, o, 0.0, 7.0)], device=device)
    finally:
        lib.vmd_set_option(b"rdf_classes", old)


def wandering_solute_case(lib, O, device=False, n=3000, box=60.0, F=12, batch=3):
    """BASELINE config 5 in small: water + a compact solute that drifts through the cell while the trajectory goes on.  The solute's
    pencil buckets, measured on one batch, overflow in the next; that must neither change a count nor cost the water selections their
    two-level build (round 4: one overflow flag for all selections did - 49 instead of 21 ms of cell build per c5 step).  Default: the
    small selection is sorted by one block per frame and nothing overflows; with buckets forced on it, only IT pays."""
    coords = water_box(O, 9, n, box, F)
    nb = 90
    rng = np.random.default_rng(5)
    shape = rng.normal(0.0, 2.5, (3, nb)).astype(np.float32)
    for f in range(F):
        centre = np.array([[6.0 + 4.4 * f], [7.0 + 3.1 * f], [30.0]], np.float32)      # about a pencil further every third frame
        coords[f][:, :nb] = np.mod(shape + centre, np.float32(box))
    solute = np.arange(nb, dtype=np.int32)
    o = oxygen(n)
    o = o[o >= nb]
    props = [("gss", solute, solute, 0.0, 9.0), ("gso", solute, o, 0.0, 12.0), ("goo", o, o, 0.0, 12.0)]
    ocell, vcell = cell_pair(O, box)
    want = {nm: oracle_rdf(O, coords, ocell, a, b, r0, r1)[0] for nm, a, b, r0, r1 in props}
    for small, classes in ((8192, 1), (0, 1), (0, 0)):
        old = (lib.vmd_set_option(b"cells_small", small), lib.vmd_set_option(b"batch_frames", batch), lib.vmd_set_option(b"rdf_classes", classes))
        try:
            ir = V.ScriptIR(lib)
            for nm, a, b, r0, r1 in props:
                ir.add_rdf(nm, a, b, (r0, r1))
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(V.MolSystem(n, unitcell=vcell), make_traj(lib, coords, vcell, device), 0, F)
            for nm in want:
                np.testing.assert_array_equal(ev.property_data(nm).counts, want[nm], err_msg=f"{nm} (cells_small {small}, classes {classes})")
            overflows, off = ev.cell_build_stats()
            if small:
                assert overflows == 0 and off == 0, (overflows, off)
            else:
                assert overflows >= 1 and off <= 1, (overflows, off)        # the solute's buckets overflow; at most IT leaves the two-level build
            ev.close()
        finally:
            lib.vmd_set_option(b"cells_small", old[0]); lib.vmd_set_option(b"batch_frames", old[1]); lib.vmd_set_option(b"rdf_classes", old[2])


def blocks_overflow_case(lib, O, device=False, n=3000, box=60.0):
    old = lib.vmd_set_option(b"cells_small", 0), lib.vmd_set_option(b"cells_cap_sample", 2)
    try:
        _blocks_overflow_case(lib, O, device, n, box)
    finally:
        lib.vmd_set_option(b"cells_small", old[0]); lib.vmd_set_option(b"cells_cap_sample", old[1])


def _blocks_overflow_case(lib, O, device=False, n=3000, box=60.0):
    """A full evaluation that keeps block partials evaluates consecutive frame blocks as ONE batch (one cell build, a pair launch per
    block, every other one on a second stream).  Here a bucket overflows in the middle blocks: nothing of the batch may reach any
    partial, the batch is repeated, and afterwards the totals, every block (through filtered sub-ranges that reuse them) and both
    ways of planning the batches give the oracle's integers."""
    F, S = 12, 3
    coords = water_box(O, 5, n, box, F)
    o = oxygen(n)
    rng = np.random.default_rng(3)
    for f in (5, 6, 7):
        coords[f][:, o] = rng.uniform(1.0, 11.0, (3, o.size)).astype(np.float32)      # all inside one 12 A pencil
    ocell, vcell = cell_pair(O, box)
    ir = V.ScriptIR(lib); ir.add_rdf("goo", o, o, (0.0, 12.0))
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(coords.shape[2], unitcell=vcell)
    per_frame = [oracle_rdf(O, coords, ocell, o, o, 0.0, 12.0, frames=[f])[0] for f in range(F)]
    import ctypes as C
    for super_, two in ((1, 1), (1, 0), (0, 0)):
        old = (lib.vmd_set_option(b"block_superbatch", super_), lib.vmd_set_option(b"block_two_streams", two))
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        try:
            full = V.ScriptEval(F, ir); full.set_block_frames(S)
            assert full.frame_range(sysm, traj, 0, F)
            nb = C.c_uint64(0)
            lib.vmd_profile_ms(b"cells_build", C.byref(nb))
            assert nb.value >= 2, "the overflowing batch was not rebuilt"
            np.testing.assert_array_equal(full.property_data("goo").counts, sum(per_frame))
            filt = V.ScriptEval(F, ir); filt.set_source(full)
            for beg, end in ((3, 9), (0, 6), (6, 12), (4, 11)):
                filt.clear_data()
                assert filt.frame_range(sysm, traj, beg, end)
                np.testing.assert_array_equal(filt.property_data("goo").counts, sum(per_frame[beg:end]), err_msg=f"[{beg},{end}) super={super_} two={two}")
            filt.clear_data()
            assert filt.frame_range(sysm, traj, 3, 9) and filt.frame_stats() == (0, 6)
        finally:
            lib.vmd_profile_enable(False)
            lib.vmd_set_option(b"block_superbatch", old[0]); lib.vmd_set_option(b"block_two_streams", old[1])


def pool_threads_case(lib, O, device=False, n_water=900, box=32.0, F=12, nthreads=5, combos=((150, 1, 1), (0, 1, 3), (0, 0, 2))):
    """frame_range the way VIAMD calls it (src/main.cpp:993-997): pool threads pull ranges of 1 - 3 frames and all call on ONE eval.
    The combining queue gathers them into rounds and refreshes the host views only when no call is waiting - when the last call
    has returned every view (values, weights, the float volume and its maximum, temporal rows and their ranges) must be what one
    call over the whole range leaves, with the waiting and the lazy refresh on and off, and after an interrupt + restart."""
    import threading
    coords, structures, mass = sdf_system(O, 11, n_water, box, F)
    n_s, N = structures.size, coords.shape[2]
    ocell, vcell = cell_pair(O, box)
    ox = np.arange(n_s, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", ox, ox, (0.0, 9.0)); ir.add_sdf("v", structures, ox, 7.0); ir.add_distance("d", structures[0], structures[1], L.DIST_MIN)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    one = V.ScriptEval(F, ir)
    assert one.frame_range(sysm, traj, 0, F)
    want = {n: one.property_data(n) for n in ("g", "v", "d")}
    counts, _ = oracle_rdf(O, coords, ocell, ox, ox, 0.0, 9.0)
    np.testing.assert_array_equal(want["g"].counts, counts)

    def pooled(ev, grain, stop_at=None):
        nxt = [0]; lock = threading.Lock(); res = []
        def work():
            while True:
                with lock:
                    b = nxt[0]; nxt[0] += grain
                if b >= F:
                    return
                if stop_at is not None and b >= stop_at:
                    ev.interrupt()
                res.append(ev.frame_range(sysm, traj, b, min(F, b + grain)))
        ths = [threading.Thread(target=work) for _ in range(nthreads)]
        [t.start() for t in ths]; [t.join() for t in ths]
        return res

    for gather, lazy, grain in combos:
        old = (lib.vmd_set_option(b"gather_us", gather), lib.vmd_set_option(b"lazy_views", lazy))
        try:
            ev = V.ScriptEval(F, ir)
            assert all(pooled(ev, grain)) and ev.frames_done() == F and ev.frame_mask().all()
            for n in ("g", "v", "d"):
                got = ev.property_data(n)
                np.testing.assert_array_equal(got.values, want[n].values, err_msg=f"{n} values (gather {gather}, lazy {lazy}, grain {grain})")
                assert got.max_value == want[n].max_value and got.min_value == want[n].min_value, n
                assert tuple(got.min_range) == tuple(want[n].min_range) and tuple(got.max_range) == tuple(want[n].max_range), n
            np.testing.assert_array_equal(ev.property_data("g").weights, want["g"].weights)
            np.testing.assert_array_equal(ev.property_data("g").counts, want["g"].counts)
            # interrupted half way, then restarted from scratch (src/main.cpp:984-990): the same views again
            ev.clear_data()
            pooled(ev, grain, stop_at=F // 2)
            assert ev.frames_done() <= F
            ev.clear_data()
            assert all(pooled(ev, grain)) and ev.frames_done() == F
            for n in ("g", "v", "d"):
                np.testing.assert_array_equal(ev.property_data(n).values, want[n].values, err_msg=f"{n} after interrupt + restart")
            # the same pattern from NATIVE threads (vmd_eval_frame_range_pooled, round 6: what bench.py --pool-threads and secondary.c1 time):
            # whole range, a ragged sub-range, one thread, more threads than frames
            for beg, end, threads in ((0, F, nthreads), (1, F - 2, 3), (0, F, 1), (0, F, 2 * F)):
                ev.clear_data()
                assert ev.frame_range_pooled(sysm, traj, beg, end, threads, grain) and ev.frames_done() == end - beg
                if (beg, end) == (0, F):
                    for n in ("g", "v", "d"):
                        np.testing.assert_array_equal(ev.property_data(n).values, want[n].values, err_msg=f"{n} from native pool threads")
                else:
                    sub = V.ScriptEval(F, ir)
                    assert sub.frame_range(sysm, traj, beg, end)
                    np.testing.assert_array_equal(ev.property_data("g").counts, sub.property_data("g").counts)
                    np.testing.assert_array_equal(ev.property_data("v").values, sub.property_data("v").values)
                    sub.close()
            ev.close()
        finally:
            lib.vmd_set_option(b"gather_us", old[0]); lib.vmd_set_option(b"lazy_views", old[1])


def readahead_case(lib, O, device=False, n_water=600, box=28.0, F=40, nthreads=6):
    """VIAMD's call pattern served by read-ahead: pool threads call frame_range with one frame or a few
    on ONE eval; the first of them evaluates a whole region of frame blocks ahead, the others only mark their frames requested, a block
    joins the totals when all of it has been asked for, and whoever returns last settles the eval.  Whatever the threads, grains and
    ranges: when the last call has returned, every view is bit for bit what ONE call over the same range leaves - a filtered range that
    ends inside a block counts nothing beyond its end, an interrupted evaluation restarts clean, a lone caller and a large range are
    served directly, blocks set by the caller (filtered evaluation) are the blocks read-ahead uses."""
    import threading
    coords, structures, mass = sdf_system(O, 13, n_water, box, F)
    n_s, N = structures.size, coords.shape[2]
    ocell, vcell = cell_pair(O, box)
    ox = np.arange(n_s, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", ox, ox, (0.0, 9.0)); ir.add_sdf("v", structures, ox, 7.0); ir.add_distance("d", structures[0], structures[1], L.DIST_MIN)
    traj = make_traj(lib, coords, vcell, device)
    traj0 = traj
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    names = ("g", "v", "d")

    def reference(beg, end):
        one = V.ScriptEval(F, ir)
        old = lib.vmd_set_option(b"readahead", 0)
        try:
            assert one.frame_range(sysm, traj, beg, end)
        finally:
            lib.vmd_set_option(b"readahead", old)
        out = {n: one.property_data(n) for n in names}
        out["mask"] = one.frame_mask().copy()
        out["_keep"] = one
        return out

    def same(ev, want, what):
        assert ev.frames_done() == int(want["mask"].sum()), what
        np.testing.assert_array_equal(ev.frame_mask(), want["mask"], err_msg=what)
        for n in names:
            got = ev.property_data(n)
            np.testing.assert_array_equal(got.values, want[n].values, err_msg=f"{n} values: {what}")
            assert got.max_value == want[n].max_value and got.min_value == want[n].min_value, (n, what)
            assert tuple(got.min_range) == tuple(want[n].min_range) and tuple(got.max_range) == tuple(want[n].max_range), (n, what)
        np.testing.assert_array_equal(ev.property_data("g").weights, want["g"].weights, err_msg=what)
        np.testing.assert_array_equal(ev.property_data("g").counts, want["g"].counts, err_msg=what)
        np.testing.assert_array_equal(ev.property_data("v").counts, want["v"].counts, err_msg=what)

    def pooled_on(ev, traj_, beg, end, grain):
        return pooled(ev, beg, end, grain, use_traj=traj_)

    def pooled(ev, beg, end, grain, nth=nthreads, stop_at=None, order=None, use_traj=None):
        traj = use_traj if use_traj is not None else traj0
        starts = list(range(beg, end, grain)) if order is None else order
        nxt = [0]; lock = threading.Lock(); res = []
        gate = threading.Barrier(nth)
        def work():
            gate.wait()
            while True:
                with lock:
                    k = nxt[0]; nxt[0] += 1
                if k >= len(starts):
                    return
                b = starts[k]
                if stop_at is not None and b >= stop_at:
                    ev.interrupt()
                res.append(ev.frame_range(sysm, traj, b, min(end, b + grain)))
        ths = [threading.Thread(target=work) for _ in range(nth)]
        [t.start() for t in ths]; [t.join() for t in ths]
        return res

    full = reference(0, F)
    counts, _ = oracle_rdf(O, coords, ocell, ox, ox, 0.0, 9.0)
    np.testing.assert_array_equal(full["g"].counts, counts)
    part = reference(7, 29)
    old = [(k, lib.vmd_set_option(k, v)) for k, v in ((b"readahead_block", 4), (b"readahead_frames", 8), (b"readahead_growth", 2),
                                                      (b"readahead_company_us", 200000), (b"readahead", 1))]
    try:
        # ---- the whole range, grain 1 and a grain that straddles blocks
        for grain in (1, 3):
            ev = V.ScriptEval(F, ir)
            assert all(pooled(ev, 0, F, grain))
            same(ev, full, f"pool of {nthreads}, grain {grain}")
            st = ev.readahead_stats()
            assert st["engaged"] == 1 and st["block_frames"] == 4 and st["regions"] >= 2 and st["committed_blocks"] + st["direct_frames"] // 4 >= 1, st
            assert st["committed_blocks"] * 4 + st["direct_frames"] == F, st            # every frame counted exactly once, one way or the other
            assert st["slow_calls"] > 0 and st["settles"] >= 1, st
            # again on the same eval (VIAMD: clear_data, re-evaluate - src/main.cpp:990-996), frames handed out from the far end
            ev.clear_data()
            assert all(pooled(ev, 0, F, grain, order=list(range(0, F, grain))[::-1]))
            same(ev, full, f"pool of {nthreads}, grain {grain}, descending")
            ev.close()
        # ---- a sub-range that starts and ends inside blocks (VIAMD's filtered evaluation, src/main.cpp:1014-1039): nothing beyond it is counted
        ev = V.ScriptEval(F, ir)
        assert all(pooled(ev, 7, 29, 1))
        same(ev, part, "sub-range [7, 29), grain 1")
        st = ev.readahead_stats()
        assert st["direct_frames"] >= 2 and st["region_frames"] >= 22, st     # frames 7 and 28 sit in blocks nobody else asked for
        # ... the rest of the trajectory afterwards, in one large call: 0..6 and 29..39 once each
        assert ev.frame_range(sysm, traj, 0, 7) and ev.frame_range(sysm, traj, 29, F)
        same(ev, full, "sub-range, then the rest in two direct calls")
        ev.close()
        # ---- interrupted half way, restarted (src/main.cpp:984-990)
        ev = V.ScriptEval(F, ir)
        pooled(ev, 0, F, 1, stop_at=F // 2)
        assert ev.frames_done() <= F
        ev.clear_data()
        assert all(pooled(ev, 0, F, 1))
        same(ev, full, "after interrupt + clear_data")
        ev.close()
        # ---- one caller, frame by frame: every call is the last one, read-ahead must stay out of the way
        lib.vmd_set_option(b"readahead_company_us", 50)
        ev = V.ScriptEval(F, ir)
        for f in range(7, 29):
            assert ev.frame_range(sysm, traj, f, f + 1)
            assert ev.frames_done() == f + 1 - 7                      # final after EVERY call
        same(ev, part, "single caller, grain 1")
        assert ev.readahead_stats()["regions"] == 0
        ev.close()
        lib.vmd_set_option(b"readahead_company_us", 200000)
        # ---- small calls first, then a large range over evaluated-ahead but never requested blocks
        ev = V.ScriptEval(F, ir)
        assert all(pooled(ev, 0, 10, 1))
        assert ev.frames_done() == 10
        assert ev.frame_range(sysm, traj, 10, F)
        same(ev, full, "pool over [0, 10), one call over [10, F)")
        ev.close()
        # ---- the caller's own frame blocks (filtered evaluation): read-ahead evaluates into THOSE, and a filtered eval reuses them
        ev = V.ScriptEval(F, ir)
        ev.set_block_frames(5)
        assert all(pooled(ev, 0, F, 1))
        same(ev, full, "caller-set blocks of 5")
        assert ev.readahead_stats()["block_frames"] == 5
        filt = V.ScriptEval(F, ir)
        filt.set_source(ev)
        assert filt.frame_range(sysm, traj, 7, 29)
        same(filt, part, "filtered eval served from blocks that read-ahead computed")
        assert filt.frame_stats()[1] >= 15                         # blocks [10, 25) at least came from the partials
        # ... and the filtered eval itself driven by the pool, frame by frame (src/main.cpp:1014-1039 under :993's grain): the blocks of its
        # regions that the source has finished are adopted from the source's partials, not evaluated again
        filt.clear_data()
        assert all(pooled(filt, 7, 29, 1))
        same(filt, part, "filtered eval, pool of threads, blocks adopted from the source")
        computed, reused = filt.frame_stats()
        # [10, 25) adopted; the ragged ends 7..9 and 25..28 are work.  How much more depends on who leads: a thread that arrives while the first
        # leader is still adopting may evaluate its own frames directly (seen once on the GPU in round 6: 17 computed, 30 reused - results bit
        # for bit what they must be, checked above) - but never more than the 22 frames that were asked for
        assert reused >= 15 and computed <= 22, (computed, reused)
        filt.close(); ev.close()
        # ---- VIAMD unedited: nobody calls set_block_frames; the full eval's own read-ahead blocks serve the filtered eval
        ev = V.ScriptEval(F, ir)
        assert all(pooled(ev, 0, F, 1))
        same(ev, full, "full eval, pool")
        filt = V.ScriptEval(F, ir)
        filt.set_source(ev)                                        # (the shim links evals of one ir like this)
        assert all(pooled(filt, 7, 29, 1))
        same(filt, part, "filtered eval adopting the full eval's read-ahead blocks")
        assert filt.frame_stats()[1] >= 16, filt.frame_stats()      # blocks of 4: [8, 28) are whole blocks of the source
        filt.close(); ev.close()
        # ---- opt-in deferred settle (round 5): ONE caller walking frame by frame is served by read-ahead like a pool, the
        # final settle runs on the eval's helper thread after a quiet period - or at once in wait_settled / finalize
        lib.vmd_set_option(b"readahead_lone", 1)
        try:
            ev = V.ScriptEval(F, ir)
            for f in range(F):
                assert ev.frame_range(sysm, traj, f, f + 1)
                assert ev.frames_done() <= f + 1                    # never ahead of what was asked for
            ev.wait_settled()
            same(ev, full, "lone caller, deferred settle, wait_settled")
            ev.wait_settled()                                          # nothing owed: a no-op
            st = ev.readahead_stats()
            assert st["engaged"] == 1 and st["regions"] >= 2 and st["committed_blocks"] >= F // 4 - 1, st
            # the same walk again on the same eval, no explicit wait: the helper settles by itself once the caller has been quiet
            ev.clear_data()
            for f in range(F):
                assert ev.frame_range(sysm, traj, f, f + 1)
            deadline = time.time() + 60.0
            while True:                                                # commits first, the views at the end of the same settle: poll the whole state
                try:
                    same(ev, full, "lone caller, deferred settle, helper thread")
                    break
                except AssertionError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.005)
            # a sub-range that starts and ends inside blocks; finalize() settles
            ev.clear_data()
            for f in range(7, 29):
                assert ev.frame_range(sysm, traj, f, f + 1)
            ev.finalize()
            same(ev, part, "lone caller over [7, 29), finalize")
            # interrupted half way: whatever was committed is a prefix-free subset; clear_data cancels the owed settle; the rerun is exact
            ev.clear_data()
            for f in range(F // 2):
                assert ev.frame_range(sysm, traj, f, f + 1)
            ev.interrupt()
            assert not ev.frame_range(sysm, traj, F // 2, F // 2 + 1)
            ev.clear_data()
            for f in range(0, F, 2):                                   # grain 2 this time
                assert ev.frame_range(sysm, traj, f, min(F, f + 2))
            ev.wait_settled()
            same(ev, full, "lone caller after interrupt + clear_data")
            ev.close()
            # freed right after the last call returned, the settle still owed: the helper ends with the eval
            ev = V.ScriptEval(F, ir)
            for f in range(9):
                assert ev.frame_range(sysm, traj, f, f + 1)
            ev.close()
            # a pool with the option on: the last leaver arms the helper instead of settling
            ev = V.ScriptEval(F, ir)
            assert all(pooled(ev, 0, F, 1))
            ev.wait_settled()
            same(ev, full, "pool of threads, deferred settle")
            ev.close()
        finally:
            lib.vmd_set_option(b"readahead_lone", 0)
        # ... and chosen per eval (vmd_eval_set_deferred_settle: what the shim's VMD_SHIM_DEFERRED_SETTLE does), the process-wide option off:
        # this eval trails and is settled on demand, a second eval walked the same way beside it keeps the strict contract
        ev, strict = V.ScriptEval(F, ir), V.ScriptEval(F, ir)
        ev.set_deferred_settle(1)
        for f in range(7, 29):
            assert ev.frame_range(sysm, traj, f, f + 1) and strict.frame_range(sysm, traj, f, f + 1)
            assert strict.frames_done() == f + 1 - 7 and ev.frames_done() <= f + 1 - 7
        same(strict, part, "strict eval beside a deferred one")
        assert ev.readahead_stats()["regions"] >= 1 and strict.readahead_stats()["regions"] == 0
        ev.wait_settled()
        same(ev, part, "per-eval deferred settle")
        ev.set_deferred_settle(-1)                                      # back to the process-wide option (off): strict again after clear_data
        ev.clear_data()
        for f in range(7, 12):
            assert ev.frame_range(sysm, traj, f, f + 1) and ev.frames_done() == f + 1 - 7
        ev.close(); strict.close()
        # ---- block partials that cannot be allocated (a large volume script beside an HBM-resident trajectory) must not fail
        # the evaluation: read-ahead steps aside, the combining queue serves the same calls, the results are the same
        lib.vmd_set_option(b"readahead_fail_alloc", 1)
        try:
            ev = V.ScriptEval(F, ir)
            assert all(pooled(ev, 0, F, 1)), lib.last_error()
            same(ev, full, "read-ahead whose block allocation failed: served by the combining queue")
            st = ev.readahead_stats()
            assert st["engaged"] == 0 and st["regions"] == 0, st
            ev.close()
        finally:
            lib.vmd_set_option(b"readahead_fail_alloc", 0)
        # ---- a source that was evaluated from ANOTHER trajectory (same script, same frame count) hands over nothing
        other = make_traj(lib, coords[::-1].copy(), vcell, device)      # the same frames in reverse order: a different trajectory instance
        ev = V.ScriptEval(F, ir)
        ev.set_block_frames(5)
        assert all(pooled(ev, 0, F, 1))
        for driver in ("one call", "pool"):
            filt = V.ScriptEval(F, ir)
            filt.set_source(ev)
            if driver == "one call":
                assert filt.frame_range(sysm, other, 0, F)
            else:
                assert all(pooled_on(filt, other, 0, F, 1))
            assert filt.frame_stats()[1] == 0, (driver, filt.frame_stats())            # nothing adopted ...
            np.testing.assert_array_equal(filt.property_data("g").counts, full["g"].counts)   # ... and a histogram does not care about frame order
            np.testing.assert_array_equal(filt.property_data("d").values[::-1], full["d"].values, err_msg=driver)     # rows follow the OTHER trajectory
            filt.close()
        ev.close()
    finally:
        for k, v in old:
            lib.vmd_set_option(k, v)


def resource_cache_case(lib, O, device=False, n_water=1500, box=38.0):
    """VIAMD creates an eval per script edit and frees the old one: the blocks, streams and events an eval gives up are cached
    process-wide and reused by the next.  Many life cycles give the oracle's integers every time, the cache stops growing after the
    first cycle, honours its bound, can be emptied, and can be switched off."""
    import ctypes as C
    coords, structures, mass = sdf_system(O, 5, n_water, box, 4)
    n_s, N = structures.size, coords.shape[2]
    ocell, vcell = cell_pair(O, box)
    ox = np.arange(n_s, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", ox, ox, (0.0, 9.0)); ir.add_sdf("v", structures, ox, 7.0); ir.add_distance("d", structures[0], structures[1], L.DIST_MIN)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    counts, _ = oracle_rdf(O, coords, ocell, ox, ox, 0.0, 9.0)
    vol, _ = oracle_sdf(O, coords, ocell, structures, mass, ox, 7.0)

    def stats():
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        lib.vmd_pool_stats(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def cycle():
        ev = V.ScriptEval(coords.shape[0], ir)
        assert ev.frame_range(sysm, traj, 0, coords.shape[0])
        np.testing.assert_array_equal(ev.property_data("g").counts, counts)
        np.testing.assert_array_equal(ev.property_data("v").counts, vol)
        ev.close()

    lib.vmd_pool_trim()
    assert stats() == (0, 0, 0)
    cycle()
    first = stats()
    assert first[0] > 0 and first[2] > 10, first                     # device blocks came back
    for _ in range(5):
        cycle()
    assert stats() == first, (stats(), first)                        # ... and every later eval lived on them: nothing new was cached
    # two evals alive at once need a second set; afterwards both sets are cached
    a = V.ScriptEval(coords.shape[0], ir); b = V.ScriptEval(coords.shape[0], ir)
    assert a.frame_range(sysm, traj, 0, 2) and b.frame_range(sysm, traj, 2, 4)
    a.close(); b.close()
    assert stats()[0] > first[0]
    lib.vmd_pool_trim()
    assert stats() == (0, 0, 0)
    old = lib.vmd_set_option(b"pool_mb", 1)                          # a bound below one volume accumulator (16.8 MB): that block is not kept
    try:
        cycle()
        assert 0 < stats()[0] <= (1 << 20)
        lib.vmd_set_option(b"pool_mb", 0)                            # off: every block goes back to the runtime
        cycle()
        assert stats() == (0, 0, 0)
    finally:
        lib.vmd_set_option(b"pool_mb", old)
    cycle()
    assert stats()[0] > 0


def class_decomposition_cases(lib, O, device=False, n_water=3000, box=40.0):
    """Co-evaluated RDFs of one range share pair passes through disjoint atom classes (BASELINE config 5: goo is a subset of
    ghv).  With and without the decomposition every property equals its own oracle histogram; sets that overlap partially,
    nested sets, a set that lists an atom twice (-> no decomposition) and a second range in the same script."""
    import ctypes as C
    coords, structures, mass = sdf_system(O, 12, n_water, box, 2, K=3, m=8)
    n_s, N = structures.size, coords.shape[2]
    o = np.arange(n_s, N, 3, dtype=np.int32)
    h = np.array([i for i in range(n_s, N) if (i - n_s) % 3], np.int32)
    heavy = np.concatenate([np.arange(0, n_s, 2, dtype=np.int32), o])          # water oxygens + every other blob atom
    a = np.concatenate([o[: o.size // 2], h[: h.size // 3]])                      # overlaps with b in a quarter of the oxygens
    b = np.concatenate([o[o.size // 4:], np.arange(1, n_s, 2, dtype=np.int32)])
    base = [("goo", o, o, 0.0, 9.0), ("goh", o, h, 0.0, 9.0), ("ghv", heavy, heavy, 0.0, 9.0), ("gab", a, b, 0.0, 9.0),
            ("gba", b, a, 0.0, 9.0), ("short", o, heavy, 0.5, 6.0)]
    launches = {}
    for classes in (1, 0):
        old = lib.vmd_set_option(b"rdf_classes", classes)
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        try:
            check_rdf(lib, O, coords, box, base, device=device)
        finally:
            lib.vmd_set_option(b"rdf_classes", old); lib.vmd_profile_enable(False)
        nl = C.c_uint64(0)
        lib.vmd_profile_ms(b"rdf_pencil", C.byref(nl))
        launches[classes] = nl.value
    assert launches[0] == len(base) and launches[1] != launches[0], launches       # one pass per property vs one per class pair
    dup = np.concatenate([o[:50], o[:5]])                                         # an atom listed twice counts twice: direct passes only
    check_rdf(lib, O, coords, box, [("gdup", dup, o, 0.0, 9.0), ("goo", o, o, 0.0, 9.0)], device=device)


# ---- SDF scenario: K rigid-ish structures of m atoms tumbling in a water box -------------------------------------------

def sdf_system(O, seed, n_water_atoms, box, frames, K=4, m=6):
    """Returns coords [F,3,N], structures [K,m], masses [N].  Atoms [0, K*m) are the structures, the rest water."""
    rng = np.random.default_rng(seed)
    n_s = K * m
    N = n_s + n_water_atoms
    water = np.stack([O.synth_frame(seed, N, box, 0.05, f, n_blob=n_s) for f in range(frames)])  # [F,3,N], blob rows zero
    template = rng.normal(0, 1.6, (m, 3))
    centers = rng.uniform(0, box, (K, 3))
    coords = water.copy()

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx

    Rk = [rot(rng.normal(size=3), rng.uniform(0, np.pi)) for _ in range(K)]
    for f in range(frames):
        for k in range(K):
            Rk[k] = rot(rng.normal(size=3), np.deg2rad(4.0)) @ Rk[k]
            centers[k] += rng.normal(0, 0.3, 3)
            pts = template @ Rk[k].T + centers[k] + rng.normal(0, 0.05, (m, 3))
            pts = np.mod(pts, box)     # wrapped: structures straddle the periodic boundary now and then
            coords[f, :, k * m:(k + 1) * m] = pts.T.astype(np.float32)
    structures = np.arange(n_s, dtype=np.int32).reshape(K, m)
    mass = np.ones(N, np.float32)
    mass[:n_s] = rng.choice([12.011, 14.007, 15.999, 1.008], n_s).astype(np.float32)
    w = np.arange(N - n_s)
    mass[n_s:] = np.where(w % 3 == 0, 15.999, 1.008).astype(np.float32)
    return coords, structures, mass


def oracle_sdf(O, coords, ocell, structures, mass, tgt, cutoff, dim=128, frames=None):
    K, m = structures.shape
    smass = mass[structures]
    ref_pose = O.sdf_ref_pose(coords[0, 0], coords[0, 1], coords[0, 2], ocell, structures[0], smass[0])
    vol = np.zeros(dim ** 3, np.uint64)
    mats = []
    for f in (range(coords.shape[0]) if frames is None else frames):
        M, R32, c32 = O.sdf_frame_align(coords[f, 0], coords[f, 1], coords[f, 2], ocell, structures, smass, ref_pose)
        O.sdf_frame_scatter(coords[f, 0], coords[f, 1], coords[f, 2], ocell, structures, R32, c32, tgt, cutoff, dim, vol)
        mats.append(M)
    return vol, np.stack(mats)


def check_sdf(lib, O, coords, box, structures, mass, tgt, cutoff, flags=L.PBC_ALL, device=False, ranges=None, allow_empty=False):
    ocell, vcell = cell_pair(O, box, flags)
    F, _, N = coords.shape
    ir = V.ScriptIR(lib)
    ir.add_sdf("v", structures, tgt, cutoff)
    ev = V.ScriptEval(F, ir)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    for beg, end in (ranges or [(0, F)]):
        assert ev.frame_range(sysm, traj, beg, end)
    pd = ev.property_data("v")
    vol, mats = oracle_sdf(O, coords, ocell, structures, mass, tgt, cutoff)
    assert pd.dim[1:] == (128, 128, 128)
    np.testing.assert_array_equal(pd.counts, vol, err_msg="SDF voxel counts differ from the oracle")
    np.testing.assert_array_equal(pd.values, vol.astype(np.float32))
    assert pd.max_value == float(vol.max())
    assert allow_empty or vol.sum() > 0
    # vis payload: world->reference matrices of a frame (density_volume.cpp:263)
    f = F - 1
    M4, ext = ev.sdf_matrices("v", sysm, traj, f)
    assert ext == np.float32(cutoff)
    np.testing.assert_allclose(M4[:, :3, :], mats[f].astype(np.float32), rtol=0, atol=0)
    if not ranges:
        # clear_data (src/main.cpp:990): a reader polling the record sees a NEW fingerprint over ZEROS at once - never the previous run's voxels -
        # although the 8.4 MB view itself is not touched (round 6: `values` points at shared zero pages until the next view has been written);
        # the re-evaluation that follows immediately (:993-997) brings the same volume back, in the view's own memory
        fp0, addr0 = pd.fingerprint, pd.c.values and C.cast(pd.c.values, C.c_void_p).value
        ev.clear_data()
        pd = ev.property_data("v")
        assert pd.fingerprint != fp0 and pd.max_value == 0.0
        assert not pd.values.any(), "clear_data must leave a zero volume under the new fingerprint"
        assert ev.frame_range(sysm, traj, 0, F)
        pd = ev.property_data("v")
        np.testing.assert_array_equal(pd.values, vol.astype(np.float32))
        assert C.cast(pd.c.values, C.c_void_p).value == addr0 and pd.max_value == float(vol.max())
    return ev, vol


def oracle_distance(O, coords, ocell, mass, a, b, kind):
    F = coords.shape[0]
    dim1 = len(a) * len(b) if kind == L.DIST_PAIR else 1
    out = np.zeros((F, dim1), np.float32)
    for f in range(F):
        x, y, z = coords[f]
        if kind == L.DIST_COM:
            out[f, 0] = O.distance_com(x, y, z, ocell, a, mass[a], b, mass[b])
        elif kind == L.DIST_MIN:
            out[f, 0] = O.distance_minmax(x, y, z, ocell, a, b, "min")
        elif kind == L.DIST_MAX:
            out[f, 0] = O.distance_minmax(x, y, z, ocell, a, b, "max")
        else:
            out[f] = O.distance_pair(x, y, z, ocell, a, b)
    return out


def oracle_distance_population(O, coords, ocell, mass, a_sets, b_sets, kind):
    cols = [oracle_distance(O, coords, ocell, mass, np.asarray(a, np.int32), np.asarray(b, np.int32), kind) for a, b in zip(a_sets, b_sets)]
    return np.concatenate(cols, axis=1)


def check_distances(lib, O, coords, box, mass, specs, flags=L.PBC_ALL, device=False, ranges=None):
    """specs: list of (name, a, b, kind) — or (name, a_sets, b_sets, kind, "pop") for a population of contexts.
    Temporal rows must equal the oracle bit for bit (fp32)."""
    ocell, vcell = cell_pair(O, box, flags)
    F, _, N = coords.shape
    ir = V.ScriptIR(lib)
    for sp in specs:
        if len(sp) == 5:
            ir.add_distance_population(sp[0], sp[1], sp[2], sp[3])
        else:
            ir.add_distance(*sp[:3], sp[3])
    ev = V.ScriptEval(F, ir)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    for beg, end in (ranges or [(0, F)]):
        assert ev.frame_range(sysm, traj, beg, end)
    for sp in specs:
        name, a, b, kind = sp[:4]
        pd = ev.property_data(name)
        if len(sp) == 5:
            ref = oracle_distance_population(O, coords, ocell, mass, a, b, kind)
        else:
            ref = oracle_distance(O, coords, ocell, mass, np.asarray(a, np.int32), np.asarray(b, np.int32), kind)
        assert pd.dim[0] == F and pd.dim[1] == ref.shape[1]
        got = pd.values.reshape(F, -1)
        np.testing.assert_array_equal(got, ref, err_msg=f"{name}: temporal values differ from the oracle")
        if ref.shape[1] > 1:
            agg = pd.aggregate
            np.testing.assert_allclose(agg["mean"], ref.mean(axis=1), rtol=1e-6)
            np.testing.assert_array_equal(agg["ext"][:, 0], ref.min(axis=1))
            np.testing.assert_array_equal(agg["ext"][:, 1], ref.max(axis=1))
        assert pd.min_range[0] == ref.min() and pd.max_range[0] == ref.max()
    return ev


def triclinic_cases(lib, O, n_water, device=False):
    """SPEC S3t: triclinic cells go through the general (brute) RDF kernel and the rint-in-fractional-space minimum image of
    SDF / distance; everything must still equal the oracle bit for bit."""
    box = (30.0, 28.0, 26.0, 6.0, -4.0, 5.0)
    A = np.array([[box[0], box[3], box[4]], [0, box[1], box[5]], [0, 0, box[2]]])
    coords, structures, mass = sdf_system(O, 31, n_water, 26.0, 3)
    F, _, N = coords.shape
    # spread the orthorhombic test system over the triclinic cell (and beyond: atoms up to half a cell outside)
    frac = coords.astype(np.float64) / 26.0 * 1.4 - 0.2
    coords = np.einsum("ij,fjn->fin", A, frac).astype(np.float32)
    n_s = structures.size
    # keep every structure compact (the alignment needs whole structures): shrink the blob atoms towards their centre
    for f in range(F):
        for k in range(structures.shape[0]):
            idx = structures[k]
            c0 = coords[f][:, idx[:1]]
            coords[f][:, idx] = c0 + 0.15 * (coords[f][:, idx] - c0)
    o = np.arange(n_s, N, 3, dtype=np.int32)
    h = np.array([i for i in range(n_s, N) if (i - n_s) % 3], np.int32)
    # pencil grid in the sheared cell (default) and the all-pairs kernel: both must reproduce the oracle's S3t arithmetic
    import ctypes as C
    for brute in (0, 1):
        old = lib.vmd_set_option(b"force_brute", brute)
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        try:
            check_rdf(lib, O, coords, box, [("goo", o, o, 0.0, 9.0), ("goh", o, h, 0.5, 8.0)], device=device, oracle_method="brute")
        finally:
            lib.vmd_set_option(b"force_brute", old)
            lib.vmd_profile_enable(False)
        n_grid, n_brute = C.c_uint64(0), C.c_uint64(0)
        lib.vmd_profile_ms(b"rdf_pencil", C.byref(n_grid)); lib.vmd_profile_ms(b"rdf_brute", C.byref(n_brute))
        assert (n_grid.value, n_brute.value) == ((0, 2) if brute else (2, 0)), "wrong kernel family for the triclinic batch"
    # strongly sheared cell (tilts at the reduced-cell limit), cell fluctuating from frame to frame, atoms outside the cell
    rng = np.random.default_rng(77)
    nO = max(200, n_water // 3)
    boxes, frames = [], []
    for f in range(2):
        Lx, Ly, Lz = 44.0 + f, 40.0 - 0.5 * f, 36.0 + 0.25 * f
        bx = (Lx, Ly, Lz, 0.5 * Lx, -0.5 * Lx + f, 0.5 * Ly - 0.5 * f)
        Af = np.array([[bx[0], bx[3], bx[4]], [0, bx[1], bx[5]], [0, 0, bx[2]]])
        frames.append((Af @ rng.uniform(-0.3, 1.3, (3, nO))).astype(np.float32))
        boxes.append(bx)
    c2 = np.stack(frames)
    a = np.arange(0, nO, 2, dtype=np.int32); b2 = np.arange(1, nO, 2, dtype=np.int32); allo = np.arange(nO, dtype=np.int32)
    check_rdf(lib, O, c2, boxes, [("gaa", allo, allo, 0.0, 8.0), ("gab", a, b2, 1.0, 7.5)], device=device, oracle_method="brute")
    check_sdf(lib, O, coords, box, structures, mass, o, 6.0, device=device)
    check_distances(lib, O, coords, box, mass, [("d", structures[0], structures[1], L.DIST_COM), ("m", structures[0], structures[2], L.DIST_MIN),
                                               ("x", structures[1], structures[2], L.DIST_MAX), ("p", structures[0][:2], o[:3], L.DIST_PAIR)],
                    device=device)


def filtered_cases(lib, O, n_water, device=False):
    """a full evaluation that keeps per-block partials, and "Eval Filt" evaluations (src/main.cpp:1014-1039)
    that reuse them.  Every sub-range result must equal a plain evaluation of that range by the oracle, bit for bit."""
    box, F, S = 34.0, 11, 4
    coords, structures, mass = sdf_system(O, 23, n_water, box, F)
    n_s, N = structures.size, coords.shape[2]
    ocell, vcell = cell_pair(O, box)
    ox = np.arange(n_s, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", ox, ox, (0.0, 9.0))
    ir.add_sdf("v", structures, ox, 7.0)
    ir.add_distance("d", structures[0], structures[1], L.DIST_MIN)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    dref = oracle_distance(O, coords, ocell, mass, structures[0], structures[1], L.DIST_MIN)

    def verify(ev, frames):
        frames = list(frames)
        counts, weights = oracle_rdf(O, coords, ocell, ox, ox, 0.0, 9.0, frames=frames)
        pd = ev.property_data("g")
        np.testing.assert_array_equal(pd.counts, counts)
        np.testing.assert_allclose(pd.weights64, weights, rtol=1e-12)
        vol, _ = oracle_sdf(O, coords, ocell, structures, mass, ox, 7.0, frames=frames)
        np.testing.assert_array_equal(ev.property_data("v").counts, vol)
        assert vol.sum() > 0
        got = ev.property_data("d").values.reshape(F, -1)
        np.testing.assert_array_equal(got[frames], dref[frames])
        mask = np.zeros(F, bool); mask[frames] = True
        np.testing.assert_array_equal(ev.frame_mask().astype(bool), mask)

    # full evaluation in one call: blocks [0,4) [4,8) [8,11) are all kept
    full = V.ScriptEval(F, ir)
    full.set_block_frames(S)
    assert full.frame_range(sysm, traj, 0, F)
    verify(full, range(F))
    assert full.frame_stats() == (F, 0)

    filt = V.ScriptEval(F, ir)
    filt.set_source(full)
    assert filt.frame_range(sysm, traj, 1, 10)            # ragged 1-3, block [4,8), ragged 8-9
    verify(filt, range(1, 10))
    assert filt.frame_stats() == (5, 4)
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 0, F)             # answered entirely from the partials
    verify(filt, range(F))
    assert filt.frame_stats() == (0, F)
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 5, 7)             # inside one block: plain evaluation
    verify(filt, range(5, 7))
    assert filt.frame_stats() == (2, 0)

    # a full evaluation fed in unaligned ranges keeps only the blocks a single call covered: [0,4) and [8,11)
    full.clear_data()
    assert full.frame_range(sysm, traj, 6, F) and full.frame_range(sysm, traj, 0, 6)
    verify(full, range(F))
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 0, F)
    verify(filt, range(F))
    assert filt.frame_stats() == (4, 7)
    filt.set_source(None)
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 2, 9)
    verify(filt, range(2, 9))
    assert filt.frame_stats() == (7, 0)

    # guard rails
    other = V.ScriptIR(lib); other.add_rdf("g", ox, ox, (0.0, 8.0))
    import pytest
    with pytest.raises(V.VmdError):
        V.ScriptEval(F, other).set_source(full)
    # a source that keeps no blocks (yet) is accepted since round 4 - read-ahead may give it some later; until then everything is computed
    lone, bare = V.ScriptEval(F, ir), V.ScriptEval(F, ir)
    lone.set_source(bare)
    assert lone.frame_range(sysm, traj, 2, 9)
    verify(lone, range(2, 9))
    assert lone.frame_stats() == (7, 0)
    lone.close(); bare.close()
    with pytest.raises(V.VmdError):
        full.set_block_frames(2)                                  # not after frames were evaluated


def _kernel_family(lib):
    import ctypes as C
    n_grid, n_brute = C.c_uint64(0), C.c_uint64(0)
    lib.vmd_profile_ms(b"rdf_pencil", C.byref(n_grid)); lib.vmd_profile_ms(b"rdf_brute", C.byref(n_brute))
    return int(n_grid.value), int(n_brute.value)


def open_boundary_cases(lib, O, n, device=False):
    """Systems without a periodic cell, and slabs (periodic in two directions): the pencil grid over the batch's bounding box
    must reproduce SPEC S3 with the open axes left alone, exactly like the all-pairs kernel does."""
    rng = np.random.default_rng(41)
    F = 3
    # a drifting, breathing cloud: the bounding box differs from frame to frame; a few far outliers stretch it
    c = np.stack([rng.normal(0, 14.0 + 2 * f, (3, n)) + np.array([[5.0 * f], [-30.0], [100.0]]) for f in range(F)]).astype(np.float32)
    c[:, :, :5] += 60.0
    a, b, allidx = np.arange(0, n, 2), np.arange(1, n, 3), np.arange(n)
    props = [("gaa", allidx, allidx, 0.0, 9.0), ("gab", a, b, 0.5, 7.0)]
    for brute in (0, 1):
        old = lib.vmd_set_option(b"force_brute", brute)
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        try:
            check_rdf(lib, O, c, None, props, device=device, oracle_method="brute")
        finally:
            lib.vmd_set_option(b"force_brute", old); lib.vmd_profile_enable(False)
        assert _kernel_family(lib) == ((0, 2) if brute else (2, 0))
    # slab: periodic in x and y (box 40 x 36), open along z; and a wire: periodic along z only
    s = c.copy()
    lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
    try:
        check_rdf(lib, O, s, (40.0, 36.0, 50.0), props, flags=3, device=device, oracle_method="brute")
        check_rdf(lib, O, s, (40.0, 36.0, 44.0), props, flags=4, device=device, oracle_method="brute")
    finally:
        lib.vmd_profile_enable(False)
    assert _kernel_family(lib) == (4, 0)
    # flat system (zero extent along z), and a cutoff larger than the whole cloud (one cell per axis)
    flat = c.copy(); flat[:, 2, :] = 3.25
    check_rdf(lib, O, flat, None, [("g", allidx, allidx, 0.0, 9.0)], device=device, oracle_method="brute")
    # the same cloud kilo-Angstroms away from the origin: x - origin is rounded at the magnitude of the raw coordinate
    far = (c + np.float32(2500.0)).astype(np.float32)
    check_rdf(lib, O, far, None, [("g", a, b, 0.0, 9.0)], device=device, oracle_method="brute")
    small = (c[:, :, :200] * 0.1).astype(np.float32)
    check_rdf(lib, O, small, None, [("g", np.arange(200), np.arange(200), 0.0, 12.0)], device=device, oracle_method="brute")


def sdf_triclinic_spread_structures(lib, O, device=False):
    """Regression (found by scripts/fuzz_emu_sdf.py): four reference structures spread over a triclinic cell.  The scatter's
    group pre-filter relies on the triangle inequality of the minimum-image metric, which rounding in fractional space only
    provides for vectors shorter than half the cell width; the filter must stand down when its reach is longer."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sdf_triclinic_spread_structures.npz"))
    box = tuple(float(v) for v in d["box"])
    check_sdf(lib, O, d["coords"], box, d["structures"], d["mass"], d["tgt"], float(d["cutoff"]), device=device)


def sheared_sc_lattice(lib, device=False):
    """Known answer that owes nothing to the oracle (tests/test_oracle.py: test_sc_lattice_in_sheared_cells): the simple cubic
    lattice Z^3 * a0 described by cells a = (n,0,0), b = (s,n,0), c = (t,u,n) (integer s,t,u: the same crystal) must show the
    exact shell multiplicities 6, 12, 8, 6, 24, 24, 12, 30, 24 per atom in the evaluator's 1024-bin histogram.  a0 = 1.0024 keeps every
    shell radius >= 0.23 bins away from a bin edge at r_max = 3.2."""
    n, a0, rmax = 8, 1.0024, 3.2
    pts = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij")).reshape(3, -1).astype(np.float32) * np.float32(a0)
    N = pts.shape[1]
    idx = np.arange(N, dtype=np.int32)
    expect = np.zeros(1024, np.uint64)
    for d2, mult in ((1, 6), (2, 12), (3, 8), (4, 6), (5, 24), (6, 24), (8, 12), (9, 30), (10, 24)):
        expect[int(np.sqrt(d2) * a0 / rmax * 1024.0)] += mult * N
    for s, t, u in ((0, 0, 0), (3, -2, 3), (-3, 3, -2)):
        moved = pts.copy()
        moved[:, ::3] += (np.array([s, n, 0], np.float32) * np.float32(a0))[:, None]
        moved[:, 1::5] -= (np.array([t, u, n], np.float32) * np.float32(a0))[:, None]
        cell = V.make_unitcell(n * a0, tilt=(s * a0, t * a0, u * a0))
        ir = V.ScriptIR(lib)
        ir.add_rdf("g", idx, idx, (0.0, rmax))          # cutoff below half the cell width: the pencil grid (sheared for s,t,u != 0)
        ev = V.ScriptEval(2, ir)
        traj = make_traj(lib, np.stack([pts, moved]), cell, device)
        assert ev.frame_range(V.MolSystem(N, unitcell=cell), traj, 0, 2)
        np.testing.assert_array_equal(ev.property_data("g").counts, 2 * expect, err_msg=f"tilt {(s, t, u)}")


def open_sc_lattice(lib, O=None, device=False):
    """Known answer for open and partly periodic systems (no oracle involved unless `O` is given, which is then checked against
    the same numbers): a finite n^3 block of a simple cubic lattice.  The number of ordered pairs with displacement (dx,dy,dz) is
    prod_k (n if axis k is periodic else n - |d_k|), exactly."""
    n, a0, rmax = 8, 1.0024, 3.2
    pts = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij")).reshape(3, -1).astype(np.float32) * np.float32(a0)
    pts = pts + np.float32(37.5)                      # away from the origin: open axes use the raw coordinates
    N = pts.shape[1]
    idx = np.arange(N, dtype=np.int32)
    r = np.arange(-3, 4)
    for flags in (0, 3, 4, 5):
        expect = np.zeros(1024, np.uint64)
        for dx in r:
            for dy in r:
                for dz in r:
                    d2 = int(dx * dx + dy * dy + dz * dz)
                    if d2 == 0 or np.sqrt(d2) * a0 >= rmax:
                        continue
                    pairs = 1
                    for k, d in enumerate((dx, dy, dz)):
                        pairs *= n if flags >> k & 1 else n - abs(int(d))
                    expect[int(np.sqrt(d2) * a0 / rmax * 1024.0)] += pairs
        cell = V.make_unitcell(n * a0, flags=flags) if flags else V.make_unitcell(None)
        ir = V.ScriptIR(lib)
        ir.add_rdf("g", idx, idx, (0.0, rmax))
        ev = V.ScriptEval(1, ir)
        assert ev.frame_range(V.MolSystem(N, unitcell=cell), make_traj(lib, pts[None], cell, device), 0, 1)
        np.testing.assert_array_equal(ev.property_data("g").counts, expect, err_msg=f"pbc flags {flags}")
        if O is not None:
            ocell = O.make_cell(n * a0, flags=flags) if flags else O.make_cell(None)
            for method in ("brute", "cells"):
                counts, hits = O.rdf_frame(pts[0], pts[1], pts[2], ocell, idx, idx, 0.0, rmax, nbins=1024, method=method)
                if hits == 2 ** 64 - 1:
                    continue                           # this oracle method does not cover the cell type
                np.testing.assert_array_equal(counts, expect, err_msg=f"oracle {method}, pbc flags {flags}")


def sdf_rotations_known_answer(lib, device=False):
    """Known answer for sdf() that owes nothing to the oracle: a rigid, asymmetric 4-atom structure (equal masses, centre of mass
    exactly representable) carried through the 24 proper rotations of the cube (signed permutation matrices: exact in float) and
    integer translations, with target atoms sitting on voxel CENTRES of the aligned 128^3 grid.  Whatever the alignment's last
    bits are, every target must land in its own voxel in every frame: counts == number of frames there, 0 elsewhere, x fastest
    (`values[z*d*d + y*d + x]`, /root/reference/src/main.cpp:5811-5815), and the world->reference matrices are the inverse
    rotations."""
    import itertools
    s, dim = 8.0, 128
    body = np.array([[2, 0, 0], [0, 1, 0], [0, 0, 0.5], [-2, -1, -0.5]], np.float64)         # sum = 0: COM at the centre
    rng = np.random.default_rng(12)
    vox = rng.integers(0, dim, (300, 3))                                                       # (ix, iy, iz) of every target
    vox = np.unique(vox, axis=0)
    q = -s + (2 * s / dim) * (vox + 0.5)                                                       # voxel centres, exact
    rots = []
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1, -1), repeat=3):
            R = np.zeros((3, 3))
            for r in range(3):
                R[r, perm[r]] = signs[r]
            if round(np.linalg.det(R)) == 1:
                rots.append(R)
    assert len(rots) == 24
    rots.sort(key=lambda R: -np.trace(R))                                                      # identity first: frame 0 is the pose
    F = len(rots)
    coords = np.zeros((F, 3, 4 + len(q)), np.float32)
    centres = rng.integers(-20, 21, (F, 3)).astype(np.float64) + 40.0
    for f, R in enumerate(rots):
        coords[f, :, :4] = (body @ R.T + centres[f]).T
        coords[f, :, 4:] = (q @ R.T + centres[f]).T
    assert np.array_equal(coords.astype(np.float64)[0, :, :4].T, body + centres[0])           # everything is exactly representable
    N = coords.shape[2]
    cell = V.make_unitcell(None)
    ir = V.ScriptIR(lib)
    ir.add_sdf("v", np.arange(4, dtype=np.int32)[None, :], np.arange(4, N, dtype=np.int32), s)
    ev = V.ScriptEval(F, ir)
    sysm = V.MolSystem(N, mass=np.ones(N, np.float32), unitcell=cell)
    traj = make_traj(lib, coords, cell, device)
    assert ev.frame_range(sysm, traj, 0, F)
    vol = ev.property_data("v").counts.reshape(dim, dim, dim)                                  # [z][y][x]
    expect = np.zeros((dim, dim, dim), np.uint64)
    expect[vox[:, 2], vox[:, 1], vox[:, 0]] = F
    np.testing.assert_array_equal(vol, expect)
    for f in (0, 7, F - 1):
        M4, ext = ev.sdf_matrices("v", sysm, traj, f)
        assert ext == np.float32(s)
        A = np.asarray(M4[0], np.float64)               # rows of the world -> reference matrix: q = R^T (x - centre)
        np.testing.assert_allclose(A[:3, :3], rots[f].T, atol=1e-6)
        np.testing.assert_allclose(A[:3, :3] @ centres[f] + A[:3, 3], 0.0, atol=1e-4)


def distance_known_answer(lib, device=False):
    """Closed-form answers for the distance family (no oracle): integer coordinates across the periodic boundary of a cube of 20.
    a = {(1,1,1), (1,1,3)} (COM (1,1,2)), b = {(18,5,2), (18,5,6)} (COM (18,5,4)): the minimum image of the COM separation is
    (3, -4, -2) -> sqrt(29); atom pairs: (3,4,1) sqrt 26, (3,4,5) sqrt 50, (3,4,1) sqrt 26, (3,4,3) sqrt 34."""
    xyz = np.array([[1, 1, 1], [1, 1, 3], [18, 5, 2], [18, 5, 6]], np.float32)
    F = 3
    coords = np.stack([(xyz + np.float32(20.0 * f)).T for f in range(F)])        # whole-box shifts change nothing
    cell = V.make_unitcell(20.0)
    ir = V.ScriptIR(lib)
    ir.add_distance("com", [0, 1], [2, 3], L.DIST_COM)
    ir.add_distance("min", [0, 1], [2, 3], L.DIST_MIN)
    ir.add_distance("max", [0, 1], [2, 3], L.DIST_MAX)
    ir.add_distance("pair", [0, 1], [2, 3], L.DIST_PAIR)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(4, mass=np.ones(4, np.float32), unitcell=cell), make_traj(lib, coords, cell, device), 0, F)
    f32 = lambda v: np.sqrt(np.float32(v))
    for f in range(F):
        assert ev.property_data("com").values.reshape(F, -1)[f, 0] == f32(29)
        assert ev.property_data("min").values.reshape(F, -1)[f, 0] == f32(26)
        assert ev.property_data("max").values.reshape(F, -1)[f, 0] == f32(50)
        np.testing.assert_array_equal(np.sort(ev.property_data("pair").values.reshape(F, -1)[f]), np.sort([f32(26), f32(50), f32(26), f32(34)]))


def export_cases(lib, O, tmp_path, device=False, n_water=1500, box=34.0):
    """SURVEY 8f-2 behind the C ABI: an evaluated script -> Gaussian cube (volume + atoms of reference structure 0, export_cube,
    src/main.cpp:5718-5830) and XVG / CSV tables (src/main.cpp:5640-5716, 5953-6040); everything read back and compared with
    the evaluator's own arrays and the vis payload (matrices, structures, extent)."""
    from viamd_amd import export
    F = 3
    coords, structures, mass = sdf_system(O, 8, n_water, box, F, K=3, m=7)
    n_s, N = structures.size, coords.shape[2]
    ox = np.arange(n_s, N, 3, dtype=np.int32)
    ocell, vcell = cell_pair(O, box)
    ir = V.ScriptIR(lib)
    ir.add_sdf("v", structures, ox, 7.0)
    ir.add_rdf("g", ox, ox, (0.0, 9.0))
    ir.add_distance("d", structures[0], structures[1], L.DIST_COM)
    ir.add_distance_population("p", [s[:2] for s in structures], [s[3:5] for s in structures], L.DIST_MIN)
    ev = V.ScriptEval(F, ir)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    assert ev.frame_range(sysm, traj, 0, F)
    # vis payload: matrices of a frame + the structures' atoms + the half extent
    mats, st, ext = ev.sdf_payload("v", sysm, traj, 1)
    np.testing.assert_array_equal(st, structures)
    M4, ext2 = ev.sdf_matrices("v", sysm, traj, 1)
    np.testing.assert_array_equal(mats, M4)
    assert ext == ext2 == np.float32(7.0)
    # cube: structure 0's atoms at frame-0 coordinates through the matrix of the requested frame, then the volume
    anum = np.where(mass > 15.0, 8, np.where(mass > 13.0, 7, np.where(mass > 11.0, 6, 1))).astype(np.uint8)
    path = tmp_path / "v.cube"
    ev.export_cube(path, "v", sysm, traj, frame=1, atomic_numbers=anum)
    back = export.read_cube(path)
    vol = ev.property_data("v").counts
    assert vol.sum() > 0
    np.testing.assert_array_equal(back["volume"], vol.astype(np.float64))        # counts stay far below 1e6: %12.6E is exact
    b = export.ANGSTROM_TO_BOHR
    assert back["dim"] == [128] * 3 and abs(back["origin"][0] + 7.0 * b) < 1e-5 and abs(back["voxel"][1] - 14.0 * b / 128) < 1e-6
    assert len(back["atoms"]) == structures.shape[1]
    want = (mats[0] @ np.concatenate([coords[0][:, np.sort(structures[0])], np.ones((1, structures.shape[1]))]))[:3].T * b
    got = np.array([a[2:] for a in back["atoms"]])
    np.testing.assert_allclose(got, want, atol=2e-5)
    assert [int(a[0]) for a in back["atoms"]] == [int(anum[i]) for i in np.sort(structures[0])]
    lines = path.read_text().split("\n")
    assert lines[0] == "EXPORTED DENSITY VOLUME FROM VIAMD, UNITS IN BOHR" and lines[1] == "OUTER LOOP: X, MIDDLE LOOP: Y, INNER LOOP: Z"
    # tables
    for fmt in ("xvg", "csv"):
        for name in ("g", "d", "p"):
            ev.export_table(tmp_path / f"{name}.{fmt}", name, fmt)
    rows = [l for l in (tmp_path / "d.xvg").read_text().split("\n") if l and l[0] not in "#@"]
    d = ev.property_data("d").values
    assert len(rows) == F and [float(r.split()[0]) for r in rows] == [0.0, 1.0, 2.0]
    np.testing.assert_allclose([float(r.split()[1]) for r in rows], d, atol=6e-7)
    assert '@ s1 legend "d (\u00c5)"' in (tmp_path / "d.xvg").read_text()       # "label (unit)": src/main.cpp:5965-5967
    assert (tmp_path / "g.csv").read_text().split("\n")[0] == "\u00c5,g,"         # x label = the x unit string, :5999
    ev.export_table(tmp_path / "d_ps.csv", "d", "csv", frame_times=np.arange(F) * 2.0, time_unit="ps")
    assert (tmp_path / "d_ps.csv").read_text().split("\n")[:2] == ["Time (ps),d (\u00c5),", "0,%.6g," % d[0]]       # :5969-5974
    csv = (tmp_path / "p.csv").read_text().split("\n")
    assert csv[0] == "Frame,p[1],p[2],p[3],"                                      # trailing comma as export_csv writes it
    p = ev.property_data("p").values.reshape(F, 3)
    np.testing.assert_allclose([[float(t) for t in r.split(",")[1:4]] for r in csv[1:1 + F]], p, rtol=1e-5)      # %.6g
    g = [l for l in (tmp_path / "g.csv").read_text().split("\n")[1:] if l]
    assert len(g) == 128
    gd = V.downsample_histogram(ev.property_data("g").values, ev.property_data("g").weights, 128, lib=lib)
    np.testing.assert_allclose([float(r.split(",")[1]) for r in g], gd, rtol=1e-5, atol=1e-12)
    x = [float(r.split(",")[0]) for r in g]
    assert x[0] == 0.0 and abs(x[-1] - 9.0) < 1e-6 and abs(x[1] - 9.0 / 127) < 1e-6     # sample_range: both ends included
    with pytest_raises(V.VmdError):
        ev.export_table(tmp_path / "v.csv", "v", "csv")


def pytest_raises(exc):
    import pytest
    return pytest.raises(exc)


def filtered_contention_case(lib, O, device=False, n=3000, box=60.0, F=12, S=2, rounds=10):
    """"Eval Filt" attached to a source that is still RUNNING.  One thread drives the full evaluation in
    small ranges (it keeps one partial accumulator per block of S frames), another keeps clearing the filtered eval and asking
    for random sub-ranges: whatever blocks happen to be finished are served from the partials, the rest is computed - every
    answer must equal the oracle's histogram of that range, bit for bit."""
    import threading
    coords = water_box(O, 9, n, box, F)
    o = oxygen(n)
    ocell, vcell = cell_pair(O, box)
    per_frame = [oracle_rdf(O, coords, ocell, o, o, 0.0, 10.0, frames=(f,))[0] for f in range(F)]
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", o, o, 10.0)
    traj = make_traj(lib, coords, vcell, device)
    sysm = V.MolSystem(n, unitcell=vcell)
    rng = np.random.default_rng(4)
    for rnd in range(2):
        full, filt = V.ScriptEval(F, ir), V.ScriptEval(F, ir)
        full.set_block_frames(S)
        filt.set_source(full)
        errors = []

        def run_full():
            try:
                for beg in range(0, F, S):                       # whole blocks, one call each (like enkiTS ranges)
                    assert full.frame_range(sysm, traj, beg, min(F, beg + S))
            except Exception as ex:                              # noqa: BLE001
                errors.append(ex)

        th = threading.Thread(target=run_full)
        th.start()
        reused = 0
        for k in range(rounds):
            a = int(rng.integers(0, F - 1)); b = int(rng.integers(a + 1, F + 1))
            filt.clear_data()
            assert filt.frame_range(sysm, traj, a, b)
            want = np.sum([per_frame[f] for f in range(a, b)], axis=0).astype(np.uint64)
            np.testing.assert_array_equal(filt.property_data("g").counts, want, err_msg=f"filtered range [{a}, {b}) while the source runs")
            reused += filt.frame_stats()[1]
        th.join()
        assert not errors, errors
        np.testing.assert_array_equal(full.property_data("g").counts, np.sum(per_frame, axis=0).astype(np.uint64))
        filt.clear_data()
        assert filt.frame_range(sysm, traj, 0, F) and filt.frame_stats() == (0, F)      # afterwards everything comes from the partials
        filt.set_source(None)


def device_view_cache_case(lib, O, n=1500, box=40.0):
    """A resident trajectory whose cells or coordinates change between evaluations: the evaluator keeps the boxes (and, for open
    axes, the bounding boxes) of an unchanged frame range on the device, keyed by vmd_device_view_t::cells_version - every
    modification must invalidate them."""
    F = 3
    coords = water_box(O, 13, n, box, F)
    o = oxygen(n)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", o, o, 9.0)
    ev = V.ScriptEval(F, ir)

    def run_and_check(traj, vcell, ocell, frames_coords):
        ev.clear_data()
        assert ev.frame_range(V.MolSystem(n, unitcell=vcell), traj, 0, F)
        want, _ = oracle_rdf(O, frames_coords, ocell, o, o, 0.0, 9.0)
        np.testing.assert_array_equal(ev.property_data("g").counts, want)

    ocell, vcell = cell_pair(O, box)
    traj = V.DeviceTrajectory(F, n, lib=lib)
    traj.upload(coords, vcell)
    run_and_check(traj, vcell, ocell, coords)
    run_and_check(traj, vcell, ocell, coords)                        # second run: served from the cached boxes
    ocell2, vcell2 = cell_pair(O, box + 3.0)                         # same coordinates in a larger cell
    traj.set_cell(vcell2)
    run_and_check(traj, vcell2, ocell2, coords)
    c2 = coords.copy()
    c2[1] = water_box(O, 14, n, box, 1)[0]                            # new coordinates in frame 1
    traj.upload_frame(1, vcell2, c2[1, 0], c2[1, 1], c2[1, 2])
    run_and_check(traj, vcell2, ocell2, c2)
    # no cell at all: the grid spans the bounding box of the batch, which moves with the coordinates
    ocell0, vcell0 = cell_pair(O, None, 0)
    t0 = V.DeviceTrajectory(F, n, lib=lib)
    t0.upload(coords, vcell0)
    run_and_check(t0, vcell0, ocell0, coords)
    c3 = coords * np.float32(1.5) + np.float32(7.0)                   # the system grows and moves: a stale bounding box would clip it
    for f in range(F):
        t0.upload_frame(f, vcell0, c3[f, 0], c3[f, 1], c3[f, 2])
    run_and_check(t0, vcell0, ocell0, c3)


def spec_switch_check(lib, oracle):
    """oracle/SPEC.md's DECISION: tags as configuration (VERDICT r02 next #3b): vmd_set_option("spec_*", 1) flips the product, the
    oracle flips with vo_set_spec / its inputs, and the two still agree bit for bit - so matching a real mdlib later is a
    setting, not a kernel edit.  Every switch is also shown to CHANGE the result on this system (the test would pass vacuously
    otherwise)."""
    coords, structures, mass = sdf_system(oracle, 5, 1500, 30.0, 3)
    N = coords.shape[2]
    # two atoms on top of each other and a pair at exactly r_max along x: the cases the open / closed interval disagree on
    coords[:, :, 700] = coords[:, :, 703]
    coords[:, :, 706] = coords[:, :, 709]
    coords[:, 0, 706] = np.mod(coords[:, 0, 709] + np.float32(6.0), np.float32(30.0))
    n_s = structures.size
    o = np.arange(n_s + (-n_s) % 3 + 1, N, 3, dtype=np.int32)          # 700, 703, 706, 709 are among them
    assert {700, 703, 706, 709} <= set(o.tolist())
    cell, ocell = V.make_unitcell(30.0), oracle.make_cell(30.0)
    members = structures.reshape(-1)
    tgt = np.unique(np.concatenate([o, members[::2]])).astype(np.int32)        # targets that ARE structure atoms: the exclusion rule matters
    F = coords.shape[0]
    coords_default = coords

    # the same system moved out of the cell: whole box lengths plus a fraction, a different shift per atom
    rng = np.random.default_rng(17)
    shifted = (coords + (rng.integers(-2, 3, size=(1, 3, N)) * np.float32(30.0) + rng.random((1, 3, N)).astype(np.float32) * np.float32(1e-3))).astype(np.float32)

    def product(coords_override=None, **opts):
        old = {k: lib.vmd_set_option(("spec_" + k).encode(), v) for k, v in opts.items()}
        coords = coords_override if coords_override is not None else coords_default
        try:
            ir = V.ScriptIR(lib)
            ir.add_rdf("g", o, o, 6.0)                                   # same set: half-shell pass
            ir.add_rdf("h", o[: o.size // 2], o, 6.0)                    # overlapping sets: full-shell pass
            ir.add_sdf("v", structures, tgt, 8.0)
            ir.add_distance("d", structures[0], structures[1], L.DIST_COM)
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(V.MolSystem(N, mass=mass, unitcell=cell), V.HostTrajectory(coords, cell), 0, F)
            out = {k: (ev.property_data(k).counts.copy() if k != "d" else None, np.array(ev.property_data(k).values)) for k in "ghvd"}
            out["hw"] = np.array(ev.property_data("h").weights64)
            return out
        finally:
            for k, v in old.items():
                lib.vmd_set_option(("spec_" + k).encode(), v)

    base = product()
    # --- D-RDF-OPEN
    got = product(rdf_closed=1)
    old = oracle.set_spec("rdf_closed", 1)
    try:
        want_g, _ = oracle_rdf(oracle, coords, ocell, o, o, 0.0, 6.0)
        want_h, _ = oracle_rdf(oracle, coords, ocell, o[: o.size // 2], o, 0.0, 6.0)
    finally:
        oracle.set_spec("rdf_closed", old)
    np.testing.assert_array_equal(got["g"][0], want_g)
    np.testing.assert_array_equal(got["h"][0], want_h)
    assert got["g"][0][0] == base["g"][0][0] + F * (o.size + 2)          # self pairs + the coincident pair (700, 703), both orders
    assert got["g"][0].sum() > base["g"][0].sum() and got["h"][0].sum() > base["h"][0].sum()
    np.testing.assert_array_equal(base["g"][0], oracle_rdf(oracle, coords, ocell, o, o, 0.0, 6.0)[0])
    # --- D-SDF-EXCL
    got = product(sdf_include_self=1)
    old = oracle.set_spec("sdf_include_self", 1)
    try:
        want_v, _ = oracle_sdf(oracle, coords, ocell, structures, mass, tgt, 8.0)
    finally:
        oracle.set_spec("sdf_include_self", old)
    np.testing.assert_array_equal(got["v"][0], want_v.reshape(-1))
    assert got["v"][0].sum() > base["v"][0].sum()
    # --- D-SDF-NORM: the float view becomes a number density, the integer accumulators stay what they are
    got = product(sdf_density=1)
    np.testing.assert_array_equal(got["v"][0], base["v"][0])
    edge = 2.0 * 8.0 / 128
    scale = np.float32(1.0 / (F * edge ** 3))
    np.testing.assert_array_equal(got["v"][1], base["v"][0].astype(np.float32) * scale)
    assert got["v"][1].max() > 0 and not np.array_equal(got["v"][1], base["v"][1])
    # --- D-RDF-NORM: the normalisation weights follow the setting, the counts never do
    for mode in (1, 2):
        got = product(rdf_norm=mode)
        old = oracle.set_spec("rdf_norm", mode)
        try:
            _, want_w = oracle_rdf(oracle, coords, ocell, o[: o.size // 2], o, 0.0, 6.0)
        finally:
            oracle.set_spec("rdf_norm", old)
        np.testing.assert_array_equal(got["h"][0], base["h"][0])
        np.testing.assert_allclose(got["hw"], want_w, rtol=1e-12)
        assert not np.allclose(got["hw"], base["hw"], rtol=1e-3)
    # --- D-WRAP: positions as they are (here: shifted out of the cell by whole and fractional box lengths), minimum image by rounding
    got = product(rdf_raw=1, coords_override=shifted)
    old = oracle.set_spec("rdf_raw", 1)
    try:
        want_g, _ = oracle_rdf(oracle, shifted, ocell, o, o, 0.0, 6.0)
        want_h, _ = oracle_rdf(oracle, shifted, ocell, o[: o.size // 2], o, 0.0, 6.0)
    finally:
        oracle.set_spec("rdf_raw", old)
    np.testing.assert_array_equal(got["g"][0], want_g)
    np.testing.assert_array_equal(got["h"][0], want_h)
    wrapped_g = product(coords_override=shifted)["g"][0]
    np.testing.assert_array_equal(wrapped_g, oracle_rdf(oracle, shifted, ocell, o, o, 0.0, 6.0)[0])
    assert abs(int(got["g"][0].sum()) - int(wrapped_g.sum())) < 1e-4 * wrapped_g.sum()      # the same physics ...
    assert not np.array_equal(got["g"][0], wrapped_g)                                       # ... in different roundings: some pair changes its bin
    # --- D-DIST-COM: geometric centres = the oracle's centre of mass with unit masses
    got = product(dist_geometric_com=1)
    want_d = oracle_distance(oracle, coords, ocell, np.ones_like(mass), structures[0], structures[1], L.DIST_COM)
    np.testing.assert_array_equal(got["d"][1].reshape(F, -1), want_d)
    assert not np.array_equal(got["d"][1], base["d"][1])


def bonded_ring_case(O, frames=4, box=24.0, seed=3):
    """D-SDF-UNWRAP with bonds (VERDICT r02 missing #3): K ring-shaped ligands of m = 8 atoms whose INDEX order jumps across the ring
    (0, 4, 1, 5, 2, 6, 3, 7 around the circle: index neighbours sit on opposite sides), each ring straddling a face of the cell.
    Chained along the index order a hop of more than half the cell would be folded the wrong way; along the bonds every hop is one
    bond length.  -> (coords [F,3,N], structures [K,m], mass, bonds [nb,2], targets, whole [F,K,m,3] = the rings before wrapping)"""
    rng = np.random.default_rng(seed)
    K, m, radius = 3, 8, 7.5                      # ring diameter 15 A > box / 2 = 12 A: opposite atoms are more than half a cell apart
    perm = np.array([0, 4, 1, 5, 2, 6, 3, 7])     # ring position of local atom a
    nwater = 600
    N = K * m + nwater
    coords = np.zeros((frames, 3, N), np.float32)
    whole = np.zeros((frames, K, m, 3))
    structures = np.arange(K * m, dtype=np.int32).reshape(K, m)
    bonds = []
    for k in range(K):
        pos_of = {int(perm[a]): a for a in range(m)}                  # ring position -> local atom
        for r in range(m):
            bonds.append((k * m + pos_of[r], k * m + pos_of[(r + 1) % m]))
    bonds = np.array(bonds, np.int32)
    rng.shuffle(bonds)
    ang = 2 * np.pi * perm / m
    base = np.stack([radius * np.cos(ang), radius * np.sin(ang), 0.6 * np.sin(3 * ang)], 1)      # a puckered, asymmetric ring
    base[:, 0] *= 1.15
    for f in range(frames):
        for k in range(K):
            a, b = rng.normal(size=3), rng.normal(size=3)
            a /= np.linalg.norm(a); b -= a * (a @ b); b /= np.linalg.norm(b)
            Rm = np.stack([a, b, np.cross(a, b)], 1)
            centre = np.array([0.3, box / 2, box / 2]) if k == 0 else rng.uniform(0, box, 3)     # ring 0 always straddles the x = 0 face
            pts = base @ Rm.T + centre + rng.normal(0, 0.05, (m, 3))
            coords[f, :, k * m:(k + 1) * m] = np.mod(pts, box).T
            whole[f, k] = pts
        coords[f, :, K * m:] = rng.uniform(0, box, (3, nwater))
    mass = np.ones(N, np.float32)
    mass[:K * m] = np.tile(rng.uniform(1.0, 16.0, m).astype(np.float32), K)      # the same weights in every ring: rigid copies align exactly
    return coords, structures, mass, bonds, np.arange(K * m, N, dtype=np.int32), whole


def check_bonded_unwrap(lib, O):
    coords, structures, mass, bonds, tgt, whole = bonded_ring_case(O)
    F, _, N = coords.shape
    ocell, vcell = cell_pair(O, 24.0, L.PBC_ALL)

    def product(with_bonds):
        ir = V.ScriptIR(lib)
        ir.add_sdf("v", structures, tgt, 9.0)
        ev = V.ScriptEval(F, ir)
        sysm = V.MolSystem(N, mass=mass, unitcell=vcell, bonds=bonds if with_bonds else None)
        assert ev.frame_range(sysm, V.HostTrajectory(coords, vcell), 0, F)
        mats = np.stack([ev.sdf_matrices("v", sysm, V.HostTrajectory(coords, vcell), f)[0] for f in range(F)])
        return ev.property_data("v").counts.copy(), mats

    with O.unwrap_tree(bonds, structures) as tree:
        assert (tree.parent[:, 0] == -1).all() and (tree.parent[:, 1:] >= 0).all()
        want_vol, want_M = oracle_sdf(O, coords, ocell, structures, mass, tgt, 9.0)
    plain_vol, plain_M = oracle_sdf(O, coords, ocell, structures, mass, tgt, 9.0)
    got_vol, got_M = product(True)
    np.testing.assert_array_equal(got_vol, want_vol)                       # bonds handed over: the bond-tree unwrap, bit for bit
    got_plain, _ = product(False)
    np.testing.assert_array_equal(got_plain, plain_vol)                    # without bonds: the index chain, as before
    assert not np.array_equal(want_vol, plain_vol)                         # and the two really differ on this system
    np.testing.assert_allclose(got_M[:, :, :3, :], want_M, atol=2e-4)
    # known answer, no oracle: every ring is a rigid copy of one template (+ 0.05 A of jitter), so with the rings made whole correctly
    # the world->reference matrices carry every (frame, ring) onto the pose of ring 0 at frame 0; a ring folded the wrong way is off
    # by Angstroms.  The whole ring is placed on the periodic image whose root atom is the wrapped one (where the unwrap starts).
    def aligned(M):
        out = np.zeros(whole.shape)
        for f in range(F):
            for k in range(structures.shape[0]):
                root = coords[f, :, structures[k, 0]].astype(np.float64)
                pts = whole[f, k] + (root - whole[f, k, 0])
                out[f, k] = pts @ M[f, k][:3, :3].T + M[f, k][:3, 3]
        return out
    q = aligned(got_M)
    assert np.sqrt(((q - q[0, 0]) ** 2).sum(-1)).max() < 0.5
    _, chain_M = product(False)
    qc = aligned(chain_M)
    assert np.sqrt(((qc - qc[0, 0]) ** 2).sum(-1)).max() > 3.0              # the index chain tears at least one of these rings apart
    return want_vol.sum()
