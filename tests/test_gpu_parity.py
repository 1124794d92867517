"""Parity tests proper: the hipcc-built product library on a real MI355X, through the C ABI, against the oracle.
Integer results must be bit-identical; normalised g(r) within 1e-5 relative (BASELINE.json north_star)."""
import threading

import numpy as np
import pytest

import cases
import viamd_amd as V
from viamd_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def box30k(oracle):
    return cases.water_box(oracle, 7, 30000, 80.0, 4)


def test_device_sqrt_and_division_are_correctly_rounded(gpu_lib, oracle):
    """SPEC S3/S4 lean on IEEE sqrtf and 1/x on the device: a distance_pair row exposes sqrtf(d2) bit for bit."""
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 30, (2, 3, 400)).astype(np.float32)
    cases.check_distances(gpu_lib, oracle, c, None, np.ones(400, np.float32),
                          [("p", np.arange(0, 200), np.arange(200, 400), L.DIST_PAIR)], flags=0)


def test_rdf_same_set_half_shell(gpu_lib, oracle, box30k):
    o = cases.oxygen(30000)
    cases.check_rdf(gpu_lib, oracle, box30k, 80.0, [("goo", o, o, 0.0, 12.0)], device=True)


def test_rdf_two_sets_ranges_shared_grid(gpu_lib, oracle, box30k):
    o, h = cases.oxygen(30000), cases.hydrogen(30000)
    cases.check_rdf(gpu_lib, oracle, box30k[:2], 80.0,
                    [("goh", o, h, 0.0, 12.0), ("goo", o, o, 0.0, 12.0), ("ring", h, o, 2.5, 9.0)], device=True)


def test_all_cell_build_paths(gpu_lib, oracle, box30k):
    cases.cell_build_cases(gpu_lib, oracle, box30k, 80.0, device=True)


def test_cell_build_bucket_overflow_is_caught_and_repeated(gpu_lib, oracle):
    cases.cell_build_overflow_case(gpu_lib, oracle, device=True, n=30000, box=80.0)


def test_batch_of_frame_blocks_overflow_is_all_or_nothing(gpu_lib, oracle):
    cases.blocks_overflow_case(gpu_lib, oracle, device=True, n=30000, box=80.0)


def test_a_wandering_solute_does_not_cost_the_solvent_its_cell_build(gpu_lib, oracle):
    cases.wandering_solute_case(gpu_lib, oracle, device=True, n=30000, box=80.0, F=24, batch=4)


def test_pool_calls_are_served_by_read_ahead(gpu_lib, oracle):
    cases.readahead_case(gpu_lib, oracle, device=True, n_water=6000, box=58.0, F=160, nthreads=12)


def test_pool_threads_with_small_ranges_leave_the_views_of_one_call(gpu_lib, oracle):
    cases.pool_threads_case(gpu_lib, oracle, device=True, n_water=9000, box=66.0, F=96, nthreads=12,
                            combos=((150, 1, 1), (150, 1, 3), (0, 1, 1), (0, 0, 2), (150, 0, 1)))


def test_eval_life_cycles_reuse_cached_blocks_streams_and_events(gpu_lib, oracle):
    cases.resource_cache_case(gpu_lib, oracle, device=True, n_water=9000, box=66.0)


def test_coevaluated_rdfs_share_pair_passes(gpu_lib, oracle):
    cases.class_decomposition_cases(gpu_lib, oracle, device=True, n_water=30000, box=70.0)


@pytest.mark.parametrize("variant,shist", [(0, 0), (2, 0), (0, 1), (3, 0)])
def test_rdf_hit_compaction_variants(gpu_lib, oracle, box30k, variant, shist):
    """variant 0: one compaction per candidate column; variant 2: pair entries (two columns share one compaction; hand-scheduled
    push / pop of their own) - same / different sets, r_min > 0, a thin shell at the cutoff, edge cases, triclinic and open cells,
    and the BASELINE config 2 shape"""
    o, h = cases.oxygen(30000), cases.hydrogen(30000)
    old = gpu_lib.vmd_set_option(b"rdf_variant", variant)
    old_sh = gpu_lib.vmd_set_option(b"rdf_shared_hist", shist)      # one LDS histogram per block instead of one per wave
    try:
        cases.check_rdf(gpu_lib, oracle, box30k[:3], 80.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0),
                                                          ("shell", o, o, 11.5, 12.0)], device=True)
        cases.rdf_edge_cases(gpu_lib, oracle, device=True)
        cases.triclinic_cases(gpu_lib, oracle, 3000, device=True)
        cases.open_boundary_cases(gpu_lib, oracle, 6000, device=True)
        N, F = 100002, 2
        t = V.DeviceTrajectory(F, N)
        t.synth(2, 100.0, 0.05)
        oo = cases.oxygen(N)
        ir = V.ScriptIR(); ir.add_rdf("g", oo, oo, 12.0)
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(V.MolSystem(N), t, 0, F)
        coords = np.stack([oracle.synth_frame(2, N, 100.0, 0.05, f) for f in range(F)])
        counts, _ = cases.oracle_rdf(oracle, coords, oracle.make_cell(100.0), oo, oo, 0.0, 12.0)
        np.testing.assert_array_equal(ev.property_data("g").counts, counts)
    finally:
        gpu_lib.vmd_set_option(b"rdf_variant", old)
        gpu_lib.vmd_set_option(b"rdf_shared_hist", old_sh)


@pytest.mark.parametrize("pop,shist", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_rdf_pop_variants(gpu_lib, oracle, box30k, pop, shist):
    """how k_rdf_pencil drains its hit stack when r_min == 0: 0 = the 9-instruction pop, 1 = margin folded into the constant + spare bin, the
    stack read with ds_read_addtid_b32 (M0 + 4 * lane; the default) - bit-identical counts, incl. BASELINE config 2's shape.  The exact path
    behind both is the out-of-line vmd_slow_flush_call (the thin shell and the edge cases put most hits on it)"""
    o, h = cases.oxygen(30000), cases.hydrogen(30000)
    old = gpu_lib.vmd_set_option(b"rdf_pop", pop)
    old_sh = gpu_lib.vmd_set_option(b"rdf_shared_hist", shist)      # 1 (default): one histogram per block, eight waves per SIMD
    try:
        cases.check_rdf(gpu_lib, oracle, box30k[:3], 80.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0),
                                                          ("shell", o, o, 11.5, 12.0)], device=True)
        cases.rdf_edge_cases(gpu_lib, oracle, device=True)
        N, F = 100002, 3
        t = V.DeviceTrajectory(F, N)
        t.synth(2, 100.0, 0.05)
        oo = cases.oxygen(N)
        ir = V.ScriptIR(); ir.add_rdf("g", oo, oo, 12.0)
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(V.MolSystem(N), t, 0, F)
        coords = np.stack([oracle.synth_frame(2, N, 100.0, 0.05, f) for f in range(F)])
        counts, _ = cases.oracle_rdf(oracle, coords, oracle.make_cell(100.0), oo, oo, 0.0, 12.0)
        np.testing.assert_array_equal(ev.property_data("g").counts, counts)
    finally:
        gpu_lib.vmd_set_option(b"rdf_pop", old)
        gpu_lib.vmd_set_option(b"rdf_shared_hist", old_sh)


def test_rdf_inline_variant_and_host_staging(gpu_lib, oracle, box30k):
    o = cases.oxygen(30000)
    cases.check_rdf(gpu_lib, oracle, box30k[:2], 80.0, [("goo", o, o, 0.0, 12.0)], variant=1, device=False)


def test_staged_host_trajectories_overlap_pipeline(gpu_lib, oracle, box30k):
    """load_frame staging and the pinned host view, with tiny batches so that both pipeline stages are used many times"""
    o, h = cases.oxygen(30000), cases.hydrogen(30000)
    old = gpu_lib.vmd_set_option(b"batch_frames", 1)
    try:
        cases.check_rdf(gpu_lib, oracle, box30k, 80.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 8.0)], device="pinned")
        cases.check_rdf(gpu_lib, oracle, box30k, 80.0, [("goo", o, o, 0.0, 12.0)], device=False)
    finally:
        gpu_lib.vmd_set_option(b"batch_frames", old)
    cases.check_rdf(gpu_lib, oracle, box30k, 80.0, [("goo", o, o, 0.0, 12.0)], device="pinned")


def test_rdf_small_boxes_and_unwrapped_input(gpu_lib, oracle):
    c = cases.water_box(oracle, 11, 3000, 26.0, 3)                    # ny = nz = 2
    o = cases.oxygen(3000)
    cases.check_rdf(gpu_lib, oracle, c, 26.0, [("goo", o, o, 0.0, 12.0), ("gall", o, np.arange(3000), 0.0, 12.0)], device=True)
    rng = np.random.default_rng(5)
    box = (50.0, 38.0, 64.0)
    c = (rng.uniform(-1.0, 2.0, (2, 3, 9000)) * np.array(box)[None, :, None]).astype(np.float32)
    a = np.arange(0, 9000, 2)
    cases.check_rdf(gpu_lib, oracle, c, box, [("g", a, a, 0.0, 9.0), ("gx", a, np.arange(1, 9000, 2), 0.5, 11.0)])


def test_rdf_box_changes_every_frame(gpu_lib, oracle):
    rng = np.random.default_rng(8)
    boxes = [60.0, 63.5, 58.2, (61.0, 59.0, 64.0), 60.5]
    c = np.stack([(rng.uniform(0, 1, (3, 12000)) * np.array(b if not np.isscalar(b) else (b,) * 3)[:, None]).astype(np.float32) for b in boxes])
    a = np.arange(0, 12000, 2)
    cases.check_rdf(gpu_lib, oracle, c, boxes, [("g", a, a, 0.0, 11.0), ("gx", a, np.arange(1, 12000, 2), 0.0, 9.0)], device=True)


def test_rdf_edge_cases(gpu_lib, oracle):
    cases.rdf_edge_cases(gpu_lib, oracle, device=True)


def test_rdf_brute_paths(gpu_lib, oracle):
    rng = np.random.default_rng(3)
    c = rng.uniform(0, 20, (5, 3, 700)).astype(np.float32)
    a, b = np.arange(0, 700, 2), np.arange(700)
    cases.check_rdf(gpu_lib, oracle, c, None, [("g", a, b, 0.0, 10.0)], oracle_method="brute")
    cases.check_rdf(gpu_lib, oracle, c, 20.0, [("g", a, a, 0.0, 10.0)], oracle_method="brute")
    cases.check_rdf(gpu_lib, oracle, c, 20.0, [("g", a, b, 1.0, 6.0)], flags=3, oracle_method="brute")


def test_c1_standin_chain_no_pbc(gpu_lib, oracle):
    """BASELINE config 1 stand-in (the 1ALA-500 file is missing): 112-atom chain, 500 frames, no cell,
    script rdf(element('O'), element('O'), 10.0)."""
    rng = np.random.default_rng(1)
    base = np.cumsum(rng.normal(0, 0.9, (112, 3)), axis=0)
    c = (base[None] + rng.normal(0, 0.3, (500, 112, 3))).transpose(0, 2, 1).astype(np.float32)
    o = np.arange(3, 112, 10)
    cases.check_rdf(gpu_lib, oracle, c, None, [("r", o, o, 0.0, 10.0)], oracle_method="brute")


def test_rdf_sharding_threads_and_reevaluation(gpu_lib, oracle, box30k):
    o = cases.oxygen(30000)
    ev = cases.check_rdf(gpu_lib, oracle, box30k, 80.0, [("goo", o, o, 0.0, 12.0)], device=True, ranges=[(3, 4), (0, 2), (2, 3)])
    ref = ev.property_data("goo").counts.copy()
    # re-entrant frame_range from pool threads with disjoint ranges on ONE eval (src/main.cpp:993-997)
    vcell = V.make_unitcell(80.0)
    traj = V.DeviceTrajectory(4, 30000)
    traj.upload(box30k, vcell)
    ev.clear_data()
    assert ev.property_data("goo").counts.sum() == 0
    sysm = V.MolSystem(30000, unitcell=vcell)
    ths = [threading.Thread(target=lambda b=b: ev.frame_range(sysm, traj, b, b + 1)) for b in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert ev.frames_done() == 4
    np.testing.assert_array_equal(ev.property_data("goo").counts, ref)


def test_full_and_filtered_eval_run_concurrently(gpu_lib, oracle, box30k):
    """VIAMD's "Eval Full" + "Eval Filt" (src/main.cpp:982-1039): two evals from ONE ir, the filtered one over a sub-range,
    both running at the same time from different pool threads."""
    o = cases.oxygen(30000)
    ir = V.ScriptIR(); ir.add_rdf("goo", o, o, 12.0)
    full, filt = V.ScriptEval(4, ir), V.ScriptEval(4, ir)
    assert full.ir_fingerprint() == filt.ir_fingerprint() == ir.fingerprint()
    vcell = V.make_unitcell(80.0)
    traj = V.DeviceTrajectory(4, 30000); traj.upload(box30k, vcell)
    sysm = V.MolSystem(30000, unitcell=vcell)
    ths = [threading.Thread(target=lambda: full.frame_range(sysm, traj, 0, 4)),
           threading.Thread(target=lambda: filt.frame_range(sysm, traj, 1, 3))]
    [t.start() for t in ths]; [t.join() for t in ths]
    ocell = oracle.make_cell(80.0)
    ref_full, _ = cases.oracle_rdf(oracle, box30k, ocell, o, o, 0.0, 12.0)
    ref_filt, wf = cases.oracle_rdf(oracle, box30k, ocell, o, o, 0.0, 12.0, frames=(1, 2))
    np.testing.assert_array_equal(full.property_data("goo").counts, ref_full)
    np.testing.assert_array_equal(filt.property_data("goo").counts, ref_filt)
    np.testing.assert_allclose(filt.property_data("goo").weights64, wf, rtol=1e-12)
    assert list(filt.frame_mask()) == [0, 1, 1, 0] and filt.frames_done() == 2
    # moving the filter window: clear + re-evaluate only the new sub-range (what the timeline slider triggers)
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 2, 4)
    ref2, _ = cases.oracle_rdf(oracle, box30k, ocell, o, o, 0.0, 12.0, frames=(2, 3))
    np.testing.assert_array_equal(filt.property_data("goo").counts, ref2)


def test_synth_kernel_matches_oracle_generator(gpu_lib, oracle):
    t = V.DeviceTrajectory(3, 100002)
    t.synth(2, 100.0, 0.05)
    for f in (0, 2):
        got, cell = t.download_frame(f)
        np.testing.assert_array_equal(got, oracle.synth_frame(2, 100002, 100.0, 0.05, f))


def test_config2_shape_against_oracle(gpu_lib, oracle):
    """BASELINE config 2 shape (100 002 atoms, L = 100, O-O, r_c = 12) on a few frames, device-generated trajectory."""
    N, F = 100002, 3
    t = V.DeviceTrajectory(F, N)
    t.synth(2, 100.0, 0.05)
    o = cases.oxygen(N)
    ir = V.ScriptIR()
    ir.add_rdf("goo", o, o, 12.0)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(N), t, 0, F)
    coords = np.stack([oracle.synth_frame(2, N, 100.0, 0.05, f) for f in range(F)])
    counts, weights = cases.oracle_rdf(oracle, coords, oracle.make_cell(100.0), o, o, 0.0, 12.0)
    pd = ev.property_data("goo")
    np.testing.assert_array_equal(pd.counts, counts)
    g = V.downsample_histogram(pd.values, pd.weights, 128)
    assert abs(g[64:].mean() - 1.0) < 0.02                         # ideal-gas-like O-O: g -> 1
    assert 7.9e6 * F < counts.sum() < 8.2e6 * F                    # 8.04e6 ordered pairs per frame


def test_full_size_properties_config3(gpu_lib, oracle):
    """BASELINE config 3 size (1M atoms, heavy = 333 334 O): size-independent properties at full size + one frame
    checked against the oracle."""
    N, F = 1000002, 4
    t = V.DeviceTrajectory(F, N)
    t.synth(3, 215.443, 0.05)
    o = cases.oxygen(N)
    ir = V.ScriptIR()
    ir.add_rdf("g", o, o, 12.0)
    whole, parts = V.ScriptEval(F, ir), V.ScriptEval(F, ir)
    sysm = V.MolSystem(N)
    assert whole.frame_range(sysm, t, 0, F)
    for b, e in ((2, 4), (0, 1), (1, 2)):
        assert parts.frame_range(sysm, t, b, e)
    c = whole.property_data("g").counts
    np.testing.assert_array_equal(c, parts.property_data("g").counts)       # any sharding -> identical integers
    assert (c % 2 == 0).all()                                                # unordered pairs counted twice
    assert 7.9e7 * F < c.sum() < 8.2e7 * F                                   # 8.04e7 ordered pairs / frame
    one = V.ScriptEval(F, ir)
    assert one.frame_range(sysm, t, 1, 2)
    f1 = oracle.synth_frame(3, N, 215.443, 0.05, 1)
    ref, _ = oracle.rdf_frame(f1[0], f1[1], f1[2], oracle.make_cell(215.443), o, o, 0.0, 12.0, method="cells")
    np.testing.assert_array_equal(one.property_data("g").counts, ref)


def test_sdf_volume_matrices_and_sharding(gpu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 4, 30000, 70.0, 6, K=7, m=10)
    n_s = structures.size
    tgt = np.arange(n_s, coords.shape[2], 3, dtype=np.int32)
    _, vol = cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, tgt, 10.0, device=True)
    assert vol.sum() > 6 * 7 * 150
    cases.check_sdf(gpu_lib, oracle, coords[:3], 70.0, structures, mass, np.arange(coords.shape[2], dtype=np.int32), 6.0,
                    ranges=[(1, 3), (0, 1)])


def test_sdf_sparse_and_dense_target_paths(gpu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 14, 30000, 70.0, 3, K=7, m=10)
    n_s, N = structures.size, coords.shape[2]
    dense = np.arange(n_s, N, 3, dtype=np.int32)
    sparse = np.arange(n_s, N, 30, dtype=np.int32)
    for flag in (0, 1):
        old = gpu_lib.vmd_set_option(b"sdf_dense", flag)
        try:
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, dense, 10.0, device=True)
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, np.arange(N, dtype=np.int32), 6.0, device=True)
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, sparse, 10.0)
        finally:
            gpu_lib.vmd_set_option(b"sdf_dense", old)
    # the scatter's target addressing: index list / arithmetic progression generated on the device, 4 / 8 atoms per thread,
    # targets with and without owners, an irregular list (no progression)
    irregular = np.sort(np.random.default_rng(5).choice(np.arange(0, N, dtype=np.int32), N // 4, replace=False)).astype(np.int32)
    # rows: 16-byte row streaming for progressions; wv: 1 per-wave compaction, 2 / >= 16 the persistent streaming kernel (on that many blocks)
    for arith, ilp, rows, wv in ((0, 4, 0, 0), (1, 8, 0, 0), (0, 16, 0, 0), (1, 4, 1, 0), (1, 4, 4, 0), (1, 4, 0, 1), (1, 4, 0, 2), (0, 8, 0, 2), (1, 4, 0, 16), (0, 4, 0, 512)):
        old = gpu_lib.vmd_set_option(b"sdf_arith", arith), gpu_lib.vmd_set_option(b"sdf_ilp", ilp), gpu_lib.vmd_set_option(b"sdf_rows", rows)
        old_wv = gpu_lib.vmd_set_option(b"sdf_wave", wv)
        try:
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, dense, 10.0, device=True)
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, irregular, 10.0, device=True)
            cases.check_sdf(gpu_lib, oracle, coords, 70.0, structures, mass, np.arange(N, dtype=np.int32), 6.0, device=True)   # stride 1, owners among the targets
        finally:
            gpu_lib.vmd_set_option(b"sdf_arith", old[0]); gpu_lib.vmd_set_option(b"sdf_ilp", old[1]); gpu_lib.vmd_set_option(b"sdf_rows", old[2])
            gpu_lib.vmd_set_option(b"sdf_wave", old_wv)


def test_sdf_rigid_motion_invariance(gpu_lib, oracle):
    """frame 1 = frame 0 rigidly rotated + translated (no cell) -> the same volume twice."""
    rng = np.random.default_rng(6)
    m, nt = 9, 20000
    pts0 = np.concatenate([rng.normal(0, 2.0, (m, 3)), rng.uniform(-15, 15, (nt, 3))]).astype(np.float32)
    ang, ax = 1.1, np.array([0.3, -0.5, 0.8]); ax /= np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Q = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    pts1 = (pts0.astype(np.float64) @ Q.T + np.array([3.0, -7.0, 11.0])).astype(np.float32)
    coords = np.stack([pts0.T, pts1.T])
    structures = np.arange(m, dtype=np.int32)[None]
    tgt = np.arange(m, m + nt, dtype=np.int32)
    mass = rng.uniform(1, 16, m + nt).astype(np.float32)
    ev, vol = cases.check_sdf(gpu_lib, oracle, coords, None, structures, mass, tgt, 10.0, flags=0)
    ir = V.ScriptIR(); ir.add_sdf("v", structures, tgt, 10.0)
    e0 = V.ScriptEval(2, ir)
    traj = V.HostTrajectory(coords, V.make_unitcell(None))
    assert e0.frame_range(V.MolSystem(m + nt, mass=mass), traj, 0, 1)
    v0 = e0.property_data("v").counts.astype(np.int64)
    v1 = vol.astype(np.int64) - v0
    assert v0.sum() > 3000 and abs(int(v0.sum()) - int(v1.sum())) <= 5
    assert np.abs(v0 - v1).sum() <= 0.02 * v0.sum()


def test_distance_family_and_coevaluation(gpu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 9, 9000, 50.0, 12)
    specs = [("d", [3], [40], L.DIST_COM), ("dcom", structures[0], structures[1], L.DIST_COM),
             ("dmin", structures[0], structures[2], L.DIST_MIN), ("dmax", structures[1], structures[3], L.DIST_MAX),
             ("dpair", structures[0][:3], structures[1][:4], L.DIST_PAIR)]
    pops = [("pcom", [s[:1] for s in structures], [s[3:4] for s in structures], L.DIST_COM, "pop"),
            ("pmax", [s[:2] for s in structures], [s[2:] for s in structures], L.DIST_MAX, "pop"),
            ("ppair", [s[:2] for s in structures], [s[3:6] for s in structures], L.DIST_PAIR, "pop")]
    cases.check_distances(gpu_lib, oracle, coords, 50.0, mass, specs + pops, device=True, ranges=[(0, 5), (5, 12)])
    # BASELINE config 5 in miniature: 3 RDF + 1 SDF + 4 distance properties on one eval, one pass over the frames
    N = coords.shape[2]
    n_s = structures.size
    o = np.arange(n_s, N, 3, dtype=np.int32)
    h = np.array([i for i in range(n_s, N) if (i - n_s) % 3], dtype=np.int32)
    ir = V.ScriptIR()
    ir.add_rdf("goo", o, o, 12.0); ir.add_rdf("goh", o, h, 12.0); ir.add_rdf("ghh", h, h, 12.0)
    ir.add_sdf("v", structures, o, 10.0)
    for name, a, b, kind in specs[:4]:
        ir.add_distance(name, a, b, kind)
    ev = V.ScriptEval(12, ir)
    vcell = V.make_unitcell(50.0)
    traj = V.DeviceTrajectory(12, N); traj.upload(coords, vcell)
    assert ev.frame_range(V.MolSystem(N, mass=mass, unitcell=vcell), traj, 0, 12)
    ocell = oracle.make_cell(50.0)
    for name, a, b in (("goo", o, o), ("goh", o, h), ("ghh", h, h)):
        ref, _ = cases.oracle_rdf(oracle, coords, ocell, a, b, 0.0, 12.0)
        np.testing.assert_array_equal(ev.property_data(name).counts, ref)
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, structures, mass, o, 10.0)
    np.testing.assert_array_equal(ev.property_data("v").counts, vol)
    for name, a, b, kind in specs[:4]:
        ref = cases.oracle_distance(oracle, coords, ocell, mass, np.asarray(a, np.int32), np.asarray(b, np.int32), kind)
        np.testing.assert_array_equal(ev.property_data(name).values.reshape(12, -1), ref)


def test_interrupt_and_errors(gpu_lib, oracle, box30k):
    o = cases.oxygen(30000)
    ir = V.ScriptIR(); ir.add_rdf("goo", o, o, 12.0)
    ev = V.ScriptEval(4, ir)
    traj = V.HostTrajectory(box30k, V.make_unitcell(80.0))
    ev.interrupt()
    assert ev.frame_range(V.MolSystem(30000), traj, 0, 4) is False and ev.frames_done() == 0
    ev.clear_data()
    assert ev.frame_range(V.MolSystem(30000), traj, 0, 4) and ev.frames_done() == 4
    tri = V.make_unitcell(80.0, flags=3); tri.xy = 5.0                   # tilted but not periodic along z: rejected
    with pytest.raises(V.VmdError, match="triclinic"):
        ev2 = V.ScriptEval(4, ir)
        ev2.frame_range(V.MolSystem(30000), V.HostTrajectory(box30k, tri), 0, 1)
    ir2 = V.ScriptIR(); ir2.add_rdf("bad", [5], [40000], 5.0)
    with pytest.raises(V.VmdError, match="references atom"):
        V.ScriptEval(4, ir2).frame_range(V.MolSystem(30000), traj, 0, 1)


def test_nccl_single_rank_reduce_is_identity(gpu_lib, oracle, box30k):
    """The RCCL merge path on one GPU through torch's process group: a world_size-1 group makes reduce_eval a no-op, so the
    collective is exercised explicitly - the library's own communicator (vmd_comm_create, 1 rank) + vmd_eval_reduce: in-place
    ncclAllReduce on the device accumulators, packed fp64 all-reduce of the host parts, finalize."""
    import ctypes as C
    from viamd_amd import _lib as LL
    o = cases.oxygen(30000)
    ev = cases.check_rdf(gpu_lib, oracle, box30k[:2], 80.0, [("goo", o, o, 0.0, 12.0)], device=True)
    ref = ev.property_data("goo").counts.copy()
    w = ev.property_data("goo").weights64.copy()
    ident = np.zeros(LL.COMM_ID_BYTES, np.uint8)
    assert gpu_lib.vmd_comm_unique_id(ident.ctypes.data_as(LL.c_uint8_p)), gpu_lib.last_error()
    comm = gpu_lib.vmd_comm_create(1, 0, ident.ctypes.data_as(LL.c_uint8_p))
    assert comm, gpu_lib.last_error()
    try:
        assert gpu_lib.vmd_comm_size(comm) == 1 and gpu_lib.vmd_comm_rank(comm) == 0
        assert gpu_lib.vmd_eval_reduce(ev.h, gpu_lib.vmd_comm_collective(comm), None), gpu_lib.last_error()
    finally:
        gpu_lib.vmd_comm_destroy(comm)
    np.testing.assert_array_equal(ev.property_data("goo").counts, ref)
    np.testing.assert_array_equal(ev.property_data("goo").weights64, w)
    assert ev.frame_mask().all()


def test_synthetic_blob_system_device_equals_host_and_script_eval(gpu_lib, oracle):
    """BASELINE config 4 in miniature through the script front-end: device-generated waters + uploaded blob rows."""
    from viamd_amd import script, synth
    n_blob, n_atoms, box, F = 200, 200 + 30000, 70.0, 5
    topo = synth.water_box_topology(n_atoms, n_blob)
    traj = synth.make_device_trajectory(V, 12, n_atoms, box, F, n_blob)
    coords = cases.host_frames(oracle, 12, n_atoms, box, F, n_blob)
    for f in (0, F - 1):
        got, _ = traj.download_frame(f)
        np.testing.assert_array_equal(got, coords[f])
    ir, info = script.compile_script("s = residue(5:11); v = sdf(s, element('O') and water, 10.0);"
                                     "g = rdf(element('O') and water, element('O') and water, 12.0);", topo)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(n_atoms, mass=topo.mass, unitcell=V.make_unitcell(box)), traj, 0, F)
    ocell = oracle.make_cell(box)
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, info["v"]["structures"], topo.mass, info["v"]["target"], 10.0)
    np.testing.assert_array_equal(ev.property_data("v").counts, vol)
    ref, _ = cases.oracle_rdf(oracle, coords, ocell, info["g"]["ref"], info["g"]["target"], 0.0, 12.0)
    np.testing.assert_array_equal(ev.property_data("g").counts, ref)


def test_triclinic_cell_all_property_kinds(gpu_lib, oracle):
    cases.triclinic_cases(gpu_lib, oracle, n_water=9000, device=True)


def test_filtered_evaluation_reuses_block_partials(gpu_lib, oracle):
    cases.filtered_cases(gpu_lib, oracle, n_water=6000, device=True)
    cases.filtered_cases(gpu_lib, oracle, n_water=3000, device=False)


def test_filtered_evaluation_at_config2_shape(gpu_lib, oracle):
    """Size-independent property at the bench shape: the sum over any partition of the timeline into sub-ranges, each answered
    by a filtered eval (block partials + ragged ends), equals the full evaluation; a sub-range equals its plain evaluation."""
    N, box, F, S = 100002, 100.0, 96, 16
    traj = V.DeviceTrajectory(F, N)
    traj.synth(2, box, 0.05)
    ox = cases.oxygen(N)
    ir = V.ScriptIR(); ir.add_rdf("g", ox, ox, 12.0)
    sysm = V.MolSystem(N, unitcell=V.make_unitcell(box))
    full = V.ScriptEval(F, ir); full.set_block_frames(S)
    assert full.frame_range(sysm, traj, 0, F)
    total = full.property_data("g").counts.copy()
    filt = V.ScriptEval(F, ir); filt.set_source(full)
    plain = V.ScriptEval(F, ir)
    acc = np.zeros_like(total)
    for beg, end in [(0, 7), (7, 50), (50, 81), (81, 96)]:
        filt.clear_data(); plain.clear_data()
        assert filt.frame_range(sysm, traj, beg, end) and plain.frame_range(sysm, traj, beg, end)
        np.testing.assert_array_equal(filt.property_data("g").counts, plain.property_data("g").counts)
        np.testing.assert_allclose(filt.property_data("g").weights64, plain.property_data("g").weights64, rtol=1e-12)
        acc += filt.property_data("g").counts
    np.testing.assert_array_equal(acc, total)
    assert filt.frame_stats() == (15, 0)            # [81,96) lies inside block [80,96): plain evaluation
    filt.clear_data()
    assert filt.frame_range(sysm, traj, 7, 50)
    assert filt.frame_stats() == (9 + 2, 32)       # frames 7-15 and 48-49 computed, blocks [16,32) [32,48) reused


def test_triclinic_config2_shape_against_oracle(gpu_lib, oracle):
    """The 100k-atom water box evaluated in a sheared cell of the same volume: the pencil grid in fractional space against
    the oracle's all-pairs S3t evaluation (1.1e9 pair tests), one frame, plus grid == brute on the device for more frames."""
    N, box, tilt = 100002, 100.0, (30.0, -20.0, 25.0)
    ocell = oracle.make_cell(box, 7, tilt)
    vcell = V.make_unitcell(box, tilt=tilt)
    traj = V.DeviceTrajectory(3, N)
    traj.synth(2, box, 0.05)
    traj.set_cell(vcell)
    ox = cases.oxygen(N)
    ir = V.ScriptIR(); ir.add_rdf("g", ox, ox, 12.0)
    sysm = V.MolSystem(N, unitcell=vcell)
    ev = V.ScriptEval(3, ir)
    assert ev.frame_range(sysm, traj, 0, 1)
    xyz, _ = traj.download_frame(0)
    counts, hits = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], ocell, ox, ox, 0.0, 12.0, method="brute")
    assert hits > 7_000_000
    np.testing.assert_array_equal(ev.property_data("g").counts, counts)
    assert ev.frame_range(sysm, traj, 1, 3)
    grid = ev.property_data("g").counts.copy()
    old = gpu_lib.vmd_set_option(b"force_brute", 1)
    try:
        ev2 = V.ScriptEval(3, ir)
        assert ev2.frame_range(sysm, traj, 0, 3)
    finally:
        gpu_lib.vmd_set_option(b"force_brute", old)
    np.testing.assert_array_equal(ev2.property_data("g").counts, grid)


def test_open_boundaries_and_slabs_on_the_grid(gpu_lib, oracle):
    cases.open_boundary_cases(gpu_lib, oracle, 6000, device=True)
    cases.open_boundary_cases(gpu_lib, oracle, 1500, device=False)


def test_sdf_triclinic_spread_structures_regression(gpu_lib, oracle):
    cases.sdf_triclinic_spread_structures(gpu_lib, oracle, device=True)


def test_cube_and_table_export_from_a_gpu_evaluation(tmp_path, gpu_lib, oracle):
    """GPU-evaluated SDF -> vmd_export_cube (C++) -> read back == counts; vis payload with the structures."""
    cases.export_cases(gpu_lib, oracle, tmp_path, device=True, n_water=30000, box=70.0)


def test_filtered_eval_against_a_running_source(gpu_lib, oracle):
    cases.filtered_contention_case(gpu_lib, oracle, device=True, n=30000, box=80.0, F=24, S=4, rounds=25)


def test_resident_trajectory_changes_invalidate_cached_boxes(gpu_lib, oracle):
    cases.device_view_cache_case(gpu_lib, oracle, n=30000, box=80.0)
