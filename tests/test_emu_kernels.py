"""Kernel + host logic on the CPU SIMT emulator (tests/emu) against the oracle — small sizes, no GPU needed.

These run the *same* .hip/.cpp sources as the product, compiled by g++ against a fake hip_runtime.h; they prove the
indexing / segment enumeration / queue machinery / batching logic, not device numerics (that is `-m gpu`)."""
import numpy as np
import pytest

import cases
import viamd_amd as V
from viamd_amd import _lib as L


@pytest.fixture(scope="module")
def box3k(oracle):
    return cases.water_box(oracle, 7, 3000, 60.0, 3)


def test_rdf_same_set_half_shell(emu_lib, oracle, box3k):
    o = cases.oxygen(3000)
    cases.check_rdf(emu_lib, oracle, box3k, 60.0, [("goo", o, o, 0.0, 12.0)])


def test_rdf_two_sets_full_shell_and_range(emu_lib, oracle, box3k):
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    cases.check_rdf(emu_lib, oracle, box3k[:2], 60.0, [("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0)])


def test_all_cell_build_paths(emu_lib, oracle, box3k):
    cases.cell_build_cases(emu_lib, oracle, box3k, 60.0)


def test_cell_build_bucket_overflow_is_caught_and_repeated(emu_lib, oracle):
    cases.cell_build_overflow_case(emu_lib, oracle)


def test_batch_of_frame_blocks_overflow_is_all_or_nothing(emu_lib, oracle):
    cases.blocks_overflow_case(emu_lib, oracle)


def test_small_launches_split_by_neighbour_pencil(emu_lib, oracle, box3k):
    """rdf_nsplit: the work item of a small launch is (chunk, part of its neighbour pencils) - same pairs, same integers: half shell
    (5 parts), two sets (9 parts), an explicit part count that does not divide the neighbour count, split pencils."""
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    for n, extra in ((-1, {}), (3, {}), (7, {"pencil_split_y": 2})):
        old = emu_lib.vmd_set_option(b"rdf_nsplit", n)
        olde = {k: emu_lib.vmd_set_option(k.encode(), v) for k, v in extra.items()}
        try:
            cases.check_rdf(emu_lib, oracle, box3k[:2], 60.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 1.0, 10.0)])
        finally:
            emu_lib.vmd_set_option(b"rdf_nsplit", old)
            for k, v in olde.items(): emu_lib.vmd_set_option(k.encode(), v)


def test_pool_threads_with_small_ranges_leave_the_views_of_one_call(emu_lib, oracle):
    cases.pool_threads_case(emu_lib, oracle)


def test_a_wandering_solute_does_not_cost_the_solvent_its_cell_build(emu_lib, oracle):
    cases.wandering_solute_case(emu_lib, oracle)


def test_pool_calls_are_served_by_read_ahead(emu_lib, oracle):
    cases.readahead_case(emu_lib, oracle)


def test_eval_life_cycles_reuse_cached_blocks_streams_and_events(emu_lib, oracle):
    cases.resource_cache_case(emu_lib, oracle)


def test_coevaluated_rdfs_share_pair_passes(emu_lib, oracle):
    cases.class_decomposition_cases(emu_lib, oracle, n_water=1800, box=38.0)


def test_rdf_inline_variant_matches(emu_lib, oracle, box3k):
    o = cases.oxygen(3000)
    cases.check_rdf(emu_lib, oracle, box3k[:1], 60.0, [("goo", o, o, 0.0, 12.0)], variant=1)


@pytest.mark.parametrize("variant,shist", [(0, 0), (2, 0), (0, 1), (3, 0)])
def test_rdf_hit_compaction_variants(emu_lib, oracle, box3k, variant, shist):
    """variant 0: one compaction per candidate column; variant 2: pair entries (two columns share one compaction, the partner that
    is no hit is dropped by the binning); variant 3: variant 0 behind a bounding-box test of the j windows - same sets, ranges with
    r_min > 0, own-pencil half shell, edge cases, triclinic and open cells"""
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    old = emu_lib.vmd_set_option(b"rdf_variant", variant)
    old_sh = emu_lib.vmd_set_option(b"rdf_shared_hist", shist)      # one LDS histogram per block instead of one per wave
    try:
        cases.check_rdf(emu_lib, oracle, box3k[:2], 60.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0),
                                                          ("shell", o, o, 11.5, 12.0)], variant=variant)
        cases.rdf_edge_cases(emu_lib, oracle)
        cases.triclinic_cases(emu_lib, oracle, 600)
        if variant == 3:
            cases.open_boundary_cases(emu_lib, oracle, 600)       # incl. a system kilo-Angstroms from the origin (reach of the box test)
    finally:
        emu_lib.vmd_set_option(b"rdf_variant", old)
        emu_lib.vmd_set_option(b"rdf_shared_hist", old_sh)


@pytest.mark.parametrize("pop,shist", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_rdf_pop_variants(emu_lib, oracle, box3k, pop, shist):
    """how the hit stack is drained when r_min == 0: the 9-instruction pop against the one with the margin folded into the constant and a
    spare bin instead of the range compare (the C++ twins here; asm against twins on the GPU).  A cutoff-sized last bin, a thin shell
    (r_min > 0 falls back to the old pop), few bins, edge cases"""
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    old = emu_lib.vmd_set_option(b"rdf_pop", pop)
    old_sh = emu_lib.vmd_set_option(b"rdf_shared_hist", shist)      # 1 (default): one histogram per block, eight waves per SIMD
    try:
        cases.check_rdf(emu_lib, oracle, box3k[:2], 60.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0),
                                                          ("shell", o, o, 11.5, 12.0)])
        cases.rdf_edge_cases(emu_lib, oracle)
    finally:
        emu_lib.vmd_set_option(b"rdf_pop", old)
        emu_lib.vmd_set_option(b"rdf_shared_hist", old_sh)


@pytest.mark.parametrize("sy,sz", [(2, 1), (3, 2)])
def test_rdf_split_pencils(emu_lib, oracle, box3k, sy, sz):
    """A/B tile shape: pencils of cross-section rmax/split, neighbour reach = split, x windows of the outer neighbours shrunk -
    the same counts as the r_max pencils (ortho periodic incl. r_min > 0 and a box with few pencils, triclinic, open / slab cells)"""
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    oy, oz = emu_lib.vmd_set_option(b"pencil_split_y", sy), emu_lib.vmd_set_option(b"pencil_split_z", sz)
    try:
        cases.check_rdf(emu_lib, oracle, box3k[:1], 60.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.0, 10.0), ("ring", h, o, 2.5, 9.0)])
        c = cases.water_box(oracle, 11, 1500, 26.0, 1)      # ny = nz = 2 at split 1: every reach wraps onto the same few pencils
        cases.check_rdf(emu_lib, oracle, c, 26.0, [("goo", cases.oxygen(1500), cases.oxygen(1500), 0.0, 12.0)], oracle_method="brute")
        cases.rdf_edge_cases(emu_lib, oracle)
        if sy == 2:          # the other cell kinds once (the random scenarios of test_fuzz_emu.py draw split walks as well)
            cases.triclinic_cases(emu_lib, oracle, 300)
            cases.open_boundary_cases(emu_lib, oracle, 300)
    finally:
        emu_lib.vmd_set_option(b"pencil_split_y", oy); emu_lib.vmd_set_option(b"pencil_split_z", oz)


def test_dense_lane_selections_walk_half_width_pencils_by_default(emu_lib, oracle):
    """Round 5: with `pencil_split_y` at its default (0 = by density) a group whose passes all put >= 0.08 atoms / A^3 in the lanes - every
    atom of a liquid, SURVEY 8d's C3-dense - is walked on half-width pencils in y; sparser scripts keep the r_max pencils.  Either way the
    oracle's integers; which grid was walked shows in the candidate-column count."""
    assert emu_lib.vmd_set_option(b"pencil_split_y", 0) == 0          # the default
    c = cases.water_box(oracle, 17, 3000, 30.0, 2)                      # 3 000 atoms in 27 000 A^3: 0.11 / A^3
    everything = np.arange(3000, dtype=np.int32)

    def columns(split, props):
        old = emu_lib.vmd_set_option(b"pencil_split_y", split)
        try:
            emu_lib.vmd_hip_rdf_columns(1)
            cases.check_rdf(emu_lib, oracle, c, 30.0, props)
            return int(emu_lib.vmd_hip_rdf_columns(1))
        finally:
            emu_lib.vmd_set_option(b"pencil_split_y", old)

    dense = [("gaa", everything, everything, 0.0, 7.0)]
    auto, fixed1, fixed2 = columns(0, dense), columns(1, dense), columns(2, dense)
    assert auto == fixed2 and fixed2 != fixed1, (auto, fixed1, fixed2)       # (which of the two walks fewer columns depends on the size: 5 % fewer for 10^6 atoms on the MI355X)
    sparse = [("goo", cases.oxygen(3000), cases.oxygen(3000), 0.0, 7.0)]       # 0.037 / A^3 in the lanes
    assert columns(0, sparse) == columns(1, sparse)
    mixed = dense + [("goh", cases.oxygen(3000), cases.hydrogen(3000), 0.0, 7.0)]   # one sparse pass in the group keeps the r_max pencils for all
    assert columns(0, mixed) == columns(1, mixed)


def test_rdf_two_pencils_per_axis(emu_lib, oracle):
    # L = 26, rc = 12 -> ny = nz = 2: both neighbour offsets map to the same pencil with different images
    c = cases.water_box(oracle, 11, 1500, 26.0, 2)
    o = cases.oxygen(1500)
    cases.check_rdf(emu_lib, oracle, c, 26.0, [("goo", o, o, 0.0, 12.0), ("gall", o, np.arange(1500), 0.0, 12.0)],
                    oracle_method="brute")


def test_rdf_noncubic_box_and_unwrapped_input(emu_lib, oracle):
    rng = np.random.default_rng(5)
    box = (50.0, 38.0, 64.0)
    c = (rng.uniform(-1.0, 2.0, (2, 3, 2400)) * np.array(box)[None, :, None]).astype(np.float32)  # outside the cell too
    a = np.arange(0, 2400, 2)
    cases.check_rdf(emu_lib, oracle, c, box, [("g", a, a, 0.0, 9.0)])


def test_rdf_box_changes_every_frame(emu_lib, oracle):
    """NPT-like trajectory: the grid of a batch comes from its smallest box, cell coordinates are scaled per frame"""
    rng = np.random.default_rng(8)
    boxes = [40.0, 43.5, 38.2, (41.0, 39.0, 44.0)]
    c = np.stack([(rng.uniform(0, 1, (3, 1500)) * np.array(b if not np.isscalar(b) else (b,) * 3)[:, None]).astype(np.float32) for b in boxes])
    a = np.arange(0, 1500, 2)
    cases.check_rdf(emu_lib, oracle, c, boxes, [("g", a, a, 0.0, 11.0), ("gx", a, np.arange(1, 1500, 2), 0.0, 9.0)])


def test_rdf_edge_cases(emu_lib, oracle):
    cases.rdf_edge_cases(emu_lib, oracle)


def test_rdf_brute_nonperiodic_and_large_cutoff(emu_lib, oracle):
    rng = np.random.default_rng(3)
    c = rng.uniform(0, 20, (3, 3, 300)).astype(np.float32)
    a, b = np.arange(0, 300, 2), np.arange(300)
    cases.check_rdf(emu_lib, oracle, c, None, [("g", a, b, 0.0, 10.0)], oracle_method="brute")          # no cell (C1-like)
    cases.check_rdf(emu_lib, oracle, c, 20.0, [("g", a, a, 0.0, 10.0)], oracle_method="brute")          # rc = L/2 -> brute + PBC
    cases.check_rdf(emu_lib, oracle, c, 20.0, [("g", a, b, 1.0, 6.0)], flags=3,
                    oracle_method="brute")                                                               # slab periodicity


def test_rdf_sharding_and_device_trajectory(emu_lib, oracle, box3k):
    o = cases.oxygen(3000)
    cases.check_rdf(emu_lib, oracle, box3k, 60.0, [("goo", o, o, 0.0, 12.0)], device=True, ranges=[(2, 3), (0, 1), (1, 2)])


def test_pinned_host_trajectory_and_small_batches(emu_lib, oracle, box3k):
    o = cases.oxygen(3000)
    old = emu_lib.vmd_set_option(b"batch_frames", 1)          # every frame its own batch: exercises the two-stage pipeline
    try:
        cases.check_rdf(emu_lib, oracle, box3k, 60.0, [("goo", o, o, 0.0, 12.0)], device="pinned")
        cases.check_rdf(emu_lib, oracle, box3k, 60.0, [("goo", o, o, 0.0, 12.0)], device=False)
    finally:
        emu_lib.vmd_set_option(b"batch_frames", old)


def test_sdf_volume_and_matrices(emu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 4, 1500, 40.0, 3)
    n_s = structures.size
    tgt = np.arange(n_s, coords.shape[2], 3, dtype=np.int32)                # water oxygens
    cases.check_sdf(emu_lib, oracle, coords, 40.0, structures, mass, tgt, 10.0)
    # targets that include the structures' own atoms exercise the exclusion rule (SPEC D-SDF-EXCL)
    cases.check_sdf(emu_lib, oracle, coords[:2], 40.0, structures, mass, np.arange(coords.shape[2], dtype=np.int32), 6.0,
                    ranges=[(1, 2), (0, 1)])


def test_sdf_sparse_and_dense_target_paths(emu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 14, 1200, 36.0, 2)
    n_s, N = structures.size, coords.shape[2]
    dense = np.arange(n_s, N, 3, dtype=np.int32)                  # a third of all atoms -> streamed with tags
    sparse = np.arange(n_s, N, 30, dtype=np.int32)                # a few percent -> index-list gather
    for flag in (0, 1):
        old = emu_lib.vmd_set_option(b"sdf_dense", flag)
        try:
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, dense, 8.0)
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, np.arange(N, dtype=np.int32), 5.0)   # owners among the targets
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, sparse, 9.0)
        finally:
            emu_lib.vmd_set_option(b"sdf_dense", old)
    # the scatter's target addressing: index list / arithmetic progression generated on the device, 4 / 8 atoms per thread,
    # targets with and without owners, an irregular list (no progression)
    irregular = np.sort(np.random.default_rng(5).choice(np.arange(0, N, dtype=np.int32), N // 4, replace=False)).astype(np.int32)
    for arith, ilp, rows, wv in ((0, 4, 0, 0), (1, 8, 0, 0), (0, 16, 0, 0), (1, 4, 1, 0), (1, 4, 4, 0), (1, 4, 0, 1), (0, 8, 0, 1), (1, 4, 0, 2), (0, 8, 0, 16)):      # rows: 16-byte row streaming; wv: 1 per-wave compaction, 2 / >= 16 persistent streaming kernel
        old = emu_lib.vmd_set_option(b"sdf_arith", arith), emu_lib.vmd_set_option(b"sdf_ilp", ilp), emu_lib.vmd_set_option(b"sdf_rows", rows)
        old_wv = emu_lib.vmd_set_option(b"sdf_wave", wv)
        try:
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, dense, 8.0)
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, irregular, 8.0)
            cases.check_sdf(emu_lib, oracle, coords, 36.0, structures, mass, np.arange(N, dtype=np.int32), 5.0)   # stride 1, owners among the targets
        finally:
            emu_lib.vmd_set_option(b"sdf_arith", old[0]); emu_lib.vmd_set_option(b"sdf_ilp", old[1]); emu_lib.vmd_set_option(b"sdf_rows", old[2])
            emu_lib.vmd_set_option(b"sdf_wave", old_wv)


def test_distance_family(emu_lib, oracle):
    coords, structures, mass = cases.sdf_system(oracle, 9, 600, 30.0, 4)
    specs = [("d", [3], [40], L.DIST_COM), ("dcom", structures[0], structures[1], L.DIST_COM),
             ("dmin", structures[0], structures[2], L.DIST_MIN), ("dmax", structures[1], structures[3], L.DIST_MAX),
             ("dpair", structures[0][:3], structures[1][:4], L.DIST_PAIR),
             # populations: `distance(1, 4) in residue(:)`, `distance_min(...) in ...`, `distance_pair(1:2, 4:6) in ...`
             ("pcom", [s[:1] for s in structures], [s[3:4] for s in structures], L.DIST_COM, "pop"),
             ("pmin", [s[:2] for s in structures], [s[2:] for s in structures], L.DIST_MIN, "pop"),
             ("ppair", [s[:2] for s in structures], [s[3:6] for s in structures], L.DIST_PAIR, "pop")]
    cases.check_distances(emu_lib, oracle, coords, 30.0, mass, specs, ranges=[(0, 2), (2, 4)])


def test_frame_mask_as_bitfield_words(emu_lib, oracle):
    """md_script_eval_frame_mask is an md_bitfield_t in VIAMD (src/main.cpp:1513, tested bit by bit :194-210): 70 frames = two words"""
    import viamd_amd as V
    F, n = 70, 30
    coords = np.random.default_rng(3).uniform(0, 20, (F, 3, n)).astype(np.float32)
    ir = V.ScriptIR(emu_lib)
    ir.add_distance("d", [0], [1], L.DIST_COM)
    ev = V.ScriptEval(F, ir)
    traj = V.HostTrajectory(coords, V.make_unitcell(20.0))
    assert ev.frame_mask_bits().tolist() == [0, 0]
    for beg, end in ((3, 9), (60, 70), (63, 65)):
        assert ev.frame_range(V.MolSystem(n, unitcell=V.make_unitcell(20.0)), traj, beg, end)
    bytes_ = ev.frame_mask()
    words = ev.frame_mask_bits()
    assert words.size == 2
    for f in range(F):
        assert bool(bytes_[f]) == bool((int(words[f // 64]) >> (f & 63)) & 1) == (3 <= f < 9 or 60 <= f < 70)


def test_synth_kernel_matches_oracle_generator(emu_lib, oracle):
    import viamd_amd as V
    t = V.DeviceTrajectory(3, 999, lib=emu_lib)
    t.synth(42, 50.0, 0.05)
    for f in range(3):
        got, cell = t.download_frame(f)
        ref = oracle.synth_frame(42, 999, 50.0, 0.05, f)
        np.testing.assert_array_equal(got, ref)
        assert cell.x == 50.0 and cell.flags == 7


def test_interrupt_and_clear(emu_lib, oracle, box3k):
    import viamd_amd as V
    o = cases.oxygen(3000)
    ir = V.ScriptIR(emu_lib)
    ir.add_rdf("goo", o, o, 12.0)
    ev = V.ScriptEval(3, ir)
    traj = V.HostTrajectory(box3k, V.make_unitcell(60.0))
    ev.interrupt()
    assert ev.frame_range(V.MolSystem(3000), traj, 0, 3) is False      # returns promptly, nothing evaluated
    assert ev.frames_done() == 0 and not ev.frame_mask().any()
    ev.clear_data()                                                    # clears the interrupt flag too (src/main.cpp:990)
    assert ev.frame_range(V.MolSystem(3000), traj, 0, 1)
    fp = ev.property_data("goo").fingerprint
    first = ev.property_data("goo").counts.copy()
    assert first.sum() > 0
    ev.clear_data()
    assert ev.property_data("goo").counts.sum() == 0 and ev.property_data("goo").fingerprint != fp
    assert ev.frame_range(V.MolSystem(3000), traj, 0, 1)
    np.testing.assert_array_equal(ev.property_data("goo").counts, first)
    assert ev.ir_fingerprint() == ir.fingerprint()


def test_triclinic_cell_all_property_kinds(emu_lib, oracle):
    cases.triclinic_cases(emu_lib, oracle, n_water=900)


def test_filtered_evaluation_reuses_block_partials(emu_lib, oracle):
    cases.filtered_cases(emu_lib, oracle, 900)


def test_open_boundaries_and_slabs_on_the_grid(emu_lib, oracle):
    cases.open_boundary_cases(emu_lib, oracle, 700)


def test_sdf_triclinic_spread_structures_regression(emu_lib, oracle):
    cases.sdf_triclinic_spread_structures(emu_lib, oracle)


def test_result_views_keep_their_eval_alive(emu_lib, oracle):
    """VIAMD caches md_script_property_data_t pointers for the eval's lifetime (src/main.cpp:1286,1303); in the Python mirror a
    view must therefore keep the eval alive: `check(...).property_data(n).counts` on a temporary used to read freed memory."""
    import gc
    import viamd_amd as V
    coords = cases.water_box(oracle, 3, 600, 30.0, 2)
    o = cases.oxygen(600)

    def views():
        ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 8.0)
        ev = V.ScriptEval(2, ir)
        assert ev.frame_range(V.MolSystem(600, unitcell=V.make_unitcell(30.0)), V.HostTrajectory(coords, V.make_unitcell(30.0)), 0, 2)
        pd = ev.property_data("g")
        assert pd.unit_str == ("\u00c5", "")                              # md_script_property_data_t::unit as VIAMD prints it
        return pd.counts, pd.values[:10], ev.frame_mask(), pd.counts.copy()

    c, v, m, ref = views()
    gc.collect()
    junk = [np.full(1024, 0xAB, np.uint64) for _ in range(64)]            # recycle freed blocks, if any
    np.testing.assert_array_equal(c, ref)
    assert m.all() and ref.sum() > 0 and len(junk) == 64 and v.shape == (10,)


def test_sheared_sc_lattice_known_answer(emu_lib):
    cases.sheared_sc_lattice(emu_lib)


def test_open_sc_lattice_known_answer(emu_lib, oracle):
    cases.open_sc_lattice(emu_lib, oracle)


def test_sdf_rotations_known_answer(emu_lib):
    cases.sdf_rotations_known_answer(emu_lib)


def test_distance_known_answer(emu_lib):
    cases.distance_known_answer(emu_lib)


def test_filtered_eval_against_a_running_source(emu_lib, oracle):
    cases.filtered_contention_case(emu_lib, oracle, n=1500, box=40.0, F=8, S=2, rounds=6)


def test_resident_trajectory_changes_invalidate_cached_boxes(emu_lib, oracle):
    cases.device_view_cache_case(emu_lib, oracle)


def test_spec_decisions_are_switches(emu_lib, oracle):
    cases.spec_switch_check(emu_lib, oracle)


def test_sdf_structures_are_made_whole_along_their_bonds(emu_lib, oracle):
    """vmd_system_t::bonds (md_system_t::bond, /root/reference/src/viamd.cpp:2257, 3088-3091): ring-shaped ligands in scrambled index
    order across a cell face - bond-tree unwrap == oracle bit for bit, differs from the index chain, and aligns every rigid copy
    onto the reference pose (known answer)."""
    assert cases.check_bonded_unwrap(emu_lib, oracle) > 0
