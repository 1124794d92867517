"""Parity at the sizes BASELINE.json names: every configuration the bench launches is compared with the
oracle at its full atom count, through the script front-end, on enough frames to cross a launch batch where that matters.
The oracle runs on all host cores (vo_rdf_run / vo_sdf_run hand frames to OpenMP threads); a few minutes in total."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import cases
import viamd_amd as V
from viamd_amd import _lib as L
from viamd_amd import script, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the workload table: the same scripts, sizes and seeds the bench line is measured on)

CORES = max(1, len(os.sched_getaffinity(0)))


def _host_frames(oracle, w, frames):
    """the synthetic trajectory of a bench workload on the host (the generator the device kernel is bit-identical to)"""
    with ThreadPoolExecutor(min(CORES, 32)) as ex:
        fr = list(ex.map(lambda f: oracle.synth_frame(w["seed"], w["atoms"], w["box"], 0.05, f, n_blob=w["blob"]), range(frames)))
    out = np.stack(fr)
    if w["blob"]:
        for b0, xyz in synth.blob_trajectory(w["seed"], w["blob"], w["box"], frames):
            out[b0:b0 + xyz.shape[0], :, :w["blob"]] = xyz
    return out


def _evaluate(w, frames, ranges=None):
    traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], frames, w["blob"])
    topo = synth.water_box_topology(w["atoms"], w["blob"])
    ir, info = script.compile_script(w["script"], topo)
    ev = V.ScriptEval(frames, ir)
    sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
    for beg, end in (ranges or [(0, frames)]):
        assert ev.frame_range(sysm, traj, beg, end)
    assert ev.frames_done() == frames
    return ev, info, topo, traj


def _check_rdf(oracle, ev, name, d, coords, ocell):
    counts, weights, hits = oracle.rdf_run(coords, [ocell] * coords.shape[0], d["ref"], d["target"], d["rmin"], d["rmax"], nthreads=CORES)
    pd = ev.property_data(name)
    np.testing.assert_array_equal(pd.counts, counts, err_msg=f"{name}: integer histogram differs from the oracle")
    np.testing.assert_array_equal(pd.values, counts.astype(np.float32))
    np.testing.assert_allclose(pd.weights64, weights, rtol=1e-12)
    g_dev = V.downsample_histogram(pd.values, pd.weights, 128)
    g_ref = oracle.downsample_histogram(counts.astype(np.float32), weights.astype(np.float32), 128)
    np.testing.assert_allclose(g_dev, g_ref, rtol=1e-5, atol=0)            # BASELINE.json: 1e-5 relative on normalised g(r)
    assert counts.sum() == hits and hits > 0
    return int(hits)


def test_config2_more_frames_than_one_launch_batch(gpu_lib, oracle):
    """C2 (100 002 atoms, O-O, r_c 12): 1 100 frames = more than one launch batch (<= 1 024 frames), the batches the bench
    actually launches, against vo_rdf_run on all cores."""
    w = bench.WORKLOADS["c2"]
    F = 1100
    ev, info, _, _ = _evaluate(w, F)
    coords = _host_frames(oracle, w, F)
    hits = _check_rdf(oracle, ev, "g", info["g"], coords, oracle.make_cell(w["box"]))
    assert 7.9e6 * F < hits < 8.2e6 * F                                     # 8.04e6 ordered pairs per frame


def test_config3_three_frames(gpu_lib, oracle):
    """C3 (1 000 002 atoms, heavy-atom RDF through `not element('H')`): 3 frames, evaluated as two ranges."""
    w = bench.WORKLOADS["c3"]
    F = 3
    ev, info, _, _ = _evaluate(w, F, ranges=[(1, 3), (0, 1)])
    coords = _host_frames(oracle, w, F)
    hits = _check_rdf(oracle, ev, "g", info["g"], coords, oracle.make_cell(w["box"]))
    assert 7.9e7 * F < hits < 8.2e7 * F


@pytest.mark.parametrize("dense", [0, 1])
def test_config4_full_size_through_the_script(gpu_lib, oracle, dense):
    """C4 (100 001 atoms: 2 000-atom blob + waters; `residue(5:11)` = 7 structures x 10 atoms, water oxygens as targets, 128^3):
    5 frames through the script front-end, index-gather and dense-tag scatter paths, volume + vis matrices vs the oracle."""
    w = bench.WORKLOADS["c4"]
    F = 5
    old = gpu_lib.vmd_set_option(b"sdf_dense", dense)
    try:
        ev, info, topo, traj = _evaluate(w, F, ranges=[(0, 2), (2, 5)])
    finally:
        gpu_lib.vmd_set_option(b"sdf_dense", old)
    d = info["v"]
    assert d["structures"].shape == (7, 10) and 30000 < d["target"].size < 34000
    coords = _host_frames(oracle, w, F)
    ocell = oracle.make_cell(w["box"])
    vol, mats = cases.oracle_sdf(oracle, coords, ocell, d["structures"], topo.mass, d["target"], d["cutoff"])
    pd = ev.property_data("v")
    np.testing.assert_array_equal(pd.counts, vol, err_msg="SDF voxel counts differ from the oracle")
    np.testing.assert_array_equal(pd.values, vol.astype(np.float32))
    assert pd.max_value == float(vol.max()) and vol.sum() > 7 * F * 150      # ~267 in-box oxygens per structure and frame
    sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
    M4, ext = ev.sdf_matrices("v", sysm, traj, F - 1)
    np.testing.assert_array_equal(M4[:, :3, :], mats[F - 1].astype(np.float32))


def test_config5_all_eight_properties_in_one_eval(gpu_lib, oracle):
    """C5 (1 001 999 atoms): 3 RDF + SDF + 4 distance properties co-evaluated by ONE eval over 2 frames, every property
    against the oracle."""
    w = bench.WORKLOADS["c5"]
    F = 2
    ev, info, topo, _ = _evaluate(w, F)
    coords = _host_frames(oracle, w, F)
    ocell = oracle.make_cell(w["box"])
    for name in ("goo", "goh", "ghv"):
        _check_rdf(oracle, ev, name, info[name], coords, ocell)
    # goo is a subset of ghv (water oxygens are heavy atoms): whatever pair work the evaluator shares, the integers must nest
    assert (ev.property_data("ghv").counts >= ev.property_data("goo").counts).all()
    d = info["v"]
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, d["structures"], topo.mass, d["target"], d["cutoff"])
    np.testing.assert_array_equal(ev.property_data("v").counts, vol)
    assert vol.sum() > 0
    kinds = {"distance": L.DIST_COM, "distance_min": L.DIST_MIN, "distance_max": L.DIST_MAX, "distance_pair": L.DIST_PAIR}
    for name in ("d1", "d2", "d3", "d4"):
        dd = info[name]
        ref = cases.oracle_distance(oracle, coords, ocell, topo.mass, np.asarray(dd["a"], np.int32), np.asarray(dd["b"], np.int32), kinds[dd["kind"]])
        np.testing.assert_array_equal(ev.property_data(name).values.reshape(F, -1), ref, err_msg=name)


# ---- size-independent properties at the sizes the bench launches (no oracle involved: the GPU path against itself through
# ---- identities of the domain; the oracle comparisons above cover a few frames of each configuration)

def _rdf_counts(w, traj, topo, ref, tgt, frames, ranges=None, opts=()):
    lib = V.default_lib()
    old = [(k, lib.vmd_set_option(k, v)) for k, v in opts]
    try:
        ir = V.ScriptIR(lib)
        ir.add_rdf("g", ref, tgt, (0.0, 12.0))
        ev = V.ScriptEval(frames, ir)
        sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
        for beg, end in (ranges or [(0, frames)]):
            assert ev.frame_range(sysm, traj, beg, end)
        return ev.property_data("g").counts.copy()
    finally:
        for k, v in old:
            lib.vmd_set_option(k, v)


def test_config3_linearity_and_symmetry_of_the_pair_histogram(gpu_lib):
    """1 000 002 atoms: with the heavy atoms H split into disjoint A and B, rdf(H, H) = rdf(A, A) + rdf(B, B) + 2 rdf(A, B) and
    rdf(A, B) = rdf(B, A) - the same-set half-shell walk, the two-set full-shell walk and both lane assignments against each other."""
    w = bench.WORKLOADS["c3"]
    F = 2
    traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
    topo = synth.water_box_topology(w["atoms"], w["blob"])
    heavy = np.arange(0, w["atoms"], 3, dtype=np.int32)
    rng = np.random.default_rng(11)
    pick = rng.random(heavy.size) < 0.37                     # uneven split: A is the sparser set
    A, B = heavy[pick], heavy[~pick]
    hh = _rdf_counts(w, traj, topo, heavy, heavy, F)
    aa = _rdf_counts(w, traj, topo, A, A, F)
    bb = _rdf_counts(w, traj, topo, B, B, F)
    ab = _rdf_counts(w, traj, topo, A, B, F)
    ba = _rdf_counts(w, traj, topo, B, A, F)
    np.testing.assert_array_equal(ab, ba)
    np.testing.assert_array_equal(hh, aa + bb + ab + ba)
    assert (hh % 2 == 0).all() and 7.9e7 * F < hh.sum() < 8.2e7 * F
    # the brute-force kernel family on a subset small enough for it: the same identity across kernel families
    a_small, b_small = A[:20000], B[:30000]
    grid = _rdf_counts(w, traj, topo, a_small, b_small, F)
    brute = _rdf_counts(w, traj, topo, a_small, b_small, F, opts=[(b"force_brute", 1)])
    np.testing.assert_array_equal(grid, brute)


def test_config3_and_config4_any_frame_partition_gives_the_same_accumulators(gpu_lib):
    """any sharding of the frames -> identical u64 counts.  c3 at 200 frames in one batch / in ragged ranges with
    37-frame batches / in reverse block order; c4 at 4 000 frames likewise (volumes)."""
    lib = V.default_lib()
    w = bench.WORKLOADS["c3"]
    F = 200
    ev1, info, topo, traj = _evaluate(w, F)
    ref = ev1.property_data("g").counts.copy()
    old = lib.vmd_set_option(b"batch_frames", 37)
    try:
        ev2, _, _, _ = _evaluate(w, F, ranges=[(150, 200), (0, 1), (1, 64), (64, 150)])
    finally:
        lib.vmd_set_option(b"batch_frames", old)
    np.testing.assert_array_equal(ev2.property_data("g").counts, ref)
    np.testing.assert_allclose(ev2.property_data("g").weights64, ev1.property_data("g").weights64, rtol=1e-12)
    del ev1, ev2, traj
    w = bench.WORKLOADS["c4"]
    F = 4000
    ev1, info, topo, traj = _evaluate(w, F)
    name = [n for n, d in info.items() if d["kind"] == "sdf"][0]
    ref = ev1.property_data(name).counts.copy()
    assert ref.sum() > 0
    old = lib.vmd_set_option(b"batch_frames", 333)
    try:
        ev2, _, _, _ = _evaluate(w, F, ranges=[(3000, 4000), (0, 7), (7, 3000)])
    finally:
        lib.vmd_set_option(b"batch_frames", old)
    np.testing.assert_array_equal(ev2.property_data(name).counts, ref)
    for n, d in info.items():
        if d["kind"].startswith("distance"):
            np.testing.assert_array_equal(ev2.property_data(n).values, ev1.property_data(n).values)
