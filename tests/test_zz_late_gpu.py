"""GPU checks of what was added after the last GPU session of round 1 (file names sort last: `pytest -x` reaches these only
after every earlier parity test has passed): closed-form known answers (no oracle involved) and the dense all-atom selection of
the C3-dense workload here; the trajectory-file paths incl. the device-side XTC decoder, which have never run on hardware, in
test_zzz_xdr_gpu.py behind them."""
import numpy as np
import pytest

import cases
import viamd_amd as V


@pytest.mark.gpu
def test_sheared_sc_lattice_known_answer(gpu_lib):
    """Exact shell multiplicities of a simple cubic crystal in cubic and sheared cells (no oracle involved)."""
    cases.sheared_sc_lattice(gpu_lib, device=True)


@pytest.mark.gpu
def test_open_sc_lattice_known_answer(gpu_lib):
    """Exact pair counts of a finite lattice block without a cell, as a slab and as wires (no oracle involved)."""
    cases.open_sc_lattice(gpu_lib, device=True)


@pytest.mark.gpu
def test_sdf_rotations_known_answer(gpu_lib):
    """Targets on voxel centres of the aligned grid through the 24 cube rotations: exact volume, inverse rotations (no oracle)."""
    cases.sdf_rotations_known_answer(gpu_lib, device=True)


@pytest.mark.gpu
def test_distance_known_answer(gpu_lib):
    cases.distance_known_answer(gpu_lib, device=True)


def _dense_all_atom_rdf(lib, oracle, seed, N, box, F, device):
    """`rdf(all, all, r)`: every atom in both selections (the C3-dense workload of bench.py): half shell over the whole frame,
    selection == identity, split cell build on the GPU sizes."""
    coords = cases.water_box(oracle, seed, N, box, F)
    everything = np.arange(N, dtype=np.int32)
    ev = cases.check_rdf(lib, oracle, coords, box, [("g", everything, everything, 0.0, 12.0)], device=device)
    c = ev.property_data("g").counts
    assert (c % 2 == 0).all() and c.sum() > 0
    return c


def test_dense_all_atom_rdf_on_emulator(emu_lib, oracle):
    _dense_all_atom_rdf(emu_lib, oracle, 5, 1500, 40.0, 2, False)


@pytest.mark.gpu
def test_dense_all_atom_rdf(gpu_lib, oracle):
    """150 000 atoms at the density of BASELINE config 3 (0.1 / A^3): ~1.1e8 ordered pairs per frame, bit-exact against the oracle."""
    N = 150000
    box = float((N / 0.1) ** (1.0 / 3.0))
    c = _dense_all_atom_rdf(gpu_lib, oracle, 6, N, box, 2, True)
    assert 1.0e8 * 2 < c.sum() < 1.2e8 * 2


@pytest.mark.gpu
def test_spec_decisions_are_switches_on_the_gpu(gpu_lib, oracle):
    """the DECISION switches through the hipcc-built kernels: closed RDF interval (a pair at exactly r_max, coincident atoms, the self
    pairs of the half-shell pass), SDF without the exclusion rule, density-scaled float view, geometric-centre distance"""
    cases.spec_switch_check(gpu_lib, oracle)


@pytest.mark.gpu
def test_sdf_structures_are_made_whole_along_their_bonds_on_the_gpu(gpu_lib, oracle):
    assert cases.check_bonded_unwrap(gpu_lib, oracle) > 0


@pytest.mark.gpu
def test_merge_through_the_torch_fallback_collective_on_device_memory(gpu_lib, oracle):
    """bench.py's N > 1 fallback (viamd_amd/dist.py: TorchCollective(device=True)): the callbacks alias the evaluator's DEVICE
    accumulators as torch tensors (__cuda_array_interface__) and all-reduce them through torch's own RCCL communicator.  A 1-rank
    process group on the 1-GPU box: the merge must leave every result as it was, travel as u32 for the volume, and say so."""
    import os
    import torch
    import torch.distributed as dist
    from viamd_amd.dist import TorchCollective, reduce_stats
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        coords, structures, mass = cases.sdf_system(oracle, 8, 3000, 40.0, 4)
        N = coords.shape[2]
        o = np.arange(structures.size, N, 3, dtype=np.int32)
        cell = V.make_unitcell(40.0)
        ir = V.ScriptIR(gpu_lib)
        ir.add_rdf("g", o, o, 10.0)
        ir.add_sdf("v", structures, o, 8.0)
        ir.add_distance("d", structures[0], structures[1], 0)
        ev = V.ScriptEval(4, ir)
        assert ev.frame_range(V.MolSystem(N, mass=mass, unitcell=cell), V.HostTrajectory(coords, cell), 0, 4)
        before = {k: ev.property_data(k).counts.copy() for k in "gv"}
        d_before = np.array(ev.property_data("d").values)
        coll = TorchCollective(device=True)
        assert gpu_lib.vmd_eval_reduce(ev.h, coll.collective(), None), gpu_lib.last_error()
        for k in "gv":
            np.testing.assert_array_equal(ev.property_data(k).counts, before[k])
        np.testing.assert_array_equal(np.array(ev.property_data("d").values), d_before)
        st = reduce_stats(ev)
        assert st["allreduce_calls"] == 3 and st["volumes_as_u32"] == 1 and not st["grouped_into_one_launch"] and st["bytes"] > 128 ** 3 * 4
        assert before["v"].sum() > 0 and ev.frame_mask().all()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_inline_asm_blocks_agree_with_their_cpp_twins_on_the_gpu(gpu_lib, oracle, tmp_path):
    """The CPU suite runs the kernels' C++ twins (VMD_NO_INLINE_ASM, tests/emu); the product runs the inline-asm blocks.  Here the
    twins are compiled for gfx950 too (tests/native/libviamd_amd_twin.so, built by __graft_entry__.build()) and both libraries
    evaluate the same inputs on the GPU: rdf (pencil kernel, open and periodic, both orders), sdf (align + scatter), distances
    and an XTC batch decoded on the device must all equal the oracle's integers, hence each other's."""
    import conftest
    import viamd_amd as V
    from viamd_amd import _lib as L
    twin = V.VmdLib(conftest.build_twin())
    assert twin.vmd_device_count() > 0
    twin.vmd_set_device(0)
    for lib in (twin, gpu_lib):
        coords, structures, mass = cases.sdf_system(oracle, 11, 9000, 52.0, 5)
        N = coords.shape[2]
        o = np.arange(structures.size, N, 3, dtype=np.int32)
        h = np.setdiff1d(np.arange(structures.size, N, dtype=np.int32), o)
        cases.check_rdf(lib, oracle, coords, 52.0, [("goo", o, o, 0.0, 12.0), ("goh", o, h, 0.5, 9.0), ("gho", h, o, 0.0, 7.0)], device=True)
        cases.check_rdf(lib, oracle, coords, None, [("open", o, o, 0.0, 10.0)], device=True)
        cases.check_sdf(lib, oracle, coords, 52.0, structures, mass, o, 10.0, device=True)
        cases.check_distances(lib, oracle, coords, 52.0, mass, [("d", structures[0], structures[1], L.DIST_COM)], device=True)
    # the device XTC decoder: same file, same batch plan, both libraries
    F, Nw, box = 12, 12000, 50.0
    wc = cases.water_box(oracle, 5, Nw, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "t.xtc"
    V.write_xtc(p, wc, cell, lib=gpu_lib)
    ow = cases.oxygen(Nw)
    got = []
    for lib in (twin, gpu_lib):
        ir = V.ScriptIR(lib); ir.add_rdf("g", ow, ow, 9.0)
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(V.MolSystem(Nw, unitcell=cell), V.XdrTrajectory(p, lib=lib), 0, F)
        assert ev.frames_device_decoded() == F
        got.append(ev.property_data("g").counts.copy())
    np.testing.assert_array_equal(got[0], got[1])
    assert got[0].sum() > 0
