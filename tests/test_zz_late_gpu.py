"""GPU checks of what was added after the last GPU session of round 1 (file name sorts last: `pytest -x` reaches these only
after every earlier parity test has passed): XTC / TRR files staged through the multi-threaded host decode into the kernels,
and the dense all-atom selection of the C3-dense workload."""
import numpy as np
import pytest

import cases
import viamd_amd as V


def _xdr_through_the_evaluator(lib, oracle, tmp_path, box, F, N, device_check):
    coords = cases.water_box(oracle, 31, N, box, F)
    cell = V.make_unitcell(box)
    o = cases.oxygen(N)
    ir = V.ScriptIR(lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    res = {}
    for fmt, write in (("xtc", V.write_xtc), ("trr", V.write_trr)):
        p = tmp_path / f"w.{fmt}"
        write(p, coords, cell, lib=lib)
        t = V.XdrTrajectory(p, lib=lib)
        decoded = np.stack([t.load_frame(f)[0] for f in range(F)])
        assert abs(decoded - coords).max() < (0.0051 if fmt == "xtc" else 1e-4)
        old = lib.vmd_set_option(b"batch_frames", max(4, F // 3))          # several staged batches, decoded on several threads
        try:
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, V.XdrTrajectory(p, lib=lib), 0, F)
        finally:
            lib.vmd_set_option(b"batch_frames", old)
        got = ev.property_data("g").counts.copy()
        counts, _ = cases.oracle_rdf(oracle, decoded, oracle.make_cell(box), o, o, 0.0, 9.0)
        np.testing.assert_array_equal(got, counts, err_msg=fmt)            # bit-exact on the coordinates the file holds
        res[fmt] = got
        if fmt == "xtc":
            # the same file with the batch decompressed on the device: one thread per frame (1), index pass + one thread per
            # chunk (2); identical integers, every frame counted as device-decoded
            for mode in (1, 2):
                old_d = lib.vmd_set_option(b"xtc_device_decode", mode)
                old_c = lib.vmd_set_option(b"xtc_chunk", 256)
                old = lib.vmd_set_option(b"batch_frames", max(4, F // 3))
                try:
                    ev = V.ScriptEval(F, ir)
                    assert ev.frame_range(sysm, V.XdrTrajectory(p, lib=lib), 0, F)
                finally:
                    lib.vmd_set_option(b"batch_frames", old)
                    lib.vmd_set_option(b"xtc_chunk", old_c)
                    lib.vmd_set_option(b"xtc_device_decode", old_d)
                assert ev.frames_device_decoded() == F
                np.testing.assert_array_equal(ev.property_data("g").counts, counts, err_msg=f"xtc, device decode variant {mode}")
    assert res["xtc"].sum() > 0 and abs(int(res["xtc"].sum()) - int(res["trr"].sum())) < 0.01 * res["trr"].sum()


def test_xdr_files_through_the_evaluator_on_emulator(emu_lib, oracle, tmp_path):
    _xdr_through_the_evaluator(emu_lib, oracle, tmp_path, 30.0, 6, 600, False)


@pytest.mark.gpu
def test_xdr_files_through_the_evaluator(gpu_lib, oracle, tmp_path):
    _xdr_through_the_evaluator(gpu_lib, oracle, tmp_path, 70.0, 24, 30000, True)


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
