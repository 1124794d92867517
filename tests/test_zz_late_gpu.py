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
