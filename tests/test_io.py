"""File-level plumbing around the hot path: PDB multi-MODEL in, cube / XVG / CSV out."""
import numpy as np

import viamd_amd as V
from viamd_amd import export, pdb, script, synth


def test_cube_roundtrip_and_layout(tmp_path):
    dim = 8
    vol = np.arange(dim ** 3, dtype=np.float32)              # values[z*d*d + y*d + x]
    p = tmp_path / "v.cube"
    export.write_cube(p, vol, dim, 10.0, atoms=[(8, 0.0, 1.0, -2.0)])
    lines = p.read_text().split("\n")
    assert lines[0] == "EXPORTED DENSITY VOLUME FROM VIAMD, UNITS IN BOHR" and lines[1].startswith("OUTER LOOP: X")
    half = 10.0 * export.ANGSTROM_TO_BOHR
    assert lines[2].split() == ["-1", "%.6f" % -half, "%.6f" % -half, "%.6f" % -half]
    # first data values walk z fastest for x = y = 0 (src/main.cpp:5808-5816): indices 0, d*d, 2*d*d, ...
    first = [float(t) for t in lines[8].split()]
    assert first == [0.0, 64.0, 128.0, 192.0, 256.0, 320.0]
    back = export.read_cube(p)
    np.testing.assert_array_equal(back["volume"], vol)
    assert back["dim"] == [dim] * 3 and abs(back["voxel"][0] - 2 * half / dim) < 1e-6


def test_config1_standin_pdb_through_script_on_emulator(tmp_path, emu_lib, oracle):
    """BASELINE config 1 plumbing: multi-MODEL PDB -> trajectory + topology -> `rdf(element('O'),element('O'),10.0)`."""
    rng = np.random.default_rng(1)
    n, F = 112, 6
    topo = synth.water_box_topology(n, n_blob=n)             # a 112-atom poly-ALA-like chain, no waters
    base = np.cumsum(rng.normal(0, 0.9, (n, 3)), axis=0)
    coords = (base[None] + rng.normal(0, 0.3, (F, n, 3))).transpose(0, 2, 1)
    p = tmp_path / "1ALA-standin.pdb"
    pdb.write_pdb(p, coords, topo)
    c2, t2, cell = pdb.read_pdb(p)
    assert c2.shape == (F, 3, n) and t2.num_residues == topo.num_residues and cell.flags == 0
    np.testing.assert_allclose(c2, coords, atol=5e-4)        # %8.3f columns
    assert list(t2.elements) == list(topo.elements)
    ir, info = script.compile_script("r = rdf(element('O'), element('O'), 10.0);", t2, lib=emu_lib)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(n, mass=t2.mass, unitcell=cell), V.HostTrajectory(c2, cell), 0, F)
    ocell = oracle.make_cell(None, 0)
    ref = np.zeros(1024, np.uint64)
    for f in range(F):
        oracle.rdf_frame(c2[f, 0], c2[f, 1], c2[f, 2], ocell, info["r"]["ref"], info["r"]["target"], 0.0, 10.0, counts=ref)
    pd = ev.property_data("r")
    np.testing.assert_array_equal(pd.counts, ref)
    # export what VIAMD would plot
    x, g = export.distribution_table(pd, 128, lib=emu_lib)
    export.write_xvg(tmp_path / "r.xvg", "r", "r (A)", "g(r)", x, [("r", g)])
    export.write_csv(tmp_path / "r.csv", "x", x, [("r", g)])
    rows = [l for l in (tmp_path / "r.xvg").read_text().split("\n") if l and l[0] not in "#@"]
    assert len(rows) == 128 and abs(float(rows[5].split()[1]) - g[5]) < 1e-5
    assert (tmp_path / "r.csv").read_text().split("\n")[0] == "x,r"


def test_pdb_periodic_cell(tmp_path):
    topo = synth.water_box_topology(30)
    coords = np.random.default_rng(0).uniform(0, 20, (2, 3, 30))
    pdb.write_pdb(tmp_path / "w.pdb", coords, topo, box=20.0)
    c, t, cell = pdb.read_pdb(tmp_path / "w.pdb")
    assert (cell.x, cell.y, cell.z, cell.flags) == (20.0, 20.0, 20.0, 7)
    assert t.residue_name(0) == "HOH" and t.num_residues == 10


def test_pdb_triclinic_cell_roundtrip(tmp_path):
    topo = synth.water_box_topology(30)
    coords = np.random.default_rng(0).uniform(0, 20, (1, 3, 30))
    pdb.write_pdb(tmp_path / "t.pdb", coords, topo, box=(30.0, 28.0, 26.0), tilt=(6.0, -4.0, 5.0))
    c, t, cell = pdb.read_pdb(tmp_path / "t.pdb")
    np.testing.assert_allclose([cell.x, cell.y, cell.z, cell.xy, cell.xz, cell.yz], [30.0, 28.0, 26.0, 6.0, -4.0, 5.0], atol=2e-2)
    assert cell.flags == 7
