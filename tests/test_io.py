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


def test_dcd_reader_roundtrip_both_byte_orders_and_cells(tmp_path, host_lib):
    emu_lib = host_lib       # pure host code: the g++ emulator build and the hipcc-built product library
    """The native DCD reader (vmd_dcd.cpp; VIAMD: md_dcd_attach_from_file, src/loader.cpp:151-152): coordinates bit for bit,
    orthorhombic cells exactly, triclinic cells through angles (degrees and cosines), no-cell files, error paths."""
    rng = np.random.default_rng(5)
    F, N = 5, 37
    coords = rng.normal(0, 20, (F, 3, N)).astype(np.float32)
    ortho = V.make_unitcell((31.5, 28.25, 40.0))
    tri = [V.make_unitcell((30.0 + f, 28.0, 26.0), tilt=(6.0, -4.0 + 0.5 * f, 5.0)) for f in range(F)]
    for big in (False, True):
        p = tmp_path / f"o{int(big)}.dcd"
        V.write_dcd(p, coords, ortho, big_endian=big)
        t = V.DcdTrajectory(p, lib=emu_lib)
        assert (t.num_frames(), t.num_atoms()) == (F, N)
        for f in (0, F - 1, 2):                                   # random access
            xyz, cell = t.load_frame(f)
            np.testing.assert_array_equal(xyz, coords[f])
            assert (cell.x, cell.y, cell.z, cell.xy, cell.xz, cell.yz, cell.flags) == (31.5, 28.25, 40.0, 0.0, 0.0, 0.0, 7)
        t.close()
    for cosines in (False, True):
        p = tmp_path / f"t{int(cosines)}.dcd"
        V.write_dcd(p, coords, tri, cosines=cosines)
        t = V.DcdTrajectory(p, lib=emu_lib)
        for f in range(F):
            _, cell = t.load_frame(f)
            got = np.array([cell.x, cell.y, cell.z, cell.xy, cell.xz, cell.yz])
            want = np.array([tri[f].x, tri[f].y, tri[f].z, tri[f].xy, tri[f].xz, tri[f].yz])
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-5)
            assert cell.flags == 7
    p = tmp_path / "nocell.dcd"
    V.write_dcd(p, coords, None)
    _, cell = V.DcdTrajectory(p, lib=emu_lib).load_frame(1)
    assert cell.flags == 0 and cell.x == 0.0
    import pytest
    bad = tmp_path / "bad.dcd"
    bad.write_bytes(b"not a dcd file at all, but long enough to hold a header ........................................................")
    with pytest.raises(V.VmdError, match="not a DCD"):
        V.DcdTrajectory(bad, lib=emu_lib)
    with pytest.raises(V.VmdError, match="cannot open"):
        V.DcdTrajectory(tmp_path / "missing.dcd", lib=emu_lib)
    with pytest.raises(V.VmdError, match="out of range"):
        V.DcdTrajectory(tmp_path / "o0.dcd", lib=emu_lib).load_frame(F)


def test_dcd_trajectory_through_the_evaluator_on_emulator(tmp_path, emu_lib, oracle):
    """A DCD file staged through load_frame gives the same histogram as the same frames handed over from memory."""
    import cases
    box = 40.0
    coords = cases.water_box(oracle, 12, 1500, box, 5)
    cell = V.make_unitcell(box)
    p = tmp_path / "w.dcd"
    V.write_dcd(p, coords, cell)
    o = cases.oxygen(1500)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(1500, unitcell=cell)
    old = emu_lib.vmd_set_option(b"batch_frames", 2)          # several staged batches
    try:
        a = V.ScriptEval(5, ir); assert a.frame_range(sysm, V.DcdTrajectory(p, lib=emu_lib), 0, 5)
        b = V.ScriptEval(5, ir); assert b.frame_range(sysm, V.HostTrajectory(coords, cell), 0, 5)
    finally:
        emu_lib.vmd_set_option(b"batch_frames", old)
    np.testing.assert_array_equal(a.property_data("g").counts, b.property_data("g").counts)
    np.testing.assert_array_equal(a.property_data("g").weights64, b.property_data("g").weights64)
    assert a.property_data("g").counts.sum() > 0


def test_staged_batches_are_decoded_on_several_threads(tmp_path, emu_lib, oracle):
    """load_frame is called concurrently for the frames of one staged batch (VIAMD's decoders are re-entrant: its pool threads
    call md_trajectory_load_frame at once, src/main.cpp:995-996): results must not depend on the thread count, and a failing
    frame must surface as an error naming it."""
    import cases
    import pytest
    box, F, N = 30.0, 40, 300
    coords = cases.water_box(oracle, 21, N, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "t.dcd"
    V.write_dcd(p, coords, cell)
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 8.0)
    sysm = V.MolSystem(N, unitcell=cell)
    res = []
    for threads in (1, 8):
        old = emu_lib.vmd_set_option(b"load_threads", threads)
        try:
            for traj in (V.DcdTrajectory(p, lib=emu_lib), V.HostTrajectory(coords, cell)):
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, traj, 0, F)
                res.append(ev.property_data("g").counts.copy())
        finally:
            emu_lib.vmd_set_option(b"load_threads", old)
    for r in res[1:]:
        np.testing.assert_array_equal(r, res[0])
    assert res[0].sum() > 0
    # a truncated file: the frame count comes from the file size, so cut inside the last frame and ask for it
    data = p.read_bytes()
    q = tmp_path / "cut.dcd"
    q.write_bytes(data[:len(data) - 100])
    t = V.DcdTrajectory(q, lib=emu_lib)
    assert t.num_frames() == F - 1
    ev = V.ScriptEval(F, ir)
    with pytest.raises(V.VmdError, match="fewer frames"):
        ev.frame_range(sysm, t, 0, F)


def test_xyz_and_lammps_dump_readers(tmp_path):
    from viamd_amd import textio
    rng = np.random.default_rng(8)
    F, N = 3, 11
    coords = rng.normal(0, 9, (F, 3, N)).astype(np.float32)
    elems = np.array(["O", "H", "H"] * 4)[:N]
    cells = [V.make_unitcell((20.0 + f, 18.0, 16.0), tilt=(2.0, -1.0, 0.5 * f)) for f in range(F)]
    p = tmp_path / "t.xyz"
    textio.write_xyz(p, coords, elems, cells)
    c2, e2, k2 = textio.read_xyz(p)
    np.testing.assert_allclose(c2, coords, atol=1e-6)
    assert list(e2) == list(elems)
    for a, b in zip(k2, cells):
        assert (a.x, a.y, a.z, a.xy, a.xz, a.yz, a.flags) == (b.x, b.y, b.z, b.xy, b.xz, b.yz, 7)
    textio.write_xyz(p, coords, elems)                       # plain XYZ: no cell
    assert textio.read_xyz(p)[2][0].flags == 0
    # LAMMPS dump, triclinic, atoms out of order, scaled coordinates in the second frame
    q = tmp_path / "d.lammpstrj"
    xy, xz, yz, L = 3.0, -2.0, 1.5, (24.0, 22.0, 20.0)
    lo = (-1.0, 2.0, 0.5)
    with open(q, "w") as f:
        for m in range(2):
            f.write(f"ITEM: TIMESTEP\n{100 * m}\nITEM: NUMBER OF ATOMS\n{N}\nITEM: BOX BOUNDS xy xz yz pp pp pp\n")
            f.write(f"{lo[0] + min(0, xy, xz, xy + xz)} {lo[0] + L[0] + max(0, xy, xz, xy + xz)} {xy}\n")
            f.write(f"{lo[1] + min(0, yz)} {lo[1] + L[1] + max(0, yz)} {xz}\n{lo[2]} {lo[2] + L[2]} {yz}\n")
            order = rng.permutation(N)
            if m == 0:
                f.write("ITEM: ATOMS id type x y z\n")
                for i in order:
                    f.write(f"{i + 1} {1 + i % 2} {float(coords[m, 0, i])!r} {float(coords[m, 1, i])!r} {float(coords[m, 2, i])!r}\n")
            else:
                f.write("ITEM: ATOMS id type xs ys zs\n")
                A = np.array([[L[0], xy, xz], [0, L[1], yz], [0, 0, L[2]]])
                s = np.linalg.solve(A, coords[m].astype(np.float64) - np.array(lo)[:, None])
                for i in order:
                    f.write(f"{i + 1} {1 + i % 2} {float(s[0, i])!r} {float(s[1, i])!r} {float(s[2, i])!r}\n")
    c3, types, k3, steps = textio.read_lammps_dump(q)
    assert steps == [0, 100] and list(types) == [1 + i % 2 for i in range(N)]
    np.testing.assert_allclose(c3, coords[:2], atol=2e-5)
    assert abs(k3[0].x - L[0]) < 1e-5 and abs(k3[0].y - L[1]) < 1e-5 and abs(k3[0].z - L[2]) < 1e-5
    assert (abs(k3[1].xy - xy), abs(k3[1].xz - xz), abs(k3[1].yz - yz)) < (1e-6, 1e-6, 1e-6) and k3[0].flags == 7


def test_gro_reader_multi_frame_and_triclinic_box(tmp_path):
    from viamd_amd import textio
    rng = np.random.default_rng(3)
    F, N = 2, 9
    coords = rng.uniform(-5, 45, (F, 3, N)).astype(np.float32)
    resid = np.repeat(np.arange(1, 4), 3); resname = np.array(["SOL"] * N); name = np.array(["OW", "HW1", "HW2"] * 3)
    cells = [V.make_unitcell((40.0, 38.0, 36.0)), V.make_unitcell((40.0, 38.0, 36.0), tilt=(5.0, -3.0, 4.0))]
    p = tmp_path / "w.gro"
    textio.write_gro(p, coords, resid, resname, name, cells)
    c2, meta, k2 = textio.read_gro(p)
    np.testing.assert_allclose(c2, coords, atol=6e-3)              # %8.3f in nm = 0.005 A
    assert list(meta["name"]) == list(name) and list(meta["resid"]) == list(resid) and meta["resname"][0] == "SOL"
    assert (k2[0].x, k2[0].y, k2[0].z, k2[0].xy, k2[0].flags) == (40.0, 38.0, 36.0, 0.0, 7)
    np.testing.assert_allclose([k2[1].xy, k2[1].xz, k2[1].yz], [5.0, -3.0, 4.0], atol=1e-4)


def test_mmcif_lammps_data_and_loader_dispatch(tmp_path, emu_lib, oracle):
    """The loader table of src/loader.cpp:22-77 (type from the extension, System / Trajectory flags) in front of the readers:
    mmCIF (two models, quoted tokens, missing label_seq_id), LAMMPS data (atom style from the Atoms comment, unsorted ids,
    tilt), and every trajectory type opening through one call."""
    import cases
    import pytest
    from viamd_amd import loader, textio
    cif = tmp_path / "m.cif"
    cif.write_text("""data_TEST
#
_cell.length_a 30.000
_cell.length_b 28.0(1)
_cell.length_c 26.000
_cell.angle_alpha 90.00
_cell.angle_beta 90.00
_cell.angle_gamma 80.00
#
loop_
_atom_site.group_PDB
_atom_site.id
_atom_site.type_symbol
_atom_site.label_atom_id
_atom_site.label_comp_id
_atom_site.label_asym_id
_atom_site.label_seq_id
_atom_site.Cartn_x
_atom_site.Cartn_y
_atom_site.Cartn_z
_atom_site.auth_seq_id
_atom_site.pdbx_PDB_model_num
ATOM 1 N N ALA A 1 1.000 2.000 3.000 5 1
ATOM 2 C CA ALA A 1 2.000 2.500 3.500 5 1
ATOM 3 C "C'" GLY A 2 3.000 3.500 4.500 6 1
HETATM 4 O O HOH B . 9.000 9.500 9.750 101 1
HETATM 5 ZN ZN 'ZN' C . 5.000 5.500 5.750 201 1
ATOM 1 N N ALA A 1 1.100 2.100 3.100 5 2
ATOM 2 C CA ALA A 1 2.100 2.600 3.600 5 2
ATOM 3 C "C'" GLY A 2 3.100 3.600 4.600 6 2
HETATM 4 O O HOH B . 9.100 9.600 9.850 101 2
HETATM 5 ZN ZN 'ZN' C . 5.100 5.600 5.850 201 2
#
loop_
_other.thing
x
""")
    coords, meta, params = textio.read_mmcif(cif)
    assert coords.shape == (2, 3, 5) and params == (30.0, 28.0, 26.0, 90.0, 90.0, 80.0)
    np.testing.assert_allclose(coords[1, :, 3], [9.1, 9.6, 9.85], rtol=1e-6)
    assert list(meta["name"]) == ["N", "CA", "C'", "O", "ZN"] and list(meta["resid"]) == [1, 1, 2, 101, 201]
    topo, xyz0, cell = loader.load_system(cif)
    assert topo.num_residues == 4 and list(topo.elements) == ["N", "C", "C", "O", "Zn"]
    np.testing.assert_allclose([cell.x, cell.y, cell.xy], [30.0, 28.0 * np.sin(np.radians(80)), 28.0 * np.cos(np.radians(80))], rtol=1e-6)
    assert abs(float(topo.mass[4]) - 65.38) < 1e-3 and cell.flags == 7

    data = tmp_path / "w.data"
    data.write_text("""LAMMPS data file via test

6 atoms
2 atom types

-1.0 19.0 xlo xhi
0.0 20.0 ylo yhi
0.0 22.0 zlo zhi
2.0 -1.0 0.5 xy xz yz

Masses

1 15.9994
2 1.008

Atoms # full

3 1 2 0.4 1.5 1.0 1.0 0 0 0
1 1 1 -0.8 1.0 1.0 1.0
2 1 2 0.4 0.5 1.0 1.0
6 2 2 0.4 6.5 6.0 6.0
4 2 1 -0.8 6.0 6.0 6.0
5 2 2 0.4 5.5 6.0 6.0

Velocities

1 0 0 0
""")
    topo, xyz0, cell = loader.load_system(data)
    assert list(topo.elements) == ["O", "H", "H", "O", "H", "H"] and topo.num_residues == 2
    np.testing.assert_array_equal(xyz0[0], np.array([1.0, 0.5, 1.5, 6.0, 5.5, 6.5], np.float32))
    assert (cell.x, cell.y, cell.z, cell.xy, cell.xz, cell.yz) == (20.0, 20.0, 22.0, 2.0, -1.0, 0.5)
    with pytest.raises(ValueError, match="atom style"):
        bad = tmp_path / "nostyle.data"
        bad.write_text(data.read_text().replace("Atoms # full", "Atoms"))
        loader.load_system(bad)
    topo2, _, _ = loader.load_system(bad, atom_style="full")
    assert topo2.num_atoms == 6

    # one call opens every trajectory type; the same frames give the same histogram whatever the container
    box, F, N = 30.0, 4, 300
    coords = cases.water_box(oracle, 77, N, box, F)
    cellb = V.make_unitcell(box)
    V.write_dcd(tmp_path / "t.dcd", coords, cellb)
    V.write_trr(tmp_path / "t.trr", coords, cellb, lib=emu_lib)
    V.write_xtc(tmp_path / "t.xtc", coords, cellb, lib=emu_lib)
    textio.write_xyz(tmp_path / "t.xyz", coords, ["O", "H", "H"] * (N // 3), [cellb] * F)
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 8.0)
    sysm = V.MolSystem(N, unitcell=cellb)
    counts = {}
    for ext in ("dcd", "trr", "xtc", "xyz"):
        traj = loader.open_trajectory(tmp_path / f"t.{ext}", lib=emu_lib)
        assert (traj.num_frames(), traj.num_atoms()) == (F, N)
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, traj, 0, F)
        counts[ext] = ev.property_data("g").counts.copy()
    np.testing.assert_array_equal(counts["dcd"], cases.check_rdf(emu_lib, oracle, coords, box, [("g", o, o, 0.0, 8.0)]).property_data("g").counts)
    assert counts["dcd"].sum() > 0 and abs(int(counts["xtc"].sum()) - int(counts["dcd"].sum())) < 0.01 * counts["dcd"].sum()
    assert loader.loader_type("a/b/c.XTC") == ("xtc", loader.FLAG_TRAJECTORY)
    with pytest.raises(ValueError, match="no trajectory"):
        loader.open_trajectory(tmp_path / "x.gro")
    with pytest.raises(ValueError, match="no system"):
        loader.load_system(tmp_path / "t.dcd")
    with pytest.raises(ValueError, match="could not determine loader type"):
        loader.loader_type("file.unknown")


def test_export_in_the_product_library_on_the_emulator(tmp_path, emu_lib, oracle):
    import cases
    cases.export_cases(emu_lib, oracle, tmp_path)
