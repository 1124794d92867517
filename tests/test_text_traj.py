"""Native text trajectory readers (viamd_amd/csrc/vmd_text.cpp: multi-MODEL PDB, XYZ / XMOL, LAMMPS dump) against the independent Python
readers of the same formats (viamd_amd/pdb.py, viamd_amd/textio.py): every frame bit for bit, cells, errors, and an evaluation straight from
the file.  Pure host code: runs on the emulator build and on the product library (which loads without a GPU)."""
import numpy as np
import pytest

import viamd_amd as V
from viamd_amd import loader, pdb, textio
from viamd_amd.script import Topology


def _topo(n):
    elems = (["O", "H", "H"] * ((n + 2) // 3))[:n]
    return Topology(elems, ["SOL"] * n, [i // 3 for i in range(n)], [e + "W" for e in elems])


def _coords(F, n, seed=3, scale=40.0):
    rng = np.random.default_rng(seed)
    c = (rng.random((F, 3, n)) * scale - 5.0).astype(np.float32)
    c[0, :, 0] = (0.0, -0.0004, 9999.999)          # zero, a value that prints as -0.000, the widest field of %8.3f
    return c


def _frames(t):
    return np.stack([t.load_frame(f)[0] for f in range(t.num_frames())])


def _cell_tuple(c):
    return (c.x, c.y, c.z, c.xy, c.xz, c.yz, c.flags)


def test_multi_model_pdb(tmp_path, host_lib):
    F, n = 7, 50
    coords, topo = _coords(F, n), _topo(n)
    p = tmp_path / "traj.pdb"
    pdb.write_pdb(p, coords, topo, box=(30.0, 28.0, 26.0), tilt=(6.0, -4.0, 5.0))
    want, _, cell = pdb.read_pdb(p)
    t = V.TextTrajectory(p, lib=host_lib)
    assert (t.num_frames(), t.num_atoms()) == (F, n)
    np.testing.assert_array_equal(_frames(t), want)                   # the text holds %8.3f: both readers must round it the same way
    for f in (0, F - 1):
        assert _cell_tuple(t.load_frame(f)[1]) == _cell_tuple(cell)
    assert loader.open_trajectory(p, lib=host_lib).num_frames() == F
    # no MODEL records, HETATM, a TER line between chains, CRLF line ends, no CRYST1: one frame, no cell
    lines = ["REMARK single model\r\n"]
    for i in range(5):
        rec = "HETATM" if i == 3 else "ATOM  "
        lines.append("%s%5d  CA  ALA A%4d    %8.3f%8.3f%8.3f  1.00  0.00           C\r\n" % (rec, i + 1, i + 1, -1.5 * i, 2.25 * i, 1000.125 + i))
        if i == 2:
            lines.append("TER\r\n")
    lines.append("END\r\n")
    q = tmp_path / "single.pdb"
    q.write_bytes("".join(lines).encode())
    t1 = V.TextTrajectory(q, lib=host_lib)
    assert (t1.num_frames(), t1.num_atoms()) == (1, 5)
    xyz, c1, _ = t1.load_frame(0)
    np.testing.assert_array_equal(xyz, pdb.read_pdb(q)[0][0])
    assert c1.flags == 0 and xyz[2, 4] == np.float32(1004.125)
    # a MODEL with a different atom count is refused when the file is opened
    bad = tmp_path / "bad.pdb"
    text = p.read_text().split("\n")
    k = max(i for i, l in enumerate(text) if l.startswith("ATOM"))
    bad.write_text("\n".join(text[:k] + text[k + 1:]))
    with pytest.raises(V.VmdError, match="atoms"):
        V.TextTrajectory(bad, lib=host_lib)


def test_xyz_and_extended_xyz(tmp_path, host_lib):
    F, n = 5, 33
    coords = _coords(F, n, seed=5, scale=12.0)
    elements = np.array((["C", "N", "O"] * n)[:n])
    cells = [V.make_unitcell((20.0 + f, 21.0, 22.0), tilt=(1.5, -0.5, 0.25 * f)) for f in range(F)]
    p = tmp_path / "t.xyz"
    textio.write_xyz(p, coords, elements, cells)
    want, _, wcells = textio.read_xyz(p)
    t = V.TextTrajectory(p, lib=host_lib)
    assert (t.num_frames(), t.num_atoms()) == (F, n)
    np.testing.assert_array_equal(_frames(t), want)
    for f in range(F):
        assert _cell_tuple(t.load_frame(f)[1]) == _cell_tuple(wcells[f])
    # numbers the fast path does not take (17+ significant digits, exponents) and blank lines between frames
    q = tmp_path / "odd.xmol"
    q.write_text("2\nframe 0\nAr 1.2345678901234567890 -2.5e-3 3E+2\nAr 0.1 1e-30 123456789012345678\n\n2\nframe 1\nAr 1 2 3\nAr -0 +7.25 .5\n")
    t2 = V.TextTrajectory(q, lib=host_lib)
    got, want2 = _frames(t2), textio.read_xyz(q)[0]
    np.testing.assert_array_equal(got, want2)
    assert got[0, 0, 0] == np.float32(1.2345678901234567890) and got[0, 2, 1] == np.float32(123456789012345678.0) and t2.load_frame(0)[1].flags == 0
    with pytest.raises(V.VmdError, match="lower triangular"):
        r = tmp_path / "r.xyz"
        r.write_text('1\nLattice="10 1 0 0 10 0 0 0 10"\nAr 0 0 0\n')
        V.TextTrajectory(r, lib=host_lib)
    with pytest.raises(V.VmdError, match="incomplete|missing"):
        r = tmp_path / "s.xyz"
        r.write_text("2\nc\nAr 0 0 0\nAr 1 1\n")
        V.TextTrajectory(r, lib=host_lib).load_frame(0)


def test_tinker_arc(tmp_path, host_lib):
    """a Tinker archive has NO comment line and its atom lines start with an index (`idx sym x y z type bonds...`); a periodic
    one carries a box line (a b c alpha beta gamma) after the count.  Both flavours, a Tinker-format .xyz, a file that changes flavour half
    way, and counts that must be refused before they are cast."""
    F, n = 4, 7
    coords = _coords(F, n, seed=9, scale=15.0)
    sym = ["O", "H", "H", "C", "N", "Cl", "Na+"]

    def write(path, box, first_title="water box"):
        with open(path, "w") as f:
            for m in range(F):
                f.write(f"{n:6d}  {first_title if m == 0 else ''}\n")
                if box:
                    f.write(f"{30.0 + m:12.6f}{31.0:12.6f}{32.0:12.6f}{90.0:12.6f}{90.0:12.6f}{90.0 if box == 'ortho' else 75.0:12.6f}\n")
                for k in range(n):
                    x, y, z = (float(v) for v in coords[m, :, k])
                    f.write(f"{k + 1:6d}  {sym[k]:3s}{x:12.6f}{y:12.6f}{z:12.6f}{1 + k % 3:6d}{(k % n) + 1:6d}{((k + 1) % n) + 1:6d}\n")

    want = np.asarray([[[np.float32(f"{float(coords[m, a, k]):.6f}") for k in range(n)] for a in range(3)] for m in range(F)], np.float32)
    for name, box in (("plain.arc", None), ("ortho.arc", "ortho"), ("tri.arc", "tri"), ("tinker.xyz", "ortho")):
        p = tmp_path / name
        write(p, box)
        t = V.TextTrajectory(p, lib=host_lib)
        assert (t.num_frames(), t.num_atoms()) == (F, n), name
        np.testing.assert_array_equal(_frames(t), want, err_msg=name)
        cell = t.load_frame(2)[1]
        if box is None:
            assert cell.flags == 0, name
        else:
            assert cell.flags == 7 and cell.x == np.float32(32.0) and cell.z == np.float32(32.0), name
            assert (cell.xy == 0.0) == (box == "ortho") and (box == "ortho" or abs(cell.xy - 31.0 * np.cos(np.deg2rad(75.0))) < 1e-4), name
    # a plain XYZ frame after Tinker frames: refused, not misread
    mixed = tmp_path / "mixed.arc"
    write(mixed, None)
    with open(mixed, "a") as f:
        f.write(f"{n}\ncomment\n" + "".join(f"{sym[k]} 0 0 {k}\n" for k in range(n)))
    with pytest.raises(V.VmdError, match="layout"):
        V.TextTrajectory(mixed, lib=host_lib)
    # counts: not integral, absurd (would overflow the cast), and the same for a LAMMPS dump
    for bad, pat in (("2.5\nc\nAr 0 0 0\n", "atom count"), ("1e300\nc\nAr 0 0 0\n", "atom count"), ("99999999999\nc\nAr 0 0 0\n", "atom count")):
        q = tmp_path / "bad.xyz"
        q.write_text(bad)
        with pytest.raises(V.VmdError, match=pat):
            V.TextTrajectory(q, lib=host_lib)
    for cnt in ("2.5", "1e300"):
        q = tmp_path / "bad.lammpstrj"
        q.write_text(f"ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n{cnt}\nITEM: BOX BOUNDS pp pp pp\n0 1\n0 1\n0 1\nITEM: ATOMS id x y z\n1 0 0 0\n")
        with pytest.raises(V.VmdError, match="atom count"):
            V.TextTrajectory(q, lib=host_lib)


def _write_dump(path, coords, ids_per_frame, scaled, tri, steps):
    F, _, n = coords.shape
    with open(path, "w") as f:
        for m in range(F):
            xlo, ylo, zlo, lx, ly, lz = -3.0, 1.0, 0.5, 30.0 + m, 28.0, 26.0
            xy, xz, yz = (4.0, -2.0, 3.0) if tri else (0.0, 0.0, 0.0)
            f.write(f"ITEM: TIMESTEP\n{steps[m]}\nITEM: NUMBER OF ATOMS\n{n}\n")
            if tri:
                bx = (xlo + min(0.0, xy, xz, xy + xz), xlo + lx + max(0.0, xy, xz, xy + xz))
                by = (ylo + min(0.0, yz), ylo + ly + max(0.0, yz))
                f.write(f"ITEM: BOX BOUNDS xy xz yz pp pp ff\n{bx[0]!r} {bx[1]!r} {xy!r}\n{by[0]!r} {by[1]!r} {xz!r}\n{zlo!r} {zlo + lz!r} {yz!r}\n")
            else:
                f.write(f"ITEM: BOX BOUNDS pp pp pp\n{xlo!r} {xlo + lx!r}\n{ylo!r} {ylo + ly!r}\n{zlo!r} {zlo + lz!r}\n")
            cols = "id type xs ys zs vx" if scaled else "type x y z id"
            f.write(f"ITEM: ATOMS {cols}\n")
            for k in ids_per_frame[m]:                       # atoms in a different order in every frame: the id column sorts them
                x, y, z = (float(v) for v in coords[m, :, k])
                if scaled:
                    sz = (z - zlo) / lz; sy = (y - ylo - sz * yz) / ly; sx = (x - xlo - sy * xy - sz * xz) / lx
                    f.write(f"{k + 1} {1 + k % 2} {sx!r} {sy!r} {sz!r} 0.0\n")
                else:
                    f.write(f"{1 + k % 2} {x!r} {y!r} {z!r} {k + 1}\n")


@pytest.mark.parametrize("scaled,tri", [(False, False), (True, False), (True, True)])
def test_lammps_dump(tmp_path, host_lib, scaled, tri):
    F, n = 4, 41
    coords = _coords(F, n, seed=9, scale=20.0)
    rng = np.random.default_rng(1)
    order = [rng.permutation(n) for _ in range(F)]
    steps = [0, 500, 1000, 2500]
    p = tmp_path / "d.lammpstrj"
    _write_dump(p, coords, order, scaled, tri, steps)
    want, _, wcells, wsteps = textio.read_lammps_dump(p)
    t = V.TextTrajectory(p, lib=host_lib)
    assert (t.num_frames(), t.num_atoms()) == (F, n)
    np.testing.assert_array_equal(_frames(t), want)
    if not scaled:
        np.testing.assert_array_equal(want, coords)                 # repr() round-trips a float32 through its double
    for f in range(F):
        _, cell, ts = t.load_frame(f)
        assert _cell_tuple(cell) == _cell_tuple(wcells[f]) and ts == float(wsteps[f])
    assert t.load_frame(0)[1].flags == (3 if tri else 7)


def test_evaluation_straight_from_a_pdb_trajectory(tmp_path, emu_lib, oracle):
    """BASELINE configs[0] in small: a multi-MODEL PDB, `rdf(element('O'), element('O'), 10.0)`, evaluated through the native reader (frames
    parsed on the staging threads) - the oracle's integers on the coordinates the file holds."""
    import cases
    from viamd_amd import script
    F, n, box = 6, 600, 24.0
    coords = cases.water_box(oracle, 4, n, box, F)
    topo = _topo(n)
    p = tmp_path / "water.pdb"
    pdb.write_pdb(p, coords, topo, box=box)
    held = pdb.read_pdb(p)[0]                                        # three decimals survive the file
    traj = loader.open_trajectory(p, lib=emu_lib)
    assert isinstance(traj, V.TextTrajectory) and traj.num_frames() == F
    ir, info = script.compile_script("g = rdf(element('O'), element('O'), 10.0);", topo, lib=emu_lib)
    ev = V.ScriptEval(F, ir)
    old = emu_lib.vmd_set_option(b"load_threads", 3)
    try:
        assert ev.frame_range(V.MolSystem(n, mass=topo.mass, unitcell=traj.load_frame(0)[1]), traj, 0, F)
    finally:
        emu_lib.vmd_set_option(b"load_threads", old)
    o = np.arange(0, n, 3, dtype=np.int32)
    counts, _ = cases.oracle_rdf(oracle, held, oracle.make_cell(box), o, o, 0.0, 10.0)
    np.testing.assert_array_equal(ev.property_data("g").counts, counts)
    assert counts.sum() > 0


def test_pdb_system_and_a_cpp_host_running_config_1_from_the_file_alone(tmp_path, emu_lib, oracle):
    import conftest
    _pdb_host_case(tmp_path, emu_lib, conftest.build_emu(), oracle)


@pytest.mark.gpu
def test_cpp_host_runs_config_1_from_a_pdb_file_on_the_gpu(tmp_path, gpu_lib, oracle):
    from viamd_amd import build
    _pdb_host_case(tmp_path, gpu_lib, build.build(), oracle, n_w=3000, box=46.0, F=12)


def _pdb_host_case(tmp_path, emu_lib, emu, oracle, n_w=150, box=22.0, F=5):
    """BASELINE configs[0] (`datasets/1ALA-500.pdb`, `rdf(element('O'),element('O'),10.0)`: the blob is missing from the reference, so a
    stand-in of the same kind - a capped alanine in water, multi-MODEL PDB): the native PDB SYSTEM reader equals the Python one (elements,
    names, residues, resSeq, masses, first-frame coordinates, cell), and a C++ program that is given nothing but the file and the script
    string (tests/native/cabi_pdb_demo.cpp) prints the sums the Python host gets through its own readers."""
    import ctypes as C
    import os
    import subprocess
    import cases
    from viamd_amd import _lib as L, script
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["N", "CA", "C", "O", "CB", "H", "HA", "HB1", "HB2", "HB3"]
    elems = ["N", "C", "C", "O", "C", "H", "H", "H", "H", "H"]
    n = 10 + 3 * n_w
    topo = Topology(elems + ["O", "H", "H"] * n_w, ["ALA"] * 10 + ["HOH"] * (3 * n_w), [0] * 10 + [1 + i // 3 for i in range(3 * n_w)],
                    names + ["OW", "HW1", "HW2"] * n_w)
    coords = cases.water_box(oracle, 8, n, box, F)
    p = tmp_path / "ala.pdb"
    pdb.write_pdb(p, coords, topo, box=box)
    want_c, want_t, want_cell = pdb.read_pdb(p)
    h = emu_lib.vmd_textsys_open(str(p).encode())
    assert h, emu_lib.last_error()
    t = emu_lib.vmd_textsys_topology(h).contents
    assert t.num_atoms == n
    assert [t.elements[i].decode() for i in range(n)] == list(want_t.elements)
    assert [t.names[i].decode() for i in range(n)] == list(want_t.names) and [t.resnames[i].decode() for i in range(n)] == list(want_t.resnames)
    np.testing.assert_array_equal(np.ctypeslib.as_array(t.residue_index, (n,)), np.asarray(want_t.residue_index))
    np.testing.assert_array_equal(np.ctypeslib.as_array(t.residue_seq_id, (n,)), np.asarray(want_t.residue_seq_id))
    np.testing.assert_array_equal(np.ctypeslib.as_array(emu_lib.vmd_textsys_mass(h), (n,)), want_t.mass)
    cell = L.Unitcell()
    xyz = np.ctypeslib.as_array(emu_lib.vmd_textsys_coords(h, C.byref(cell)), (3, n)).copy()
    np.testing.assert_array_equal(xyz, want_c[0])
    assert _cell_tuple(cell) == _cell_tuple(want_cell)
    emu_lib.vmd_textsys_close(h)
    # the C++ host
    text = "g = rdf(element('O'), element('O'), 10.0); d = distance(resname('ALA'), residue(5));"
    exe = str(tmp_path / "cabi_pdb_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "native", "cabi_pdb_demo.cpp"), "-I" + os.path.join(ROOT, "include"), emu,
                           "-Wl,-rpath," + os.path.dirname(emu), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", exe])
    out = subprocess.run([exe, str(p), text], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")
    assert lines[0].startswith(f"atoms={n} frames={F} residues={1 + n_w} first=N/N/ALA last_mass=1.008 cell={box:.3f},{box:.3f},{box:.3f}"), lines[0]
    got = dict(l.split(" ", 1) for l in lines[1:])
    ir, info = script.compile_script(text, want_t, lib=emu_lib)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(n, mass=want_t.mass, unitcell=want_cell), V.HostTrajectory(want_c, want_cell), 0, F)
    assert float(got["g"].split("sum=")[1]) == float(ev.property_data("g").counts.sum()) > 0
    o = np.array([i for i in range(n) if want_t.elements[i] == "O"], np.int32)
    np.testing.assert_array_equal(ev.property_data("g").counts, cases.oracle_rdf(oracle, want_c, oracle.make_cell(box), o, o, 0.0, 10.0)[0])
    assert abs(float(got["d"].split("sum=")[1]) - float(ev.property_data("d").values.astype(np.float64).sum())) < 1e-5
