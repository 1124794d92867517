"""Randomised hunt between the native text readers (csrc/vmd_text.cpp) and the Python readers (viamd_amd/pdb.py, textio.py): random multi-MODEL PDB,
XYZ and LAMMPS dump files with random number formats - every frame must come out bit for bit the same.  usage: python scripts/fuzz_text.py N SEED [emu]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import viamd_amd as V
from viamd_amd import pdb, textio

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "emu":
    import conftest
    lib = V.VmdLib(conftest.build_emu())
else:
    lib = V.default_lib()
rng = np.random.default_rng(seed)


def number(v):
    """one of the ways a float shows up in a text trajectory"""
    k = rng.integers(0, 7)
    if k == 0: return repr(float(v))
    if k == 1: return "%.3f" % v
    if k == 2: return "%.8e" % v
    if k == 3: return "%.17g" % v
    if k == 4: return "%d" % int(v)
    if k == 5: return "%+.5f" % v
    return "%.12f" % v


fails = 0
with tempfile.TemporaryDirectory() as d:
    for case in range(n_cases):
        F, n = int(rng.integers(1, 6)), int(rng.integers(1, 40))
        coords = (rng.standard_normal((F, 3, n)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        kind = case % 3
        try:
            if kind == 0:
                p = os.path.join(d, f"f{case}.pdb")
                nl = "\r\n" if rng.random() < 0.3 else "\n"
                with open(p, "w", newline="") as f:
                    if rng.random() < 0.5:
                        f.write("CRYST1%9.3f%9.3f%9.3f%7.2f%7.2f%7.2f P 1           1%s" % (30 + rng.random(), 31.0, 32.5, 90.0, 90.0 + 20 * (rng.random() < 0.5), 90.0, nl))
                    for m in range(F):
                        if F > 1 or rng.random() < 0.5: f.write("MODEL     %4d%s" % (m + 1, nl))
                        for i in range(n):
                            c = np.clip(coords[m, :, i], -999.0, 9999.0)
                            f.write("%s%5d  CA  ALA A%4d    %8.3f%8.3f%8.3f  1.00  0.00%s%s" % ("HETATM" if rng.random() < 0.2 else "ATOM  ", i + 1, i + 1, c[0], c[1], c[2],
                                                                                                 "           C" if rng.random() < 0.7 else "", nl))
                            if rng.random() < 0.1: f.write("TER" + nl)
                        f.write(("ENDMDL" if F > 1 or rng.random() < 0.5 else "END") + nl)
                want = pdb.read_pdb(p)[0]
            elif kind == 1:
                p = os.path.join(d, f"f{case}.xyz")
                with open(p, "w") as f:
                    for m in range(F):
                        f.write(f"{n}\ncomment {m}\n")
                        for i in range(n):
                            f.write("Ar " + " ".join(number(v) for v in coords[m, :, i]) + (" extra 1.0\n" if rng.random() < 0.2 else "\n"))
                        if rng.random() < 0.3: f.write("\n")
                want = textio.read_xyz(p)[0]
            else:
                p = os.path.join(d, f"f{case}.lammpstrj")
                with open(p, "w") as f:
                    for m in range(F):
                        f.write(f"ITEM: TIMESTEP\n{m * 10}\nITEM: NUMBER OF ATOMS\n{n}\nITEM: BOX BOUNDS pp pp pp\n0 50\n-1.5 40\n0.25 30.5\nITEM: ATOMS id x y z type\n")
                        for i in rng.permutation(n):
                            f.write(f"{i + 1} " + " ".join(number(v) for v in coords[m, :, i]) + " 1\n")
                want = textio.read_lammps_dump(p)[0]
            t = V.TextTrajectory(p, lib=lib)
            got = np.stack([t.load_frame(f)[0] for f in range(t.num_frames())])
            if got.shape != want.shape or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                fails += 1; print("MISMATCH", case, kind, got.shape, want.shape, flush=True)
        except Exception as ex:
            fails += 1; print("ERROR", case, kind, repr(ex)[:200], flush=True)
print(f"{n_cases} random text trajectories (seed {seed}): {fails} failures")
