"""Damaged text trajectories: valid multi-MODEL PDB / XYZ / LAMMPS dump files with random bytes flipped, runs of bytes deleted or inserted, and the
tail cut off, handed to the native readers of csrc/vmd_text.cpp (and the PDB system reader).  Nothing to compare with - a damaged file has no right
answer; what is checked is that the reader either opens it and serves every frame it announces, or refuses with a message - no crash, and under
AddressSanitizer (scripts/sanitize_emu.sh builds the library; run this with VIAMD_EMU_SANITIZE=address,undefined and the runtime preloaded) no
access outside the mapping.  usage: python scripts/fuzz_text_damaged.py N SEED"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
import viamd_amd as V
from viamd_amd.texttraj import TextTrajectory
from viamd_amd.eval import VmdError

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
lib = V.VmdLib(conftest.build_emu())
rng = np.random.default_rng(seed)


def pdb(F, n):
    out = ["CRYST1   30.000   30.000   30.000  90.00  90.00  90.00 P 1           1\n"]
    for f in range(F):
        out.append("MODEL     %4d\n" % (f + 1))
        for i in range(n):
            x, y, z = rng.uniform(-99, 99, 3)
            out.append("ATOM  %5d  O   HOH A%4d    %8.3f%8.3f%8.3f  1.00  0.00           O\n" % (i + 1, i + 1, x, y, z))
        out.append("ENDMDL\n")
    out.append("END\n")
    return "".join(out).encode()


def xyz(F, n):
    out = []
    for f in range(F):
        out.append("%d\nframe %d\n" % (n, f))
        for i in range(n):
            out.append("O %.5f %.5f %.5f\n" % tuple(rng.uniform(-99, 99, 3)))
    return "".join(out).encode()


def lammps(F, n):
    out = []
    for f in range(F):
        out.append("ITEM: TIMESTEP\n%d\nITEM: NUMBER OF ATOMS\n%d\nITEM: BOX BOUNDS pp pp pp\n0 30\n0 30\n0 30\nITEM: ATOMS id type x y z\n" % (f * 100, n))
        for i in rng.permutation(n):
            out.append("%d 1 %.5f %.5f %.5f\n" % ((i + 1,) + tuple(rng.uniform(0, 30, 3))))
    return "".join(out).encode()


def damage(b):
    b = bytearray(b)
    for _ in range(int(rng.integers(1, 6))):
        k = rng.integers(0, 5)
        if not b: break
        p = int(rng.integers(0, len(b)))
        if k == 0: b[p] = int(rng.integers(0, 256))
        elif k == 1: del b[p:p + int(rng.integers(1, 40))]
        elif k == 2: b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
        elif k == 3: b[p:p] = rng.choice([b"\n", b"\r\n", b" ", b"1e999", b"-", b"nan", b"ITEM: ", b"MODEL", b"ENDMDL\n", b"99999999999999999999"])
        else: del b[p:]
    return bytes(b)


opened = refused = frames = 0
with tempfile.TemporaryDirectory() as d:
    for case in range(n_cases):
        F, n = int(rng.integers(1, 5)), int(rng.integers(1, 30))
        kind = case % 3
        data = damage([pdb, xyz, lammps][kind](F, n))
        ext = ["pdb", "xyz", "lammpstrj"][kind]
        p = os.path.join(d, f"c{case}.{ext}")
        open(p, "wb").write(data)
        try:
            t = TextTrajectory(p, lib=lib)
        except VmdError:
            refused += 1
        else:
            opened += 1
            nf, na = t.num_frames(), t.num_atoms()
            assert 0 <= nf < 10 ** 6 and 0 <= na < 10 ** 6, (case, nf, na)
            for f in range(nf):
                try:
                    t.load_frame(f); frames += 1
                except VmdError:
                    pass
            t.close()
        if kind == 0:          # the PDB system reader on the same bytes
            h = lib.vmd_textsys_open(p.encode())
            if h: lib.vmd_textsys_close(h)
        os.remove(p)
print(f"{n_cases} damaged files: {opened} opened ({frames} frames served), {refused} refused, no crash")
