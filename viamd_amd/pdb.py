"""Multi-MODEL PDB reader / writer (the first trajectory decoder; BASELINE config 1 is a PDB).

Only what the evaluator needs: ATOM/HETATM coordinates per MODEL, CRYST1 cell, element / residue columns."""
import numpy as np

from .eval import make_unitcell
from .script import Topology
from .synth import MASS


def write_pdb(path, coords, topo, box=None, tilt=(0.0, 0.0, 0.0)):
    """coords float [F, 3, N]; topo: Topology; box: None or (x, y, z); tilt = (xy, xz, yz) for a triclinic cell."""
    coords = np.asarray(coords)
    with open(path, "w") as f:
        if box is not None:
            b = (box, box, box) if np.isscalar(box) else box
            va, vb, vc = np.array([b[0], 0, 0.0]), np.array([tilt[0], b[1], 0.0]), np.array([tilt[1], tilt[2], b[2]])
            ang = lambda u, v: np.rad2deg(np.arccos(np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v))))
            f.write("CRYST1%9.3f%9.3f%9.3f%7.2f%7.2f%7.2f P 1           1\n" % (
                np.linalg.norm(va), np.linalg.norm(vb), np.linalg.norm(vc), ang(vb, vc), ang(va, vc), ang(va, vb)))
        for m in range(coords.shape[0]):
            f.write("MODEL     %4d\n" % (m + 1))
            for i in range(coords.shape[2]):
                f.write("ATOM  %5d %-4s %-3s A%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % (
                    (i + 1) % 100000, str(topo.names[i])[:4], str(topo.resnames[i])[:3], (int(topo.residue_index[i]) + 1) % 10000,
                    coords[m, 0, i], coords[m, 1, i], coords[m, 2, i], 1.0, 0.0, str(topo.elements[i])[:2].rjust(2)))
            f.write("ENDMDL\n")
        f.write("END\n")


def read_pdb(path):
    """Returns (coords float32 [F, 3, N], Topology, unitcell)."""
    frames, cur = [], []
    elems, names, resn, resi = [], [], [], []
    box, tilt = None, (0.0, 0.0, 0.0)
    first = True
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec == "CRYST1":
                a, b, c = float(line[6:15]), float(line[15:24]), float(line[24:33])
                al, be, ga = (np.deg2rad(float(line[33:40])), np.deg2rad(float(line[40:47])), np.deg2rad(float(line[47:54])))
                # (a, b, c, alpha, beta, gamma) -> lower-triangular basis a=(x,0,0), b=(xy,y,0), c=(xz,yz,z)
                xy, xz = b * np.cos(ga), c * np.cos(be)
                ly = np.sqrt(b * b - xy * xy)
                yz = (b * c * np.cos(al) - xy * xz) / ly
                lz = np.sqrt(c * c - xz * xz - yz * yz)
                box = (a, ly, lz)
                tilt = tuple(0.0 if abs(v) < 1e-6 else float(v) for v in (xy, xz, yz))
            elif rec in ("ATOM  ", "HETATM"):
                cur.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                if first:
                    el = line[76:78].strip() or line[12:16].strip()[:1]
                    elems.append(el.capitalize()); names.append(line[12:16].strip()); resn.append(line[17:20].strip())
                    resi.append((line[21], int(line[22:26])))
            elif rec == "ENDMDL" or (rec.startswith("END") and cur):
                if cur:
                    frames.append(np.array(cur, np.float32).T)
                    cur, first = [], False
    if cur:
        frames.append(np.array(cur, np.float32).T)
    coords = np.stack(frames)
    # residue index: a new residue whenever (chain, resSeq) changes
    ridx, last, k = [], None, -1
    for r in resi:
        if r != last:
            k += 1
            last = r
        ridx.append(k)
    mass = np.array([MASS.get(e, 12.0) for e in elems], np.float32)
    topo = Topology(elems, resn, ridx, names, mass=mass, residue_seq_id=[r[1] for r in resi])
    cell = make_unitcell(box, tilt=tilt) if box is not None else make_unitcell(None)
    return coords, topo, cell
