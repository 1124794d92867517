"""File-type dispatch in front of the evaluator, mirroring VIAMD's loader table (/root/reference/src/loader.cpp:22-77: names,
extensions and System / Trajectory flags per type; `load::init` picks the type from the extension, `load::load` calls the
mdlib reader, :111-159).

    topo, coords0, cell = loader.load_system("box.gro")            # LoaderFlag_System: atoms + first frame
    traj = loader.open_trajectory("run.xtc")                        # LoaderFlag_Trajectory: md_trajectory_i stand-in
    ev.frame_range(MolSystem(traj.num_atoms(), mass=topo.mass, unitcell=cell), traj, 0, traj.num_frames())

Trajectories go through the native readers (random access, decoded on the evaluator's staging threads): DCD, XTC, TRR and - since
round 4 - the text formats (multi-MODEL PDB, XYZ / XMOL, LAMMPS dump: csrc/vmd_text.cpp).  Systems (topologies) are read in Python.
"""
import os

import numpy as np

from . import pdb, textio
from .script import Topology
from .synth import MASS

FLAG_SYSTEM, FLAG_TRAJECTORY = 1, 2

# (type name, extensions, flags) — loader.cpp:22-77
LOADERS = [
    ("pdb", ("pdb",), FLAG_SYSTEM | FLAG_TRAJECTORY),
    ("gro", ("gro",), FLAG_SYSTEM),
    ("xyz", ("xyz", "xmol", "arc"), FLAG_SYSTEM | FLAG_TRAJECTORY),
    ("cif", ("cif",), FLAG_SYSTEM),
    ("lammps_data", ("data",), FLAG_SYSTEM),
    ("lammpstrj", ("lammpstrj",), FLAG_TRAJECTORY),
    ("xtc", ("xtc",), FLAG_TRAJECTORY),
    ("trr", ("trr",), FLAG_TRAJECTORY),
    ("dcd", ("dcd",), FLAG_TRAJECTORY),
]


def loader_type(path):
    """-> (type name, flags) from the file extension (load::init, loader.cpp:79-84); ValueError when unknown."""
    ext = os.path.splitext(str(path))[1].lstrip(".").lower()
    for name, exts, flags in LOADERS:
        if ext in exts:
            return name, flags
    raise ValueError(f"could not determine loader type from file extension '{ext}'")


def _guess_element(name):
    """Element from an atom name / type label: leading letters, two-letter symbols only when they are known."""
    s = "".join(ch for ch in str(name) if ch.isalpha())
    two = s[:2].capitalize()
    # atom names like CA / CD / NE / HG / CO are carbons, nitrogens and hydrogens of residues far more often than metals:
    # a two-letter symbol is taken only when its first letter is not itself a common organic element, or the name IS the symbol
    if len(s) >= 2 and two in MASS and (s[:1].upper() not in "CNOHSP" or (len(s) == 2 and str(name) == two)):
        return two
    return s[:1].upper() if s else "X"


def _residue_index(keys):
    out, last, k = [], None, -1
    for key in keys:
        if key != last:
            k, last = k + 1, key
        out.append(k)
    return out


def _element_from_mass(m):
    if m <= 0:
        return "X"
    return min(MASS, key=lambda e: abs(MASS[e] - m))


def load_system(path, atom_style=None):
    """-> (Topology, coords float32 [3, N] of the first frame, Unitcell) for every type with LoaderFlag_System."""
    kind, flags = loader_type(path)
    if not flags & FLAG_SYSTEM:
        raise ValueError(f"'{path}': a {kind} file holds no system (atoms), only a trajectory")
    if kind == "pdb":
        coords, topo, cell = pdb.read_pdb(path)
        return topo, coords[0], cell
    if kind == "gro":
        coords, meta, cells = textio.read_gro(path)
        elems = [_guess_element(n) for n in meta["name"]]
        topo = Topology(elems, meta["resname"], _residue_index(zip(meta["resid"], meta["resname"])), meta["name"],
                        mass=np.array([MASS.get(e, 12.0) for e in elems], np.float32), residue_seq_id=meta["resid"])
        return topo, coords[0], cells[0]
    if kind == "xyz":
        coords, elements, cells = textio.read_xyz(path)
        elems = [_guess_element(e) for e in elements]
        topo = Topology(elems, mass=np.array([MASS.get(e, 12.0) for e in elems], np.float32))
        return topo, coords[0], cells[0]
    if kind == "cif":
        coords, meta, params = textio.read_mmcif(path)
        topo = Topology(meta["element"], meta["resname"], _residue_index(zip(meta["chain"], meta["resid"], meta["resname"])),
                        meta["name"], mass=np.array([MASS.get(e, 12.0) for e in meta["element"]], np.float32),
                        residue_seq_id=meta["resid"])
        cell = textio.cell_from_parameters(*params) if params else textio.make_unitcell(None)
        return topo, coords[0], cell
    coords, meta, cell = textio.read_lammps_data(path, atom_style)
    elems = [_element_from_mass(m) for m in meta["mass"]]
    mol = meta["mol"] if meta["mol"].any() else np.zeros(len(elems), np.int64)
    topo = Topology(elems, residue_index=_residue_index(mol), names=[str(t) for t in meta["type"]], mass=meta["mass"])
    return topo, coords[0], cell


def open_trajectory(path, lib=None):
    """-> an object with interface() / num_frames() / num_atoms() for ScriptEval.frame_range, for every type with
    LoaderFlag_Trajectory (md_*_attach_from_file / the multi-frame text readers)."""
    from .dcd import DcdTrajectory
    from .xdr import XdrTrajectory
    kind, flags = loader_type(path)
    if not flags & FLAG_TRAJECTORY:
        raise ValueError(f"'{path}': a {kind} file holds no trajectory")
    if kind == "dcd":
        return DcdTrajectory(path, lib=lib)
    if kind in ("xtc", "trr"):
        return XdrTrajectory(path, lib=lib)
    # multi-MODEL PDB, XYZ / XMOL, LAMMPS dump: the native text reader (mapped file, frame index, parsed on the staging threads)
    from .texttraj import TextTrajectory
    return TextTrajectory(path, format={"pdb": "pdb", "xyz": "xyz"}.get(kind, "lammpstrj"), lib=lib)
