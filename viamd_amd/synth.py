"""Seeded synthetic systems of SURVEY.md 8d (BASELINE configs 2-5): an O,H,H water box, optionally with a
protein-like blob of bonded 10-atom residues in front.  Waters are generated on the device by the counter-based
generator (vmd_devtraj_synth, bit-identical to oracle S9); the blob's rigid tumbling is generated here on the host
(it needs sin/cos, which are not bit-reproducible across CPU and GPU) and uploaded into the leading atom rows."""
import numpy as np

from .script import Topology

# standard atomic weights of the elements that occur in biomolecular / materials trajectories (md_atom_mass analogue:
# masses weight the centre of mass of the SDF alignment, /root/reference/src/viamd.cpp:2253)
MASS = {"H": 1.008, "He": 4.0026, "Li": 6.94, "Be": 9.0122, "B": 10.81, "C": 12.011, "N": 14.007, "O": 15.999, "F": 18.998,
        "Ne": 20.180, "Na": 22.990, "Mg": 24.305, "Al": 26.982, "Si": 28.085, "P": 30.974, "S": 32.06, "Cl": 35.45,
        "Ar": 39.948, "K": 39.098, "Ca": 40.078, "Ti": 47.867, "Cr": 51.996, "Mn": 54.938, "Fe": 55.845, "Co": 58.933,
        "Ni": 58.693, "Cu": 63.546, "Zn": 65.38, "Se": 78.971, "Br": 79.904, "Rb": 85.468, "Sr": 87.62, "Mo": 95.95,
        "Ag": 107.87, "Cd": 112.41, "I": 126.90, "Cs": 132.91, "Ba": 137.33, "Pt": 195.08, "Au": 196.97, "Hg": 200.59,
        "Pb": 207.2}


def water_box_topology(n_atoms, n_blob=0, atoms_per_residue=10):
    """Elements / residue names / residue indices / masses of the synthetic system: [blob residues][O,H,H waters]."""
    elements = np.empty(n_atoms, dtype="<U2")
    resnames = np.empty(n_atoms, dtype="<U4")
    resid = np.empty(n_atoms, np.int64)
    blob_elems = np.array(["N", "C", "C", "O", "C", "H", "H", "H", "C", "H"])
    b = np.arange(n_blob)
    elements[:n_blob] = blob_elems[b % atoms_per_residue % blob_elems.size]
    resnames[:n_blob] = "ALA"
    resid[:n_blob] = b // atoms_per_residue
    n_blob_res = -(-n_blob // atoms_per_residue) if n_blob else 0
    w = np.arange(n_atoms - n_blob)
    elements[n_blob:] = np.where(w % 3 == 0, "O", "H")
    resnames[n_blob:] = "HOH"
    resid[n_blob:] = n_blob_res + w // 3
    mass = np.array([MASS[e] for e in ("C", "N", "O", "H")])
    lut = {"C": 0, "N": 1, "O": 2, "H": 3}
    m = mass[np.vectorize(lut.get)(elements)].astype(np.float32)
    return Topology(elements, resnames, resid, mass=m)


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def blob_trajectory(seed, n_blob, box, frames, atoms_per_residue=10, chunk=512):
    """Yields (frame_beg, xyz float32 [F, 3, n_blob]): a bonded chain of residues tumbling rigidly (random rotation
    walk 2 deg/frame, translation walk 0.2 A/frame) with internal jitter sigma 0.1 A, wrapped into the box."""
    rng = np.random.default_rng(seed)
    n_res = -(-n_blob // atoms_per_residue)
    steps = rng.normal(size=(n_res, 3))
    steps *= 3.8 / np.linalg.norm(steps, axis=1, keepdims=True)
    centers = np.cumsum(steps, axis=0)
    centers -= centers.mean(axis=0)
    template = (np.repeat(centers, atoms_per_residue, axis=0) + rng.normal(0, 1.2, (n_res * atoms_per_residue, 3)))[:n_blob]
    R = _rot(rng.normal(size=3), rng.uniform(0, np.pi))
    t = np.full(3, 0.5 * box)
    for f0 in range(0, frames, chunk):
        nf = min(chunk, frames - f0)
        out = np.empty((nf, 3, n_blob), np.float32)
        for f in range(nf):
            if f0 + f > 0:
                R = _rot(rng.normal(size=3), np.deg2rad(2.0)) @ R
                t = t + rng.normal(0, 0.2, 3)
            pts = template @ R.T + t + rng.normal(0, 0.1, (n_blob, 3))
            out[f] = np.mod(pts, box).T
        yield f0, out


def make_device_trajectory(V, seed, n_atoms, box, frames, n_blob=0, sigma=0.05, lib=None, shard=None):
    """DeviceTrajectory of the synthetic system, generated in place in HBM.  shard = (beg, end): only that block of frames
    (and frame 0) is resident - one rank's part of a frame-sharded trajectory, same content as the whole one."""
    traj = V.DeviceTrajectory(frames, n_atoms, lib=lib, shard=shard)
    traj.synth(seed, box, sigma, n_blob=n_blob)
    if n_blob:
        beg, end = (0, frames) if shard is None else shard
        for f0, xyz in blob_trajectory(seed, n_blob, box, end):          # a sequential random walk: generated from frame 0 on
            lo, hi = max(f0, beg), min(f0 + xyz.shape[0], end)
            if lo < hi:
                traj.upload_atoms(lo, 0, xyz[lo - f0:hi - f0])
            if f0 == 0 and beg > 0:
                traj.upload_atoms(0, 0, xyz[:1])
    return traj
