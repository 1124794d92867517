"""DCD trajectories (CHARMM / NAMD): `DcdTrajectory` wraps the native reader of viamd_amd/csrc/vmd_dcd.cpp (VIAMD attaches
these files through md_dcd_attach_from_file, /root/reference/src/loader.cpp:151-152); `write_dcd` produces files in the same
layout (tests, export of synthetic trajectories)."""
import ctypes as C
import struct

import numpy as np

from . import _lib as L
from .eval import VmdError


class DcdTrajectory:
    """A DCD file as md_trajectory_i stand-in: frames are read (pread) straight into the evaluator's pinned staging buffer."""

    def __init__(self, path, lib=None):
        self.lib = lib or L.default_lib()
        self.h = self.lib.vmd_dcdtraj_open(str(path).encode())
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._iface = self.lib.vmd_dcdtraj_interface(self.h)

    def interface(self):
        return self._iface

    def num_frames(self):
        i = self._iface.contents
        return int(i.num_frames(i.inst))

    def num_atoms(self):
        i = self._iface.contents
        return int(i.num_atoms(i.inst))

    def load_frame(self, frame):
        """-> (xyz float32 [3, N], Unitcell)"""
        n = self.num_atoms()
        out = np.zeros((3, n), np.float32)
        hdr = L.FrameHeader()
        i = self._iface.contents
        if not i.load_frame(i.inst, int(frame), C.byref(hdr), out[0].ctypes.data_as(L.c_float_p),
                            out[1].ctypes.data_as(L.c_float_p), out[2].ctypes.data_as(L.c_float_p)):
            raise VmdError(self.lib.last_error())
        return out, hdr.unitcell

    def close(self):
        if self.h:
            self.lib.vmd_dcdtraj_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cell_to_dcd(cell, cosines=False):
    """Unitcell {x,y,z,xy,xz,yz} -> the six doubles of a DCD unit-cell record: A, gamma, B, beta, alpha, C."""
    a = np.array([cell.x, 0.0, 0.0])
    b = np.array([cell.xy, cell.y, 0.0])
    c = np.array([cell.xz, cell.yz, cell.z])
    A, B, Cc = np.linalg.norm(a), np.linalg.norm(b), np.linalg.norm(c)
    if A == 0 or B == 0 or Cc == 0:
        return [0.0] * 6
    cg, cb, ca = a @ b / (A * B), a @ c / (A * Cc), b @ c / (B * Cc)
    if cosines:
        ang = [cg, cb, ca]
    else:
        ang = [90.0 if v == 0.0 else float(np.degrees(np.arccos(v))) for v in (cg, cb, ca)]
    return [A, ang[0], B, ang[1], ang[2], Cc]


def write_dcd(path, coords, cells=None, big_endian=False, cosines=False, title="written by viamd_amd"):
    """coords float32 [F, 3, N] (or a trajectory object with num_frames() / num_atoms() / frame(f) -> [3, N]);
    cells: None, one Unitcell or one per frame."""
    if hasattr(coords, "frame"):
        F, N, get = coords.num_frames(), coords.num_atoms(), coords.frame
    else:
        coords = np.asarray(coords, np.float32)
        F, _, N = coords.shape
        get = lambda f: coords[f]
    e = ">" if big_endian else "<"
    if cells is not None and isinstance(cells, L.Unitcell):
        cells = [cells] * F

    def rec(payload):
        return struct.pack(e + "i", len(payload)) + payload + struct.pack(e + "i", len(payload))

    icntrl = [0] * 20
    icntrl[0] = F            # NSET
    icntrl[1] = 1            # ISTART
    icntrl[2] = 1            # NSAVC
    icntrl[10] = 1 if cells is not None else 0
    icntrl[19] = 24          # CHARMM version: marks the CHARMM record layout
    with open(path, "wb") as fh:
        fh.write(rec(b"CORD" + struct.pack(e + "20i", *icntrl)))
        t = title.encode()[:80].ljust(80)
        fh.write(rec(struct.pack(e + "i", 1) + t))
        fh.write(rec(struct.pack(e + "i", N)))
        for f in range(F):
            if cells is not None:
                fh.write(rec(struct.pack(e + "6d", *cell_to_dcd(cells[f], cosines))))
            xyz = np.asarray(get(f), np.float32)
            for a in range(3):
                fh.write(rec(xyz[a].astype(e + "f4").tobytes()))
