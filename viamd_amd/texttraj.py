"""Text trajectories - multi-MODEL PDB, XYZ / XMOL, LAMMPS dump - through the native reader of viamd_amd/csrc/vmd_text.cpp (VIAMD attaches
them through mdlib, /root/reference/src/loader.cpp:22-77, 111-159).  The file is mapped and indexed once; the evaluator pulls frames through
load_frame on its staging threads.  `viamd_amd.pdb.read_pdb`, `textio.read_xyz` and `textio.read_lammps_dump` are independent Python
implementations of the same formats: the tests hold the two against each other bit for bit."""
import ctypes as C

import numpy as np

from . import _lib as L
from .eval import VmdError


class TextTrajectory:
    """A PDB / XYZ / LAMMPS-dump file as md_trajectory_i stand-in (format from the extension, or `format="pdb" | "xyz" | "lammpstrj"`)."""

    def __init__(self, path, format=None, lib=None):
        self.lib = lib or L.default_lib()
        self.h = self.lib.vmd_texttraj_open(str(path).encode(), format.encode() if format else None)
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._iface = self.lib.vmd_texttraj_interface(self.h)

    def interface(self):
        return self._iface

    def num_frames(self):
        i = self._iface.contents
        return int(i.num_frames(i.inst))

    def num_atoms(self):
        i = self._iface.contents
        return int(i.num_atoms(i.inst))

    def load_frame(self, frame):
        """-> (xyz float32 [3, N], Unitcell, timestamp)"""
        n = self.num_atoms()
        out = np.zeros((3, n), np.float32)
        hdr = L.FrameHeader()
        i = self._iface.contents
        if not i.load_frame(i.inst, int(frame), C.byref(hdr), out[0].ctypes.data_as(L.c_float_p),
                            out[1].ctypes.data_as(L.c_float_p), out[2].ctypes.data_as(L.c_float_p)):
            raise VmdError(self.lib.last_error())
        return out, hdr.unitcell, float(hdr.timestamp)

    def close(self):
        if self.h:
            self.lib.vmd_texttraj_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
