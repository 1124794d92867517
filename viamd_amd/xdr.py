"""GROMACS XTC / TRR trajectories: `XdrTrajectory` wraps the native reader of viamd_amd/csrc/vmd_xdr.cpp (VIAMD attaches these
files through md_xtc_attach_from_file / md_trr_attach_from_file, /root/reference/src/loader.cpp:147-150); `write_xtc` /
`write_trr` drive the native writer of the same file (tests, export of synthetic trajectories, `bench.py --traj xtc`)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .eval import VmdError

KIND_XTC, KIND_TRR = 0, 1


class XdrTrajectory:
    """An XTC or TRR file as md_trajectory_i stand-in: frames are decompressed straight into the evaluator's pinned
    staging buffer, several frames of a staged batch at a time (load_frame is re-entrant)."""

    def __init__(self, path, lib=None):
        self.lib = lib or L.default_lib()
        self.h = self.lib.vmd_xdrtraj_open(str(path).encode())
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._iface = self.lib.vmd_xdrtraj_interface(self.h)

    def interface(self):
        return self._iface

    @property
    def kind(self):
        return "xtc" if self.lib.vmd_xdrtraj_kind(self.h) == KIND_XTC else "trr"

    def num_frames(self):
        i = self._iface.contents
        return int(i.num_frames(i.inst))

    def num_atoms(self):
        i = self._iface.contents
        return int(i.num_atoms(i.inst))

    def frame_step(self, frame):
        return int(self.lib.vmd_xdrtraj_frame_step(self.h, int(frame)))

    def load_frame(self, frame, with_header=False):
        """-> (xyz float32 [3, N] in Angstrom, Unitcell[, FrameHeader])"""
        n = self.num_atoms()
        out = np.zeros((3, n), np.float32)
        hdr = L.FrameHeader()
        i = self._iface.contents
        if not i.load_frame(i.inst, int(frame), C.byref(hdr), out[0].ctypes.data_as(L.c_float_p),
                            out[1].ctypes.data_as(L.c_float_p), out[2].ctypes.data_as(L.c_float_p)):
            raise VmdError(self.lib.last_error())
        return (out, hdr.unitcell, hdr) if with_header else (out, hdr.unitcell)

    def save_checkpoints(self, path):
        """Write the decoder checkpoints the device decodes of this trajectory have left so far (vmd_ckcache_save)."""
        if not self.lib.vmd_ckcache_save(self._iface, str(path).encode()):
            raise VmdError(self.lib.last_error())

    def load_checkpoints(self, path, device=0):
        """Install decoder checkpoints written by an earlier process (vmd_ckcache_load) -> frames covered (0: not this trajectory)."""
        n = int(self.lib.vmd_ckcache_load(self._iface, str(path).encode(), int(device)))
        if n < 0:
            raise VmdError(self.lib.last_error())
        return n

    def close(self):
        if self.h:
            self.lib.vmd_xdrtraj_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CompressedDeviceTrajectory:
    """Every frame of an XTC trajectory, still compressed, resident in HBM (vmd_rawtraj_*): evaluations decompress their batches
    on the device straight from that copy - the file is read and crosses PCIe once, not once per evaluation."""

    def __init__(self, src):
        self.src = src                                   # kept alive: the native object borrows its interface
        self.lib = src.lib
        self.h = self.lib.vmd_rawtraj_create(src.interface())
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._iface = self.lib.vmd_rawtraj_interface(self.h)

    def interface(self):
        return self._iface

    def num_frames(self):
        return self.src.num_frames()

    def num_atoms(self):
        return self.src.num_atoms()

    def device_bytes(self):
        return int(self.lib.vmd_rawtraj_device_bytes(self.h))

    def close(self):
        if self.h:
            self.lib.vmd_rawtraj_free(self.h)
            self.h = None
        self.src.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _write(path, kind, coords, cells, precision, dt, lib):
    lib = lib or L.default_lib()
    if hasattr(coords, "frame"):
        F, N, get = coords.num_frames(), coords.num_atoms(), coords.frame
    else:
        coords = np.asarray(coords, np.float32)
        F, _, N = coords.shape
        get = lambda f: coords[f]
    if cells is not None and isinstance(cells, L.Unitcell):
        cells = [cells] * F
    w = lib.vmd_xdrwriter_open(str(path).encode(), kind, N, float(precision))
    if not w:
        raise VmdError(lib.last_error())
    try:
        for f in range(F):
            xyz = np.ascontiguousarray(get(f), np.float32)
            cell = C.byref(cells[f]) if cells is not None else None
            if not lib.vmd_xdrwriter_write_frame(w, f, float(f * dt), cell, xyz[0].ctypes.data_as(L.c_float_p),
                                                 xyz[1].ctypes.data_as(L.c_float_p), xyz[2].ctypes.data_as(L.c_float_p)):
                raise VmdError(lib.last_error())
    finally:
        if not lib.vmd_xdrwriter_close(w):
            raise VmdError(f"closing {path} failed")


def write_xtc(path, coords, cells=None, precision=1000.0, dt=1.0, lib=None):
    """coords float32 [F, 3, N] in Angstrom (or a trajectory object with num_frames() / num_atoms() / frame(f) -> [3, N]);
    cells: None, one Unitcell or one per frame; precision in 1/nm (1000 = GROMACS' default, 0.01 A resolution)."""
    _write(path, KIND_XTC, coords, cells, precision, dt, lib)


def write_trr(path, coords, cells=None, dt=1.0, lib=None):
    _write(path, KIND_TRR, coords, cells, 0.0, dt, lib)
