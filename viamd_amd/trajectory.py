"""Trajectory stand-ins for md_trajectory_i: frames come either from host memory through a
load_frame callback (the only call VIAMD makes, src/viamd.cpp:465-467) or are pre-staged in HBM."""
import ctypes as C

import numpy as np

from . import _lib as L
from .eval import VmdError, make_unitcell


class HostTrajectory:
    """Frames in host memory: coords float32 [F, 3, N]; cells: one Unitcell per frame (or one for all)."""

    def __init__(self, coords, cells):
        self.coords = np.ascontiguousarray(coords, dtype=np.float32)
        assert self.coords.ndim == 3 and self.coords.shape[1] == 3
        F = self.coords.shape[0]
        if isinstance(cells, L.Unitcell):
            cells = [cells] * F
        assert len(cells) == F
        self.cells = list(cells)
        self.loads = 0

        def num_frames(_):
            return self.coords.shape[0]

        def num_atoms(_):
            return self.coords.shape[2]

        def load_frame(_, idx, hdr, x, y, z):
            if idx < 0 or idx >= self.coords.shape[0]:
                return False
            n = self.coords.shape[2]
            f = self.coords[idx]
            for dst, row in ((x, 0), (y, 1), (z, 2)):
                if dst:
                    C.memmove(dst, f[row].ctypes.data, n * 4)
            if hdr:
                hdr.contents.num_atoms = n
                hdr.contents.index = idx
                hdr.contents.timestamp = float(idx)
                hdr.contents.unitcell = self.cells[idx]
            self.loads += 1
            return True

        self._cb = (L.NUM_FRAMES_FN(num_frames), L.NUM_ATOMS_FN(num_atoms), L.LOAD_FRAME_FN(load_frame))
        self.c = L.TrajectoryI(None, self._cb[0], self._cb[1], self._cb[2], L.DEVICE_VIEW_FN(), L.HOST_VIEW_FN())

    def interface(self):
        return C.byref(self.c)

    def num_frames(self):
        return self.coords.shape[0]

    def num_atoms(self):
        return self.coords.shape[2]


class PinnedHostTrajectory:
    """Frames in pinned host memory in the evaluator's own SoA layout (a frame cache): every batch crosses PCIe once,
    DMA'd directly, overlapped with the kernels of the previous batch."""

    def __init__(self, num_frames, num_atoms, lib=None):
        self.lib = lib or L.default_lib()
        self.h = self.lib.vmd_hosttraj_create(int(num_frames), int(num_atoms))
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._num_frames, self._num_atoms = int(num_frames), int(num_atoms)

    def close(self):
        if self.h:
            self.lib.vmd_hosttraj_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def interface(self):
        return self.lib.vmd_hosttraj_interface(self.h)

    def num_frames(self):
        return self._num_frames

    def num_atoms(self):
        return self._num_atoms

    def frame(self, f):
        """numpy view [3, num_atoms] of frame f (writable)"""
        rs = C.c_size_t(0)
        p = self.lib.vmd_hosttraj_frame_ptr(self.h, int(f), C.byref(rs))
        return np.ctypeslib.as_array(p, shape=(3, rs.value))[:, :self._num_atoms]

    def upload(self, coords, cells):
        coords = np.asarray(coords, np.float32)
        if isinstance(cells, L.Unitcell):
            cells = [cells] * coords.shape[0]
        for f in range(coords.shape[0]):
            self.frame(f)[:] = coords[f]
            self.lib.vmd_hosttraj_set_cell(self.h, f, C.byref(cells[f]))

    def copy_from_device(self, dev, frame_beg=0, frame_end=None):
        frame_end = self._num_frames if frame_end is None else frame_end
        if not self.lib.vmd_hosttraj_copy_from_device(self.h, dev.h, int(frame_beg), int(frame_end)):
            raise VmdError(self.lib.last_error())


class DeviceTrajectory:
    """Trajectory resident in HBM as float[F][3][npad] (SoA per frame, npad multiple of 64) — SURVEY.md 8d."""

    def __init__(self, num_frames, num_atoms, lib=None, shard=None):
        """shard = (frame_beg, frame_end): this rank keeps only that block of the `num_frames` frames resident (plus frame 0,
        where the SDF reference pose is taken); frame indices stay global."""
        self.lib = lib or L.default_lib()
        self.first, self.last = (0, int(num_frames)) if shard is None else (int(shard[0]), int(shard[1]))
        self.h = self.lib.vmd_devtraj_create_shard(int(num_frames), self.first, self.last, int(num_atoms))
        if not self.h:
            raise VmdError(self.lib.last_error())
        self._num_frames, self._num_atoms = int(num_frames), int(num_atoms)

    def close(self):
        if self.h:
            self.lib.vmd_devtraj_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def interface(self):
        return self.lib.vmd_devtraj_interface(self.h)

    def num_frames(self):
        return self._num_frames

    def num_atoms(self):
        return self._num_atoms

    def upload_frame(self, frame, cell, x, y, z):
        x, y, z = (np.ascontiguousarray(a, np.float32) for a in (x, y, z))
        ok = self.lib.vmd_devtraj_upload_frame(self.h, int(frame), C.byref(cell), x.ctypes.data_as(L.c_float_p),
                                               y.ctypes.data_as(L.c_float_p), z.ctypes.data_as(L.c_float_p))
        if not ok:
            raise VmdError(self.lib.last_error())

    def upload(self, coords, cells):
        coords = np.ascontiguousarray(coords, np.float32)
        if isinstance(cells, L.Unitcell):
            cells = [cells] * coords.shape[0]
        for f in range(coords.shape[0]):
            self.upload_frame(f, cells[f], coords[f, 0], coords[f, 1], coords[f, 2])

    def upload_atoms(self, frame_beg, first_atom, xyz):
        """Overwrite a block of atoms in consecutive frames: xyz float32 [F, 3, n]."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        if not self.lib.vmd_devtraj_upload_atoms(self.h, int(frame_beg), xyz.shape[0], int(first_atom), xyz.shape[2],
                                                 xyz.ctypes.data_as(L.c_float_p)):
            raise VmdError(self.lib.last_error())

    def synth(self, seed, L_box, sigma=0.05, n_blob=0, frame_beg=0, frame_end=None):
        """Fill frames with the seeded synthetic water box of SURVEY.md 8d (generated on the device)."""
        frame_beg = max(int(frame_beg), self.first)
        frame_end = self.last if frame_end is None else frame_end
        if not self.lib.vmd_devtraj_synth(self.h, int(seed), float(L_box), float(sigma), int(n_blob), int(frame_beg), int(frame_end)):
            raise VmdError(self.lib.last_error())

    def set_cell(self, cell, frame_beg=0, frame_end=None):
        """Replace the unit cell of frames [beg, end); coordinates are wrapped on use (e.g. a sheared cell of the same volume)."""
        frame_end = self._num_frames if frame_end is None else frame_end
        if not self.lib.vmd_devtraj_set_cell(self.h, int(frame_beg), int(frame_end), C.byref(cell)):
            raise VmdError(self.lib.last_error())

    def download_frame(self, frame):
        out = np.zeros((3, self._num_atoms), np.float32)
        hdr = L.FrameHeader()
        iface = self.interface().contents
        ok = iface.load_frame(iface.inst, int(frame), C.byref(hdr), out[0].ctypes.data_as(L.c_float_p),
                              out[1].ctypes.data_as(L.c_float_p), out[2].ctypes.data_as(L.c_float_p))
        if not ok:
            raise VmdError(self.lib.last_error())
        return out, hdr.unitcell

    def device_ptr(self):
        fs, rs = C.c_size_t(0), C.c_size_t(0)
        p = self.lib.vmd_devtraj_device_ptr(self.h, C.byref(fs), C.byref(rs))
        return p, fs.value, rs.value
