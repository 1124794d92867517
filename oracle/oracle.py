"""ctypes binding of oracle/libvmd_oracle.so — TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see SPEC.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VMD_ORACLE_LIB: another build of the same source (scripts/sanitize_emu.sh points it at an ASan/UBSan build)
_LIB_PATH = os.environ.get("VMD_ORACLE_LIB") or os.path.join(_HERE, "libvmd_oracle.so")

_FAST_PATH = os.path.join(_HERE, "libvmd_cpu_fast.so")

PBC_ALL = 7


class Cell(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float),
                ("xy", C.c_float), ("xz", C.c_float), ("yz", C.c_float),
                ("flags", C.c_uint32)]


class Synth(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_atoms", C.c_uint32), ("n_blob", C.c_uint32),
                ("L", C.c_float), ("sigma", C.c_float)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("vmd_oracle.c", "vmd_oracle.h", "Makefile", "vmd_cpu_fast.c")]
    if (not force and os.path.exists(_LIB_PATH) and os.path.exists(_FAST_PATH)
            and all(min(os.path.getmtime(_LIB_PATH), os.path.getmtime(_FAST_PATH)) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        fp, ip, u64p, dp, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_double), C.POINTER(C.c_uint8))
        cp = C.POINTER(Cell)
        L = _lib
        L.vo_wrap.restype = C.c_float
        L.vo_wrap.argtypes = [C.c_float, C.c_float]
        for name in ("vo_rdf_frame_brute", "vo_rdf_frame_cells"):
            f = getattr(L, name)
            f.restype = C.c_uint64
            f.argtypes = [fp, fp, fp, cp, ip, C.c_size_t, ip, C.c_size_t, C.c_float, C.c_float, C.c_int, u64p]
        L.vo_rdf_weights_frame.restype = None
        L.vo_rdf_weights_frame.argtypes = [cp, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, dp]
        L.vo_rdf_run.restype = C.c_uint64
        L.vo_rdf_run.argtypes = [fp, cp, C.c_size_t, C.c_size_t, ip, C.c_size_t, ip, C.c_size_t,
                                 C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, u64p, dp]
        L.vo_sdf_ref_pose.restype = None
        L.vo_sdf_ref_pose.argtypes = [fp, fp, fp, cp, ip, fp, C.c_size_t, dp]
        L.vo_sdf_frame_align.restype = None
        L.vo_sdf_frame_align.argtypes = [fp, fp, fp, cp, ip, fp, C.c_size_t, C.c_size_t, dp, dp, fp, fp]
        L.vo_jacobi4.restype = None
        L.vo_jacobi4.argtypes = [dp, dp]
        L.vo_sdf_frame_scatter.restype = C.c_uint64
        L.vo_sdf_frame_scatter.argtypes = [fp, fp, fp, cp, ip, C.c_size_t, C.c_size_t, fp, fp, ip, C.c_size_t,
                                           C.c_float, C.c_int, u64p]
        L.vo_sdf_run.restype = C.c_uint64
        L.vo_sdf_run.argtypes = [fp, cp, C.c_size_t, C.c_size_t, ip, fp, C.c_size_t, C.c_size_t, ip, C.c_size_t,
                                 C.c_float, C.c_int, C.c_int, u64p]
        L.vo_set_com.restype = None
        L.vo_set_com.argtypes = [fp, fp, fp, cp, ip, fp, C.c_size_t, fp]
        L.vo_distance_com.restype = C.c_float
        L.vo_distance_com.argtypes = [fp, fp, fp, cp, ip, fp, C.c_size_t, ip, fp, C.c_size_t]
        for name in ("vo_distance_min", "vo_distance_max"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [fp, fp, fp, cp, ip, C.c_size_t, ip, C.c_size_t]
        L.vo_distance_pair.restype = None
        L.vo_distance_pair.argtypes = [fp, fp, fp, cp, ip, C.c_size_t, ip, C.c_size_t, fp]
        L.vo_downsample_histogram.restype = None
        L.vo_downsample_histogram.argtypes = [fp, C.c_int, fp, fp, C.c_int]
        L.vo_compute_histogram.restype = None
        L.vo_compute_histogram.argtypes = [fp, C.c_int, C.c_float, C.c_float, fp, C.c_int, fp, fp]
        L.vo_compute_histogram_masked.restype = None
        L.vo_compute_histogram_masked.argtypes = [fp, C.c_int, C.c_float, C.c_float, fp, C.c_int, u8p, C.c_int, C.c_int]
        L.vo_synth_uniform.restype = C.c_float
        L.vo_synth_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.vo_synth_frame.restype = None
        L.vo_synth_frame.argtypes = [C.POINTER(Synth), C.c_uint32, fp, fp, fp]
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def make_cell(L, flags=PBC_ALL, tilt=(0.0, 0.0, 0.0)):
    if L is None:
        return Cell(0, 0, 0, 0, 0, 0, 0)
    if np.isscalar(L):
        L = (L, L, L)
    return Cell(float(L[0]), float(L[1]), float(L[2]), float(tilt[0]), float(tilt[1]), float(tilt[2]), flags)


def _as_idx(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _xyz(x, y, z):
    return (np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32), np.ascontiguousarray(z, np.float32))


def rdf_frame(x, y, z, cell, ref, tgt, rmin, rmax, nbins=1024, counts=None, method="brute"):
    x, y, z = _xyz(x, y, z)
    ref, tgt = _as_idx(ref), _as_idx(tgt)
    if counts is None:
        counts = np.zeros(nbins, np.uint64)
    fn = lib().vo_rdf_frame_brute if method == "brute" else lib().vo_rdf_frame_cells
    hits = fn(_f(x), _f(y), _f(z), C.byref(cell), _i(ref), ref.size, _i(tgt), tgt.size, rmin, rmax, nbins, _u64(counts))
    if hits == 2 ** 64 - 1:
        if method == "cells":
            return rdf_frame(x, y, z, cell, ref, tgt, rmin, rmax, nbins, counts, "brute")
        raise ValueError("oracle rejected the configuration (triclinic cell)")
    return counts, hits


def rdf_weights(cell, nref, ntgt, rmin, rmax, nbins=1024, weights=None):
    if weights is None:
        weights = np.zeros(nbins, np.float64)
    lib().vo_rdf_weights_frame(C.byref(cell), nref, ntgt, rmin, rmax, nbins, _d(weights))
    return weights


def rdf_run(traj, cells, ref, tgt, rmin, rmax, nbins=1024, nthreads=1, use_cells=True):
    """traj: float32 [F,3,npad]; cells: list of Cell (len F).  Returns counts u64, weights f64, hits."""
    traj = np.ascontiguousarray(traj, np.float32)
    F, _, npad = traj.shape
    carr = (Cell * F)(*cells)
    ref, tgt = _as_idx(ref), _as_idx(tgt)
    counts = np.zeros(nbins, np.uint64)
    weights = np.zeros(nbins, np.float64)
    hits = lib().vo_rdf_run(_f(traj), carr, F, npad, _i(ref), ref.size, _i(tgt), tgt.size, rmin, rmax, nbins,
                            nthreads, 1 if use_cells else 0, _u64(counts), _d(weights))
    return counts, weights, hits


def sdf_ref_pose(x, y, z, cell, idx, mass):
    x, y, z = _xyz(x, y, z)
    idx = _as_idx(idx)
    mass = np.ascontiguousarray(mass, np.float32)
    out = np.zeros((idx.size, 3), np.float64)
    lib().vo_sdf_ref_pose(_f(x), _f(y), _f(z), C.byref(cell), _i(idx), _f(mass), idx.size, _d(out))
    return out


def sdf_frame_align(x, y, z, cell, struct_idx, struct_mass, ref_pose):
    """struct_idx [K,m]; returns M [K,3,4] f64, R32 [K,3,3] f32, c32 [K,3] f32"""
    x, y, z = _xyz(x, y, z)
    struct_idx = _as_idx(struct_idx)
    K, m = struct_idx.shape
    struct_mass = np.ascontiguousarray(struct_mass, np.float32)
    ref_pose = np.ascontiguousarray(ref_pose, np.float64)
    M = np.zeros((K, 3, 4), np.float64)
    R32 = np.zeros((K, 3, 3), np.float32)
    c32 = np.zeros((K, 3), np.float32)
    lib().vo_sdf_frame_align(_f(x), _f(y), _f(z), C.byref(cell), _i(struct_idx), _f(struct_mass), K, m,
                             _d(ref_pose), _d(M), _f(R32), _f(c32))
    return M, R32, c32


def sdf_frame_scatter(x, y, z, cell, struct_idx, R32, c32, tgt, s, dim=128, vol=None):
    x, y, z = _xyz(x, y, z)
    struct_idx = _as_idx(struct_idx)
    K, m = struct_idx.shape
    tgt = _as_idx(tgt)
    if vol is None:
        vol = np.zeros(dim * dim * dim, np.uint64)
    R32 = np.ascontiguousarray(R32, np.float32)
    c32 = np.ascontiguousarray(c32, np.float32)
    hits = lib().vo_sdf_frame_scatter(_f(x), _f(y), _f(z), C.byref(cell), _i(struct_idx), K, m, _f(R32), _f(c32),
                                      _i(tgt), tgt.size, s, dim, _u64(vol))
    return vol, hits


def sdf_run(traj, cells, struct_idx, struct_mass, tgt, s, dim=128, nthreads=1):
    """traj float32 [F,3,npad]; returns vol u64[dim^3], hits"""
    traj = np.ascontiguousarray(traj, np.float32)
    F, _, npad = traj.shape
    carr = (Cell * F)(*cells)
    struct_idx = _as_idx(struct_idx)
    K, m = struct_idx.shape
    struct_mass = np.ascontiguousarray(struct_mass, np.float32)
    tgt = _as_idx(tgt)
    vol = np.zeros(dim ** 3, np.uint64)
    hits = lib().vo_sdf_run(_f(traj), carr, F, npad, _i(struct_idx), _f(struct_mass), K, m, _i(tgt), tgt.size, s, dim, nthreads,
                            _u64(vol))
    return vol, hits


def jacobi4(A):
    A = np.array(A, np.float64, order="C").copy()
    V = np.zeros((4, 4), np.float64)
    lib().vo_jacobi4(_d(A), _d(V))
    return np.diag(A).copy(), V


def set_com(x, y, z, cell, idx, mass):
    x, y, z = _xyz(x, y, z)
    idx = _as_idx(idx)
    mass = None if mass is None else np.ascontiguousarray(mass, np.float32)
    out = np.zeros(3, np.float32)
    lib().vo_set_com(_f(x), _f(y), _f(z), C.byref(cell), _i(idx), _f(mass), idx.size, _f(out))
    return out


def distance_com(x, y, z, cell, a, ma, b, mb):
    x, y, z = _xyz(x, y, z)
    a, b = _as_idx(a), _as_idx(b)
    ma = None if ma is None else np.ascontiguousarray(ma, np.float32)
    mb = None if mb is None else np.ascontiguousarray(mb, np.float32)
    return lib().vo_distance_com(_f(x), _f(y), _f(z), C.byref(cell), _i(a), _f(ma), a.size, _i(b), _f(mb), b.size)


def distance_minmax(x, y, z, cell, a, b, which):
    x, y, z = _xyz(x, y, z)
    a, b = _as_idx(a), _as_idx(b)
    fn = lib().vo_distance_min if which == "min" else lib().vo_distance_max
    return fn(_f(x), _f(y), _f(z), C.byref(cell), _i(a), a.size, _i(b), b.size)


def distance_pair(x, y, z, cell, a, b):
    x, y, z = _xyz(x, y, z)
    a, b = _as_idx(a), _as_idx(b)
    out = np.zeros(a.size * b.size, np.float32)
    lib().vo_distance_pair(_f(x), _f(y), _f(z), C.byref(cell), _i(a), a.size, _i(b), b.size, _f(out))
    return out


_fast = None


def rdf_run_fast(traj, cells, ref_idx, tgt_idx, rmin, rmax, nbins=1024, nthreads=1, simd=True):
    """bench.py's tuned CPU baseline (oracle/vmd_cpu_fast.c: half shell, AVX-512 where the CPU has it): -> (counts u64, hits) with
    the integers of rdf_run, or None when the cell kind is not supported (orthorhombic + fully periodic only).  NOT the checker."""
    global _fast
    if _fast is None:
        build()
        _fast = C.CDLL(_FAST_PATH)
        _fast.vf_rdf_run.restype = C.c_uint64
        _fast.vf_rdf_run.argtypes = [C.POINTER(C.c_float), C.POINTER(Cell), C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.c_size_t,
                                     C.POINTER(C.c_int32), C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        _fast.vf_have_avx512.restype = C.c_int
    traj = np.ascontiguousarray(traj, np.float32)
    F, _, npad = traj.shape
    ref_idx = np.ascontiguousarray(ref_idx, np.int32)
    tgt_idx = np.ascontiguousarray(tgt_idx, np.int32)
    same = ref_idx.shape == tgt_idx.shape and np.array_equal(ref_idx, tgt_idx)
    carr = (Cell * F)(*cells)
    counts = np.zeros(nbins, np.uint64)
    ip = C.POINTER(C.c_int32)
    h = _fast.vf_rdf_run(traj.ctypes.data_as(C.POINTER(C.c_float)), carr, F, npad, ref_idx.ctypes.data_as(ip), ref_idx.size,
                         tgt_idx.ctypes.data_as(ip), tgt_idx.size, 1 if same else 0, rmin, rmax, nbins, nthreads, 1 if simd else 0,
                         counts.ctypes.data_as(C.POINTER(C.c_uint64)))
    if h == 2 ** 64 - 1:
        return None
    return counts, int(h)


def have_avx512():
    rdf_run_fast(np.zeros((0, 3, 1), np.float32), [], np.zeros(1, np.int32), np.zeros(1, np.int32), 0.0, 1.0)
    return bool(_fast.vf_have_avx512())


class unwrap_tree:
    """with unwrap_tree(bonds, structures): ...   installs the bond trees of the K structures (vo_bond_tree per structure) for the
    sdf_* calls inside the block: structures are made whole along their bonds (D-SDF-UNWRAP) instead of along their index order."""

    def __init__(self, bonds, structures):
        L = lib()
        L.vo_bond_tree.restype = None
        L.vo_bond_tree.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.vo_set_unwrap_tree.restype = None
        L.vo_set_unwrap_tree.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, C.c_size_t]
        bonds = np.ascontiguousarray(bonds, np.int32).reshape(-1, 2)
        structures = np.ascontiguousarray(structures, np.int32)
        K, m = structures.shape
        self.order = np.zeros((K, m), np.int32)
        self.parent = np.zeros((K, m), np.int32)
        for k in range(K):
            L.vo_bond_tree(_i(bonds), bonds.shape[0], _i(structures[k]), m, _i(self.order[k]), _i(self.parent[k]))
        self.K, self.m = K, m

    def __enter__(self):
        lib().vo_set_unwrap_tree(_i(self.order), _i(self.parent), self.K, self.m)
        return self

    def __exit__(self, *a):
        lib().vo_set_unwrap_tree(None, None, 0, 0)


def set_spec(key, value):
    """DECISION switch of SPEC.md ("rdf_closed", "sdf_include_self", "rdf_raw", "rdf_norm" = 0 / 1 / 2); returns the previous value."""
    L = lib()
    L.vo_set_spec.restype = C.c_int
    L.vo_set_spec.argtypes = [C.c_char_p, C.c_int]
    old = L.vo_set_spec(key.encode(), int(value))
    if old < 0:
        raise KeyError(key)
    return old


def downsample_histogram(values, weights, num_dst_bins):
    values = np.ascontiguousarray(values, np.float32)
    weights = None if weights is None else np.ascontiguousarray(weights, np.float32)
    dst = np.zeros(num_dst_bins, np.float32)
    lib().vo_downsample_histogram(_f(dst), num_dst_bins, _f(values), _f(weights), values.size)
    return dst


def compute_histogram(values, num_bins, rmin, rmax):
    values = np.ascontiguousarray(values, np.float32)
    bins = np.zeros(num_bins, np.float32)
    lo, hi = C.c_float(0), C.c_float(0)
    lib().vo_compute_histogram(_f(bins), num_bins, rmin, rmax, _f(values), values.size, C.byref(lo), C.byref(hi))
    return bins, lo.value, hi.value


def compute_histogram_masked(values, dim, mask, num_bins, rmin, rmax, aggregate=False):
    values = np.ascontiguousarray(values, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    hdim = 1 if aggregate else dim
    bins = np.zeros(hdim * num_bins, np.float32)
    lib().vo_compute_histogram_masked(_f(bins), num_bins, rmin, rmax, _f(values), dim,
                                      mask.ctypes.data_as(C.POINTER(C.c_uint8)), mask.size, 1 if aggregate else 0)
    return bins


def synth_frame(seed, n_atoms, L, sigma, frame, n_blob=0, npad=None):
    npad = npad or n_atoms
    out = np.zeros((3, npad), np.float32)
    cfg = Synth(seed, n_atoms, n_blob, L, sigma)
    lib().vo_synth_frame(C.byref(cfg), frame, _f(out[0]), _f(out[1]), _f(out[2]))
    return out
