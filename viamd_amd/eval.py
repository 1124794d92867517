"""Host-side mirror of the md_script evaluation surface VIAMD uses, over the C ABI.

Names follow the reference: ScriptIR ~ md_script_ir_t, ScriptEval ~ md_script_eval_t (create / clear_data /
frame_range / interrupt / property_data / frame_mask), see /root/reference/src/main.cpp:951-1039, :1286, :1513.
Everything here is plumbing: the arithmetic runs in the HIP kernels behind libviamd_amd.so.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _idx(a):
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
    return a, a.ctypes.data_as(L.c_int32_p)


def make_unitcell(box, flags=L.PBC_ALL, tilt=(0.0, 0.0, 0.0)):
    """box: None (no cell), scalar (cubic) or (x,y,z); tilt = (xy, xz, yz) of a triclinic cell
    (basis a=(x,0,0), b=(xy,y,0), c=(xz,yz,z), as VIAMD reads md_unitcell_t, src/viamd.cpp:1837-1843)."""
    if box is None:
        return L.Unitcell(0, 0, 0, 0, 0, 0, 0)
    if np.isscalar(box):
        box = (box, box, box)
    return L.Unitcell(float(box[0]), float(box[1]), float(box[2]), float(tilt[0]), float(tilt[1]), float(tilt[2]), flags)


class VmdError(RuntimeError):
    pass


class MolSystem:
    """The slice of md_system_t the evaluator reads: atom count, masses, unit cell (src/main.cpp:642)."""

    def __init__(self, num_atoms, mass=None, unitcell=None, bonds=None):
        """bonds: int32 [nbonds, 2] atom index pairs (md_system_t::bond) or None: sdf() structures are then made whole along
        their index order instead of along the bond graph."""
        self.num_atoms = int(num_atoms)
        self.mass = np.ascontiguousarray(mass if mass is not None else np.ones(num_atoms), dtype=np.float32)
        assert self.mass.size == self.num_atoms
        self.c = L.System()
        self.c.atom_count = self.num_atoms
        self.c.mass = self.mass.ctypes.data_as(L.c_float_p)
        self.c.unitcell = unitcell if unitcell is not None else make_unitcell(None)
        self.bonds = None
        if bonds is not None and len(bonds):
            self.bonds = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 2)
            self.c.bonds = self.bonds.ctypes.data_as(C.POINTER(C.c_int32))
            self.c.bond_count = self.bonds.shape[0]


class ScriptIR:
    """Property descriptors (md_script_ir_t stand-in)."""

    def __init__(self, lib=None):
        self.lib = lib or L.default_lib()
        self.h = self.lib.vmd_ir_create()
        if not self.h:
            raise VmdError("vmd_ir_create failed")

    def close(self):
        if self.h:
            self.lib.vmd_ir_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, ok):
        if not ok:
            raise VmdError(self.lib.last_error())

    def add_rdf(self, name, ref, target, cutoff, rmin=0.0):
        """`name = rdf(ref, target, cutoff)` (src/main.cpp:528); cutoff may be (rmin, rmax)."""
        if not np.isscalar(cutoff):
            rmin, cutoff = cutoff
        r, rp = _idx(ref)
        t, tp = _idx(target)
        self._check(self.lib.vmd_ir_add_rdf(self.h, name.encode(), rp, r.size, tp, t.size, float(rmin), float(cutoff)))

    def add_sdf(self, name, structures, target, cutoff):
        """`name = sdf(structures, target, cutoff)`; structures: [K, m] atom indices."""
        s = np.ascontiguousarray(structures, dtype=np.int32)
        if s.ndim != 2:
            raise ValueError("structures must be a [K, m] index array")
        t, tp = _idx(target)
        self._check(self.lib.vmd_ir_add_sdf(self.h, name.encode(), s.ctypes.data_as(L.c_int32_p), s.shape[0], s.shape[1],
                                            tp, t.size, float(cutoff)))

    def add_distance(self, name, a, b, kind=L.DIST_COM):
        """`name = distance|distance_min|distance_max|distance_pair(a, b)` (src/main.cpp:2817-2858)."""
        a_, ap = _idx(a)
        b_, bp = _idx(b)
        self._check(self.lib.vmd_ir_add_distance(self.h, name.encode(), int(kind), ap, a_.size, bp, b_.size))

    def add_distance_population(self, name, a_sets, b_sets, kind=L.DIST_COM):
        """`name = distance*(a, b) in <contexts>`: one (a, b) pair of index sets per context -> dim[1] = number of contexts."""
        assert len(a_sets) == len(b_sets) and len(a_sets) > 0
        a = np.concatenate([np.asarray(x, np.int32).reshape(-1) for x in a_sets]).astype(np.int32)
        b = np.concatenate([np.asarray(x, np.int32).reshape(-1) for x in b_sets]).astype(np.int32)
        ao = np.concatenate([[0], np.cumsum([len(x) for x in a_sets])]).astype(np.int32)
        bo = np.concatenate([[0], np.cumsum([len(x) for x in b_sets])]).astype(np.int32)
        self._check(self.lib.vmd_ir_add_distance_population(self.h, name.encode(), int(kind), len(a_sets),
                                                            a.ctypes.data_as(L.c_int32_p), ao.ctypes.data_as(L.c_int32_p),
                                                            b.ctypes.data_as(L.c_int32_p), bo.ctypes.data_as(L.c_int32_p)))

    def valid(self):
        return bool(self.lib.vmd_ir_valid(self.h))

    def fingerprint(self):
        return int(self.lib.vmd_ir_fingerprint(self.h))

    def property_count(self):
        return int(self.lib.vmd_ir_property_count(self.h))

    def property_names(self):
        n = self.property_count()
        p = self.lib.vmd_ir_property_names(self.h)
        return [p[i].decode() for i in range(n)]

    def property_flags(self, name):
        return int(self.lib.vmd_ir_property_flags(self.h, name.encode()))


class _EvalArray(np.ndarray):
    """ndarray view into memory owned by a ScriptEval: keeps the eval alive for as long as the view (or a slice of it) is."""

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class PropertyDataView:
    """Zero-copy numpy views of one md_script_property_data_t.  The arrays stay valid for the lifetime of the
    eval (md_script_property_data_t addresses are stable, src/main.cpp:1286,1303) and hold a reference to it."""

    def __init__(self, c, ev=None, name=None):
        self.c = c  # ctypes PropertyData (owned by the eval)
        self._ev, self._name = ev, name

    @property
    def dim(self):
        return tuple(self.c.dim)

    @property
    def fingerprint(self):
        return int(self.c.fingerprint)

    def _arr(self, ptr, n, dtype):
        if not ptr or n == 0:
            return None
        a = np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).view(_EvalArray)
        a._owner = self._ev
        return a

    @property
    def values(self):
        return self._arr(self.c.values, self.c.num_values, np.float32)

    @property
    def weights(self):
        return self._arr(self.c.weights, self.c.dim[2], np.float32) if self.c.weights else None

    @property
    def counts(self):
        if not self.c.counts:
            return None
        if self._ev is not None and not self._ev.lib.vmd_eval_refresh_counts(self._ev.h, self._name.encode()):
            raise VmdError(self._ev.lib.last_error())
        n = self.c.dim[2] if self.c.weights else self.c.dim[1] * self.c.dim[2] * self.c.dim[3]
        return self._arr(self.c.counts, n, np.uint64)

    @property
    def weights64(self):
        return self._arr(self.c.weights64, self.c.dim[2], np.float64) if self.c.weights64 else None

    @property
    def unit_str(self):
        """(x unit, y unit) as VIAMD prints md_script_property_data_t::unit (src/main.cpp:1314-1315)"""
        return tuple((u or b"").decode("utf-8") for u in self.c.unit_str)

    @property
    def min_range(self):
        return tuple(self.c.min_range)

    @property
    def max_range(self):
        return tuple(self.c.max_range)

    @property
    def max_value(self):
        return float(self.c.max_value)

    @property
    def min_value(self):
        return float(self.c.min_value)

    @property
    def aggregate(self):
        if not self.c.aggregate:
            return None
        a = self.c.aggregate.contents
        n = a.num_values
        def own(arr):
            arr = arr.view(_EvalArray)
            arr._owner = self._ev
            return arr
        return {"mean": own(np.ctypeslib.as_array(a.population_mean, shape=(n,))),
                "var": own(np.ctypeslib.as_array(a.population_var, shape=(n,))),
                "ext": own(np.ctypeslib.as_array(a.population_ext, shape=(n * 2,))).reshape(n, 2)}


class ScriptEval:
    """md_script_eval_t stand-in: owns the accumulators, evaluates frame ranges on the GPU."""

    def __init__(self, num_frames, ir):
        self.lib = ir.lib
        self.ir = ir
        self.h = self.lib.vmd_eval_create(int(num_frames), ir.h)
        if not self.h:
            raise VmdError(self.lib.last_error())

    def close(self):
        if self.h:
            self.lib.vmd_eval_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear_data(self):
        self.lib.vmd_eval_clear_data(self.h)

    def interrupt(self):
        self.lib.vmd_eval_interrupt(self.h)

    def ir_fingerprint(self):
        return int(self.lib.vmd_eval_ir_fingerprint(self.h))

    def frame_range(self, sys, traj, frame_beg, frame_end):
        """md_script_eval_frame_range(eval, ir, &sys, traj, beg, end); False on interrupt; raises on error."""
        sysp = C.byref(sys.c) if sys is not None else None
        # deferred-settle mode: the helper thread evaluates from copies of these records after the call has returned: the
        # objects behind them - coordinate arrays, ctypes callbacks - stay referenced until the next call, wait_settled, clear_data or free
        self._last_inputs = (sys, traj)
        ok = self.lib.vmd_eval_frame_range(self.h, self.ir.h, sysp, traj.interface(), int(frame_beg), int(frame_end))
        if not ok:
            if self.lib.vmd_eval_frames_done(self.h) < self.num_frames() and self._interrupted():
                return False
            raise VmdError(self.lib.last_error())
        return True

    def _interrupted(self):
        # an interrupt leaves no error message behind
        return True if not self.lib.last_error() else False

    def property_data(self, name):
        p = self.lib.vmd_eval_property_data(self.h, name.encode())
        if not p:
            return None
        return PropertyDataView(p.contents, self, name)

    def set_block_frames(self, block_frames):
        """Keep one partial accumulator per block of `block_frames` frames in HBM (filtered evaluation)."""
        if not self.lib.vmd_eval_set_block_frames(self.h, int(block_frames)):
            raise VmdError(self.lib.last_error())

    def set_source(self, source):
        """Serve whole frame blocks of later frame_range calls from `source`'s block partials (None detaches).
        The VIAMD "Eval Filt" pattern: a second eval over a timeline sub-range (src/main.cpp:1014-1039)."""
        if not self.lib.vmd_eval_set_source(self.h, source.h if source is not None else None):
            raise VmdError(self.lib.last_error())
        self._source = source          # keep it alive

    def frame_stats(self):
        """(frames evaluated by kernels, frames served from block partials) since the last clear_data."""
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.lib.vmd_eval_frame_stats(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def readahead_stats(self):
        """What read-ahead did for this eval (vmd_eval_readahead_stats): a dict of the counters."""
        st = L.ReadAheadStats()
        self.lib.vmd_eval_readahead_stats(self.h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in L.ReadAheadStats._fields_}

    def cell_build_stats(self):
        """(bucket overflows of the two-level cell build, selections that left it for the single-level builds)"""
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.lib.vmd_eval_cell_build_stats(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def frames_device_decoded(self):
        """Frames whose coordinates were decompressed on the device (load_raw + k_xtc_decode) since the last clear_data."""
        return int(self.lib.vmd_eval_frames_device_decoded(self.h))

    def frames_section_decoded(self):
        """... and how many of those were decoded in sections from the trajectory's decoder checkpoints (a re-evaluation)."""
        return int(self.lib.vmd_eval_frames_section_decoded(self.h))

    def frames_mapped(self):
        """... and how many reached the device by DMA straight out of the mapped file (no host copy of the compressed bytes)."""
        return int(self.lib.vmd_eval_frames_mapped(self.h))

    def num_frames(self):
        return int(self.lib.vmd_eval_num_frames(self.h))

    def frames_done(self):
        return int(self.lib.vmd_eval_frames_done(self.h))

    def frame_mask(self):
        n = self.num_frames()
        m = np.ctypeslib.as_array(self.lib.vmd_eval_frame_mask(self.h), shape=(n,)).view(_EvalArray)
        m._owner = self
        return m

    def frame_mask_bits(self):
        """the mask as md_bitfield_t storage: uint64 words, bit f & 63 of word f // 64"""
        nw = int(self.lib.vmd_eval_frame_mask_bits(self.h, None, 0))
        w = np.zeros(nw, np.uint64)
        self.lib.vmd_eval_frame_mask_bits(self.h, w.ctypes.data_as(L.c_uint64_p), nw)
        return w

    def set_frame_mask(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        self.lib.vmd_eval_set_frame_mask(self.h, m.ctypes.data_as(L.c_uint8_p), m.size)

    def frame_range_pooled(self, sys, traj, frame_beg, frame_end, num_threads=16, grain=1):
        """VIAMD's call pattern (src/main.cpp:993-997): native pool threads pulling ranges of `grain` frames, all calling frame_range on this eval"""
        sysp = C.byref(sys.c) if sys is not None else None
        self._last_inputs = (sys, traj)
        ok = self.lib.vmd_eval_frame_range_pooled(self.h, self.ir.h, sysp, traj.interface(), int(frame_beg), int(frame_end), int(num_threads), int(grain))
        if not ok:
            if self.lib.vmd_eval_frames_done(self.h) < self.num_frames() and self._interrupted():
                return False
            raise VmdError(self.lib.last_error())
        return True

    def set_deferred_settle(self, mode=1):
        """this eval's own choice of the deferred-settle mode: 1 on, 0 off, -1 follow the process-wide option readahead_lone"""
        if not self.lib.vmd_eval_set_deferred_settle(self.h, int(mode)):
            raise VmdError(self.lib.last_error())

    def wait_settled(self):
        """deferred-settle mode (option readahead_lone): bring totals and views up to what the calls so far asked for, now"""
        if not self.lib.vmd_eval_wait_settled(self.h):
            raise VmdError(self.lib.last_error())

    def defer_volume_views(self, defer=True):
        """a rank of a multi-GPU evaluation: the float view of a volume is derived once, by finalize() / the merge, not after every range"""
        if not self.lib.vmd_eval_defer_volume_views(self.h, bool(defer)):
            raise VmdError(self.lib.last_error())

    def finalize(self):
        if not self.lib.vmd_eval_finalize(self.h):
            raise VmdError(self.lib.last_error())

    def accum_views(self):
        n = self.lib.vmd_eval_accum_views(self.h, None, 0)
        arr = (L.AccumView * n)()
        self.lib.vmd_eval_accum_views(self.h, arr, n)
        return list(arr)

    def sdf_matrices(self, name, sys, traj, frame):
        """vis.sdf.matrices / vis.sdf.extent of md_script_vis_eval_payload (density_volume.cpp:183-204)."""
        K = C.c_size_t(0)
        ext = C.c_float(0)
        sysp = C.byref(sys.c) if sys is not None else None
        if not self.lib.vmd_eval_sdf_matrices(self.h, name.encode(), sysp, traj.interface(), int(frame), None, C.byref(K), C.byref(ext)):
            raise VmdError(self.lib.last_error())
        mats = np.zeros((K.value, 16), np.float32)
        if not self.lib.vmd_eval_sdf_matrices(self.h, name.encode(), sysp, traj.interface(), int(frame),
                                              mats.ctypes.data_as(L.c_float_p), C.byref(K), C.byref(ext)):
            raise VmdError(self.lib.last_error())
        # column-major mat4 -> numpy [K,4,4] with M[k] @ [x,y,z,1]
        return mats.reshape(-1, 4, 4).transpose(0, 2, 1).copy(), float(ext.value)


def _sdf_payload(self, name, sys, traj, frame):
    """md_script_vis_eval_payload(ATOMS | SDF) as VIAMD consumes it: (matrices [K,4,4] with M[k] @ [x,y,z,1], structures [K,m]
    atom indices, extent) - density_volume.cpp:183-204, 263-269."""
    pl = L.SdfPayload()
    sysp = C.byref(sys.c) if sys is not None else None
    if not self.lib.vmd_eval_sdf_payload(self.h, name.encode(), sysp, traj.interface(), int(frame), C.byref(pl)):
        raise VmdError(self.lib.last_error())
    K, m = pl.num_structures, pl.atoms_per_structure
    mats = np.ctypeslib.as_array(pl.matrices, shape=(K, 16)).copy().reshape(K, 4, 4).transpose(0, 2, 1).copy()
    structures = np.ctypeslib.as_array(pl.structures, shape=(K, m)).copy()
    return mats, structures, float(pl.extent)


def _export_cube(self, path, name, sys, traj, frame=0, atomic_numbers=None):
    """export_cube (src/main.cpp:5718-5830) in C++ behind the ABI: the volume + the atoms of reference structure 0."""
    an = None if atomic_numbers is None else np.ascontiguousarray(atomic_numbers, np.uint8)
    sysp = C.byref(sys.c) if sys is not None else None
    if not self.lib.vmd_export_cube(str(path).encode(), self.h, name.encode(), sysp, traj.interface(), int(frame),
                                    an.ctypes.data_as(L.c_uint8_p) if an is not None else None):
        raise VmdError(self.lib.last_error())


def _export_table(self, path, name, fmt="xvg", frame_times=None, num_bins=0, time_unit=None):
    """the XVG / CSV table VIAMD's export window writes for a temporal or distribution property (src/main.cpp:5953-6040)"""
    ft = None if frame_times is None else np.ascontiguousarray(frame_times, np.float64)
    if not self.lib.vmd_export_property_table(str(path).encode(), self.h, name.encode(), fmt.encode(),
                                              ft.ctypes.data_as(L.c_double_p) if ft is not None else None,
                                              time_unit.encode() if time_unit else None, int(num_bins)):
        raise VmdError(self.lib.last_error())


ScriptEval.sdf_payload = _sdf_payload
ScriptEval.export_cube = _export_cube
ScriptEval.export_table = _export_table


def downsample_histogram(values, weights, num_bins, lib=None):
    """What VIAMD plots for a distribution property: g = sum(values)/sum(weights) per display bin (src/main.cpp:232-250)."""
    lib = lib or L.default_lib()
    v = np.ascontiguousarray(values, np.float32)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    out = np.zeros(num_bins, np.float32)
    lib.vmd_downsample_histogram(out.ctypes.data_as(L.c_float_p), num_bins, v.ctypes.data_as(L.c_float_p),
                                 w.ctypes.data_as(L.c_float_p) if w is not None else None, v.size)
    return out


def compute_histogram_masked(values, dim, mask, num_bins, rmin, rmax, aggregate=False, lib=None):
    lib = lib or L.default_lib()
    v = np.ascontiguousarray(values, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros((1 if aggregate else dim) * num_bins, np.float32)
    lib.vmd_compute_histogram_masked(out.ctypes.data_as(L.c_float_p), num_bins, float(rmin), float(rmax),
                                     v.ctypes.data_as(L.c_float_p), dim, m.ctypes.data_as(L.c_uint8_p), m.size, bool(aggregate))
    return out


def compute_histogram_masked_y(values, dim, mask, num_bins, rmin, rmax, aggregate=False, lib=None):
    """-> (bins, (y_min, y_max)): compute_histogram_masked plus the value range VIAMD stores with the histogram (src/main.cpp:212-229)."""
    lib = lib or L.default_lib()
    v = np.ascontiguousarray(values, np.float32)
    m = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros((1 if aggregate else dim) * num_bins, np.float32)
    yr = np.zeros(2, np.float32)
    lib.vmd_compute_histogram_masked_y(out.ctypes.data_as(L.c_float_p), num_bins, float(rmin), float(rmax), v.ctypes.data_as(L.c_float_p),
                                       dim, m.ctypes.data_as(L.c_uint8_p), m.size, bool(aggregate), yr.ctypes.data_as(L.c_float_p))
    return out, (float(yr[0]), float(yr[1]))


def compute_histogram(values, num_bins, rmin, rmax, lib=None):
    """-> (bins, min, max): VIAMD's unmasked compute_histogram (src/main.cpp:139-170)."""
    lib = lib or L.default_lib()
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(num_bins, np.float32)
    lo, hi = C.c_float(0), C.c_float(0)
    lib.vmd_compute_histogram(out.ctypes.data_as(L.c_float_p), num_bins, float(rmin), float(rmax), v.ctypes.data_as(L.c_float_p), v.size,
                              C.cast(C.byref(lo), L.c_float_p), C.cast(C.byref(hi), L.c_float_p))
    return out, lo.value, hi.value


def scale_histogram(bins, weights, lib=None):
    """bins / weights where the weight is not zero (src/main.cpp:252-261); returns a new array."""
    lib = lib or L.default_lib()
    b = np.array(bins, np.float32)
    w = np.ascontiguousarray(weights, np.float32)
    lib.vmd_scale_histogram(b.ctypes.data_as(L.c_float_p), w.ctypes.data_as(L.c_float_p), b.size)
    return b
