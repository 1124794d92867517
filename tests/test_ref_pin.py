"""a10 pinned to the reference ITSELF: compute_histogram, compute_histogram_masked, downsample_histogram and scale_histogram are
plain C++ inside /root/reference/src/main.cpp:139-261.  oracle/make_ref.py cuts those line ranges out of the reference where it
lies and compiles them (oracle/_ref/libviamd_ref.so, git-ignored, travels to the GPU box prebuilt); here the oracle's restatement
(vo_*) and the product's host functions (vmd_*) are run against that compiled reference code on random and adversarial inputs -
bit for bit, NaNs included.  This is the formula the 1e-5 tolerance of normalised g(r) is measured through."""
import ctypes as C

import numpy as np
import pytest

import viamd_amd as V
from viamd_amd import eval as E

from oracle import make_ref

fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def ref():
    path = make_ref.build()
    if path is None:
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/libviamd_ref.so is present")
    lib = C.CDLL(path)
    lib.ref_compute_histogram.argtypes = [fp, C.c_int, C.c_float, C.c_float, fp, C.c_int, fp, fp]
    lib.ref_compute_histogram_masked.argtypes = [fp, C.c_int, C.c_float, C.c_float, fp, C.c_int, u8p, C.c_int, C.c_int, fp]
    lib.ref_downsample_histogram.argtypes = [fp, C.c_int, fp, fp, C.c_int]
    lib.ref_scale_histogram.argtypes = [fp, fp, C.c_int]
    for f in (lib.ref_compute_histogram, lib.ref_compute_histogram_masked, lib.ref_downsample_histogram, lib.ref_scale_histogram):
        f.restype = None
    return lib


def _f(a):
    return a.ctypes.data_as(fp)


def same(a, b):
    """bit-identical float arrays (NaN == NaN, -0 != +0)"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _value_sets(rng):
    yield rng.normal(3.0, 1.5, 4000).astype(np.float32), 0.5, 6.0
    yield rng.uniform(-1, 1, 777).astype(np.float32), -1.0, 1.0                      # values on both range ends
    yield np.array([0.0, 1.0, 1.0, 0.25, 0.999999, 1.0000001, -1e-9], np.float32), 0.0, 1.0
    yield rng.uniform(10, 11, 50).astype(np.float32), 0.0, 1.0                       # nothing inside the range
    yield np.float32(7.5) * np.ones(30, np.float32), 7.5, 7.5                        # empty range: inv_range = inf (unmasked) / 0 (masked)
    yield (rng.integers(0, 50, 5000) / np.float32(7)).astype(np.float32), 0.0, 7.0   # many values exactly on bin edges


def test_compute_histogram_against_the_reference(ref, oracle, host_lib):
    rng = np.random.default_rng(5)
    for values, lo, hi in _value_sets(rng):
        for nb in (1, 7, 128, 1024):
            want = np.zeros(nb, np.float32)
            mn, mx = C.c_float(-1), C.c_float(-1)
            ref.ref_compute_histogram(_f(want), nb, lo, hi, _f(values), values.size, C.byref(mn), C.byref(mx))
            got_o, omn, omx = oracle.compute_histogram(values, nb, lo, hi)
            got_p, pmn, pmx = E.compute_histogram(values, nb, lo, hi, lib=host_lib)
            assert same(got_o, want) and same(got_p, want), (lo, hi, nb)
            assert same([omn, omx], [mn.value, mx.value]) and same([pmn, pmx], [mn.value, mx.value])


def test_compute_histogram_masked_against_the_reference(ref, oracle, host_lib):
    rng = np.random.default_rng(6)
    for values, lo, hi in _value_sets(rng):
        for dim in (1, 3):
            F = values.size // dim
            v = values[:F * dim]
            for mask in (np.ones(F, np.uint8), (rng.random(F) < 0.4).astype(np.uint8), np.zeros(F, np.uint8)):
                for agg in (False, True):
                    for nb in (5, 128):
                        hd = 1 if agg else dim
                        want = np.zeros(hd * nb, np.float32)
                        yr = np.full(2, -7.0, np.float32)
                        ref.ref_compute_histogram_masked(_f(want), nb, lo, hi, _f(v), dim, mask.ctypes.data_as(u8p), F, int(agg), _f(yr))
                        got_o = oracle.compute_histogram_masked(v, dim, mask, nb, lo, hi, agg)
                        got_p, pyr = E.compute_histogram_masked_y(v, dim, mask, nb, lo, hi, agg, lib=host_lib)
                        assert same(got_o, want), ("oracle", lo, hi, dim, agg, nb)
                        assert same(got_p, want), ("product", lo, hi, dim, agg, nb)
                        if mask.any():                     # without a frame the reference leaves y_min / y_max as they were
                            assert same(pyr, yr), (pyr, yr)
                        assert same(E.compute_histogram_masked(v, dim, mask, nb, lo, hi, agg, lib=host_lib), want)


def test_downsample_and_scale_against_the_reference(ref, oracle, host_lib):
    rng = np.random.default_rng(7)
    for n_src in (1024, 1000, 128, 7):
        src = rng.integers(0, 500, n_src).astype(np.float32)
        w = rng.uniform(0, 40, n_src).astype(np.float32)
        w[rng.random(n_src) < 0.1] = 0.0                      # empty source bins: 0 / 0 and x / 0 go through as the reference has them
        for n_dst in sorted({1, 2, 7, 128, n_src // 3 or 1, n_src}):
            if n_dst > n_src:
                continue
            for weights in (w, None):
                want = np.zeros(n_dst, np.float32)
                ref.ref_downsample_histogram(_f(want), n_dst, _f(src), _f(weights) if weights is not None else None, n_src)
                assert same(oracle.downsample_histogram(src, weights, n_dst), want)
                assert same(V.downsample_histogram(src, weights, n_dst, lib=host_lib), want)
        want = src.copy()
        ref.ref_scale_histogram(_f(want), _f(w), n_src)
        assert same(E.scale_histogram(src, w, lib=host_lib), want)


def test_g_of_r_of_an_evaluated_rdf_through_the_reference_downsample(ref, oracle, emu_lib):
    """End to end on the a10 formula: counts and weights of an evaluated rdf() (emulator build of the product) -> the REFERENCE's
    downsample_histogram == the product's, bit for bit, and g -> 1 for the ideal gas within the statistical error."""
    import cases
    coords = cases.water_box(oracle, 9, 3000, 40.0, 4)
    o = cases.oxygen(3000)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 12.0)
    ev = V.ScriptEval(4, ir)
    assert ev.frame_range(V.MolSystem(3000, unitcell=V.make_unitcell(40.0)), V.HostTrajectory(coords, V.make_unitcell(40.0)), 0, 4)
    pd = ev.property_data("g")
    vals, wts = np.array(pd.values, np.float32), np.array(pd.weights, np.float32)
    want = np.zeros(128, np.float32)
    ref.ref_downsample_histogram(_f(want), 128, _f(vals), _f(wts), vals.size)
    got = V.downsample_histogram(vals, wts, 128, lib=emu_lib)
    assert same(got, want)
    assert abs(float(np.mean(want[64:])) - 1.0) < 0.05
