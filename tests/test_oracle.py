"""The oracle against mathematics (known-answer fixtures in tests/golden, analytic expectations) and against itself
(brute force vs cell list).  PARITY UNPINNED: the reference has no tests or golden vectors for this path
(SURVEY.md F4/8c), so these are what pins the oracle."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        return json.load(f)


def test_sc_lattice_shell_counts(oracle, golden):
    g = golden["sc_lattice"]
    n = g["n"]
    pts = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij")).reshape(3, -1).astype(np.float32) * g["a"]
    idx = np.arange(pts.shape[1])
    rmax, nbins = 3.2, 320         # bin width 0.01 with an exactly representable inverse (1/3.2 = 0.3125)
    for method in ("brute", "cells"):
        counts, hits = oracle.rdf_frame(pts[0], pts[1], pts[2], oracle.make_cell(float(n)), idx, idx, 0.0, rmax, nbins=nbins,
                                        method=method)
        expect = np.zeros(nbins, np.uint64)
        for d2, mult in g["shells"].items():
            expect[int(np.sqrt(np.float32(int(d2))) * np.float32(100.0))] += mult * pts.shape[1]
        np.testing.assert_array_equal(counts, expect)
        assert hits == expect.sum()


def test_two_atoms_across_the_boundary(oracle, golden):
    for c in golden["two_atom"]:
        xyz = np.array([c["a"], c["b"]], np.float32).T.copy()
        counts, hits = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], oracle.make_cell(float(c["L"])), [0], [1], 0.0, 10.0, nbins=1000)
        assert hits == 1
        d = np.sqrt(np.float32(c["d2"]))
        assert counts[int(d * np.float32(0.1) * np.float32(1000.0))] == 1


def test_self_pairs_and_open_interval(oracle):
    # same set: the i == j pair (d = 0) drops out because the interval is open at r_min = 0 (SPEC D-RDF-OPEN)
    xyz = np.array([[1.0, 2.0, 3.0], [1.0, 2.0, 9.0]], np.float32).T.copy()
    counts, hits = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], oracle.make_cell(30.0), [0, 1], [0, 1], 0.0, 10.0)
    assert hits == 2
    counts, hits = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], oracle.make_cell(30.0), [0, 1], [0, 1], 0.0, 6.0)
    assert hits == 0      # d == r_max exactly is outside


def test_ideal_gas_g_of_r_is_one(oracle):
    rng = np.random.default_rng(1)
    N, Lb = 20000, 60.0
    xyz = rng.uniform(0, Lb, (3, N)).astype(np.float32)
    a, b = np.arange(0, N, 2), np.arange(1, N, 2)
    cell = oracle.make_cell(Lb)
    counts, _ = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], cell, a, b, 0.0, 12.0, method="cells")
    w = oracle.rdf_weights(cell, a.size, b.size, 0.0, 12.0)
    g = oracle.downsample_histogram(counts.astype(np.float32), w.astype(np.float32), 16)
    assert np.all(np.abs(g[2:] - 1.0) < 0.05)            # expected counts per display bin >= 1e3
    assert abs(counts.sum() / w.sum() - 1.0) < 0.01


@pytest.mark.parametrize("box,flags", [(40.0, 7), ((50.0, 37.0, 41.0), 7), (None, 0), (30.0, 3)])
def test_cells_equal_brute(oracle, box, flags):
    rng = np.random.default_rng(2)
    N = 2500
    ext = 40.0 if box is None else np.array(box if not np.isscalar(box) else (box,) * 3)[:, None]
    xyz = (rng.uniform(-0.5, 1.5, (3, N)) * ext).astype(np.float32)
    cell = oracle.make_cell(box, flags)
    a, b = np.arange(0, N, 3), np.arange(N)
    for rmin, rmax in ((0.0, 11.0), (2.0, 7.5)):
        cb, hb = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], cell, a, b, rmin, rmax, method="brute")
        cc, hc = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], cell, a, b, rmin, rmax, method="cells")
        assert hb == hc
        np.testing.assert_array_equal(cb, cc)


def test_wrap_range_and_edges(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(3)
    for Lb in (1.0, 12.5, 100.0, 215.443):
        for x in np.concatenate([rng.uniform(-5 * Lb, 5 * Lb, 2000), [0.0, Lb, -Lb, np.nextafter(np.float32(Lb), 0), -1e-30, 3 * Lb]]):
            w = lib.vo_wrap(float(np.float32(x)), Lb)
            assert 0.0 <= w < np.float32(Lb)
            k = round((float(np.float32(x)) - w) / Lb)
            assert abs(float(np.float32(x)) - k * Lb - w) <= 1e-5 * max(1.0, abs(x))


def test_run_driver_threads_and_sharding_invariance(oracle):
    F, N, Lb = 6, 3000, 50.0
    traj = np.stack([oracle.synth_frame(5, N, Lb, 0.05, f) for f in range(F)])
    cells = [oracle.make_cell(Lb)] * F
    o = np.arange(0, N, 3)
    c1, w1, h1 = oracle.rdf_run(traj, cells, o, o, 0.0, 12.0, nthreads=1)
    c4, w4, h4 = oracle.rdf_run(traj, cells, o, o, 0.0, 12.0, nthreads=4)
    cb, wb, hb = oracle.rdf_run(traj, cells, o, o, 0.0, 12.0, nthreads=2, use_cells=False)
    assert h1 == h4 == hb
    np.testing.assert_array_equal(c1, c4)
    np.testing.assert_array_equal(c1, cb)
    np.testing.assert_allclose(w1, w4, rtol=1e-13)
    ca, _, _ = oracle.rdf_run(traj[:2], cells[:2], o, o, 0.0, 12.0)
    cz, _, _ = oracle.rdf_run(traj[2:], cells[2:], o, o, 0.0, 12.0)
    np.testing.assert_array_equal(ca + cz, c1)
    assert (c1 % 2 == 0).all()                  # same-set histogram: every unordered pair twice


def test_jacobi_eigen_solver(oracle):
    rng = np.random.default_rng(4)
    for _ in range(20):
        A = rng.normal(size=(4, 4))
        A = A + A.T
        w, V = oracle.jacobi4(A)
        np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(A), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(V @ np.diag(w) @ V.T, A, atol=1e-12)
        np.testing.assert_allclose(V.T @ V, np.eye(4), atol=1e-13)


def test_alignment_recovers_rigid_motion_and_sdf_is_invariant(oracle):
    """a rigidly rotated + translated copy of frame 0 gives the same aligned picture."""
    rng = np.random.default_rng(6)
    m, nt = 9, 4000
    ref = rng.normal(0, 2.0, (m, 3))
    tgt = rng.uniform(-15, 15, (nt, 3))
    mass = rng.uniform(1, 16, m).astype(np.float32)
    pts0 = np.concatenate([ref, tgt]).astype(np.float32)
    ang, ax = 1.1, np.array([0.3, -0.5, 0.8]); ax /= np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Q = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    pts1 = (pts0.astype(np.float64) @ Q.T + np.array([3.0, -7.0, 11.0])).astype(np.float32)
    cell = oracle.make_cell(None)
    sidx = np.arange(m, dtype=np.int32)[None, :]
    pose = oracle.sdf_ref_pose(pts0[:, 0], pts0[:, 1], pts0[:, 2], cell, sidx[0], mass)
    M0, R0, c0 = oracle.sdf_frame_align(pts0[:, 0], pts0[:, 1], pts0[:, 2], cell, sidx, mass[None], pose)
    M1, R1, c1 = oracle.sdf_frame_align(pts1[:, 0], pts1[:, 1], pts1[:, 2], cell, sidx, mass[None], pose)
    np.testing.assert_allclose(M0[0][:, :3], np.eye(3), atol=1e-6)          # frame 0 aligns onto itself
    np.testing.assert_allclose(M1[0][:, :3], Q.T, atol=1e-5)                # and the rotation is undone
    q0 = (M0[0][:, :3] @ pts0[m:].T.astype(np.float64)).T + M0[0][:, 3]
    q1 = (M1[0][:, :3] @ pts1[m:].T.astype(np.float64)).T + M1[0][:, 3]
    np.testing.assert_allclose(q0, q1, atol=2e-4)
    t = np.arange(m, m + nt, dtype=np.int32)
    v0, h0 = oracle.sdf_frame_scatter(pts0[:, 0], pts0[:, 1], pts0[:, 2], cell, sidx, R0, c0, t, 10.0, 32)
    v1, h1 = oracle.sdf_frame_scatter(pts1[:, 0], pts1[:, 1], pts1[:, 2], cell, sidx, R1, c1, t, 10.0, 32)
    assert h0 > 500 and abs(int(h0) - int(h1)) <= 3
    assert np.abs(v0.astype(np.int64) - v1.astype(np.int64)).sum() <= 0.02 * h0    # only atoms within fp32 eps of a voxel face move


def test_downsample_and_compute_histogram_fixtures(oracle, golden):
    d = golden["downsample"]
    for c in d["cases"]:
        got = oracle.downsample_histogram(d["values"], d["weights"], c["num_dst_bins"])
        np.testing.assert_array_equal(got, np.array([float(Fraction(s)) for s in c["expected"]], np.float32))
    h = golden["compute_histogram"]
    bins, lo, hi = oracle.compute_histogram(h["values"], h["num_bins"], h["min"], h["max"])
    exp = np.array([float(Fraction(s)) for s in h["expected"]], np.float32)
    np.testing.assert_array_equal(bins, exp)
    assert lo == exp.min() and hi == exp.max()
    # masked variant on a temporal property of population 1 equals the plain one over the masked frames
    vals = np.array(h["values"], np.float32)
    mask = np.ones(vals.size, np.uint8); mask[3] = 0
    got = oracle.compute_histogram_masked(vals, 1, mask, h["num_bins"], h["min"], h["max"])
    ref, _, _ = oracle.compute_histogram(np.delete(vals, 3), h["num_bins"], h["min"], h["max"])
    np.testing.assert_array_equal(got, ref)


def test_synth_is_deterministic_and_in_box(oracle):
    a = oracle.synth_frame(2, 3000, 100.0, 0.05, 17)
    b = oracle.synth_frame(2, 3000, 100.0, 0.05, 17)
    np.testing.assert_array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 100.0
    f0 = oracle.synth_frame(2, 3000, 100.0, 0.05, 0)
    oh = f0[:, 1::3] - f0[:, 0::3]
    oh -= 100.0 * np.round(oh / 100.0)
    assert np.abs(oh).max() <= 0.5501            # O-H offsets of frame 0 stay below 1 Angstrom


def test_triclinic_minimum_image_against_27_image_search(oracle):
    """SPEC S3t (round in fractional space) against an fp64 search over all 27 neighbouring images."""
    rng = np.random.default_rng(9)
    x, y, z, xy, xz, yz = 30.0, 28.0, 26.0, 6.0, -4.0, 5.0
    A = np.array([[x, xy, xz], [0, y, yz], [0, 0, z]])
    n = 600
    pts = (A @ rng.uniform(-0.5, 1.5, (3, n))).astype(np.float32)        # inside and outside the cell
    cell = oracle.make_cell((x, y, z), 7, (xy, xz, yz))
    idx = np.arange(n)
    counts, hits = oracle.rdf_frame(pts[0], pts[1], pts[2], cell, idx, idx, 0.0, 10.0)     # 10 < half the smallest width
    d = pts.astype(np.float64)[:, :, None] - pts.astype(np.float64)[:, None, :]
    best = np.full((n, n), np.inf)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            for k in (-1, 0, 1):
                s = A @ np.array([i, j, k], float)
                # fold the difference by whole cell vectors first (points lie up to 1.5 cells out)
                dd = d - (A @ np.round(np.linalg.solve(A, d.reshape(3, -1)))).reshape(3, n, n) + s[:, None, None]
                best = np.minimum(best, np.sqrt((dd ** 2).sum(axis=0)))
    sel = (best > 0) & (best < 10.0)
    ref = np.bincount((best[sel] / 10.0 * 1024).astype(int), minlength=1024)[:1024]
    assert abs(int(hits) - int(sel.sum())) <= 2
    assert np.abs(counts.astype(np.int64) - ref).sum() <= max(4, 2e-3 * hits)        # fp32 vs fp64 at bin edges only
    # distances use the same arithmetic
    a, b = [3], [77]
    dmin = oracle.distance_minmax(pts[0], pts[1], pts[2], cell, a, b, "min")
    assert abs(dmin - best[3, 77]) < 1e-4
    dcom = oracle.distance_com(pts[0], pts[1], pts[2], cell, a, None, b, None)
    assert abs(dcom - best[3, 77]) < 1e-4


def test_sc_lattice_in_sheared_cells(oracle, golden):
    """The same crystal described by a sheared cell: Z^3 is invariant under a = (n,0,0), b = (s,n,0), c = (t,u,n) with integer
    s, t, u, and the n^3 points of the cube are one representative per class of Z^3 / <a,b,c>.  The triclinic minimum image
    (SPEC S3t) must therefore reproduce the exact shell multiplicities 6, 12, 8, 6, 24, 24, 12, 30 - a known answer that owes
    nothing to the oracle's own 27-image search.  The spacing 1.003 keeps every shell radius >= 0.2 bins away from a bin edge."""
    g = golden["sc_lattice"]
    n, a0 = g["n"], 1.003
    pts = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij")).reshape(3, -1).astype(np.float32) * np.float32(a0)
    idx = np.arange(pts.shape[1])
    rmax, nbins = 3.2, 320
    expect = np.zeros(nbins, np.uint64)
    for d2, mult in g["shells"].items():
        if np.sqrt(int(d2)) * a0 < rmax:
            expect[int(np.sqrt(int(d2)) * a0 * 100.0)] += mult * pts.shape[1]
    for s, t, u in ((0, 0, 0), (3, 0, 0), (3, -2, 3), (-3, 3, -2), (1, 3, -3)):
        cell = oracle.make_cell(n * a0, tilt=(s * a0, t * a0, u * a0))
        # atoms anywhere in space describe the same crystal: shift some by lattice vectors of the sheared cell
        moved = pts.copy()
        moved[:, ::3] += (np.array([s, n, 0], np.float32) * np.float32(a0))[:, None]
        moved[:, 1::5] -= (np.array([t, u, n], np.float32) * np.float32(a0))[:, None]
        for xyz in (pts, moved):
            for method in ("brute", "cells"):
                counts, hits = oracle.rdf_frame(xyz[0], xyz[1], xyz[2], cell, idx, idx, 0.0, rmax, nbins=nbins, method=method)
                np.testing.assert_array_equal(counts, expect, err_msg=f"tilt {(s, t, u)}, {method}")
                assert hits == expect.sum()


@pytest.mark.parametrize("simd", [True, False])
def test_tuned_cpu_baseline_counts_what_the_oracle_counts(oracle, simd):
    """oracle/vmd_cpu_fast.c (bench.py's cpu_baseline "simd" leg: half shell, AVX-512 hit compaction) is NOT the checker - it is
    checked: the same integers as vo_rdf_run, same-set and cross passes, on a box with atoms on cell faces and coincident atoms."""
    import cases
    coords = cases.water_box(oracle, 3, 3000, 41.0, 2)
    coords[:, :, 10] = coords[:, :, 13]                      # coincident atoms (d = 0: outside the open interval)
    coords[0, 0, 16] = 0.0; coords[0, 1, 19] = 41.0 - 1e-5     # on / next to a cell face
    cell = oracle.make_cell(41.0)
    o, h = cases.oxygen(3000), cases.hydrogen(3000)
    for ref, tgt, rmin, rmax in ((o, o, 0.0, 12.0), (o, h, 0.0, 12.0), (h[:700], o, 1.5, 9.0)):
        want, _, hits = oracle.rdf_run(coords, [cell] * 2, ref, tgt, rmin, rmax, nthreads=2)
        got = oracle.rdf_run_fast(coords, [cell] * 2, ref, tgt, rmin, rmax, nthreads=2, simd=simd)
        assert got is not None
        np.testing.assert_array_equal(got[0], want)
        assert got[1] == hits and hits > 0
    tri = oracle.make_cell(41.0, tilt=(5.0, 0.0, 0.0))
    assert oracle.rdf_run_fast(coords, [tri] * 2, o, o, 0.0, 12.0) is None          # triclinic: not its business
