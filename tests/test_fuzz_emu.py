"""Seeded random RDF scenarios on the SIMT-emulator build against the oracle's all-pairs method: anisotropic / fluctuating /
triclinic / partly periodic / open cells, coordinates far outside the cell, clustered and sparse selections, overlapping
selections sharing one eval (the class decomposition), r_min > 0, cutoffs on both sides of the half-cell limit (grid and all-pairs
kernels), every cell-build path.  The scenarios are fixed by their seed; a failure prints it."""
import numpy as np
import pytest

import cases
from viamd_amd import _lib as L


def scenario(seed, scale=1):
    """scale > 1 (the GPU twin, tests/test_fuzz_gpu.py): scale x the atoms in a cell scale^(1/3) x as wide (the same density)"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 700)) * scale
    F = int(rng.integers(1, 4))
    kind = rng.choice(["ortho", "tri", "partial", "open"], p=[0.4, 0.25, 0.2, 0.15])
    Ls = rng.uniform(16.0, 60.0, 3) * float(scale) ** (1.0 / 3.0)
    flags = L.PBC_ALL
    if kind == "partial":
        flags = int(rng.choice([1, 2, 3, 4, 5, 6]))
    boxes, frames = [], []
    cluster = rng.random() < 0.35
    for f in range(F):
        s = 1.0 + 0.03 * rng.uniform(-1, 1, 3) if rng.random() < 0.6 else np.ones(3)      # cell fluctuating from frame to frame
        Lf = Ls * s
        if kind == "tri":
            xy, xz = rng.uniform(-0.5, 0.5, 2) * Lf[0]
            yz = rng.uniform(-0.5, 0.5) * Lf[1]
            bx = (float(Lf[0]), float(Lf[1]), float(Lf[2]), float(xy), float(xz), float(yz))
            A = np.array([[bx[0], bx[3], bx[4]], [0, bx[1], bx[5]], [0, 0, bx[2]]])
        else:
            bx = (float(Lf[0]), float(Lf[1]), float(Lf[2]))
            A = np.diag(Lf)
        frac = rng.uniform(-0.4, 1.4, (3, n)) if rng.random() < 0.5 else rng.uniform(0.0, 1.0, (3, n))
        if cluster:                                 # a third of the atoms in a tight blob: uneven pencils, full buckets
            k = n // 3
            frac[:, :k] = rng.uniform(0, 1, (3, 1)) + 0.04 * rng.normal(size=(3, k))
        if rng.random() < 0.3:                      # whole periodic images away
            frac += rng.integers(-2, 3, (3, n)) * (np.array([[flags & 1], [(flags >> 1) & 1], [(flags >> 2) & 1]]) if kind != "open" else 0)
        frames.append((A @ frac).astype(np.float32))
        boxes.append(bx)
    coords = np.stack(frames)
    box = None if kind == "open" else (boxes if F > 1 else boxes[0])
    # selections drawn from a few base sets so that properties of one eval overlap (class decomposition)
    allidx = np.arange(n, dtype=np.int32)
    base = [allidx, allidx[::2], allidx[1::3], allidx[rng.random(n) < 0.3], allidx[: max(2, n // 5)], allidx[rng.random(n) < 0.7]]
    base = [b for b in base if b.size >= 1]
    wmin = float(min(Ls) * 0.97)
    props = []
    wmin_cut = wmin / float(scale) ** (1.0 / 3.0) if scale > 1 else wmin      # cutoffs stay at molecular scale when the cell grows
    shared_rmax = float(rng.uniform(3.0, 0.48 * wmin_cut))
    for i in range(int(rng.integers(1, 4))):
        a = base[int(rng.integers(len(base)))]
        b = a if rng.random() < 0.45 else base[int(rng.integers(len(base)))]
        u = rng.random()
        rmax = shared_rmax if u < 0.6 else float(rng.uniform(2.0, 0.48 * wmin_cut)) if u < 0.9 else float(rng.uniform(0.5 * wmin, 0.8 * wmin))
        if scale > 1 and u >= 0.9 and n > 6000:
            rmax = shared_rmax                      # the all-pairs kernel is meant for small systems
        if kind == "tri":
            rmax = min(rmax, 0.3 * wmin)            # the perpendicular widths of a sheared cell are smaller than its edges
        rmin = 0.0 if rng.random() < 0.6 else float(rng.uniform(0.1, 0.6) * rmax)
        props.append((f"p{i}", a, b, rmin, rmax))
    opts = {}
    if rng.random() < 0.25:
        opts["cells_pencil"] = 0
    if rng.random() < 0.2:
        opts["rdf_classes"] = 0
    if rng.random() < 0.2:
        opts["rdf_variant"] = int(rng.choice([1, 2, 3]))
    if rng.random() < 0.15:
        opts["pencil_split_y"] = 2
    if rng.random() < 0.2:
        opts["batch_frames"] = 1
    if seed % 5 == 0:
        opts["cells_rec3"] = 0                      # 16-byte bucket records (no draw from rng: the scenarios stay what they were)
    if seed % 7 == 0:
        opts["cells_bin_lds"] = 0                   # level 1 of the cell build without the block-local sort
    if seed % 3 == 1:
        opts["rdf_nsplit"] = -1                     # small launches: a chunk's neighbour pencils dealt to separate work items
    if seed % 4 == 1:
        opts["defer_sync"] = 1                      # the next batch queued before the host waits for the current one (several batches
        opts.setdefault("batch_frames", 2)          # needed; no draw from rng)
    # round 6 (no draws from rng: the scenarios stay what they were): the base sets above are mostly PERIODIC index lists (all, every 2nd,
    # every 3rd from 1, a prefix), which the cell build computes instead of reading - every 6th scenario reads the lists as before; every
    # 4th sends small selections through the pencil buckets as well (cells_small = 0), every 9th measures capacities at batch ends only
    if seed % 7 == 3:
        opts["rdf_shared_hist"] = 0                 # wave-private LDS histograms (the default until round 6; seven waves per SIMD)
    if seed % 5 == 3:
        opts["rdf_pop"] = 0                         # the 9-instruction pop (default: margin folded into the constant, ds_read_addtid_b32)
    if seed % 6 == 2:
        opts["cells_sel_pattern"] = 0
    if seed % 4 == 3:
        opts["cells_small"] = 0
    if seed % 9 == 4:
        opts["cells_cap_sample"] = 2
    return coords, box, flags, props, opts, kind


@pytest.mark.parametrize("chunk", range(6))
def test_random_rdf_scenarios(emu_lib, oracle, chunk):
    for seed in range(chunk * 6, chunk * 6 + 6):
        coords, box, flags, props, opts, kind = scenario(seed)
        old = {k: emu_lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
        try:
            cases.check_rdf(emu_lib, oracle, coords, box, props, flags=flags, oracle_method="brute")
        except Exception as ex:
            raise AssertionError(f"scenario seed {seed} ({kind}, N {coords.shape[2]}, F {coords.shape[0]}, flags {flags}, opts {opts}, "
                                 f"props {[(p[0], p[1].size, p[2].size, p[3], p[4]) for p in props]}): {ex}") from ex
        finally:
            for k, v in old.items():
                emu_lib.vmd_set_option(k.encode(), v)


def sdf_scenario(seed):
    rng = np.random.default_rng(5000 + seed)
    K, m = int(rng.integers(1, 7)), int(rng.integers(3, 13))
    n_s = K * m
    n = n_s + int(rng.integers(30, 900))
    F = int(rng.integers(1, 4))
    kind = rng.choice(["ortho", "tri", "partial", "open"], p=[0.45, 0.25, 0.15, 0.15])
    Ls = rng.uniform(18.0, 50.0, 3)
    flags = L.PBC_ALL if kind in ("ortho", "tri") else int(rng.choice([1, 2, 3, 4, 5, 6])) if kind == "partial" else 0
    if kind == "tri":
        box = (float(Ls[0]), float(Ls[1]), float(Ls[2]), float(rng.uniform(-0.5, 0.5) * Ls[0]), float(rng.uniform(-0.5, 0.5) * Ls[0]),
               float(rng.uniform(-0.5, 0.5) * Ls[1]))
        A = np.array([[box[0], box[3], box[4]], [0, box[1], box[5]], [0, 0, box[2]]])
    else:
        box = (float(Ls[0]), float(Ls[1]), float(Ls[2]))
        A = np.diag(Ls)
    template = rng.normal(0, rng.uniform(0.8, 1.8), (m, 3))
    centers = (A @ rng.uniform(0, 1, (3, K))).T

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx

    coords = np.zeros((F, 3, n), np.float32)
    wrap = rng.random() < 0.5 and kind in ("ortho", "partial")
    for f in range(F):
        pts = (A @ rng.uniform(-0.2, 1.2, (3, n))).astype(np.float64)
        for k in range(K):
            c = centers[k] + rng.normal(0, 0.4, 3) * f
            p = template @ rot(rng.normal(size=3), rng.uniform(0, np.pi)).T + c + rng.normal(0, 0.05, (m, 3))
            if wrap:                                    # structures straddle the periodic faces atom by atom
                for ax in range(3):
                    if flags & (1 << ax):
                        p[:, ax] = np.mod(p[:, ax], Ls[ax])
            pts[:, k * m:(k + 1) * m] = p.T
        coords[f] = pts.astype(np.float32)
    structures = np.arange(n_s, dtype=np.int32).reshape(K, m)
    if rng.random() < 0.3:                              # structure atoms not in index order
        structures = np.stack([rng.permutation(r) for r in structures]).astype(np.int32)
    mass = rng.choice([12.011, 14.007, 15.999, 1.008], n).astype(np.float32)
    u = rng.random()
    allidx = np.arange(n, dtype=np.int32)
    if u < 0.3:                                         # arithmetic progression (the generated-index fast path), structure atoms excluded
        tgt = np.arange(n_s + int(rng.integers(0, 3)), n, int(rng.integers(1, 4)), dtype=np.int32)
    elif u < 0.5:                                       # progression through the structures: a structure's own atoms must be skipped
        tgt = np.arange(int(rng.integers(0, 3)), n, int(rng.integers(1, 4)), dtype=np.int32)
    elif u < 0.8:
        tgt = allidx[rng.random(n) < rng.uniform(0.2, 0.9)]
    else:
        tgt = allidx
    if tgt.size == 0:
        tgt = allidx
    wmin = float(min(Ls))
    cutoff = float(rng.uniform(2.0, (0.3 if kind == "tri" else 0.45) * wmin))
    opts = {}
    if rng.random() < 0.3:
        opts["sdf_dense"] = int(rng.choice([0, 1]))
    if rng.random() < 0.3:
        opts["sdf_arith"] = 0
    if rng.random() < 0.3:
        opts["sdf_ilp"] = int(rng.choice([1, 2, 8]))
    if rng.random() < 0.2:
        opts["batch_frames"] = 1
    if seed % 3 == 0:
        opts["sdf_wave"] = 1                        # per-wave compaction (no draw from rng: the scenarios stay what they were)
    if seed % 5 == 2:
        opts["sdf_wave"] = 2                        # the persistent streaming scatter kernel
    if seed % 4 == 3:
        opts["defer_sync"] = 1
        opts.setdefault("batch_frames", 2)
    dist = []
    for i, kd in enumerate((L.DIST_COM, L.DIST_MIN, L.DIST_MAX, L.DIST_PAIR)):
        a = rng.choice(n, int(rng.integers(1, 9)), replace=False).astype(np.int32)
        b = rng.choice(n, int(rng.integers(1, 9)), replace=False).astype(np.int32)
        dist.append((f"d{i}", a, b, kd))
    return coords, (None if kind == "open" else box), flags, structures, mass, tgt, cutoff, opts, dist, kind


@pytest.mark.parametrize("chunk", range(4))
def test_random_sdf_and_distance_scenarios(emu_lib, oracle, chunk):
    for seed in range(chunk * 5, chunk * 5 + 5):
        coords, box, flags, structures, mass, tgt, cutoff, opts, dist, kind = sdf_scenario(seed)
        old = {k: emu_lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
        try:
            cases.check_distances(emu_lib, oracle, coords, box, mass, dist, flags=flags)
            cases.check_sdf(emu_lib, oracle, coords, box, structures, mass, tgt, cutoff, flags=flags, allow_empty=True)
        except Exception as ex:
            raise AssertionError(f"scenario seed {seed} ({kind}, N {coords.shape[2]}, F {coords.shape[0]}, flags {flags}, K x m {structures.shape}, "
                                 f"targets {tgt.size}, cutoff {cutoff:.3f}, opts {opts}): {ex}") from ex
        finally:
            for k, v in old.items():
                emu_lib.vmd_set_option(k.encode(), v)


def test_seed_8941_of_the_round_4_gpu_campaign(emu_lib, oracle):
    """Found by scripts/fuzz_gpu.py on the MI355X (profiles/r04n): three co-evaluated RDFs with different cutoffs on selections of ~1 000 atoms, a
    cell that changes from frame to frame, one frame per batch.  On one group's grid the selection is sorted by the single-block build, on
    another's (too many fine cells for LDS) through pencil buckets - which overflowed in the second frame; the evaluator widened the buckets of
    selections by `used_pencil`, which only remembers the LAST build of a batch, found nobody to widen and gave up after four attempts."""
    coords, box, flags, props, opts, kind = scenario(8941, scale=20)
    old = {k: emu_lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
    try:
        cases.check_rdf(emu_lib, oracle, coords, box, props, flags=flags, oracle_method="brute")
    finally:
        for k, v in old.items():
            emu_lib.vmd_set_option(k.encode(), v)
