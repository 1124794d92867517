"""Randomised check of block partials / filtered evaluation on the emulator build: random block sizes, random ways of feeding
the full evaluation, random sub-range queries; every answer must equal a plain evaluation of the same frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, cases
import viamd_amd as V
from viamd_amd import _lib as L
from oracle import oracle as O

on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"          # the product library on a real GPU instead of the emulator build
lib = V.default_lib() if on_gpu else V.VmdLib(conftest.build_emu())
scale = 25 if on_gpu else 1
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
box = 30.0
bad = 0
for it in range(ncases):
    F = int(rng.integers(3, 24))
    coords, structures, mass = cases.sdf_system(O, int(rng.integers(0, 9999)), 240 * scale, box * (scale ** (1.0 / 3.0)), F, K=2, m=4)
    N = coords.shape[2]
    ocell, vcell = cases.cell_pair(O, box * (scale ** (1.0 / 3.0)))
    ox = np.arange(structures.size, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("g", ox, ox, (0.0, 8.0)); ir.add_sdf("v", structures, ox, 6.0); ir.add_distance("d", structures[0], structures[1], L.DIST_COM)
    traj = cases.make_traj(lib, coords, vcell, bool(rng.integers(0, 2)))
    sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
    S = int(rng.integers(1, F + 3))
    full = V.ScriptEval(F, ir); full.set_block_frames(S)
    old = lib.vmd_set_option(b"batch_frames", int(rng.choice([0, 1, 2, 5])))
    old_sb = (lib.vmd_set_option(b"block_superbatch", int(rng.integers(0, 2))), lib.vmd_set_option(b"block_two_streams", int(rng.integers(0, 2))))
    try:
        # feed the full evaluation in random contiguous pieces, in random order
        cuts = sorted(set([0, F] + [int(c) for c in rng.integers(1, F, int(rng.integers(0, 4)))]))
        pieces = list(zip(cuts[:-1], cuts[1:])); rng.shuffle(pieces)
        for a, b in pieces:
            assert full.frame_range(sysm, traj, a, b)
        plain = V.ScriptEval(F, ir); filt = V.ScriptEval(F, ir); filt.set_source(full)
        for q in range(3):
            a = int(rng.integers(0, F)); b = int(rng.integers(a + 1, F + 1))
            plain.clear_data(); filt.clear_data()
            assert plain.frame_range(sysm, traj, a, b) and filt.frame_range(sysm, traj, a, b)
            for nm in ("g", "v"):
                if not np.array_equal(plain.property_data(nm).counts, filt.property_data(nm).counts):
                    raise AssertionError(f"{nm} differs for [{a},{b}) S={S} pieces={pieces}")
            np.testing.assert_allclose(plain.property_data("g").weights64, filt.property_data("g").weights64, rtol=1e-12)
            np.testing.assert_array_equal(plain.property_data("d").values[a:b], filt.property_data("d").values[a:b])
            np.testing.assert_array_equal(plain.frame_mask(), filt.frame_mask())
            c, r = filt.frame_stats()
            assert c + r == b - a
    except AssertionError as e:
        bad += 1
        print("MISMATCH case", it, F, S, str(e)[:300])
    finally:
        lib.vmd_set_option(b"batch_frames", old)
        lib.vmd_set_option(b"block_superbatch", old_sb[0]); lib.vmd_set_option(b"block_two_streams", old_sb[1])
print(f"{ncases} cases, {bad} failures")
