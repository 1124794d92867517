"""Randomised parity hunt on the SIMT-emulator build (CPU): random cells (orthorhombic / triclinic / open / partly periodic),
cutoffs, selections and batch splits, RDF counts against the oracle.  usage: python scripts/fuzz_emu.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, cases
import viamd_amd as V
from viamd_amd import _lib as L
from oracle import oracle as O

lib = V.VmdLib(conftest.build_emu())
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(ncases):
    kind = rng.choice(["ortho", "tri", "open", "mixed"])
    n = int(rng.integers(40, int(os.environ.get("FUZZ_NMAX", "700"))))
    F = int(rng.integers(1, 4))
    Lx, Ly, Lz = rng.uniform(18, 60, 3)
    rmax = float(rng.uniform(2.0, 0.49 * min(Lx, Ly, Lz)))
    rmin = float(rng.choice([0.0, rng.uniform(0, 0.5 * rmax)]))
    flags, tilt = 7, (0.0, 0.0, 0.0)
    if kind == "tri":
        tilt = (rng.uniform(-0.5, 0.5) * Lx, rng.uniform(-0.5, 0.5) * Lx, rng.uniform(-0.5, 0.5) * Ly)
        rmax = min(rmax, 0.3 * min(Lx, Ly, Lz))        # widths shrink with the tilt; the host falls back to brute if too large
    elif kind == "open":
        flags = 0
    elif kind == "mixed":
        flags = int(rng.integers(1, 7))
    box = (float(Lx), float(Ly), float(Lz)) + tuple(float(t) for t in tilt) if kind == "tri" else (float(Lx), float(Ly), float(Lz))
    A = np.array([[Lx, tilt[0], tilt[1]], [0, Ly, tilt[2]], [0, 0, Lz]])
    spread = rng.choice([1.0, 1.6])                     # atoms inside the cell, or up to 30 % outside
    coords = np.stack([(A @ rng.uniform(0.5 - 0.5 * spread, 0.5 + 0.5 * spread, (3, n))) for _ in range(F)]).astype(np.float32)
    if rng.random() < 0.3:
        coords[:, :, : n // 4] = coords[:, :, : n // 4] * 0.2 + np.array([Lx, Ly, Lz])[None, :, None] * 0.4   # a dense blob
    if rng.random() < 0.2:
        coords[:, int(rng.integers(0, 3)), :] = np.float32(rng.uniform(0, 10))                               # planar
    if rng.random() < 0.25:                             # an unwrapped / far-away system: whole cells (or kilo-Angstroms) off the origin
        coords += (A @ rng.integers(-30, 30, (3, 1))).astype(np.float32)[None] if flags == 7 else np.float32(rng.uniform(-3000, 3000))
    same = rng.random() < 0.5
    a = np.sort(rng.choice(n, int(rng.integers(1, n + 1)), replace=False)).astype(np.int32)
    b = a if same else np.sort(rng.choice(n, int(rng.integers(1, n + 1)), replace=False)).astype(np.int32)
    ranges = None
    if F > 1 and rng.random() < 0.5:
        k = int(rng.integers(1, F))
        ranges = [(k, F), (0, k)]
    old = [lib.vmd_set_option(b"cells_fused", int(rng.integers(0, 2))), lib.vmd_set_option(b"cells_split", int(rng.choice([1, 2]))),
           lib.vmd_set_option(b"rdf_nsub", int(rng.choice([0, 1, 3])))]
    desc = f"case {it}: {kind} n={n} F={F} box={tuple(round(v, 2) for v in box)} flags={flags} r=({rmin:.2f},{rmax:.2f}) same={same} |a|={a.size} |b|={b.size}"
    try:
        cases.check_rdf(lib, O, coords, box, [("g", a, b, rmin, rmax)], flags=flags, ranges=ranges, variant=int(rng.integers(0, 2)),
                        oracle_method="brute")
    except AssertionError as e:
        bad += 1
        print("MISMATCH", desc, str(e)[:300].replace("\n", " "))
    except Exception as e:
        bad += 1
        print("ERROR", desc, repr(e)[:300])
    finally:
        lib.vmd_set_option(b"cells_fused", old[0]); lib.vmd_set_option(b"cells_split", old[1]); lib.vmd_set_option(b"rdf_nsub", old[2])
print(f"{ncases} cases, {bad} failures")
