"""Randomised parity hunt for SDF / distance on the SIMT-emulator build: random cells (orthorhombic / triclinic / open / partly
periodic), cutoffs, target sets.  usage: python scripts/fuzz_emu_sdf.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, cases
import viamd_amd as V
from viamd_amd import _lib as L
from oracle import oracle as O

lib = V.VmdLib(conftest.build_emu())
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
bad = 0
for it in range(ncases):
    kind = rng.choice(["ortho", "tri", "open", "mixed"])
    Lc = float(rng.uniform(24, 40))
    K, m = int(rng.integers(1, 5)), int(rng.integers(3, 8))
    nw = int(rng.integers(100, 500)) * 3
    F = int(rng.integers(1, 4))
    coords, structures, mass = cases.sdf_system(O, int(rng.integers(0, 10_000)), nw, Lc, F, K=K, m=m)
    N = coords.shape[2]
    flags, tilt = 7, (0.0, 0.0, 0.0)
    box3 = (Lc, Lc * float(rng.uniform(0.9, 1.1)), Lc * float(rng.uniform(0.9, 1.1)))
    if kind == "tri":
        tilt = tuple(float(rng.uniform(-0.4, 0.4)) * box3[0] for _ in range(2)) + (float(rng.uniform(-0.4, 0.4)) * box3[1],)
    elif kind == "open":
        flags = 0
    elif kind == "mixed":
        flags = int(rng.integers(1, 7))
    A = np.array([[box3[0], tilt[0], tilt[1]], [0, box3[1], tilt[2]], [0, 0, box3[2]]])
    frac = coords.astype(np.float64) / Lc * float(rng.choice([1.0, 1.4])) - float(rng.choice([0.0, 0.2]))
    coords = np.einsum("ij,fjn->fin", A, frac).astype(np.float32)
    for f in range(F):                       # keep every structure compact (the alignment needs whole structures)
        for k in range(K):
            idx = structures[k]
            c0 = coords[f][:, idx[:1]]
            coords[f][:, idx] = c0 + 0.15 * (coords[f][:, idx] - c0)
    box = box3 + tilt if kind == "tri" else box3
    n_s = structures.size
    tgt = np.arange(n_s, N, int(rng.choice([1, 3])), dtype=np.int32) if rng.random() < 0.7 else np.arange(N, dtype=np.int32)
    cutoff = float(rng.uniform(3.0, 0.3 * Lc))
    desc = f"case {it}: {kind} K={K} m={m} N={N} F={F} box={tuple(round(v, 2) for v in box)} flags={flags} cutoff={cutoff:.2f} |tgt|={tgt.size}"
    if only >= 0 and it != only:
        continue
    if only >= 0:
        np.savez("/tmp/fuzz_case.npz", coords=coords, structures=structures, mass=mass, tgt=tgt, cutoff=cutoff, box=np.array(box), flags=flags)
    try:
        cases.check_sdf(lib, O, coords, box, structures, mass, tgt, cutoff, flags=flags)
        specs = [("d", structures[0], structures[-1], L.DIST_COM), ("mn", structures[0], tgt[:40], L.DIST_MIN),
                 ("mx", structures[-1][:2], tgt[5:30], L.DIST_MAX), ("p", structures[0][:2], tgt[:3], L.DIST_PAIR)]
        if K > 1:       # populations: `distance*(...) in <contexts>`, one context per structure
            h = max(1, m // 2)
            specs += [("pc", [s_[:h] for s_ in structures], [s_[h:] for s_ in structures], L.DIST_COM, "pop"),
                      ("pm", [s_[:1] for s_ in structures], [s_[1:] for s_ in structures], L.DIST_MIN, "pop"),
                      ("px", [s_[:h] for s_ in structures], [s_[h:] for s_ in structures], L.DIST_MAX, "pop"),
                      ("pp", [s_[:2] for s_ in structures], [s_[2:] for s_ in structures], L.DIST_PAIR, "pop")]
        rg = None
        if F > 1 and rng.random() < 0.5:
            kk = int(rng.integers(1, F)); rg = [(kk, F), (0, kk)]
        cases.check_distances(lib, O, coords, box, mass, specs, flags=flags, ranges=rg)
    except AssertionError as e:
        if not str(e):
            continue                      # the harness's own "volume is not empty" check: nothing to compare in this case
        bad += 1
        print("MISMATCH", desc, str(e)[:300].replace("\n", " "))
    except Exception as e:
        bad += 1
        print("ERROR", desc, repr(e)[:300])
print(f"{ncases} cases, {bad} failures")
