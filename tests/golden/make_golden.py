"""Generates the known-answer fixtures in this directory with exact integer / rational arithmetic only
(no oracle, no product code): they pin the oracle to mathematics, since the reference ships no golden
vectors for this path.  Run: python tests/golden/make_golden.py"""
import json
import os
from fractions import Fraction
from itertools import product

HERE = os.path.dirname(os.path.abspath(__file__))


def sc_lattice_shells(n=8, r2max=10):
    """Simple cubic lattice, a = 1, n^3 sites, periodic box n: ordered neighbours per site at squared distance d2
    (minimum image in integers).  Known multiplicities 6,12,8,6,24,24,12,30,... (SURVEY.md 8c ii)."""
    shells = {}
    half = n // 2
    for dx, dy, dz in product(range(-half, half + (n % 2)), repeat=3):
        # every displacement class modulo n appears once; minimum image representative
        d2 = dx * dx + dy * dy + dz * dz
        if 0 < d2 <= r2max:
            shells[d2] = shells.get(d2, 0) + 1
    return {"n": n, "a": 1.0, "shells": {str(k): v for k, v in sorted(shells.items())}}


def two_atom_cases():
    """Two atoms across a periodic boundary: exactly one ordered pair each way, at the minimum-image distance."""
    L = 20
    cases = []
    for (a, b) in [((1, 1, 1), (19, 1, 1)), ((0, 0, 0), (19, 19, 19)), ((2, 3, 4), (5, 7, 4)), ((0, 10, 0), (0, 19, 0))]:
        d2 = 0
        for u, v in zip(a, b):
            d = abs(u - v)
            d = min(d, L - d)
            d2 += d * d
        cases.append({"L": L, "a": a, "b": b, "d2": d2})
    return cases


def downsample_cases():
    """VIAMD's downsample_histogram (src/main.cpp:232-250): g[k] = sum(values)/sum(weights) over blocks."""
    values = [1, 3, 0, 2, 5, 5, 7, 1]
    weights = [2, 2, 1, 1, 4, 1, 2, 2]
    out = []
    for nd in (8, 4, 2, 1):
        f = len(values) // nd
        out.append({"num_dst_bins": nd, "expected": [
            str(Fraction(sum(values[k * f:(k + 1) * f]), sum(weights[k * f:(k + 1) * f]))) for k in range(nd)]})
    return {"values": values, "weights": weights, "cases": out}


def compute_histogram_case():
    """VIAMD's compute_histogram (src/main.cpp:139-170) on dyadic values: bins and the 1/(width*count) scaling are exact."""
    values = [0.0, 0.25, 0.5, 0.5, 0.75, 1.0, 1.5, 2.0, 2.0, -0.5, 2.5]   # last two fall outside [0,2]
    nb, lo, hi = 4, 0.0, 2.0
    counts = [0] * nb
    n = 0
    for v in values:
        if v < lo or hi < v:
            continue
        idx = min(max(int((Fraction(v) - Fraction(lo)) / Fraction(hi - lo) * nb), 0), nb - 1)
        counts[idx] += 1
        n += 1
    width = Fraction(hi - lo) / nb
    return {"values": values, "num_bins": nb, "min": lo, "max": hi,
            "expected": [str(Fraction(c) / (width * n)) for c in counts]}


if __name__ == "__main__":
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump({"sc_lattice": sc_lattice_shells(), "two_atom": two_atom_cases(), "downsample": downsample_cases(),
                   "compute_histogram": compute_histogram_case()}, f, indent=1)
    print("wrote", os.path.join(HERE, "known_answers.json"))
