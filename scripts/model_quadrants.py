"""Candidate-column count of the pencil pair kernel under three j layouts (numpy model of the geometry):
  a) today: one x window per (i chunk of 64, neighbour pencil), half-width r
  b) j atoms of a pencil split into 2 x 2 quadrants of its cross-section, one window per (chunk, pencil, quadrant) shrunk to
     sqrt(r^2 - gap^2) with gap = distance between the chunk's y/z box and the quadrant's rectangle (quadrants out of reach skipped)
  c) the bound: columns that contain at least one hit
c3-like: rho = 0.0333 selected atoms / A^3, r = 12 A, pencils ~12.67 A (17 x 17 of a 215.44 A box), half shell.
usage: python scripts/model_quadrants.py [npencils_sampled]"""
import sys

import numpy as np

rng = np.random.default_rng(1)
L, r, n = 215.443, 12.0, 333334
ny = nz = 17
w = L / ny
pos = rng.uniform(0, L, (n, 3))
py, pz = (pos[:, 1] / w).astype(int), (pos[:, 2] / w).astype(int)
pen = pz * ny + py
order = np.lexsort((pos[:, 0], pen))
pos, pen = pos[order], pen[order]
start = np.searchsorted(pen, np.arange(ny * nz + 1))
half = [(0, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]          # (dy, dz) of the half shell
nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cols_a = cols_b = cols_hit = hits = segs_a = segs_b = 0
chunks = 0
for p in rng.choice(ny * nz, nsample, replace=False):
    ipz, ipy = divmod(p, ny)
    P = pos[start[p]:start[p + 1]]
    for c0 in range(0, len(P), 64):
        I = P[c0:c0 + 64]
        chunks += 1
        xlo, xhi = I[:, 0].min(), I[:, 0].max()
        ylo, yhi, zlo, zhi = I[:, 1].min(), I[:, 1].max(), I[:, 2].min(), I[:, 2].max()
        for dy, dz in half:
            qy, qz = (ipy + dy) % ny, (ipz + dz) % nz
            J = pos[start[qz * ny + qy]:start[qz * ny + qy + 1]].copy()
            # bring the neighbour pencil next to the chunk (periodic in y, z; x images through the minimum image below)
            J[:, 1] += ((ipy + dy) - qy) * w
            J[:, 2] += ((ipz + dz) - qz) * w
            own = dy == 0 and dz == 0
            def window(Jq, rad):
                dxlo, dxhi = xlo - rad, xhi + rad
                x = Jq[:, 0]
                m = np.zeros(len(Jq), bool)
                for s in (-L, 0.0, L):
                    m |= (x + s >= dxlo) & (x + s <= dxhi)
                return m
            if own:
                J = J[J[:, 0] > xlo]                    # j > i, roughly: half of the own pencil
            ma = window(J, r)
            cols_a += ma.sum(); segs_a += 1
            d = J[ma][:, None, :] - I[None, :, :]
            d[:, :, 0] -= L * np.round(d[:, :, 0] / L)
            h = ((d ** 2).sum(-1) < r * r)
            if own:
                h &= (J[ma][:, None, 0] > I[None, :, 0])
            hits += h.sum(); cols_hit += h.any(1).sum()
            # quadrants of the neighbour pencil's cross-section
            y0, z0 = (ipy + dy) * w, (ipz + dz) * w
            for hy in (0, 1):
                for hz in (0, 1):
                    ry0, ry1 = y0 + hy * w / 2, y0 + (hy + 1) * w / 2
                    rz0, rz1 = z0 + hz * w / 2, z0 + (hz + 1) * w / 2
                    gy = max(0.0, ry0 - yhi, ylo - ry1)
                    gz = max(0.0, rz0 - zhi, zlo - rz1)
                    g2 = gy * gy + gz * gz
                    if g2 >= r * r:
                        continue
                    Jq = J[(J[:, 1] >= ry0) & (J[:, 1] < ry1) & (J[:, 2] >= rz0) & (J[:, 2] < rz1)]
                    cols_b += window(Jq, np.sqrt(r * r - g2)).sum(); segs_b += 1
print(f"chunks {chunks}: per chunk  a) {cols_a / chunks:7.1f} columns in {segs_a / chunks:4.1f} segments   b) {cols_b / chunks:7.1f} columns in {segs_b / chunks:4.1f} segments"
      f"   c) {cols_hit / chunks:7.1f} columns with a hit;  hits per chunk {hits / chunks:8.1f}")
print(f"candidate lanes per counted (unordered) hit: a) {64 * cols_a / hits:5.2f}   b) {64 * cols_b / hits:5.2f}   c) {64 * cols_hit / hits:5.2f}")
print(f"columns b / a = {cols_b / cols_a:5.3f};  columns with a hit / a = {cols_hit / cols_a:5.3f}")
