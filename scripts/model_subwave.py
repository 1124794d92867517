"""Step 2 of VERDICT r04 next #2: does a sub-wave (4 rows x 16 lanes, DPP row_newbcast) pair kernel beat k_rdf_pencil?

Geometry (numpy, one frame of the workload's density) x instruction cost (profiles/r05a_subwave_calibration.txt, measured on MI355X in a
harness with k_rdf_pencil's push / pop / LDS footprint at 7 waves per SIMD):

  today   wave = chunk of 64 i atoms of a pencil; per neighbour pencil ONE x window [xlo - r, xhi + r] of the whole chunk;
          a column = 64 i lanes x 1 j; cost per column = PK(hits per column)
  rows    the same chunk as FOUR rows of 16 consecutive atoms; each row its own window [xlo_r - r, xhi_r + r] in the same neighbour
          pencil and its own j stream; the rows advance in lock step, 16 columns per j block: steps = max over rows of ceil(n_r / 16);
          a column = 4 x (16 i lanes x 1 j); cost per column = DPP(hits per column)
  rows/S  the same with pencils of 1/S the cross-section edge (the kernel's pencil_split walk: 2S+1 neighbours per axis, windows of the
          neighbours two pencils away shrunk to sqrt(r^2 - gap^2))

PK and DPP are straight lines through the calibration rows (ps per column at chip level against hits per column).
usage: python scripts/model_subwave.py [rho_selected=0.0333] [pencils_sampled=40]"""
import sys

import numpy as np

rho = float(sys.argv[1]) if len(sys.argv) > 1 else 333334 / 215.443 ** 3
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 40
r = 12.0
L = 215.443 if rho < 0.05 else 129.27        # keep the atom count near 3.3e5 .. 2e5 so that the model runs in seconds
n = int(round(rho * L ** 3))
rng = np.random.default_rng(5)

# calibration, occupancy 7 (profiles/r05a_subwave_calibration.txt): (hits per column, ps per column)
PK = np.array([(9.01, 17.556), (9.17, 17.566)])
DPP = np.array([(13.19, 20.841), (14.04, 21.294), (14.42, 21.276), (15.86, 21.788), (20.82, 23.428)])
dpp_fit = np.polyfit(DPP[:, 0], DPP[:, 1], 1)
# the packed loop has one calibration density; its slope is the per-hit cost of the shared push / pop machinery = DPP's slope
pk_fit = np.array([dpp_fit[0], PK[:, 1].mean() - dpp_fit[0] * PK[:, 0].mean()])
print(f"cost lines (ps per column, chip): PK = {pk_fit[1]:.2f} + {pk_fit[0]:.3f} h   DPP = {dpp_fit[1]:.2f} + {dpp_fit[0]:.3f} h")


def build(S):
    npen = int(L / (r / S))                 # pencils per axis: cross-section edge >= r / S
    w = L / npen
    pos = rng.uniform(0, L, (n, 3))
    py, pz = np.minimum((pos[:, 1] / w).astype(int), npen - 1), np.minimum((pos[:, 2] / w).astype(int), npen - 1)
    pen = pz * npen + py
    order = np.lexsort((pos[:, 0], pen))
    pos, pen = pos[order], pen[order]
    start = np.searchsorted(pen, np.arange(npen * npen + 1))
    return npen, w, pos, start


def half_shell(S):
    out = []
    for dz in range(0, S + 1):
        for dy in range(-S, S + 1):
            if dz == 0 and dy < 0:
                continue
            out.append((dy, dz))
    return out


def run(S, rows, streams=False):
    """streams: every row walks the concatenation of ITS windows on its own (per-lane j addresses, per-lane image shift broadcast with
    the j atom); the wave runs until the longest row is through: steps = max over rows of ceil(sum of its windows / 16)"""
    npen, w, pos, start = build(S)
    cols = hits = 0.0
    ideal = 0.0
    for p in rng.choice(npen * npen, min(nsample * S * S, npen * npen), replace=False):
        ipz, ipy = divmod(p, npen)
        P = pos[start[p]:start[p + 1]]
        for c0 in range(0, len(P), 64):
            I = P[c0:c0 + 64]
            tiles = [I[k:k + 16] for k in range(0, len(I), 16)] if rows else [I]
            row_total = [0] * len(tiles)
            for dy, dz in half_shell(S):
                qy, qz = (ipy + dy) % npen, (ipz + dz) % npen
                J = pos[start[qz * npen + qy]:start[qz * npen + qy + 1]]
                gy, gz = max(abs(dy) - 1, 0) * w, max(abs(dz) - 1, 0) * w
                rr = r * r - gy * gy - gz * gz
                if rr <= 0:
                    continue
                rad = np.sqrt(rr)
                own = dy == 0 and dz == 0
                xj = J[:, 0]
                Jy = J[:, 1] + ((ipy + dy) - qy) * w
                Jz = J[:, 2] + ((ipz + dz) - qz) * w
                for s in (-L, 0.0, L):                           # x images: each its own segment
                    lens = []
                    for T in tiles:
                        lo, hi = T[:, 0].min() - rad, T[:, 0].max() + rad
                        m = (xj + s >= lo) & (xj + s <= hi)
                        if own:
                            m &= (xj + s > T[:, 0].min())           # j > i: roughly the half of the own pencil above the tile's first atom
                        k = int(m.sum())
                        lens.append(k)
                        if k:
                            d2 = (xj[m][:, None] + s - T[None, :, 0]) ** 2 + (Jy[m][:, None] - T[None, :, 1]) ** 2 + (Jz[m][:, None] - T[None, :, 2]) ** 2
                            h = d2 < r * r
                            if own:
                                h &= (xj[m][:, None] + s > T[None, :, 0])
                            hits += h.sum()
                    if max(lens) == 0:
                        continue
                    if streams:
                        for t_, k in enumerate(lens):
                            row_total[t_] += k
                        ideal += sum(lens) / 4.0
                    elif rows:
                        cols += 16 * max(-(-k // 16) for k in lens)      # lock step: the longest row, whole j blocks
                        ideal += sum(lens) / 4.0
                    else:
                        cols += -(-lens[0] // 4) * 4                      # groups of 4 columns (the tail runs unpacked: counted as a group)
                        ideal += lens[0]
            if streams:
                cols += 16 * max(-(-k // 16) for k in row_total)
    return cols, hits, ideal


base = None
for name, S, rows, streams in (("today: 64 i lanes, packed filter", 1, False, False), ("rows: 4 x 16, DPP filter, same pencils, lock step per window", 1, True, False),
                               ("rows/2: half-width pencils, lock step per window", 2, True, False), ("today on half-width pencils", 2, False, False),
                               ("streams: 4 x 16, DPP, same pencils, rows independent", 1, True, True),
                               ("streams/2: half-width pencils, rows independent", 2, True, True),
                               ("streams/3: third-width pencils, rows independent", 3, True, True)):
    cols, hits, ideal = run(S, rows, streams)
    h = hits / cols
    fit = dpp_fit if rows else pk_fit
    ps_col = fit[1] + fit[0] * h
    t = cols * ps_col / hits                 # ps per hit
    if base is None:
        base = t
    print(f"{name:62s} R_eff {64 / h:6.2f} (windows alone {64 * ideal / hits:5.2f})  hits/col {h:5.2f}  ps/col {ps_col:6.2f}  ps/hit {t:6.3f}  = {t / base:5.3f} x today")
