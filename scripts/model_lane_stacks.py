#!/usr/bin/env python
"""Model of VERDICT r03 #3 before building it: lane-private LDS hit stacks for k_rdf_pencil instead of the wave-wide compaction.

Every wave64 VALU instruction of this kernel issues in ~4.2 SIMD cycles whatever its class (profiles/r02_valu_calibration.txt: 570 - 590 G
instructions/s chip-wide for v_cmp, v_mbcnt, v_lshl_add, v_pk_* with an SGPR operand; the packed ops carry two columns), so the model
counts VALU instructions per candidate column (64 i lanes x 1 j).  Inputs measured on c3 (BENCH_r03): 15 % of the lanes of a
column are hits, 46 % of the columns have no hit at all (the push is skipped by s_cbranch_vccz), 7.5 VALU instructions per column today.

  today          filter 3 (12 packed ops per 4 columns) + v_cmp 1 + [hit column: 2 v_mbcnt + v_lshl_add = 3] + pops (9 VALU per 64 hits, always
                 full) + 3 SALU-side stack checks per 4 columns
  lane-private   filter 3 + v_cmpx 1 + [hit column: exec-masked v_add of the lane's slot address = 1] + 1 v_cmp per 4 columns (any lane's
                 stack nearly full?) + pops by ROW: row r of the stacks holds an entry only for the lanes with more than r hits since the last
                 drain, so a pop (10 VALU: the 9 of today + the row mask) bins fewer than 64 hits
  ring           the same with a per-lane ring buffer and a uniform tail: only rows every lane has filled are popped (always 64 hits), at the
                 price of the wrap (v_and, +1 per hit column) and a cross-lane minimum of the heads (6 DPP steps + readlane) per check

LDS: 160 KB per CU and 7 blocks of 4 waves leave 22.5 KB per block; with one histogram per block (4 KB, rdf_shared_hist) a wave can have
R = 16 rows of 64 x 4 bytes.  The hits of a lane are drawn independently per column here (p = 0.15 given the column has a hit at all:
p_col = 0.15 / 0.54) - real neighbours are spatially correlated, which makes the stacks MORE uneven than this."""
import numpy as np

rng = np.random.default_rng(1)
P_HIT, P_EMPTY = 0.15, 0.46
p_lane = P_HIT / (1.0 - P_EMPTY)
NCOL = 400000


def columns(n):
    nonempty = rng.random(n) >= P_EMPTY
    hits = rng.random((n, 64)) < p_lane
    hits &= nonempty[:, None]
    return hits


def today(h):
    hitcols = h.any(axis=1).mean()
    pops = h.sum() / 64.0 / len(h)
    return 3 + 1 + 3 * hitcols + 9 * pops, dict(hit_columns=hitcols, pops_per_column=pops, pop_fill=1.0)


def lane_private(h, R, headroom=4):
    cnt = np.zeros(64, np.int64)
    rows = 0
    hits_popped = 0
    checks = 0
    for k in range(0, len(h), 4):
        cnt += h[k:k + 4].sum(axis=0)
        checks += 1
        if cnt.max() >= R - headroom:
            rows += cnt.max(); hits_popped += cnt.sum(); cnt[:] = 0
    hitcols = h.any(axis=1).mean()
    n = len(h)
    valu = 3 + 1 + 1 * hitcols + checks / n + 10.0 * rows / n
    return valu, dict(pop_fill=hits_popped / max(rows, 1) / 64.0, pops_per_column=rows / n)


def ring(h, R, every=16):
    head = np.zeros(64, np.int64)
    tail = 0
    pops = 0
    mins = 0
    forced = 0
    for k in range(0, len(h), 4):
        head += h[k:k + 4].sum(axis=0)
        if (k // 4) % (every // 4) == 0 or head.max() - tail >= R - 4:
            mins += 1
            full = head.min() - tail
            pops += full; tail += full
            if head.max() - tail >= R - 4:            # still no room: the uneven lanes force a masked drain of everything
                forced += head.max() - tail; tail = head.max(); head[:] = tail
    hitcols = h.any(axis=1).mean()
    n = len(h)
    valu = 3 + 1 + 2 * hitcols + 9.0 * pops / n + 10.0 * forced / n + 8.0 * mins / n
    return valu, dict(full_pops_per_column=pops / n, forced_rows_per_column=forced / n, min_reductions_per_column=mins / n)


h = columns(NCOL)
t, info = today(h)
print(f"today                      {t:5.2f} VALU / column   {info}")
for R in (16, 24, 32, 64):
    v, i = lane_private(h, R)
    print(f"lane-private, R = {R:2d} rows   {v:5.2f} VALU / column  ({100 * (v / t - 1):+5.1f} %)   LDS {R * 256} B per wave   {i}")
for R in (16, 32):
    v, i = ring(h, R)
    print(f"ring, R = {R:2d} rows           {v:5.2f} VALU / column  ({100 * (v / t - 1):+5.1f} %)   LDS {R * 256} B per wave   {i}")
print("SALU per column: today 5 per hit column (2 x s_mov exec, s_bcnt1, s_lshl2_add, branch) + 3 per 4 columns; lane-private 2 per hit column")
