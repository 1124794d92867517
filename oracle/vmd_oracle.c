/*
 * oracle/vmd_oracle.c — CPU restatement (TEST INFRASTRUCTURE, PARITY UNPINNED; see vmd_oracle.h, SPEC.md).
 * Build with -ffp-contract=off: every fused operation below is an explicit fmaf().
 */
#include "vmd_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VO_PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------------------ S2 */
float vo_wrap(float x, float L) {
    const float invL = 1.0f / L;
    const float t = x * invL;
    const float f = floorf(t);
    float xw = fmaf(-f, L, x);
    if (xw < 0.0f) xw = xw + L;
    if (xw >= L) xw = xw - L;
    return xw;
}

static inline int vo_triclinic(const vo_cell_t* c) { return c && (c->xy != 0.0f || c->xz != 0.0f || c->yz != 0.0f); }

typedef struct {
    float L[3];
    float hL[3];
    int pbc[3];
    int tri;            /* triclinic: tilt factors non-zero (needs all three axes periodic) */
    float xy, xz, yz;   /* basis: a = (x,0,0), b = (xy,y,0), c = (xz,yz,z)  (src/viamd.cpp:1837-1843) */
    float iL[3];        /* fl(1/L) */
} vo_box_t;

static vo_box_t vo_box(const vo_cell_t* c) {
    vo_box_t b;
    memset(&b, 0, sizeof(b));
    if (c) {
        b.L[0] = c->x; b.L[1] = c->y; b.L[2] = c->z;
        for (int a = 0; a < 3; ++a) {
            b.pbc[a] = ((c->flags >> a) & 1u) && b.L[a] > 0.0f;
            b.hL[a] = 0.5f * b.L[a];
            b.iL[a] = 1.0f / b.L[a];
        }
        b.tri = vo_triclinic(c);
        b.xy = c->xy; b.xz = c->xz; b.yz = c->yz;
    }
    return b;
}

/* SPEC S3t (triclinic): fractional coordinates and back, fp32 */
static inline void vo_frac(const vo_box_t* b, float x, float y, float z, float s[3]) {
    s[2] = z * b->iL[2];
    s[1] = fmaf(-b->yz, s[2], y) * b->iL[1];
    s[0] = fmaf(-b->xz, s[2], fmaf(-b->xy, s[1], x)) * b->iL[0];
}
static inline void vo_cart(const vo_box_t* b, const float s[3], float d[3]) {
    d[2] = s[2] * b->L[2];
    d[1] = fmaf(b->yz, s[2], s[1] * b->L[1]);
    d[0] = fmaf(b->xz, s[2], fmaf(b->xy, s[1], s[0] * b->L[0]));
}
/* minimum image of a Cartesian displacement in a triclinic cell: round in fractional space (valid for |d| below half the
 * smallest cell width, the usual restriction) */
static inline void vo_mi_tri(const vo_box_t* b, float d[3]) {
    float s[3];
    vo_frac(b, d[0], d[1], d[2], s);
    s[0] = s[0] - rintf(s[0]); s[1] = s[1] - rintf(s[1]); s[2] = s[2] - rintf(s[2]);
    vo_cart(b, s, d);
}
static inline void vo_mi_tri_d(const vo_box_t* b, double d[3]) {
    double s[3];
    s[2] = d[2] / (double)b->L[2];
    s[1] = (d[1] - (double)b->yz * s[2]) / (double)b->L[1];
    s[0] = (d[0] - (double)b->xy * s[1] - (double)b->xz * s[2]) / (double)b->L[0];
    s[0] = s[0] - rint(s[0]); s[1] = s[1] - rint(s[1]); s[2] = s[2] - rint(s[2]);
    d[2] = s[2] * (double)b->L[2];
    d[1] = s[1] * (double)b->L[1] + (double)b->yz * s[2];
    d[0] = s[0] * (double)b->L[0] + (double)b->xy * s[1] + (double)b->xz * s[2];
}
/* SPEC S3t wrap: fractional coordinates folded into [0,1), back to Cartesian.  u[3] (optional) receives the unsheared
 * coordinates s_k * L_k the cell grid bins by. */
static inline void vo_wrap_tri(const vo_box_t* b, float x, float y, float z, float r[3], float u[3]) {
    float s[3];
    vo_frac(b, x, y, z, s);
    for (int a = 0; a < 3; ++a) {
        s[a] = s[a] - floorf(s[a]);
        if (!(s[a] < 1.0f)) s[a] = 0.0f;
    }
    r[2] = s[2] * b->L[2];
    r[1] = fmaf(b->yz, s[2], s[1] * b->L[1]);
    r[0] = fmaf(b->xz, s[2], fmaf(b->xy, s[1], s[0] * b->L[0]));
    if (u) { u[0] = s[0] * b->L[0]; u[1] = s[1] * b->L[1]; u[2] = s[2] * b->L[2]; }
}
/* SPEC S3t pair: wrapped Cartesian positions; the lattice image n is chosen by rounding the displacement in fractional
 * space, the displacement itself is the Cartesian difference minus the lattice vector n (same form as S3) */
static inline float vo_pair_d2_tri(const vo_box_t* b, const float ri[3], const float rj[3]) {
    const float d0[3] = {ri[0] - rj[0], ri[1] - rj[1], ri[2] - rj[2]};
    float s[3];
    vo_frac(b, d0[0], d0[1], d0[2], s);
    const float nx = rintf(s[0]), ny = rintf(s[1]), nz = rintf(s[2]);
    const float shx = fmaf(nz, b->xz, fmaf(ny, b->xy, nx * b->L[0]));
    const float shy = fmaf(nz, b->yz, ny * b->L[1]);
    const float shz = nz * b->L[2];
    const float dx = d0[0] - shx, dy = d0[1] - shy, dz = d0[2] - shz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* S3 on one axis */
static inline float vo_mi(float d, float L, float hL, int pbc) {
    if (pbc) {
        float s = d > hL ? L : (d < -hL ? -L : 0.0f);
        d = d - s;
    }
    return d;
}

static inline float vo_d2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

/* S4 binning of one distance */
typedef struct {
    float rmin, rmax, inv_range, fnbins;
    int nbins;
} vo_bin_t;

static vo_bin_t vo_bin_setup(float rmin, float rmax, int nbins) {
    vo_bin_t b;
    b.rmin = rmin; b.rmax = rmax; b.nbins = nbins;
    b.inv_range = 1.0f / (rmax - rmin);
    b.fnbins = (float)nbins;
    return b;
}

/* The DECISION: tags of SPEC.md as run-time switches (vo_set_spec; the product mirrors every one of them as
 * vmd_set_option("spec_...")): 0 = the default written in SPEC.md, 1 = the alternative mdlib may turn out to use. */
static int g_spec_rdf_closed = 0;        /* D-RDF-OPEN: 1 = closed interval r_min <= d <= r_max (self pairs at d = 0 count when r_min = 0) */
static int g_spec_sdf_include_self = 0;  /* D-SDF-EXCL: 1 = target atoms that belong to structure k are NOT skipped */
static int g_spec_rdf_raw = 0;           /* D-WRAP: 1 = positions enter the pair computation as they are (no wrap into the cell); the minimum
                                            image is taken by rounding, d = fmaf(-rintf(fl(d * fl(1/L))), L, d) per periodic axis (S3t for
                                            triclinic cells takes unwrapped positions as it stands) */
static int g_spec_rdf_norm = 0;          /* D-RDF-NORM: 0 = cell volume when fully periodic, else the cutoff sphere; 1 = always the cutoff
                                            sphere; 2 = per reference atom (rho = N_tgt / V: the weights do not carry N_ref) */
int vo_set_spec(const char* key, int value) {
    int* o = NULL;
    int multi = 0;
    if (!strcmp(key, "rdf_closed")) o = &g_spec_rdf_closed;
    else if (!strcmp(key, "sdf_include_self")) o = &g_spec_sdf_include_self;
    else if (!strcmp(key, "rdf_raw")) o = &g_spec_rdf_raw;
    else if (!strcmp(key, "rdf_norm")) { o = &g_spec_rdf_norm; multi = 1; }
    if (!o) return -1;
    const int old = *o;
    *o = multi ? value : (value ? 1 : 0);
    return old;
}

static inline int vo_bin_of(const vo_bin_t* b, float d2) {
    const float d = sqrtf(d2);
    if (g_spec_rdf_closed ? !(b->rmin <= d && d <= b->rmax) : !(b->rmin < d && d < b->rmax)) return -1;
    int bin = (int)(((d - b->rmin) * b->inv_range) * b->fnbins);
    if (bin < 0) bin = 0;
    if (bin > b->nbins - 1) bin = b->nbins - 1;
    return bin;
}

static void vo_gather_wrapped(const float* x, const float* y, const float* z, const vo_box_t* bx,
                              const int32_t* idx, size_t n, float* ox, float* oy, float* oz) {
    for (size_t i = 0; i < n; ++i) {
        const int32_t a = idx ? idx[i] : (int32_t)i;
        if (bx->tri) {          /* S3t: wrapped through fractional space */
            float r[3];
            vo_wrap_tri(bx, x[a], y[a], z[a], r, NULL);
            ox[i] = r[0]; oy[i] = r[1]; oz[i] = r[2];
            continue;
        }
        ox[i] = bx->pbc[0] ? vo_wrap(x[a], bx->L[0]) : x[a];
        oy[i] = bx->pbc[1] ? vo_wrap(y[a], bx->L[1]) : y[a];
        oz[i] = bx->pbc[2] ? vo_wrap(z[a], bx->L[2]) : z[a];
    }
}

/* ------------------------------------------------------------------------------------------------ S4 */
uint64_t vo_rdf_frame_brute(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                            const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                            float rmin, float rmax, int nbins, uint64_t* counts) {
    const vo_box_t bx = vo_box(cell);
    if (bx.tri && !(bx.pbc[0] && bx.pbc[1] && bx.pbc[2])) return UINT64_MAX;
    const vo_bin_t bn = vo_bin_setup(rmin, rmax, nbins);
    float* buf = (float*)malloc(sizeof(float) * 3 * (nref + ntgt) + 64);
    float *rx = buf, *ry = rx + nref, *rz = ry + nref, *tx = rz + nref, *ty = tx + ntgt, *tz = ty + ntgt;
    if (g_spec_rdf_raw) {
        for (size_t i = 0; i < nref; ++i) { const int32_t a = ref_idx ? ref_idx[i] : (int32_t)i; rx[i] = x[a]; ry[i] = y[a]; rz[i] = z[a]; }
        for (size_t i = 0; i < ntgt; ++i) { const int32_t a = tgt_idx ? tgt_idx[i] : (int32_t)i; tx[i] = x[a]; ty[i] = y[a]; tz[i] = z[a]; }
    } else {
        vo_gather_wrapped(x, y, z, &bx, ref_idx, nref, rx, ry, rz);
        vo_gather_wrapped(x, y, z, &bx, tgt_idx, ntgt, tx, ty, tz);
    }
    uint64_t hits = 0;
    for (size_t i = 0; i < nref; ++i) {
        const float xi = rx[i], yi = ry[i], zi = rz[i];
        for (size_t j = 0; j < ntgt; ++j) {
            float d2;
            if (bx.tri) {
                const float si[3] = {xi, yi, zi}, sj[3] = {tx[j], ty[j], tz[j]};
                d2 = vo_pair_d2_tri(&bx, si, sj);
            } else if (g_spec_rdf_raw) {
                float d[3] = {xi - tx[j], yi - ty[j], zi - tz[j]};
                for (int a = 0; a < 3; ++a) if (bx.pbc[a]) d[a] = fmaf(-rintf(d[a] * bx.iL[a]), bx.L[a], d[a]);
                d2 = vo_d2(d[0], d[1], d[2]);
            } else {
                const float dx = vo_mi(xi - tx[j], bx.L[0], bx.hL[0], bx.pbc[0]);
                const float dy = vo_mi(yi - ty[j], bx.L[1], bx.hL[1], bx.pbc[1]);
                const float dz = vo_mi(zi - tz[j], bx.L[2], bx.hL[2], bx.pbc[2]);
                d2 = vo_d2(dx, dy, dz);
            }
            const int bin = vo_bin_of(&bn, d2);
            if (bin >= 0) { counts[bin] += 1; hits += 1; }
        }
    }
    free(buf);
    return hits;
}

/* uniform grid with edge >= rmax on every axis; returns 0 on success */
typedef struct {
    int n[3];
    float org[3];
    float inv[3];
} vo_grid_t;

static inline int vo_cell_coord(float v, float org, float inv, int n) {
    int c = (int)((v - org) * inv);
    if (c < 0) c = 0;
    if (c > n - 1) c = n - 1;
    return c;
}

uint64_t vo_rdf_frame_cells(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                            const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                            float rmin, float rmax, int nbins, uint64_t* counts) {
    if (vo_triclinic(cell) || g_spec_rdf_raw) return UINT64_MAX;       /* the grid is orthorhombic, on wrapped positions only: callers fall back to brute */
    const vo_box_t bx = vo_box(cell);
    const vo_bin_t bn = vo_bin_setup(rmin, rmax, nbins);
    if (nref == 0 || ntgt == 0) return 0;

    float* buf = (float*)malloc(sizeof(float) * 3 * (nref + 2 * ntgt) + 64);
    float *rx = buf, *ry = rx + nref, *rz = ry + nref, *tx = rz + nref, *ty = tx + ntgt, *tz = ty + ntgt;
    float *sx = tz + ntgt, *sy = sx + ntgt, *sz = sy + ntgt;
    vo_gather_wrapped(x, y, z, &bx, ref_idx, nref, rx, ry, rz);
    vo_gather_wrapped(x, y, z, &bx, tgt_idx, ntgt, tx, ty, tz);

    vo_grid_t g;
    const float* tc[3] = {tx, ty, tz};
    const float* rc[3] = {rx, ry, rz};
    for (int a = 0; a < 3; ++a) {
        float lo = 0.0f, ext = bx.L[a];
        if (!bx.pbc[a]) {
            float hi = -FLT_MAX; lo = FLT_MAX;
            for (size_t i = 0; i < ntgt; ++i) { if (tc[a][i] < lo) lo = tc[a][i]; if (tc[a][i] > hi) hi = tc[a][i]; }
            for (size_t i = 0; i < nref; ++i) { if (rc[a][i] < lo) lo = rc[a][i]; if (rc[a][i] > hi) hi = rc[a][i]; }
            ext = hi - lo;
        }
        int n = (int)floorf(ext / rmax);
        if (n > 256) n = 256;
        /* a cell must stay wider than rmax after fp rounding of the cell coordinate */
        while (n > 1 && ((float)n / ext) * rmax > 0.9999f) n -= 1;
        if (n < 1) n = 1;
        if (bx.pbc[a] && n < 3) { free(buf); return UINT64_MAX; }  /* stencil would alias: caller uses brute */
        g.n[a] = n; g.org[a] = lo; g.inv[a] = ext > 0.0f ? (float)n / ext : 0.0f;
    }
    const size_t ncell = (size_t)g.n[0] * g.n[1] * g.n[2];
    uint32_t* start = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t* tcell = (uint32_t*)malloc(sizeof(uint32_t) * ntgt);
    for (size_t i = 0; i < ntgt; ++i) {
        const int cx = vo_cell_coord(tx[i], g.org[0], g.inv[0], g.n[0]);
        const int cy = vo_cell_coord(ty[i], g.org[1], g.inv[1], g.n[1]);
        const int cz = vo_cell_coord(tz[i], g.org[2], g.inv[2], g.n[2]);
        const uint32_t c = (uint32_t)((cz * g.n[1] + cy) * g.n[0] + cx);
        tcell[i] = c;
        start[c + 1] += 1;
    }
    for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * ncell);
    memcpy(cur, start, sizeof(uint32_t) * ncell);
    for (size_t i = 0; i < ntgt; ++i) {
        const uint32_t p = cur[tcell[i]]++;
        sx[p] = tx[i]; sy[p] = ty[i]; sz[p] = tz[i];
    }
    /* the conservative candidate filter; the exact open-interval test is in vo_bin_of */
    const float r2_up = nextafterf(rmax * rmax, FLT_MAX) * 1.0001f;

    uint64_t hits = 0;
    float d2buf[256];
    for (size_t i = 0; i < nref; ++i) {
        const float xi = rx[i], yi = ry[i], zi = rz[i];
        const int cx = vo_cell_coord(xi, g.org[0], g.inv[0], g.n[0]);
        const int cy = vo_cell_coord(yi, g.org[1], g.inv[1], g.n[1]);
        const int cz = vo_cell_coord(zi, g.org[2], g.inv[2], g.n[2]);
        for (int oz = -1; oz <= 1; ++oz) {
            int nz_ = cz + oz;
            if (bx.pbc[2]) nz_ = (nz_ + g.n[2]) % g.n[2]; else if (nz_ < 0 || nz_ >= g.n[2]) continue;
            for (int oy = -1; oy <= 1; ++oy) {
                int ny_ = cy + oy;
                if (bx.pbc[1]) ny_ = (ny_ + g.n[1]) % g.n[1]; else if (ny_ < 0 || ny_ >= g.n[1]) continue;
                for (int ox = -1; ox <= 1; ++ox) {
                    int nx_ = cx + ox;
                    if (bx.pbc[0]) nx_ = (nx_ + g.n[0]) % g.n[0]; else if (nx_ < 0 || nx_ >= g.n[0]) continue;
                    const size_t c = (size_t)(nz_ * g.n[1] + ny_) * g.n[0] + nx_;
                    size_t jb = start[c];
                    const size_t je = start[c + 1];
                    while (jb < je) {
                        const size_t nj = je - jb < 256 ? je - jb : 256;
                        for (size_t j = 0; j < nj; ++j) {
                            const float dx = vo_mi(xi - sx[jb + j], bx.L[0], bx.hL[0], bx.pbc[0]);
                            const float dy = vo_mi(yi - sy[jb + j], bx.L[1], bx.hL[1], bx.pbc[1]);
                            const float dz = vo_mi(zi - sz[jb + j], bx.L[2], bx.hL[2], bx.pbc[2]);
                            d2buf[j] = vo_d2(dx, dy, dz);
                        }
                        for (size_t j = 0; j < nj; ++j) {
                            if (d2buf[j] < r2_up) {
                                const int bin = vo_bin_of(&bn, d2buf[j]);
                                if (bin >= 0) { counts[bin] += 1; hits += 1; }
                            }
                        }
                        jb += nj;
                    }
                }
            }
        }
    }
    free(cur); free(tcell); free(start); free(buf);
    return hits;
}

void vo_rdf_weights_frame(const vo_cell_t* cell, size_t nref, size_t ntgt, float rmin, float rmax, int nbins,
                          double* weights) {
    const vo_box_t bx = vo_box(cell);
    double V;
    if (bx.pbc[0] && bx.pbc[1] && bx.pbc[2] && g_spec_rdf_norm != 1) V = (double)bx.L[0] * (double)bx.L[1] * (double)bx.L[2];
    else V = (4.0 / 3.0) * VO_PI * (double)rmax * (double)rmax * (double)rmax;
    const double rho = (g_spec_rdf_norm == 2 ? 1.0 : (double)nref) * (double)ntgt / V;
    const double w = ((double)rmax - (double)rmin) / (double)nbins;
    for (int b = 0; b < nbins; ++b) {
        const double r0 = (double)rmin + w * b;
        const double r1 = (double)rmin + w * (b + 1);
        weights[b] += rho * (4.0 / 3.0) * VO_PI * (r1 * r1 * r1 - r0 * r0 * r0);
    }
}

uint64_t vo_rdf_run(const float* traj, const vo_cell_t* cells, size_t nframes, size_t npad,
                    const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                    float rmin, float rmax, int nbins, int nthreads, int use_cells,
                    uint64_t* counts, double* weights) {
    uint64_t total = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        uint64_t* priv = (uint64_t*)malloc(sizeof(uint64_t) * nbins);
        double* wpriv = (double*)malloc(sizeof(double) * nbins);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long f = 0; f < (long)nframes; ++f) {
            const float* fx = traj + (size_t)f * 3 * npad;
            const float* fy = fx + npad;
            const float* fz = fy + npad;
            memset(priv, 0, sizeof(uint64_t) * nbins);
            memset(wpriv, 0, sizeof(double) * nbins);
            uint64_t h = UINT64_MAX;
            if (use_cells) h = vo_rdf_frame_cells(fx, fy, fz, &cells[f], ref_idx, nref, tgt_idx, ntgt, rmin, rmax, nbins, priv);
            if (h == UINT64_MAX) {
                memset(priv, 0, sizeof(uint64_t) * nbins);
                h = vo_rdf_frame_brute(fx, fy, fz, &cells[f], ref_idx, nref, tgt_idx, ntgt, rmin, rmax, nbins, priv);
            }
            vo_rdf_weights_frame(&cells[f], nref, ntgt, rmin, rmax, nbins, wpriv);
#ifdef _OPENMP
#pragma omp critical
#endif
            {
                for (int b = 0; b < nbins; ++b) { counts[b] += priv[b]; if (weights) weights[b] += wpriv[b]; }
                total += h;
            }
        }
        free(priv); free(wpriv);
    }
    return total;
}

/* ------------------------------------------------------------------------------------------------ S5 */
static inline double vo_mi_rint(double d, double L, int pbc) {
    if (pbc) d = d - L * rint(d / L);
    return d;
}

/* D-SDF-UNWRAP as a switch: with a bond tree installed (vo_set_unwrap_tree) structure k is made whole along its bonds - atom order[t]
 * hangs on atom parent[order[t]] (local indices, parent < 0: the root, taken as it is) - instead of along the index order.  Sums
 * (COM, covariance) always run in index order, so the chain of a whole, index-ordered residue gives the same bits either way. */
static const int32_t* g_tree_order = NULL;   /* [K][m] visiting order */
static const int32_t* g_tree_parent = NULL;  /* [K][m] local parent of every atom */
static size_t g_tree_K = 0, g_tree_m = 0;
void vo_set_unwrap_tree(const int32_t* order, const int32_t* parent, size_t K, size_t m) {
    g_tree_order = order; g_tree_parent = parent; g_tree_K = K; g_tree_m = m;
}

/* The tree mdlib's md_util_unwrap follows (/root/reference/src/viamd.cpp:2257), restated: breadth-first from local atom 0 over the
 * bonds among the m atoms of the structure, neighbours in increasing local index; atoms the walk does not reach are visited
 * afterwards in index order, each hanging on its index predecessor.  bonds: global atom index pairs. */
void vo_bond_tree(const int32_t* bonds, size_t nbonds, const int32_t* idx, size_t m, int32_t* order, int32_t* parent) {
    int* seen = (int*)calloc(m, sizeof(int));
    size_t head = 0, tail = 0;
    for (size_t a = 0; a < m; ++a) parent[a] = -2;
    if (m) { order[tail++] = 0; seen[0] = 1; parent[0] = -1; }
    while (head < tail) {
        const int32_t a = order[head++];
        for (size_t c = 0; c < m; ++c) {                 /* neighbours in increasing local index */
            if (seen[c]) continue;
            int bonded = 0;
            for (size_t b = 0; b < nbonds && !bonded; ++b)
                bonded = (bonds[2 * b] == idx[a] && bonds[2 * b + 1] == idx[c]) || (bonds[2 * b] == idx[c] && bonds[2 * b + 1] == idx[a]);
            if (bonded) { seen[c] = 1; parent[c] = a; order[tail++] = (int32_t)c; }
        }
    }
    for (size_t a = 1; a < m; ++a) if (!seen[a]) { parent[a] = (int32_t)a - 1; order[tail++] = (int32_t)a; }
    free(seen);
}

/* unwrap + mass weighted COM, fp64 sequential; k = structure number (selects the bond tree, if one is installed) */
static void vo_unwrap_com(const float* x, const float* y, const float* z, const vo_box_t* bx,
                          const int32_t* idx, const float* mass, size_t m, size_t k, double* p /*m*3*/, double com[3]) {
    const int tree = g_tree_order && g_tree_parent && k < g_tree_K && g_tree_m == m;
    for (size_t t = 0; t < m; ++t) {
        const size_t a = tree ? (size_t)g_tree_order[k * m + t] : t;
        const long par = tree ? (long)g_tree_parent[k * m + a] : (long)a - 1;
        const int32_t i = idx[a];
        double px = (double)x[i], py = (double)y[i], pz = (double)z[i];
        if (par >= 0 && bx->tri) {
            double d[3] = {px - p[3 * par + 0], py - p[3 * par + 1], pz - p[3 * par + 2]};
            vo_mi_tri_d(bx, d);
            px = p[3 * par + 0] + d[0]; py = p[3 * par + 1] + d[1]; pz = p[3 * par + 2] + d[2];
        } else if (par >= 0) {
            px = p[3 * par + 0] + vo_mi_rint(px - p[3 * par + 0], (double)bx->L[0], bx->pbc[0]);
            py = p[3 * par + 1] + vo_mi_rint(py - p[3 * par + 1], (double)bx->L[1], bx->pbc[1]);
            pz = p[3 * par + 2] + vo_mi_rint(pz - p[3 * par + 2], (double)bx->L[2], bx->pbc[2]);
        }
        p[3 * a + 0] = px; p[3 * a + 1] = py; p[3 * a + 2] = pz;
    }
    double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (size_t a = 0; a < m; ++a) {
        const double w = mass ? (double)mass[a] : 1.0;
        sw = sw + w;
        sx = sx + w * p[3 * a + 0]; sy = sy + w * p[3 * a + 1]; sz = sz + w * p[3 * a + 2];
    }
    com[0] = sx / sw; com[1] = sy / sw; com[2] = sz / sw;
}

void vo_sdf_ref_pose(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                     const int32_t* idx, const float* mass, size_t m, double* ref_pose) {
    const vo_box_t bx = vo_box(cell);
    double com[3];
    vo_unwrap_com(x, y, z, &bx, idx, mass, m, 0, ref_pose, com);
    for (size_t a = 0; a < m; ++a) {
        ref_pose[3 * a + 0] = ref_pose[3 * a + 0] - com[0];
        ref_pose[3 * a + 1] = ref_pose[3 * a + 1] - com[1];
        ref_pose[3 * a + 2] = ref_pose[3 * a + 2] - com[2];
    }
}

/* cyclic Jacobi on a symmetric 4x4; A is destroyed (diagonal = eigenvalues), V = eigenvectors in columns.
 * Fixed order (p,q) = (0,1)(0,2)(0,3)(1,2)(1,3)(2,3), at most 24 sweeps, stops when all off-diagonals are 0. */
void vo_jacobi4(double A[4][4], double V[4][4]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 3; ++p) for (int q = p + 1; q < 4; ++q) off = off + fabs(A[p][q]);
        if (off == 0.0) break;
        for (int p = 0; p < 3; ++p) {
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double at = fabs(theta);
                double t = 1.0 / (at + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                /* A <- J^T A J */
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                A[p][q] = 0.0; A[q][p] = 0.0;
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
}

/* Horn: rotation R (row-major 3x3) with R*cur ~ ref, from weighted covariance S[a][b] = sum w cur_a ref_b */
static void vo_horn_rotation(const double S[3][3], double R[9]) {
    double N[4][4], V[4][4];
    N[0][0] = S[0][0] + S[1][1] + S[2][2];
    N[0][1] = S[1][2] - S[2][1];
    N[0][2] = S[2][0] - S[0][2];
    N[0][3] = S[0][1] - S[1][0];
    N[1][1] = S[0][0] - S[1][1] - S[2][2];
    N[1][2] = S[0][1] + S[1][0];
    N[1][3] = S[2][0] + S[0][2];
    N[2][2] = S[1][1] - S[0][0] - S[2][2];
    N[2][3] = S[1][2] + S[2][1];
    N[3][3] = S[2][2] - S[0][0] - S[1][1];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < i; ++j) N[i][j] = N[j][i];
    vo_jacobi4(N, V);
    int best = 0;
    for (int i = 1; i < 4; ++i) if (N[i][i] > N[best][best]) best = i;
    double qw = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
    const double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw = qw / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
    R[0] = 1.0 - 2.0 * (qy * qy + qz * qz);
    R[1] = 2.0 * (qx * qy - qw * qz);
    R[2] = 2.0 * (qx * qz + qw * qy);
    R[3] = 2.0 * (qx * qy + qw * qz);
    R[4] = 1.0 - 2.0 * (qx * qx + qz * qz);
    R[5] = 2.0 * (qy * qz - qw * qx);
    R[6] = 2.0 * (qx * qz - qw * qy);
    R[7] = 2.0 * (qy * qz + qw * qx);
    R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
}

void vo_sdf_frame_align(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                        const int32_t* struct_idx, const float* struct_mass, size_t K, size_t m,
                        const double* ref_pose, double* M_out, float* R32_out, float* c32_out) {
    const vo_box_t bx = vo_box(cell);
    double* p = (double*)malloc(sizeof(double) * 3 * m);
    for (size_t k = 0; k < K; ++k) {
        const int32_t* idx = struct_idx + k * m;
        const float* mass = struct_mass ? struct_mass + k * m : NULL;
        double com[3];
        vo_unwrap_com(x, y, z, &bx, idx, mass, m, k, p, com);
        double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (size_t a = 0; a < m; ++a) {
            const double w = mass ? (double)mass[a] : 1.0;
            const double c0 = p[3 * a + 0] - com[0], c1 = p[3 * a + 1] - com[1], c2 = p[3 * a + 2] - com[2];
            const double r0 = ref_pose[3 * a + 0], r1 = ref_pose[3 * a + 1], r2 = ref_pose[3 * a + 2];
            const double wc0 = w * c0, wc1 = w * c1, wc2 = w * c2;
            S[0][0] = S[0][0] + wc0 * r0; S[0][1] = S[0][1] + wc0 * r1; S[0][2] = S[0][2] + wc0 * r2;
            S[1][0] = S[1][0] + wc1 * r0; S[1][1] = S[1][1] + wc1 * r1; S[1][2] = S[1][2] + wc1 * r2;
            S[2][0] = S[2][0] + wc2 * r0; S[2][1] = S[2][1] + wc2 * r1; S[2][2] = S[2][2] + wc2 * r2;
        }
        double R[9];
        vo_horn_rotation(S, R);
        if (M_out) {
            double* M = M_out + 12 * k;
            for (int r = 0; r < 3; ++r) {
                M[4 * r + 0] = R[3 * r + 0]; M[4 * r + 1] = R[3 * r + 1]; M[4 * r + 2] = R[3 * r + 2];
                M[4 * r + 3] = -(R[3 * r + 0] * com[0] + R[3 * r + 1] * com[1] + R[3 * r + 2] * com[2]);
            }
        }
        if (R32_out) for (int i = 0; i < 9; ++i) R32_out[9 * k + i] = (float)R[i];
        if (c32_out) for (int i = 0; i < 3; ++i) c32_out[3 * k + i] = (float)com[i];
    }
    free(p);
}

static inline float vo_mi_rintf(float d, float L, int pbc);
/* SPEC S5/S6 fp32 minimum image of a Cartesian displacement (orthorhombic: per axis by rint; triclinic: S3t) */
static inline void vo_mi3_rintf(const vo_box_t* bx, float d[3]) {
    if (bx->tri) { vo_mi_tri(bx, d); return; }
    d[0] = vo_mi_rintf(d[0], bx->L[0], bx->pbc[0]);
    d[1] = vo_mi_rintf(d[1], bx->L[1], bx->pbc[1]);
    d[2] = vo_mi_rintf(d[2], bx->L[2], bx->pbc[2]);
}
static inline float vo_mi_rintf(float d, float L, int pbc) {
    if (pbc) {
        const float invL = 1.0f / L;
        d = fmaf(-rintf(d * invL), L, d);
    }
    return d;
}

uint64_t vo_sdf_frame_scatter(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                              const int32_t* struct_idx, size_t K, size_t m,
                              const float* R32, const float* c32,
                              const int32_t* tgt_idx, size_t ntgt, float s, int dim, uint64_t* vol) {
    const vo_box_t bx = vo_box(cell);
    const float vscale = (float)dim / (2.0f * s);
    const float fdim = (float)dim;
    uint64_t hits = 0;
    for (size_t k = 0; k < K; ++k) {
        const float* R = R32 + 9 * k;
        const float* c = c32 + 3 * k;
        const int32_t* sidx = struct_idx + k * m;
        for (size_t t = 0; t < ntgt; ++t) {
            const int32_t i = tgt_idx ? tgt_idx[t] : (int32_t)t;
            int own = 0;
            if (!g_spec_sdf_include_self)
                for (size_t a = 0; a < m; ++a) if (sidx[a] == i) { own = 1; break; }
            if (own) continue;
            float dv[3] = {x[i] - c[0], y[i] - c[1], z[i] - c[2]};
            vo_mi3_rintf(&bx, dv);
            const float dx = dv[0], dy = dv[1], dz = dv[2];
            const float qx = fmaf(R[2], dz, fmaf(R[1], dy, R[0] * dx));
            const float qy = fmaf(R[5], dz, fmaf(R[4], dy, R[3] * dx));
            const float qz = fmaf(R[8], dz, fmaf(R[7], dy, R[6] * dx));
            const float tx = (qx + s) * vscale;
            const float ty = (qy + s) * vscale;
            const float tz = (qz + s) * vscale;
            if (tx >= 0.0f && tx < fdim && ty >= 0.0f && ty < fdim && tz >= 0.0f && tz < fdim) {
                const int vx = (int)tx, vy = (int)ty, vz = (int)tz;
                vol[((size_t)vz * dim + vy) * dim + vx] += 1;
                hits += 1;
            }
        }
    }
    return hits;
}

/* Multi-threaded SDF driver (same threading model as vo_rdf_run): per frame alignment + scatter into a private
 * hit list, merged into the shared volume at frame end.  ref pose is taken from frame 0 of `traj`. */
uint64_t vo_sdf_run(const float* traj, const vo_cell_t* cells, size_t nframes, size_t npad,
                    const int32_t* struct_idx, const float* struct_mass, size_t K, size_t m,
                    const int32_t* tgt_idx, size_t ntgt, float s, int dim, int nthreads, uint64_t* vol) {
    uint64_t total = 0;
    if (nthreads < 1) nthreads = 1;
    double* ref_pose = (double*)malloc(sizeof(double) * 3 * m);
    vo_sdf_ref_pose(traj, traj + npad, traj + 2 * npad, &cells[0], struct_idx, struct_mass, m, ref_pose);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* R32 = (float*)malloc(sizeof(float) * 9 * K);
        float* c32 = (float*)malloc(sizeof(float) * 3 * K);
        size_t cap = 1 << 16, nh = 0;
        uint32_t* hitlist = (uint32_t*)malloc(sizeof(uint32_t) * cap);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long f = 0; f < (long)nframes; ++f) {
            const float* x = traj + (size_t)f * 3 * npad;
            const float* y = x + npad;
            const float* z = y + npad;
            vo_sdf_frame_align(x, y, z, &cells[f], struct_idx, struct_mass, K, m, ref_pose, NULL, R32, c32);
            const vo_box_t bx = vo_box(&cells[f]);
            const float vscale = (float)dim / (2.0f * s);
            const float fdim = (float)dim;
            nh = 0;
            for (size_t k = 0; k < K; ++k) {
                const float* R = R32 + 9 * k;
                const float* c = c32 + 3 * k;
                const int32_t* sidx = struct_idx + k * m;
                for (size_t t = 0; t < ntgt; ++t) {
                    const int32_t i = tgt_idx ? tgt_idx[t] : (int32_t)t;
                    float dv[3] = {x[i] - c[0], y[i] - c[1], z[i] - c[2]};
                    vo_mi3_rintf(&bx, dv);
                    const float dx = dv[0], dy = dv[1], dz = dv[2];
                    const float qx = fmaf(R[2], dz, fmaf(R[1], dy, R[0] * dx));
                    const float qy = fmaf(R[5], dz, fmaf(R[4], dy, R[3] * dx));
                    const float qz = fmaf(R[8], dz, fmaf(R[7], dy, R[6] * dx));
                    const float tx = (qx + s) * vscale, ty = (qy + s) * vscale, tz = (qz + s) * vscale;
                    if (tx >= 0.0f && tx < fdim && ty >= 0.0f && ty < fdim && tz >= 0.0f && tz < fdim) {
                        int own = 0;
                        for (size_t a = 0; a < m; ++a) if (sidx[a] == i) { own = 1; break; }
                        if (own) continue;
                        if (nh == cap) { cap *= 2; hitlist = (uint32_t*)realloc(hitlist, sizeof(uint32_t) * cap); }
                        hitlist[nh++] = (uint32_t)(((size_t)(int)tz * dim + (int)ty) * dim + (int)tx);
                    }
                }
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            {
                for (size_t h = 0; h < nh; ++h) vol[hitlist[h]] += 1;
                total += nh;
            }
        }
        free(hitlist); free(R32); free(c32);
    }
    free(ref_pose);
    return total;
}

/* ------------------------------------------------------------------------------------------------ S6 */
void vo_set_com(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                const int32_t* idx, const float* mass, size_t n, float out[3]) {
    const vo_box_t bx = vo_box(cell);
    double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    double p0[3] = {0, 0, 0};
    for (size_t a = 0; a < n; ++a) {
        const int32_t i = idx[a];
        double px = (double)x[i], py = (double)y[i], pz = (double)z[i];
        if (a == 0) { p0[0] = px; p0[1] = py; p0[2] = pz; }
        else if (bx.tri) {
            double d[3] = {px - p0[0], py - p0[1], pz - p0[2]};
            vo_mi_tri_d(&bx, d);
            px = p0[0] + d[0]; py = p0[1] + d[1]; pz = p0[2] + d[2];
        } else {
            px = p0[0] + vo_mi_rint(px - p0[0], (double)bx.L[0], bx.pbc[0]);
            py = p0[1] + vo_mi_rint(py - p0[1], (double)bx.L[1], bx.pbc[1]);
            pz = p0[2] + vo_mi_rint(pz - p0[2], (double)bx.L[2], bx.pbc[2]);
        }
        const double w = mass ? (double)mass[a] : 1.0;
        sw = sw + w; sx = sx + w * px; sy = sy + w * py; sz = sz + w * pz;
    }
    out[0] = (float)(sx / sw); out[1] = (float)(sy / sw); out[2] = (float)(sz / sw);
}

float vo_distance_com(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, const float* ma, size_t na, const int32_t* b, const float* mb, size_t nb) {
    const vo_box_t bx = vo_box(cell);
    float ca[3], cb[3];
    vo_set_com(x, y, z, cell, a, ma, na, ca);
    vo_set_com(x, y, z, cell, b, mb, nb, cb);
    float dv[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
    vo_mi3_rintf(&bx, dv);
    return sqrtf(vo_d2(dv[0], dv[1], dv[2]));
}

static float vo_pair_d2(const float* x, const float* y, const float* z, const vo_box_t* bx, int32_t i, int32_t j) {
    if (bx->tri) {
        float ri[3], rj[3];
        vo_wrap_tri(bx, x[i], y[i], z[i], ri, NULL);
        vo_wrap_tri(bx, x[j], y[j], z[j], rj, NULL);
        return vo_pair_d2_tri(bx, ri, rj);
    }
    const float xi = bx->pbc[0] ? vo_wrap(x[i], bx->L[0]) : x[i];
    const float yi = bx->pbc[1] ? vo_wrap(y[i], bx->L[1]) : y[i];
    const float zi = bx->pbc[2] ? vo_wrap(z[i], bx->L[2]) : z[i];
    const float xj = bx->pbc[0] ? vo_wrap(x[j], bx->L[0]) : x[j];
    const float yj = bx->pbc[1] ? vo_wrap(y[j], bx->L[1]) : y[j];
    const float zj = bx->pbc[2] ? vo_wrap(z[j], bx->L[2]) : z[j];
    const float dx = vo_mi(xi - xj, bx->L[0], bx->hL[0], bx->pbc[0]);
    const float dy = vo_mi(yi - yj, bx->L[1], bx->hL[1], bx->pbc[1]);
    const float dz = vo_mi(zi - zj, bx->L[2], bx->hL[2], bx->pbc[2]);
    return vo_d2(dx, dy, dz);
}

float vo_distance_min(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, size_t na, const int32_t* b, size_t nb) {
    const vo_box_t bx = vo_box(cell);
    float best = FLT_MAX;
    for (size_t i = 0; i < na; ++i) for (size_t j = 0; j < nb; ++j) {
        const float d2 = vo_pair_d2(x, y, z, &bx, a[i], b[j]);
        if (d2 < best) best = d2;
    }
    return sqrtf(best);
}

float vo_distance_max(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, size_t na, const int32_t* b, size_t nb) {
    const vo_box_t bx = vo_box(cell);
    float best = 0.0f;
    for (size_t i = 0; i < na; ++i) for (size_t j = 0; j < nb; ++j) {
        const float d2 = vo_pair_d2(x, y, z, &bx, a[i], b[j]);
        if (d2 > best) best = d2;
    }
    return sqrtf(best);
}

void vo_distance_pair(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, size_t na, const int32_t* b, size_t nb, float* out) {
    const vo_box_t bx = vo_box(cell);
    for (size_t i = 0; i < na; ++i) for (size_t j = 0; j < nb; ++j)
        out[i * nb + j] = sqrtf(vo_pair_d2(x, y, z, &bx, a[i], b[j]));
}

/* ------------------------------------------------------------------------------------------------ S8 */
/* restated from /root/reference/src/main.cpp:232-250 */
void vo_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights,
                             int num_src_bins) {
    memset(dst_bins, 0, sizeof(float) * (size_t)num_dst_bins);
    int factor = num_src_bins / num_dst_bins;
    if (factor < 1) factor = 1;
    for (int dst_idx = 0; dst_idx < num_dst_bins; ++dst_idx) {
        double bin = 0.0, weight = 0.0;
        for (int i = 0; i < factor; ++i) {
            const int src_idx = dst_idx * factor + i;
            bin += src_bins[src_idx];
            weight += src_weights ? src_weights[src_idx] : 1.0;
        }
        dst_bins[dst_idx] = (float)(bin / weight);
    }
}

/* restated from /root/reference/src/main.cpp:139-170 */
void vo_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values,
                          int num_values, float* bin_val_min, float* bin_val_max) {
    memset(bins, 0, sizeof(float) * (size_t)num_bins);
    const float range_ext = range_max - range_min;
    const float inv_range = 1.0f / range_ext;
    int count = 0;
    for (int i = 0; i < num_values; ++i) {
        if (values[i] < range_min || range_max < values[i]) continue;
        int idx = (int)(((values[i] - range_min) * inv_range) * num_bins);
        if (idx < 0) idx = 0;
        if (idx > num_bins - 1) idx = num_bins - 1;
        bins[idx] += 1.0f;
        count += 1;
    }
    if (count == 0) {
        if (bin_val_min) *bin_val_min = 0;
        if (bin_val_max) *bin_val_max = 0;
        return;
    }
    float min_val = FLT_MAX, max_val = -FLT_MAX;
    const float width = range_ext / num_bins;
    const float scl = 1.0f / (width * count);
    for (int i = 0; i < num_bins; ++i) {
        bins[i] *= scl;
        /* the reference's MIN(min_val, x) / MAX(max_val, x) macros = (min_val < x ? min_val : x): a NaN bin REPLACES the running
         * value (found by tests/test_ref_pin.py against the compiled reference: an empty range gives NaN, NaN there) */
        min_val = min_val < bins[i] ? min_val : bins[i];
        max_val = max_val > bins[i] ? max_val : bins[i];
    }
    if (bin_val_min) *bin_val_min = min_val;
    if (bin_val_max) *bin_val_max = max_val;
}

/* restated from /root/reference/src/main.cpp:172-230; bins has (aggregate ? 1 : dim) * num_bins entries */
void vo_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                 int dim, const uint8_t* frame_mask, int num_frames, int aggregate) {
    const int hdim = aggregate ? 1 : dim;
    memset(bins, 0, sizeof(float) * (size_t)hdim * (size_t)num_bins);
    int nset = 0;
    for (int f = 0; f < num_frames; ++f) nset += frame_mask[f] ? 1 : 0;
    if (nset * dim == 0) return;
    const float range_ext = range_max - range_min;
    const float inv_range = range_ext > 0.0f ? 1.0f / range_ext : 0.0f;
    int* count = (int*)calloc((size_t)hdim, sizeof(int));
    for (int f = 0; f < num_frames; ++f) {
        if (!frame_mask[f]) continue;
        const int val_idx = dim * f;
        for (int i = 0; i < dim; ++i) {
            const float val = values[val_idx + i];
            if (val < range_min || range_max < val) continue;
            int bin_idx = (int)(((val - range_min) * inv_range) * num_bins);
            if (bin_idx < 0) bin_idx = 0;
            if (bin_idx > num_bins - 1) bin_idx = num_bins - 1;
            if (aggregate) { bins[bin_idx] += 1.0f; count[0] += 1; }
            else { bins[num_bins * i + bin_idx] += 1.0f; count[i] += 1; }
        }
    }
    const float width = range_ext / num_bins;
    for (int i = 0; i < hdim; ++i) {
        const float scl = 1.0f / (width * count[i]);
        for (int j = 0; j < num_bins; ++j) bins[num_bins * i + j] *= scl;
    }
    free(count);
}

/* ------------------------------------------------------------------------------------------------ S9 */
static inline uint64_t vo_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float vo_synth_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t atom) {
    const uint64_t key = vo_mix64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)stream);
    const uint64_t h = vo_mix64(key ^ (((uint64_t)frame << 32) | (uint64_t)atom));
    return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}

void vo_synth_frame(const vo_synth_t* cfg, uint32_t frame, float* x, float* y, float* z) {
    float* out[3] = {x, y, z};
    const float sig = (float)((double)cfg->sigma * sqrt((double)frame));
    for (uint32_t i = cfg->n_blob; i < cfg->n_atoms; ++i) {
        const uint32_t w = i - cfg->n_blob;
        const uint32_t mol = w / 3u, site = w % 3u;
        for (uint32_t c = 0; c < 3; ++c) {
            float p0 = vo_synth_uniform(cfg->seed, 1u + c, 0u, mol) * cfg->L;
            if (site) {
                const float off = (vo_synth_uniform(cfg->seed, 4u + 3u * (site - 1u) + c, 0u, mol) - 0.5f) * 1.1f;
                p0 = p0 + off;
            }
            const float g = (((vo_synth_uniform(cfg->seed, 10u + c, frame, i) + vo_synth_uniform(cfg->seed, 13u + c, frame, i)) +
                              (vo_synth_uniform(cfg->seed, 16u + c, frame, i) + vo_synth_uniform(cfg->seed, 19u + c, frame, i))) - 2.0f) * 1.7320508f;
            const float t = sig * g;
            out[c][i] = vo_wrap(p0 + t, cfg->L);
        }
    }
}
