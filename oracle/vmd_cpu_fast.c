/* oracle/vmd_cpu_fast.c - a TUNED CPU implementation of the RDF pair count, for bench.py's cpu_baseline only.
 *
 * TEST INFRASTRUCTURE, and not the checker: parity is judged against vmd_oracle.c (the plain restatement of SPEC.md); this file
 * exists because a scalar, full-shell cell list is a weak stand-in for "mdlib's CPU path on the node's cores" (VERDICT r02 weak #7).
 * It computes the same integers - tests/test_oracle.py: vf_rdf_run == vo_rdf_run bit for bit - with the arithmetic of SPEC S2-S4
 * (wrap, minimum image, fused d2, correctly rounded sqrt, the open-interval test on d), but organised for speed:
 *   - cells of edge >= r_max, targets sorted into SoA runs per cell;
 *   - HALF shell for rdf(a, a): 13 neighbour cells + the own cell with j > i, every hit counted twice (d2 is symmetric under i <-> j);
 *   - AVX-512: 16 target atoms per instruction through the minimum image and the cutoff filter; the ~15 % that pass are compressed
 *     (vcompressps) into a buffer and binned 16 at a time (vsqrtps is correctly rounded, so the bins are the oracle's);
 *   - one frame per OpenMP thread, private histograms, like vo_rdf_run.
 * Orthorhombic, fully periodic cells only (the BASELINE configs); anything else returns UINT64_MAX and the caller uses the oracle. */
#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vmd_oracle.h"

static inline float vf_wrap(float x, float L) {           /* SPEC S2, as vo_wrap */
    const float invL = 1.0f / L;
    const float f = floorf(x * invL);
    float xw = fmaf(-f, L, x);
    if (xw < 0.0f) xw = xw + L;
    if (xw >= L) xw = xw - L;
    return xw;
}

typedef struct {
    float rmin, rmax, inv_range, fnbins;
    int nbins;
} vf_bin_t;

static inline int vf_bin_of(const vf_bin_t* b, float d2) {  /* SPEC S4, as vo_bin_of (open interval) */
    const float d = sqrtf(d2);
    if (!(b->rmin < d && d < b->rmax)) return -1;
    int bin = (int)(((d - b->rmin) * b->inv_range) * b->fnbins);
    if (bin < 0) bin = 0;
    if (bin > b->nbins - 1) bin = b->nbins - 1;
    return bin;
}

/* bins the n buffered squared distances; inc = 1 or 2 */
__attribute__((target("avx512f,avx512dq"))) static uint64_t vf_flush512(const vf_bin_t* bn, const float* d2, int n, uint64_t* hist, uint64_t inc) {
    uint64_t hits = 0;
    const __m512 rmin = _mm512_set1_ps(bn->rmin), rmax = _mm512_set1_ps(bn->rmax), inv = _mm512_set1_ps(bn->inv_range), fnb = _mm512_set1_ps(bn->fnbins);
    const __m512i top = _mm512_set1_epi32(bn->nbins - 1), zero = _mm512_setzero_si512();
    for (int k = 0; k < n; k += 16) {
        const __mmask16 live = (__mmask16)(n - k >= 16 ? 0xffff : (1u << (n - k)) - 1u);
        const __m512 d = _mm512_sqrt_ps(_mm512_maskz_loadu_ps(live, d2 + k));
        const __mmask16 m = live & _mm512_cmp_ps_mask(rmin, d, _CMP_LT_OQ) & _mm512_cmp_ps_mask(d, rmax, _CMP_LT_OQ);
        const __m512 t = _mm512_mul_ps(_mm512_mul_ps(_mm512_sub_ps(d, rmin), inv), fnb);
        __m512i b = _mm512_cvttps_epi32(t);
        b = _mm512_min_epi32(_mm512_max_epi32(b, zero), top);
        int32_t bins[16];
        _mm512_storeu_si512((void*)bins, b);
        unsigned mm = m;
        while (mm) { const int l = __builtin_ctz(mm); mm &= mm - 1; hist[bins[l]] += inc; hits += inc; }
    }
    return hits;
}

static uint64_t vf_flush_scalar(const vf_bin_t* bn, const float* d2, int n, uint64_t* hist, uint64_t inc) {
    uint64_t hits = 0;
    for (int k = 0; k < n; ++k) { const int b = vf_bin_of(bn, d2[k]); if (b >= 0) { hist[b] += inc; hits += inc; } }
    return hits;
}

#define VF_BUF 1024

/* one reference atom against targets [jb, je): candidates with d2 <= r2_up go to the buffer */
__attribute__((target("avx512f,avx512dq"))) static inline int vf_row512(float xi, float yi, float zi, const float* sx, const float* sy, const float* sz,
                                                                     size_t jb, size_t je, const float L[3], const float hL[3], float r2_up, float* buf, int n) {
    const __m512 vx = _mm512_set1_ps(xi), vy = _mm512_set1_ps(yi), vz = _mm512_set1_ps(zi);
    const __m512 Lx = _mm512_set1_ps(L[0]), Ly = _mm512_set1_ps(L[1]), Lz = _mm512_set1_ps(L[2]);
    const __m512 hx = _mm512_set1_ps(hL[0]), hy = _mm512_set1_ps(hL[1]), hz = _mm512_set1_ps(hL[2]);
    const __m512 nhx = _mm512_set1_ps(-hL[0]), nhy = _mm512_set1_ps(-hL[1]), nhz = _mm512_set1_ps(-hL[2]);
    const __m512 r2 = _mm512_set1_ps(r2_up), z0 = _mm512_setzero_ps();
    for (size_t j = jb; j < je; j += 16) {
        const __mmask16 live = (__mmask16)(je - j >= 16 ? 0xffff : (1u << (je - j)) - 1u);
        __m512 dx = _mm512_sub_ps(vx, _mm512_maskz_loadu_ps(live, sx + j));
        __m512 dy = _mm512_sub_ps(vy, _mm512_maskz_loadu_ps(live, sy + j));
        __m512 dz = _mm512_sub_ps(vz, _mm512_maskz_loadu_ps(live, sz + j));
        /* SPEC S3: s = d > L/2 ? L : (d < -L/2 ? -L : 0); d = d - s */
        __m512 s = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(dx, hx, _CMP_GT_OQ), _mm512_mask_sub_ps(z0, _mm512_cmp_ps_mask(dx, nhx, _CMP_LT_OQ), z0, Lx), Lx);
        dx = _mm512_sub_ps(dx, s);
        s = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(dy, hy, _CMP_GT_OQ), _mm512_mask_sub_ps(z0, _mm512_cmp_ps_mask(dy, nhy, _CMP_LT_OQ), z0, Ly), Ly);
        dy = _mm512_sub_ps(dy, s);
        s = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(dz, hz, _CMP_GT_OQ), _mm512_mask_sub_ps(z0, _mm512_cmp_ps_mask(dz, nhz, _CMP_LT_OQ), z0, Lz), Lz);
        dz = _mm512_sub_ps(dz, s);
        const __m512 d2 = _mm512_fmadd_ps(dz, dz, _mm512_fmadd_ps(dy, dy, _mm512_mul_ps(dx, dx)));
        const __mmask16 m = live & _mm512_cmp_ps_mask(d2, r2, _CMP_LE_OQ);
        _mm512_mask_compressstoreu_ps(buf + n, m, d2);
        n += __builtin_popcount(m);
    }
    return n;
}

static inline int vf_row_scalar(float xi, float yi, float zi, const float* sx, const float* sy, const float* sz, size_t jb, size_t je,
                                const float L[3], const float hL[3], float r2_up, float* buf, int n) {
    for (size_t j = jb; j < je; ++j) {
        float dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
        dx = dx - (dx > hL[0] ? L[0] : (dx < -hL[0] ? -L[0] : 0.0f));
        dy = dy - (dy > hL[1] ? L[1] : (dy < -hL[1] ? -L[1] : 0.0f));
        dz = dz - (dz > hL[2] ? L[2] : (dz < -hL[2] ? -L[2] : 0.0f));
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        if (d2 <= r2_up) buf[n++] = d2;
    }
    return n;
}

/* one frame; returns the ordered hit count, UINT64_MAX when the cell is not orthorhombic + fully periodic or too small for the grid */
uint64_t vf_rdf_frame(const float* x, const float* y, const float* z, const vo_cell_t* cell, const int32_t* ref_idx, size_t nref,
                      const int32_t* tgt_idx, size_t ntgt, int same, float rmin, float rmax, int nbins, int simd, uint64_t* counts) {
    if (!cell || cell->xy != 0.0f || cell->xz != 0.0f || cell->yz != 0.0f || (cell->flags & 7u) != 7u) return UINT64_MAX;
    const float L[3] = {cell->x, cell->y, cell->z};
    const float hL[3] = {0.5f * L[0], 0.5f * L[1], 0.5f * L[2]};
    vf_bin_t bn = {rmin, rmax, 1.0f / (rmax - rmin), (float)nbins, nbins};
    int n[3];
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        int k = (int)floorf(L[a] / rmax);
        if (k > 256) k = 256;
        while (k > 1 && ((float)k / L[a]) * rmax > 0.9999f) k -= 1;
        if (k < 3) return UINT64_MAX;
        n[a] = k; inv[a] = (float)k / L[a];
    }
    const size_t ncell = (size_t)n[0] * n[1] * n[2];
    const size_t nsets = same ? 1 : 2;
    float* buf = (float*)malloc(sizeof(float) * 3 * (nref + (same ? 0 : ntgt)) + 64 * nsets);
    uint32_t* start[2];
    float* S[2][3];
    const int32_t* idx[2] = {ref_idx, tgt_idx};
    const size_t cnt[2] = {nref, ntgt};
    float* cursor = buf;
    for (size_t s = 0; s < nsets; ++s) {
        const size_t m = cnt[s];
        S[s][0] = cursor; S[s][1] = cursor + m; S[s][2] = cursor + 2 * m; cursor += 3 * m + 16;
        start[s] = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
        uint32_t* c_of = (uint32_t*)malloc(sizeof(uint32_t) * (m ? m : 1));
        float* w = (float*)malloc(sizeof(float) * 3 * (m ? m : 1));
        for (size_t i = 0; i < m; ++i) {
            const int32_t a = idx[s] ? idx[s][i] : (int32_t)i;
            const float wx = vf_wrap(x[a], L[0]), wy = vf_wrap(y[a], L[1]), wz = vf_wrap(z[a], L[2]);
            w[3 * i] = wx; w[3 * i + 1] = wy; w[3 * i + 2] = wz;
            int cx = (int)(wx * inv[0]), cy = (int)(wy * inv[1]), cz = (int)(wz * inv[2]);
            cx = cx < 0 ? 0 : (cx > n[0] - 1 ? n[0] - 1 : cx); cy = cy < 0 ? 0 : (cy > n[1] - 1 ? n[1] - 1 : cy); cz = cz < 0 ? 0 : (cz > n[2] - 1 ? n[2] - 1 : cz);
            c_of[i] = (uint32_t)((cz * n[1] + cy) * n[0] + cx);
            start[s][c_of[i] + 1] += 1;
        }
        for (size_t c = 0; c < ncell; ++c) start[s][c + 1] += start[s][c];
        uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * ncell);
        memcpy(cur, start[s], sizeof(uint32_t) * ncell);
        for (size_t i = 0; i < m; ++i) {
            const uint32_t p = cur[c_of[i]]++;
            S[s][0][p] = w[3 * i]; S[s][1][p] = w[3 * i + 1]; S[s][2][p] = w[3 * i + 2];
        }
        free(cur); free(c_of); free(w);
    }
    const int t = same ? 0 : 1;                            /* which set plays the targets */
    const float r2_up = nextafterf(rmax * rmax, FLT_MAX) * 1.0001f;
    float d2buf[VF_BUF + 16];
    int nb = 0;
    uint64_t hits = 0;
    const uint64_t inc = same ? 2 : 1;
    const int use512 = simd && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    /* half shell (same set): the 13 neighbour cells "above" + the own cell with j > i; full shell otherwise */
    for (int cz = 0; cz < n[2]; ++cz) for (int cy = 0; cy < n[1]; ++cy) for (int cx = 0; cx < n[0]; ++cx) {
        const size_t c = (size_t)(cz * n[1] + cy) * n[0] + cx;
        const size_t ib = start[0][c], ie = start[0][c + 1];
        if (ib == ie) continue;
        for (int oz = same ? 0 : -1; oz <= 1; ++oz) for (int oy = -1; oy <= 1; ++oy) for (int ox = -1; ox <= 1; ++ox) {
            if (same && (oz == 0 && (oy < 0 || (oy == 0 && ox < 0)))) continue;
            const int qx = (cx + ox + n[0]) % n[0], qy = (cy + oy + n[1]) % n[1], qz = (cz + oz + n[2]) % n[2];
            const size_t q = (size_t)(qz * n[1] + qy) * n[0] + qx;
            const int own = same && ox == 0 && oy == 0 && oz == 0;
            const size_t jb0 = start[t][q], je = start[t][q + 1];
            for (size_t i = ib; i < ie; ++i) {
                for (size_t jb = own ? i + 1 : jb0; jb < je; jb += 256) {           /* <= 256 candidates per call: the buffer cannot overrun */
                    const size_t jm = je - jb < 256 ? je : jb + 256;
                    nb = use512 ? vf_row512(S[0][0][i], S[0][1][i], S[0][2][i], S[t][0], S[t][1], S[t][2], jb, jm, L, hL, r2_up, d2buf, nb)
                                : vf_row_scalar(S[0][0][i], S[0][1][i], S[0][2][i], S[t][0], S[t][1], S[t][2], jb, jm, L, hL, r2_up, d2buf, nb);
                    if (nb >= VF_BUF - 256) {
                        hits += use512 ? vf_flush512(&bn, d2buf, nb, counts, inc) : vf_flush_scalar(&bn, d2buf, nb, counts, inc);
                        nb = 0;
                    }
                }
            }
        }
    }
    hits += use512 ? vf_flush512(&bn, d2buf, nb, counts, inc) : vf_flush_scalar(&bn, d2buf, nb, counts, inc);
    for (size_t s = 0; s < nsets; ++s) free(start[s]);
    free(buf);
    return hits;
}

/* frames [0, nframes) of traj (float[F][3][npad]) on nthreads OpenMP threads, one frame per thread at a time (dynamic, like VIAMD
 * hands frame ranges to its pool, /root/reference/src/task_system.cpp:73-81); counts += ordered pairs per bin */
uint64_t vf_rdf_run(const float* traj, const vo_cell_t* cells, size_t nframes, size_t npad, const int32_t* ref_idx, size_t nref,
                    const int32_t* tgt_idx, size_t ntgt, int same, float rmin, float rmax, int nbins, int nthreads, int simd, uint64_t* counts) {
    uint64_t total = 0;
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        uint64_t* priv = (uint64_t*)calloc((size_t)nbins, sizeof(uint64_t));
        uint64_t mine = 0;
#pragma omp for schedule(dynamic, 1)
        for (long f = 0; f < (long)nframes; ++f) {
            const float* fx = traj + (size_t)f * 3 * npad;
            const uint64_t h = vf_rdf_frame(fx, fx + npad, fx + 2 * npad, &cells[f], ref_idx, nref, tgt_idx, ntgt, same, rmin, rmax, nbins, simd, priv);
            if (h == UINT64_MAX) {
#pragma omp atomic write
                bad = 1;
            } else mine += h;
        }
#pragma omp critical
        {
            for (int b = 0; b < nbins; ++b) counts[b] += priv[b];
            total += mine;
        }
        free(priv);
    }
    return bad ? UINT64_MAX : total;
}

int vf_have_avx512(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }
