// scripts/subwave_calib.hip - round 5, VERDICT r04 "next" #2, step 1: does a sub-wave i tile pay on gfx950?
//
// Measures, inside the SAME harness (same LDS footprint as k_rdf_pencil: 7 blocks of 4 waves per CU, same hand-scheduled push
// and pop, same hit stack), the cost per candidate column and per HIT of
//   PK   today's inner loop: 64 i atoms in the lanes, j atoms wave-uniform through s_load_dwordx4, v_pk_* filter
//        (12 VOP3P per 4 columns), vmd_push_hot4, vmd_pop_hot;
//   DPP  the proposed one: a wave is FOUR rows of 16 lanes, each row its own tile of 16 i atoms and its own j stream; a VGPR
//        triple holds 16 j atoms per row (lane c of a row = j atom c of that row's block) and column k takes its j operand through
//        DPP row_newbcast:k (v_sub_f32_dpp), so the filter is VGPR-only: 3 v_sub_f32_dpp + v_mul_f32 + 2 v_fma_f32 per column;
//        the next j block comes through 3 global_load_dword per 16 columns, issued one block ahead;
//   VV   DPP's filter with plain VGPR operands (no DPP, no loads): what the DPP modifier itself costs;
// plus the plain instruction streams v_sub_f32 / v_sub_f32_dpp row_newbcast.
// Geometry decides the hit density, so the coordinates are real: PK draws 64 i atoms in a pencil chunk (12.67 x 12.67 x 12 A) and
// j atoms in its box window; DPP draws 16 i atoms per row in a cube of edge `a` and j atoms in the cube's window, optionally
// trimmed to the atoms within `trim` of the cube (trim = r: the Minkowski sum, the best any window shape can do).  The host
// replays the arithmetic and reports hit lanes per column and the fraction of columns with a hit; the kernel's own hit count
// (the stack traffic) must agree.
//
// build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/subwave_calib scripts/subwave_calib.hip && /tmp/subwave_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((vector_size(8)));
typedef float f4 __attribute__((vector_size(16)));
#define UNIFORM_AS __attribute__((address_space(4)))
typedef const UNIFORM_AS float cf32;

#define NBINS 1024
#define QCAP 384
#define LDS_ADDRESS(p) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))

struct wave_t { unsigned qbase, qtop, hbase; };

// vmd_push_hot4 of vmd_kernels.hip, verbatim
__device__ __forceinline__ void push4(wave_t& w, float d0, float d1, float d2, float d3, float r2) {
    unsigned t, n;
    asm volatile(
        "s_nop 0\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d0]\n\t"
        "s_cbranch_vccz .Lp4_0_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d0]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lp4_0_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d1]\n\t"
        "s_cbranch_vccz .Lp4_1_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d1]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lp4_1_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
        "s_cbranch_vccz .Lp4_2_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d2]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lp4_2_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d3]\n\t"
        "s_cbranch_vccz .Lp4_3_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d3]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lp4_3_%=:\n\t"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n)
        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [r2] "s"(r2)
        : "vcc", "scc", "memory");
}

// vmd_pop_hot of vmd_kernels.hip (the exact-path parking left out: every popped hit is binned by the fast path)
__device__ __forceinline__ void pop(wave_t& w, unsigned lane4, unsigned inc, float fast_k, float fast_c, float half, unsigned nb) {
    unsigned long long m;
    float t, d2;
    int b;
    asm volatile(
        "v_add_u32 %[b], %[q], %[l4]\n\t"
        "ds_read_b32 %[d2], %[b]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sqrt_f32 %[t], %[d2]\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[t], %[k], %[t], %[c]\n\t"
        "v_cvt_flr_i32_f32 %[b], %[t]\n\t"
        "v_fract_f32 %[t], %[t]\n\t"
        "v_add_f32 %[t], -0.5, %[t]\n\t"
        "v_cmp_lt_f32 %[m], |%[t]|, %[half]\n\t"
        "v_cmp_gt_u32 vcc, %[nb], %[b]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "v_lshl_add_u32 %[b], %[b], 2, %[hb]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[b], %[inc]\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [m] "=&s"(m), [t] "=&v"(t), [b] "=&v"(b), [d2] "=&v"(d2)
        : [q] "s"(w.qtop), [l4] "v"(lane4), [k] "s"(fast_k), [c] "v"(fast_c), [half] "s"(half), [nb] "s"(nb), [hb] "s"(w.hbase), [inc] "v"(inc)
        : "vcc", "scc", "memory");
}

__device__ __forceinline__ void drain_full(wave_t& w, unsigned lane4, unsigned inc, float fast_k, float fast_c, float half, unsigned& pops) {
    while (w.qtop - w.qbase >= 256u) {
        w.qtop -= 256u;
        pop(w, lane4, inc, fast_k, fast_c, half, NBINS);
        pops += 1;
    }
}

enum Mode { M_PK = 0, M_DPP = 1, M_VV = 2, M_SUB = 3, M_SUB_DPP = 4, M_DPP_NOLOAD = 5 };

// one DPP filter group: columns K0 .. K0+3 of the j block (xj, yj, zj), four interleaved chains
#define DPP_GROUP(K0, K1, K2, K3)                                                                                                   \
    asm volatile(                                                                                                                   \
        "v_sub_f32_dpp %[x0], %[xj], %[xi] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[x1], %[xj], %[xi] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[x2], %[xj], %[xi] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[x3], %[xj], %[xi] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y0], %[yj], %[yi] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y1], %[yj], %[yi] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y2], %[yj], %[yi] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y3], %[yj], %[yi] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_mul_f32 %[x0], %[x0], %[x0]\n\t"                                                                                        \
        "v_mul_f32 %[x1], %[x1], %[x1]\n\t"                                                                                        \
        "v_mul_f32 %[x2], %[x2], %[x2]\n\t"                                                                                        \
        "v_mul_f32 %[x3], %[x3], %[x3]\n\t"                                                                                        \
        "v_fma_f32 %[x0], %[y0], %[y0], %[x0]\n\t"                                                                                 \
        "v_fma_f32 %[x1], %[y1], %[y1], %[x1]\n\t"                                                                                 \
        "v_fma_f32 %[x2], %[y2], %[y2], %[x2]\n\t"                                                                                 \
        "v_fma_f32 %[x3], %[y3], %[y3], %[x3]\n\t"                                                                                 \
        "v_sub_f32_dpp %[y0], %[zj], %[zi] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y1], %[zj], %[zi] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y2], %[zj], %[zi] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_sub_f32_dpp %[y3], %[zj], %[zi] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"                                      \
        "v_fma_f32 %[x0], %[y0], %[y0], %[x0]\n\t"                                                                                 \
        "v_fma_f32 %[x1], %[y1], %[y1], %[x1]\n\t"                                                                                 \
        "v_fma_f32 %[x2], %[y2], %[y2], %[x2]\n\t"                                                                                 \
        "v_fma_f32 %[x3], %[y3], %[y3], %[x3]\n\t"                                                                                 \
        : [x0] "=&v"(q0), [x1] "=&v"(q1), [x2] "=&v"(q2), [x3] "=&v"(q3), [y0] "=&v"(t0), [y1] "=&v"(t1), [y2] "=&v"(t2), [y3] "=&v"(t3)  \
        : [xj] "v"(xj), [yj] "v"(yj), [zj] "v"(zj), [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi));

struct params_t {
    const float* icoord;      // PK: 3 x 64 (x row, y row, z row); DPP: the same (lane = 16 * row + c)
    const float* jcoord;      // PK: nblk blocks of {x[16], y[16], z[16]} (wave-uniform); DPP: nblk blocks of {x[64], y[64], z[64]} (lane = 16 * row + c)
    int nblk;                 // j blocks (16 columns each) cycled through
    int trips;                // j blocks processed per wave
    float r2, fast_k, fast_c, half;
    uint64_t* ticks;          // per wave
    unsigned long long* hits; // per wave: entries pushed
    unsigned* sink;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_cols(params_t p) {
    __shared__ unsigned s_hist[4][NBINS];
    __shared__ float s_queue[4][QCAP];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    wave_t w;
    w.qbase = LDS_ADDRESS(s_queue[wave]); w.qtop = w.qbase; w.hbase = LDS_ADDRESS(s_hist[wave]);
    for (int b = lane; b < NBINS; b += 64) s_hist[wave][b] = 0u;
    __builtin_amdgcn_wave_barrier();
    const float xi = p.icoord[lane], yi = p.icoord[64 + lane], zi = p.icoord[128 + lane];
    const unsigned lane4 = 4u * lane, inc = 2u;
    float fast_c; asm volatile("v_mov_b32 %0, %1" : "=v"(fast_c) : "s"(p.fast_c));
    const float r2 = p.r2, fast_k = p.fast_k, half = p.half;
    unsigned pops = 0;
    unsigned long long pushed = 0;
    float acc = 0.0f;
    __syncthreads();
    const uint64_t t_begin = __builtin_readcyclecounter();
    if (MODE == M_PK) {
        cf32* jc = (cf32*)p.jcoord;
        const f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
        int blk = 0;
        for (int trip = 0; trip < p.trips; ++trip) {
            cf32* base = jc + 48 * blk;
            // four groups of four columns; loads of the next group issued before the current one is processed (as vmd_segment_loop)
            f4 xa = *(const UNIFORM_AS f4*)(base), ya = *(const UNIFORM_AS f4*)(base + 16), za = *(const UNIFORM_AS f4*)(base + 32);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4 xb = xa, yb = ya, zb = za;
                if (g < 3) { xb = *(const UNIFORM_AS f4*)(base + 4 * (g + 1)); yb = *(const UNIFORM_AS f4*)(base + 16 + 4 * (g + 1)); zb = *(const UNIFORM_AS f4*)(base + 32 + 4 * (g + 1)); }
                f2 d2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f2 xj2 = {xa[2 * h], xa[2 * h + 1]}, yj2 = {ya[2 * h], ya[2 * h + 1]}, zj2 = {za[2 * h], za[2 * h + 1]};
                    const f2 dx = xi2 - xj2, dy = yi2 - yj2, dz = zi2 - zj2;
                    d2[h] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                }
                const unsigned q0 = w.qtop;
                push4(w, d2[0][0], d2[0][1], d2[1][0], d2[1][1], r2);
                pushed += (w.qtop - q0) >> 2;
                drain_full(w, lane4, inc, fast_k, fast_c, half, pops);
                xa = xb; ya = yb; za = zb;
            }
            blk = blk + 1 == p.nblk ? 0 : blk + 1;
        }
    } else if (MODE == M_DPP || MODE == M_VV || MODE == M_DPP_NOLOAD) {
        const float* jc = p.jcoord;
        int blk = 0;
        float xj = jc[lane], yj = jc[64 + lane], zj = jc[128 + lane];
        for (int trip = 0; trip < p.trips; ++trip) {
            const int nblk_next = blk + 1 == p.nblk ? 0 : blk + 1;
            float xn = xj, yn = yj, zn = zj;
            if (MODE != M_DPP_NOLOAD && MODE != M_VV) {      // next j block: three coalesced dword loads, one block ahead
                const float* nb = jc + 192 * nblk_next;
                xn = __builtin_nontemporal_load(nb + lane); yn = __builtin_nontemporal_load(nb + 64 + lane); zn = __builtin_nontemporal_load(nb + 128 + lane);
            }
            float q0, q1, q2, q3, t0, t1, t2, t3;
#define GROUP_TAIL()                                                                  \
            { const unsigned qq = w.qtop; push4(w, q0, q1, q2, q3, r2); pushed += (w.qtop - qq) >> 2; \
              drain_full(w, lane4, inc, fast_k, fast_c, half, pops); }
            if (MODE == M_VV) {
                // same arithmetic, j operand = this lane's own j (no DPP): 6 VGPR-only ops per column
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    asm volatile(
                        "v_sub_f32 %[x0], %[xj], %[xi]\n\tv_sub_f32 %[x1], %[xj], %[xi]\n\tv_sub_f32 %[x2], %[xj], %[xi]\n\tv_sub_f32 %[x3], %[xj], %[xi]\n\t"
                        "v_sub_f32 %[y0], %[yj], %[yi]\n\tv_sub_f32 %[y1], %[yj], %[yi]\n\tv_sub_f32 %[y2], %[yj], %[yi]\n\tv_sub_f32 %[y3], %[yj], %[yi]\n\t"
                        "v_mul_f32 %[x0], %[x0], %[x0]\n\tv_mul_f32 %[x1], %[x1], %[x1]\n\tv_mul_f32 %[x2], %[x2], %[x2]\n\tv_mul_f32 %[x3], %[x3], %[x3]\n\t"
                        "v_fma_f32 %[x0], %[y0], %[y0], %[x0]\n\tv_fma_f32 %[x1], %[y1], %[y1], %[x1]\n\tv_fma_f32 %[x2], %[y2], %[y2], %[x2]\n\tv_fma_f32 %[x3], %[y3], %[y3], %[x3]\n\t"
                        "v_sub_f32 %[y0], %[zj], %[zi]\n\tv_sub_f32 %[y1], %[zj], %[zi]\n\tv_sub_f32 %[y2], %[zj], %[zi]\n\tv_sub_f32 %[y3], %[zj], %[zi]\n\t"
                        "v_fma_f32 %[x0], %[y0], %[y0], %[x0]\n\tv_fma_f32 %[x1], %[y1], %[y1], %[x1]\n\tv_fma_f32 %[x2], %[y2], %[y2], %[x2]\n\tv_fma_f32 %[x3], %[y3], %[y3], %[x3]\n\t"
                        : [x0] "=&v"(q0), [x1] "=&v"(q1), [x2] "=&v"(q2), [x3] "=&v"(q3), [y0] "=&v"(t0), [y1] "=&v"(t1), [y2] "=&v"(t2), [y3] "=&v"(t3)
                        : [xj] "v"(xj), [yj] "v"(yj), [zj] "v"(zj), [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi));
                    GROUP_TAIL()
                }
            } else {
                DPP_GROUP(0, 1, 2, 3) GROUP_TAIL()
                DPP_GROUP(4, 5, 6, 7) GROUP_TAIL()
                DPP_GROUP(8, 9, 10, 11) GROUP_TAIL()
                DPP_GROUP(12, 13, 14, 15) GROUP_TAIL()
            }
            xj = xn; yj = yn; zj = zn;
            blk = nblk_next;
        }
        acc += xj;
    } else {
        // plain streams: 64 instructions per trip, 8 independent accumulators
        float a0 = xi, a1 = yi, a2 = zi, a3 = xi + 1.0f, a4 = yi + 1.0f, a5 = zi + 1.0f, a6 = xi + 2.0f, a7 = yi + 2.0f;
        const float d = 1.0e-9f;
        for (int trip = 0; trip < p.trips; ++trip) {
#define S8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
            if (MODE == M_SUB) {
#define X(n) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##n) : "v"(d));
                S8(X) S8(X) S8(X) S8(X) S8(X) S8(X) S8(X) S8(X)
#undef X
            } else {
                // the DPP operand is never the result of the previous instruction (as in the filter: the j VGPRs are loaded far ahead)
#define X(n) asm volatile("v_sub_f32_dpp %0, %1, %0 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(a##n) : "v"(zi));
                S8(X) S8(X) S8(X) S8(X) S8(X) S8(X) S8(X) S8(X)
#undef X
            }
        }
        acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    const uint64_t t_end = __builtin_readcyclecounter();
    acc += (float)pops + (float)w.qtop + (float)s_hist[wave][lane];
    if (acc == 1.2345f) p.sink[0] = 1u;
    if (lane == 0) { p.ticks[blockIdx.x * 4 + wave] = t_end - t_begin; p.hits[blockIdx.x * 4 + wave] = pushed; }
}

// ------------------------------------------------------------------------------------------------ host
static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double urand() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return (double)(g_state >> 11) * (1.0 / 9007199254740992.0); }

struct geom_t { std::vector<float> icoord, jcoord; double hit_lanes, hit_cols, cols; };

static float d2f(float xi, float yi, float zi, float xj, float yj, float zj) {
    const float dx = xj - xi, dy = yj - yi, dz = zj - zi;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// today's chunk: 64 i atoms in a 12.67 x 12.67 cross-section over `len` A of pencil, j atoms anywhere in the box window
static geom_t geom_pk(int nblk, float r, float cross, float len) {
    geom_t g; g.icoord.resize(192); g.jcoord.resize((size_t)nblk * 48);
    for (int l = 0; l < 64; ++l) { g.icoord[l] = (float)(100.0 + urand() * len); g.icoord[64 + l] = (float)(100.0 + urand() * cross); g.icoord[128 + l] = (float)(100.0 + urand() * cross); }
    double hl = 0, hc = 0;
    for (int b = 0; b < nblk; ++b) for (int c = 0; c < 16; ++c) {
        const float xj = (float)(100.0 - r + urand() * (len + 2 * r)), yj = (float)(100.0 - r + urand() * (cross + 2 * r)), zj = (float)(100.0 - r + urand() * (cross + 2 * r));
        g.jcoord[48 * b + c] = xj; g.jcoord[48 * b + 16 + c] = yj; g.jcoord[48 * b + 32 + c] = zj;
        int h = 0;
        for (int l = 0; l < 64; ++l) h += d2f(g.icoord[l], g.icoord[64 + l], g.icoord[128 + l], xj, yj, zj) < r * r;
        hl += h; hc += h > 0;
    }
    g.cols = 16.0 * nblk; g.hit_lanes = hl / g.cols; g.hit_cols = hc / g.cols;
    return g;
}

// sub-wave: per row a tile of 16 i atoms in a box ax x ay x az, j atoms in its window, kept only within `trim` of the tile box
static geom_t geom_dpp(int nblk, float r, float ax, float ay, float az, float trim) {
    geom_t g; g.icoord.resize(192); g.jcoord.resize((size_t)nblk * 192);
    float ox[4], oy[4], oz[4];
    for (int row = 0; row < 4; ++row) {
        ox[row] = (float)(50.0 + 40.0 * row); oy[row] = (float)(60.0 + 7.0 * row); oz[row] = (float)(80.0 - 9.0 * row);
        for (int c = 0; c < 16; ++c) { const int l = 16 * row + c; g.icoord[l] = (float)(ox[row] + urand() * ax); g.icoord[64 + l] = (float)(oy[row] + urand() * ay); g.icoord[128 + l] = (float)(oz[row] + urand() * az); }
    }
    double hl = 0, hc = 0;
    for (int b = 0; b < nblk; ++b) for (int c = 0; c < 16; ++c) {
        int h = 0;
        for (int row = 0; row < 4; ++row) {
            float xj, yj, zj;
            for (;;) {
                xj = (float)(ox[row] - r + urand() * (ax + 2 * r)); yj = (float)(oy[row] - r + urand() * (ay + 2 * r)); zj = (float)(oz[row] - r + urand() * (az + 2 * r));
                const float gx = fmaxf(fmaxf(ox[row] - xj, xj - (ox[row] + ax)), 0.0f), gy = fmaxf(fmaxf(oy[row] - yj, yj - (oy[row] + ay)), 0.0f), gz = fmaxf(fmaxf(oz[row] - zj, zj - (oz[row] + az)), 0.0f);
                if (trim <= 0.0f || gx * gx + gy * gy + gz * gz <= trim * trim) break;
            }
            const int l = 16 * row + c;
            g.jcoord[192 * b + l] = xj; g.jcoord[192 * b + 64 + l] = yj; g.jcoord[192 * b + 128 + l] = zj;
            for (int ci = 0; ci < 16; ++ci) { const int li = 16 * row + ci; h += d2f(g.icoord[li], g.icoord[64 + li], g.icoord[128 + li], xj, yj, zj) < r * r; }
        }
        hl += h; hc += h > 0;
    }
    g.cols = 16.0 * nblk; g.hit_lanes = hl / g.cols; g.hit_cols = hc / g.cols;
    return g;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct result_t { double ms, clock_ghz, ps_col, cyc_col, hits_col; };

template <int MODE>
static result_t run(const geom_t& g, int nblk, int trips, int blocks, float r) {
    float *d_i, *d_j; uint64_t* d_t; unsigned long long* d_h; unsigned* d_s;
    CK(hipMalloc((void**)&d_i, g.icoord.size() * 4)); CK(hipMalloc((void**)&d_j, g.jcoord.size() * 4 + 256));
    CK(hipMalloc((void**)&d_t, 8 * 4 * blocks)); CK(hipMalloc((void**)&d_h, 8 * 4 * blocks)); CK(hipMalloc((void**)&d_s, 64));
    CK(hipMemcpy(d_i, g.icoord.data(), g.icoord.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_j, g.jcoord.data(), g.jcoord.size() * 4, hipMemcpyHostToDevice));
    params_t p;
    p.icoord = d_i; p.jcoord = d_j; p.nblk = nblk; p.trips = trips; p.r2 = r * r;
    // bins of width r / 1024 from 0: t = sqrt(d2) * k + c; `half` wide open so that every hit is "sure" (the parking is left out)
    p.fast_k = 1024.0f / r; p.fast_c = 0.0f; p.half = 0.75f; p.ticks = d_t; p.hits = d_h; p.sink = d_s;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    params_t pw = p; pw.trips = std::max(1, trips / 16);
    hipLaunchKernelGGL(k_cols<MODE>, dim3(blocks), dim3(256), 0, 0, pw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_cols<MODE>, dim3(blocks), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> t(4 * blocks); std::vector<unsigned long long> h(4 * blocks);
    CK(hipMemcpy(t.data(), d_t, 8 * 4 * blocks, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), d_h, 8 * 4 * blocks, hipMemcpyDeviceToHost));
    uint64_t tmax = 0; double hsum = 0;
    for (int i = 0; i < 4 * blocks; ++i) { tmax = std::max(tmax, t[i]); hsum += (double)h[i]; }
    result_t r_;
    r_.ms = ms; r_.clock_ghz = (double)tmax / (ms * 1.0e6);
    const bool stream = MODE == M_SUB || MODE == M_SUB_DPP;
    const double units_wave = stream ? 64.0 * trips : 16.0 * trips;     // instructions / columns per wave
    const double units = units_wave * 4.0 * blocks;
    r_.ps_col = ms * 1.0e9 / units;                                       // chip-level picoseconds per wave-instruction / column
    r_.cyc_col = (double)tmax / (units_wave * (blocks / 256.0));          // SIMD cycles per unit at this occupancy (256 CUs)
    r_.hits_col = stream ? 0.0 : hsum / units;
    CK(hipFree(d_i)); CK(hipFree(d_j)); CK(hipFree(d_t)); CK(hipFree(d_h)); CK(hipFree(d_s));
    return r_;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    const float r = 12.0f;
    const int nblk = 64;            // 1024 columns of geometry, cycled
    printf("# device %s, %d CUs; r = %.1f A; %d j blocks of 16 columns cycled; occupancy = blocks of 4 waves per CU\n", prop.name, ncu, r, nblk);
    printf("# streams (64 instructions per trip)\n");
    printf("%-44s %4s %10s %12s %9s\n", "kind", "occ", "cyc/inst", "Ginst/s chip", "clock GHz");
    geom_t g0 = geom_pk(nblk, r, 12.67f, 12.0f);
    for (int occ : {2, 4, 7, 8}) {
        result_t a = run<M_SUB>(g0, nblk, 4000, ncu * occ, r);
        printf("%-44s %4d %10.3f %12.1f %9.3f\n", "v_sub_f32 v,v,v", occ, a.cyc_col, 1000.0 / a.ps_col, a.clock_ghz);
        result_t b = run<M_SUB_DPP>(g0, nblk, 4000, ncu * occ, r);
        printf("%-44s %4d %10.3f %12.1f %9.3f\n", "v_sub_f32_dpp v,v,v row_newbcast:k", occ, b.cyc_col, 1000.0 / b.ps_col, b.clock_ghz);
    }
    printf("# columns: filter + hand-scheduled push + pop, geometry-driven hits.  R = candidate lanes per hit lane; ps = chip-level\n");
    printf("%-58s %4s %7s %8s %8s %9s %9s %9s %9s %7s\n", "kind", "occ", "R", "hits/col", "hitcols", "cyc/col", "ps/col", "ps/hit", "hits(dev)", "GHz");
    const int trips = 3000;
    for (int occ : {4, 7}) {
        for (int rep = 0; rep < 2; ++rep) {
            geom_t g = geom_pk(nblk, r, 12.67f, rep == 0 ? 12.0f : 12.0f);
            result_t a = run<M_PK>(g, nblk, trips, ncu * occ, r);
            printf("%-58s %4d %7.2f %8.2f %8.3f %9.2f %9.3f %9.4f %9.2f %7.3f\n", "PK  64 i x s_load j, v_pk filter (today)", occ, 64.0 / g.hit_lanes, g.hit_lanes, g.hit_cols,
                   a.cyc_col, a.ps_col, a.ps_col / g.hit_lanes, a.hits_col, a.clock_ghz);
        }
        struct { const char* name; float ax, ay, az, trim; } tiles[] = {
            {"DPP 16 i x 4 rows, 7.8 A cube, box window", 7.8f, 7.8f, 7.8f, 0.0f},
            {"DPP 16 i x 4 rows, 7.8 A cube, window trimmed to r+3", 7.8f, 7.8f, 7.8f, 15.0f},
            {"DPP 16 i x 4 rows, 7.8 A cube, Minkowski (trim = r)", 7.8f, 7.8f, 7.8f, 12.0f},
            {"DPP 16 i x 4 rows, 3 x 12.67 x 12.67 slab (today's sort)", 3.0f, 12.67f, 12.67f, 0.0f},
            {"DPP 16 i x 4 rows, 6.2 x 6.15 x 12.67 half pencil", 6.2f, 6.15f, 12.67f, 0.0f},
        };
        for (auto& tl : tiles) {
            geom_t g = geom_dpp(nblk, r, tl.ax, tl.ay, tl.az, tl.trim);
            result_t a = run<M_DPP>(g, nblk, trips, ncu * occ, r);
            printf("%-58s %4d %7.2f %8.2f %8.3f %9.2f %9.3f %9.4f %9.2f %7.3f\n", tl.name, occ, 64.0 / g.hit_lanes, g.hit_lanes, g.hit_cols,
                   a.cyc_col, a.ps_col, a.ps_col / g.hit_lanes, a.hits_col, a.clock_ghz);
            if (&tl == &tiles[0]) {
                result_t b = run<M_DPP_NOLOAD>(g, nblk, trips, ncu * occ, r);
                printf("%-58s %4d %7.2f %8.2f %8.3f %9.2f %9.3f %9.4f %9.2f %7.3f\n", "    same, j block kept in registers (no loads)", occ, 64.0 / g.hit_lanes, g.hit_lanes, g.hit_cols,
                       b.cyc_col, b.ps_col, b.ps_col / g.hit_lanes, b.hits_col, b.clock_ghz);
                result_t c = run<M_VV>(g, nblk, trips, ncu * occ, r);
                printf("%-58s %4d %7s %8s %8s %9.2f %9.3f %9s %9.2f %7.3f\n", "    VV: same filter without the DPP modifier (other hits)", occ, "-", "-", "-",
                       c.cyc_col, c.ps_col, "-", c.hits_col, c.clock_ghz);
            }
        }
    }
    return 0;
}
