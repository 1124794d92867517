// scripts/valu_calib.hip — issue-rate calibration of the gfx950 vector ALU for the instruction mix of k_rdf_pencil.
//
// bench.py priced the pair kernel against "one wave64 VALU instruction per 4 cycles per SIMD" while
// /opt/skills/guides/MI355X_MICROARCH.md lists v_fma_f32 (wave64) at 2 cycles on a SIMD-32.  This program measures it:
// every kernel is a dependence-free stream of ONE instruction kind (8 independent accumulators, 64 instructions per loop
// trip), run at 1 / 2 / 4 / 8 waves per SIMD on every CU.  Reported per kind and occupancy:
//   cyc/inst (wave)  = s_memtime ticks of one wave / instructions it issued              (issue interval seen by ONE wave)
//   cyc/inst (SIMD)  = wall time x clock / instructions issued per SIMD                   (throughput of the SIMD)
// with clock = s_memtime ticks of the longest wave / wall time (the chip clocks down under dense VALU streams: compare the
// Ginst/s column, which needs no clock).  The "filter" kinds are the 4-column candidate filter of vmd_segment_loop in its packed
// (v_pk_*) and unpacked form, the "push" kind adds the hit compaction; hipcc puts an s_nop between most of their asm statements
// (it cannot see into them), so those rows are LOWER bounds of what the hand-scheduled kernel code reaches.
//
// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib scripts/valu_calib.hip && /tmp/valu_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((vector_size(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Kind {
    K_FMA = 0, K_PK_FMA, K_PK_ADD, K_PK_MUL, K_PK_ADD_S, K_ADD_S, K_CMP, K_MBCNT, K_LSHL_ADD, K_SQRT, K_CVT_FLR, K_FRACT,
    K_SALU, K_VALU_SALU, K_FILTER_PK, K_FILTER_UNPK, K_FILTER_PK_SHIFT, K_PUSH, K_DS_WRITE, K_DS_ADD, K_SUB_VVV, K_MUL_VVV, K_FMA_S, K_CMP_VV, K_CMP_VV_SDST, K_LSHLREV_VV, K_LSHL_ADD_VV, K_MBCNT_V, K_COUNT
};
static const char* kind_name[K_COUNT] = {
    "v_fma_f32 v,v,v,v", "v_pk_fma_f32", "v_pk_add_f32 v,v,v", "v_pk_mul_f32 v,v,v", "v_pk_add_f32 v,v,s[2]", "v_add_f32 v,s,v",
    "v_cmp_gt_f32 vcc,s,v", "v_mbcnt_lo+hi (2 inst)", "v_lshl_add_u32", "v_sqrt_f32", "v_cvt_flr_i32_f32", "v_fract_f32",
    "s_add_u32 (SALU only)", "v_fma_f32 + s_add_u32 alternating (per pair)",
    "filter 4 col packed (6 pk + 4 cmp = 10 inst)", "filter 4 col unpacked (12 + 4 cmp = 16 inst)",
    "filter 4 col packed + image shift (9 pk + 4 cmp = 13 inst)",
    "filter 4 col packed + push, ~14 % lanes hit (10 + 4x(3 VALU + ds_write + 4 SALU))", "ds_write_b32 (stride 4)", "ds_add_u32 (random bins)",
    "v_sub_f32 v,v,v", "v_mul_f32 v,v,v", "v_fma_f32 v,s,v,v",
    "v_cmp_gt_f32 vcc,v,v", "v_cmp_gt_f32 s[2],v,v", "v_lshlrev_b32 v,2,v", "v_lshl_add_u32 v,v,2,v", "v_mbcnt_lo+hi with VGPR masks (2 inst)"};
// instructions counted per unrolled unit (for cyc/inst)
static const int kind_insts[K_COUNT] = {1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 10, 16, 13, 10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2};

template <int KIND>
__global__ __launch_bounds__(256) void k_stream(int iters, uint64_t* ticks, float* sink, float sj0, float sj1, float sj2, float sj3) {
    __shared__ float s_q[4][1024];
    float a0 = threadIdx.x * 1.0e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0000001f, d = 1.0e-9f;
    const f2 pc = {c, c}, pd = {d, d};
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned qbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&s_q[wave][0];
    unsigned qtop = qbase;
    const unsigned lane4 = qbase + 4u * lane;
    // candidate filter operands: i atom per lane, j atoms wave-uniform (SGPRs), cutoff chosen for ~14 % hits
    const float xi = (lane & 7) * 1.7f, yi = ((lane >> 3) & 7) * 1.7f, zi = lane * 0.11f;
    const f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
    const f2 xjA = {sj0, sj1}, xjB = {sj2, sj3}, yjA = {sj1, sj2}, yjB = {sj3, sj0}, zjA = {sj2, sj0}, zjB = {sj1, sj3};
    const f2 sh = {sj3 * 0.01f, sj3 * 0.01f};
    const float r2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sj0 * sj0 * 6.0f + 30.0f)));
    float acc = 0.0f;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##n) : "v"(c), "v"(d));
            REP64(X)
#undef X
        } else if (KIND == K_PK_FMA) {
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(pc), "v"(pd));
            REP64(X)
#undef X
        } else if (KIND == K_PK_ADD) {
#define X(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "v"(pd));
            REP64(X)
#undef X
        } else if (KIND == K_PK_MUL) {
#define X(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(pc));
            REP64(X)
#undef X
        } else if (KIND == K_PK_ADD_S) {
#define X(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "s"(xjA));
            REP64(X)
#undef X
        } else if (KIND == K_ADD_S) {
#define X(n) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a##n) : "s"(sj0));
            REP64(X)
#undef X
        } else if (KIND == K_CMP) {
#define X(n) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(a##n), "s"(sj0) : "vcc");
            REP64(X)
#undef X
        } else if (KIND == K_MBCNT) {
#define X(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "=&v"(i##n) : "s"(s0), "s"(s1));
            REP64(X)
#undef X
        } else if (KIND == K_LSHL_ADD) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i##n) : "s"(s0));
            REP64(X)
#undef X
        } else if (KIND == K_SQRT) {
#define X(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##n));
            REP64(X)
#undef X
        } else if (KIND == K_CVT_FLR) {
#define X(n) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(i##n) : "v"(a##n));
            REP64(X)
#undef X
        } else if (KIND == K_FRACT) {
#define X(n) asm volatile("v_fract_f32 %0, %0" : "+v"(a##n));
            REP64(X)
#undef X
        } else if (KIND == K_SALU) {
#define X(n) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s##n) : : "scc");
            REP64(X)
#undef X
        } else if (KIND == K_VALU_SALU) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 3" : "+v"(a##n), "+s"(s##n) : "v"(c), "v"(d) : "scc");
            REP64(X)
#undef X
        } else if (KIND == K_FILTER_PK || KIND == K_FILTER_PK_SHIFT || KIND == K_PUSH) {
            // 8 groups of 4 columns per trip
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f2 dxA, dyA, dzA, dxB, dyB, dzB, qA, qB;
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dxA) : "v"(xi2), "s"(xjA));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dyA) : "v"(yi2), "s"(yjA));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dzA) : "v"(zi2), "s"(zjA));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dxB) : "v"(xi2), "s"(xjB));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dyB) : "v"(yi2), "s"(yjB));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dzB) : "v"(zi2), "s"(zjB));
                if (KIND == K_FILTER_PK_SHIFT) {
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dxA) : "v"(sh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dyA) : "v"(sh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dzA) : "v"(sh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dxB) : "v"(sh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dyB) : "v"(sh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dzB) : "v"(sh));
                }
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(qA) : "v"(dxA));
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(qB) : "v"(dxB));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(qA) : "v"(dyA));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(qB) : "v"(dyB));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(qA) : "v"(dzA));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(qB) : "v"(dzB));
                if (KIND == K_PUSH) {
                    // the hand-scheduled 4-column push of vmd_push_hot4 (compare, prefix, masked ds_write, stack advance)
                    unsigned t, n;
                    const float d0 = qA[0], d1 = qA[1], d2 = qB[0], d3 = qB[1];
                    asm volatile(
                        "s_nop 0\n\t"
                        "v_cmp_gt_f32 vcc, %[r2], %[d0]\n\t"
                        "s_cbranch_vccz .Lc0_%=\n\t"
                        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
                        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d0]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
                        ".Lc0_%=:\n\t"
                        "v_cmp_gt_f32 vcc, %[r2], %[d1]\n\t"
                        "s_cbranch_vccz .Lc1_%=\n\t"
                        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
                        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d1]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
                        ".Lc1_%=:\n\t"
                        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
                        "s_cbranch_vccz .Lc2_%=\n\t"
                        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
                        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d2]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
                        ".Lc2_%=:\n\t"
                        "v_cmp_gt_f32 vcc, %[r2], %[d3]\n\t"
                        "s_cbranch_vccz .Lc3_%=\n\t"
                        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\tv_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
                        "s_mov_b64 exec, vcc\n\tds_write_b32 %[t], %[d3]\n\ts_mov_b64 exec, -1\n\ts_bcnt1_i32_b64 %[n], vcc\n\ts_lshl2_add_u32 %[q], %[n], %[q]\n"
                        ".Lc3_%=:\n\t"
                        : [q] "+s"(qtop), [t] "=&v"(t), [n] "=&s"(n)
                        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [r2] "s"(r2)
                        : "vcc", "scc", "memory");
                    if (qtop - qbase >= 4u * 512u) qtop = qbase;     // stand-in for the pop: keep the stack inside its buffer
                } else {
                    asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(qA[0]), "s"(r2) : "vcc");
                    asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(qA[1]), "s"(r2) : "vcc");
                    asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(qB[0]), "s"(r2) : "vcc");
                    asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(qB[1]), "s"(r2) : "vcc");
                }
                acc += qA[0] + qB[1];
            }
        } else if (KIND == K_FILTER_UNPK) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float q[4];
                const float xj[4] = {sj0, sj1, sj2, sj3}, yj[4] = {sj1, sj2, sj3, sj0}, zj[4] = {sj2, sj0, sj1, sj3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float dx, dy, dz;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(xi), "s"(xj[k]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(yi), "s"(yj[k]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(zi), "s"(zj[k]));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(q[k]) : "v"(dx));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(q[k]) : "v"(dy));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(q[k]) : "v"(dz));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(q[k]), "s"(r2) : "vcc");
                acc += q[0] + q[3];
            }
        } else if (KIND == K_SUB_VVV) {
#define X(n) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##n) : "v"(d));
            REP64(X)
#undef X
        } else if (KIND == K_MUL_VVV) {
#define X(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##n) : "v"(c));
            REP64(X)
#undef X
        } else if (KIND == K_FMA_S) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##n) : "s"(sj0), "v"(d));
            REP64(X)
#undef X
        } else if (KIND == K_CMP_VV) {
#define X(n) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(a##n), "v"(c) : "vcc");
            REP64(X)
#undef X
        } else if (KIND == K_CMP_VV_SDST) {
            unsigned long long m;
#define X(n) asm volatile("v_cmp_gt_f32 %0, %2, %1" : "=s"(m) : "v"(a##n), "v"(c));
            REP64(X)
#undef X
            s0 += (unsigned)m;
        } else if (KIND == K_LSHLREV_VV) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(i##n));
            REP64(X)
#undef X
        } else if (KIND == K_LSHL_ADD_VV) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i##n) : "v"(i0));
            REP64(X)
#undef X
        } else if (KIND == K_MBCNT_V) {
#define X(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "=&v"(i##n) : "v"(i0), "v"(i1));
            REP64(X)
#undef X
        } else if (KIND == K_DS_WRITE) {
#define X(n) asm volatile("ds_write_b32 %0, %1" : : "v"(lane4), "v"(a##n) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == K_DS_ADD) {
            const unsigned addr = qbase + 4u * ((lane * 37u + it * 11u) & 1023u);
#define X(n) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(i##n) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
    acc += (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)qtop + s_q[wave][lane];
    if (acc == 1.2345f) sink[0] = acc;
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}

typedef void (*kern_t)(int, uint64_t*, float*, float, float, float, float);
template <int K> static kern_t get() { return k_stream<K>; }
static kern_t kern_of(int k) {
    switch (k) {
#define C(K) case K: return get<K>();
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27)
#undef C
    }
    return nullptr;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %.0f MHz (reported)\n", prop.name, ncu, prop.clockRate / 1000.0);
    printf("# one trip = 64 units; cyc/inst(wave) from s_memtime of one wave; cyc/inst(SIMD) = SIMD-cycles per instruction at that occupancy\n");
    printf("%-82s %5s %10s %10s %10s %9s\n", "kind", "w/SIMD", "cyc/i wave", "cyc/i SIMD", "Ginst/s/chip", "clock GHz");
    uint64_t* d_ticks; float* d_sink;
    const int max_blocks = ncu * 8;
    (void)hipMalloc((void**)&d_ticks, sizeof(uint64_t) * 4 * max_blocks);
    (void)hipMalloc((void**)&d_sink, 64);
    std::vector<uint64_t> ticks(4 * max_blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int k = 0; k < K_COUNT; ++k) {
        kern_t fn = kern_of(k);
        for (int occ : {1, 2, 4, 8}) {
            const int blocks = ncu * occ;      // 256-thread blocks: 4 waves = one per SIMD; occ blocks per CU
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, 200, d_ticks, d_sink, 1.5f, 3.25f, 5.0f, 7.75f);   // warm up
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, iters, d_ticks, d_sink, 1.5f, 3.25f, 5.0f, 7.75f);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "kernel %d failed\n", k); return 2; }
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(ticks.data(), d_ticks, sizeof(uint64_t) * 4 * blocks, hipMemcpyDeviceToHost);
            uint64_t tmax = 0; double tsum = 0.0;
            for (int i = 0; i < 4 * blocks; ++i) { tmax = std::max(tmax, ticks[i]); tsum += (double)ticks[i]; }
            const double tavg = tsum / (4.0 * blocks);
            const double units = 64.0 * iters * ((k >= K_FILTER_PK && k <= K_PUSH) ? 8.0 / 64.0 : 1.0);   // filter kinds: 8 groups per trip
            const double insts_wave = units * kind_insts[k];
            const double clock_ghz = (double)tmax / (ms * 1.0e6);
            const double cyc_wave = tavg / insts_wave;
            const double cyc_simd = (double)tmax / (insts_wave * occ);
            const double ginst = insts_wave * 4.0 * blocks / (ms * 1.0e6);
            printf("%-82s %5d %10.3f %10.3f %10.1f %9.3f\n", kind_name[k], occ, cyc_wave, cyc_simd, ginst, clock_ghz);
        }
    }
    return 0;
}
