// viamd_amd/csrc/vmd_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the RDF / SDF / distance
// hot path + the thin C-ABI launch layer declared in include/vmd_hip.h.
//
// Arithmetic contract: oracle/SPEC.md (S2 wrap, S3 pair displacement, S4 binning, S5 alignment/scatter,
// S6 distances).  This file is compiled with -ffp-contract=off: the only fused operations are the explicit
// fmaf() calls the spec names, so integer results are bit-identical to the CPU restatement.
//
// Reference functions replaced (sources live in the empty submodule ext/mdlib):
//   md_spatial_hash build/query  -> k_cells_bin_sorted / k_cells_pen_scan / k_cells_pen_sort (two-level build; single-level builds
//                                   k_cells_fused, k_cells_split_*, k_cells_count / _scan / _scatter for what it does not take) + the
//                                   segment walk of k_rdf_pencil
//   rdf()                        -> k_rdf_pencil (periodic, grid) / k_rdf_brute (general)
//   sdf() + density volume       -> k_sdf_align (fp64 Horn/Jacobi) + k_sdf_scatter
//   distance*()                  -> k_distance_com / k_distance_minmax / k_distance_pair
//
// Design notes (DESIGN.md has the long form):
//  * wave64 everywhere; a wave is the unit of work in the pair kernel (private LDS histogram + private LDS
//    hit queue per wave, no block barrier in the hot loop).
//  * i atoms live in lanes, j atoms are wave-uniform: their coordinates come through the scalar cache
//    (s_load) and feed VALU ops as SGPR operands, so the candidate filter is 7 VALU ops per pair.
//  * hits are compacted through the per-wave LDS queue so that sqrt + binning + ds_add run on full waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "vmd_hip.h"

#define VMD_WAVE 64
#define VMD_MAX_BINS 1024
#define VMD_QUEUE_CAP 384          // floats: < 64 pending + 4 undrained candidate columns of 64 (variant 2: < 64 + 64 pair entries in two
                                   // planes of 128 floats, drained after every pair-column), + the 64-entry slow stack
#define VMD_PAIR_B 128             // variant 2: float offset of the second value of a pair entry (at most 64 + 64 entries are ever pending)
#define VMD_JUNK 3.0e38f           // partner value of a pair entry that must never be counted: far beyond any r_max
#define VMD_FAR 1.0e18f            // coordinate of a padding lane: never within any cutoff, squares stay finite

// Wave-uniform read-only data (j coordinates, cell offsets, boxes) is read through the constant address space so
// that hipcc emits s_load (scalar cache, SGPR operands) instead of per-lane global_load.  Legal because those
// arrays are only written by earlier kernels.  (The SIMT emulator under tests/emu defines this to nothing.)
// wave-wide predicate mask straight from the compare (HIP's __ballot goes through a 0/1 VGPR and a second v_cmp)
#ifndef VMD_BALLOT
#define VMD_BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)
#endif
// floor-to-int and fractional part in one instruction each (v_cvt_flr_i32_f32, v_fract_f32)
#ifndef VMD_NO_INLINE_ASM
__device__ __forceinline__ int vmd_floor_to_int(float t) { int b; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(b) : "v"(t)); return b; }
__device__ __forceinline__ float vmd_fract(float t) { return __builtin_amdgcn_fractf(t); }
// keeps a wave-uniform value in a VGPR (a VOP3 instruction takes only one SGPR operand; without this the compiler
// re-materialises the second one with a v_mov in the inner loop)
__device__ __forceinline__ float vmd_in_vgpr(float v) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; }
#else
__device__ __forceinline__ int vmd_floor_to_int(float t) { return (int)floorf(t); }
__device__ __forceinline__ float vmd_fract(float t) { return t - floorf(t); }
__device__ __forceinline__ float vmd_in_vgpr(float v) { return v; }
#endif
#ifndef VMD_UNIFORM_AS
#define VMD_UNIFORM_AS __attribute__((address_space(4)))
#endif
typedef VMD_UNIFORM_AS const float vmd_cf32;
typedef VMD_UNIFORM_AS const uint32_t vmd_cu32;
typedef float vmd_f2 __attribute__((vector_size(8)));   // two fp32 lanes of one VGPR/SGPR pair (v_pk_*_f32)
typedef float vmd_f4 __attribute__((vector_size(16), aligned(4)));
// four consecutive wave-uniform floats at byte offset `off` (32-bit: s_load_dwordx4 sdst, sbase, soffset)
__device__ __forceinline__ vmd_f4 vmd_uniform_load4(vmd_cf32* base, unsigned off) {
    return *(VMD_UNIFORM_AS const vmd_f4*)((VMD_UNIFORM_AS const char*)base + off);
}

// ------------------------------------------------------------------------------------------------ helpers

// SPEC S2
// invL = fl(1.0f / L), computed once per frame on the host (IEEE division, identical to the device's)
__device__ __forceinline__ float vmd_wrap(float x, float L, float invL) {
    const float t = x * invL;
    const float f = floorf(t);
    float xw = fmaf(-f, L, x);
    if (xw < 0.0f) xw = xw + L;
    if (xw >= L) xw = xw - L;
    return xw;
}

__device__ __forceinline__ float vmd_d2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

// SPEC S3 by comparison (general kernels)
__device__ __forceinline__ float vmd_mi_cmp(float d, float L, float hL, bool pbc) {
    if (pbc) {
        const float s = d > hL ? L : (d < -hL ? -L : 0.0f);
        d = d - s;
    }
    return d;
}

// SPEC S5/S6 fp32 minimum image by rint
__device__ __forceinline__ float vmd_mi_rintf(float d, float L, float invL, bool pbc) {
    if (pbc) d = fmaf(-rintf(d * invL), L, d);
    return d;
}

__device__ __forceinline__ double vmd_mi_rint(double d, double L, bool pbc) {
    if (pbc) d = d - L * rint(d / L);
    return d;
}

// One frame's unit cell as the kernels see it.  boxes[b*9 + ...] = {Lx,Ly,Lz, 1/Lx,1/Ly,1/Lz, xy,xz,yz}; pbc bit 3 = triclinic
// (basis a = (x,0,0), b = (xy,y,0), c = (xz,yz,z), all three axes periodic; SPEC S3t).
#define VMD_BOX_STRIDE 9
#define VMD_PBC_TRICLINIC 8u
struct vmd_box_t {
    float Lx, Ly, Lz, iLx, iLy, iLz, xy, xz, yz;
    bool px, py, pz, tri;
};
__device__ __forceinline__ vmd_box_t vmd_load_box(const float* boxes, int b, uint32_t pbc) {
    const float* q = boxes + (size_t)VMD_BOX_STRIDE * b;
    vmd_box_t bx;
    bx.Lx = q[0]; bx.Ly = q[1]; bx.Lz = q[2]; bx.iLx = q[3]; bx.iLy = q[4]; bx.iLz = q[5]; bx.xy = q[6]; bx.xz = q[7]; bx.yz = q[8];
    bx.px = pbc & 1u; bx.py = pbc & 2u; bx.pz = pbc & 4u; bx.tri = pbc & VMD_PBC_TRICLINIC;
    return bx;
}
// SPEC S3t: Cartesian -> fractional and back, fp32
__device__ __forceinline__ void vmd_frac(const vmd_box_t& b, float x, float y, float z, float& sx, float& sy, float& sz) {
    sz = z * b.iLz;
    sy = fmaf(-b.yz, sz, y) * b.iLy;
    sx = fmaf(-b.xz, sz, fmaf(-b.xy, sy, x)) * b.iLx;
}
__device__ __forceinline__ void vmd_cart(const vmd_box_t& b, float sx, float sy, float sz, float& dx, float& dy, float& dz) {
    dz = sz * b.Lz;
    dy = fmaf(b.yz, sz, sy * b.Ly);
    dx = fmaf(b.xz, sz, fmaf(b.xy, sy, sx * b.Lx));
}
// SPEC S3t wrap: fractional coordinates folded into [0,1), back to Cartesian; (ux,uy,uz) = s_k * L_k are the unsheared
// coordinates the cell grid bins by
__device__ __forceinline__ void vmd_wrap_tri(const vmd_box_t& b, float x, float y, float z, float& xw, float& yw, float& zw,
                                             float& ux, float& uy, float& uz) {
    float sx, sy, sz;
    vmd_frac(b, x, y, z, sx, sy, sz);
    sx = sx - floorf(sx); if (!(sx < 1.0f)) sx = 0.0f;
    sy = sy - floorf(sy); if (!(sy < 1.0f)) sy = 0.0f;
    sz = sz - floorf(sz); if (!(sz < 1.0f)) sz = 0.0f;
    ux = sx * b.Lx; uy = sy * b.Ly; uz = sz * b.Lz;
    zw = uz;
    yw = fmaf(b.yz, sz, uy);
    xw = fmaf(b.xz, sz, fmaf(b.xy, sy, ux));
}
// SPEC S3t lattice vector n = (nx, ny, nz) in Cartesian components
__device__ __forceinline__ void vmd_lattice_shift(float Lx, float Ly, float Lz, float xy, float xz, float yz, float nx, float ny, float nz,
                                                  float& shx, float& shy, float& shz) {
    shx = fmaf(nz, xz, fmaf(ny, xy, nx * Lx));
    shy = fmaf(nz, yz, ny * Ly);
    shz = nz * Lz;
}
// SPEC S3t pair on wrapped Cartesian positions: image by rounding the displacement in fractional space
__device__ __forceinline__ float vmd_pair_d2_tri(const vmd_box_t& b, float xi, float yi, float zi, float xj, float yj, float zj) {
    const float d0x = xi - xj, d0y = yi - yj, d0z = zi - zj;
    float sx, sy, sz;
    vmd_frac(b, d0x, d0y, d0z, sx, sy, sz);
    float shx, shy, shz;
    vmd_lattice_shift(b.Lx, b.Ly, b.Lz, b.xy, b.xz, b.yz, rintf(sx), rintf(sy), rintf(sz), shx, shy, shz);
    return vmd_d2(d0x - shx, d0y - shy, d0z - shz);
}
// the coordinates pair kernels work on: wrapped Cartesian (S2 per axis for orthorhombic / open cells, S3t for triclinic)
__device__ __forceinline__ void vmd_pair_coords(const vmd_box_t& b, float x, float y, float z, float& ox, float& oy, float& oz) {
    if (b.tri) { float ux, uy, uz; vmd_wrap_tri(b, x, y, z, ox, oy, oz, ux, uy, uz); return; }
    ox = b.px ? vmd_wrap(x, b.Lx, b.iLx) : x;
    oy = b.py ? vmd_wrap(y, b.Ly, b.iLy) : y;
    oz = b.pz ? vmd_wrap(z, b.Lz, b.iLz) : z;
}
__device__ __forceinline__ float vmd_pair_d2_general(const vmd_box_t& b, float xi, float yi, float zi, float xj, float yj, float zj) {
    if (b.tri) return vmd_pair_d2_tri(b, xi, yi, zi, xj, yj, zj);
    const float dx = vmd_mi_cmp(xi - xj, b.Lx, 0.5f * b.Lx, b.px);
    const float dy = vmd_mi_cmp(yi - yj, b.Ly, 0.5f * b.Ly, b.py);
    const float dz = vmd_mi_cmp(zi - zj, b.Lz, 0.5f * b.Lz, b.pz);
    return vmd_d2(dx, dy, dz);
}
// SPEC S5/S6 minimum image of a Cartesian displacement by rounding (fp32 / fp64)
__device__ __forceinline__ void vmd_mi3_rintf(const vmd_box_t& b, float& dx, float& dy, float& dz) {
    if (b.tri) {
        float sx, sy, sz;
        vmd_frac(b, dx, dy, dz, sx, sy, sz);
        sx = sx - rintf(sx); sy = sy - rintf(sy); sz = sz - rintf(sz);
        vmd_cart(b, sx, sy, sz, dx, dy, dz);
        return;
    }
    dx = vmd_mi_rintf(dx, b.Lx, b.iLx, b.px);
    dy = vmd_mi_rintf(dy, b.Ly, b.iLy, b.py);
    dz = vmd_mi_rintf(dz, b.Lz, b.iLz, b.pz);
}
__device__ __forceinline__ void vmd_mi3_rint(const vmd_box_t& b, double& dx, double& dy, double& dz) {
    if (b.tri) {
        double sz = dz / (double)b.Lz;
        double sy = (dy - (double)b.yz * sz) / (double)b.Ly;
        double sx = (dx - (double)b.xy * sy - (double)b.xz * sz) / (double)b.Lx;
        sx = sx - rint(sx); sy = sy - rint(sy); sz = sz - rint(sz);
        dz = sz * (double)b.Lz;
        dy = sy * (double)b.Ly + (double)b.yz * sz;
        dx = sx * (double)b.Lx + (double)b.xy * sy + (double)b.xz * sz;
        return;
    }
    dx = vmd_mi_rint(dx, (double)b.Lx, b.px);
    dy = vmd_mi_rint(dy, (double)b.Ly, b.py);
    dz = vmd_mi_rint(dz, (double)b.Lz, b.pz);
}

__device__ __forceinline__ int vmd_cell_coord(float v, float inv, int n) {
    int c = (int)(v * inv);
    c = c < 0 ? 0 : c;
    return c > n - 1 ? n - 1 : c;
}

__device__ __forceinline__ float vmd_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float vmd_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float vmd_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ unsigned vmd_lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// SPEC S4: bin of one squared distance, -1 when outside the open interval
struct vmd_binning_t {
    float rmin, rmax, inv_range, fnbins;
    int nbins;
    // fast path of vmd_bin_add: t' = fma(v_sqrt_f32(d2), fast_k, fast_c) approximates the spec's scaled distance to within
    // fast_delta/4; it is trusted when it lies at least fast_delta away from every integer (bin edge, and through bins 0 and
    // nbins-1 also from r_min / r_max); everything else takes the exact path
    float fast_k, fast_c, fast_half;   // fast_half = 0.5 - fast_delta
    float fast_far;                    // t' >= fast_far: beyond r_max by a whole bin, i.e. certainly not a hit (pair entries carry such partners)
    // the same test with fast_delta folded into the constant (r_min = 0 only, vmd_pop_hot0): t2 = fma(v_sqrt_f32(d2), fast_k, fast_delta) is
    // t' + delta, so "t' at least delta away from every integer" reads  fract(t2) > 2 delta  and floor(t2) is the bin - one compare instead of
    // add + compare, and no range test: t2 > 0 always, and t2 >= nbins (d > r_max, still below the padded cutoff) lands in a spare bin
    float fast_c2, fast_2d;            // fast_delta, 2 * fast_delta; fast_2d >= 1 (nothing is ever sure) when the fold does not apply
    int closed;                        // DECISION(D-RDF-OPEN) flipped: hit iff r_min <= d <= r_max.  Only vmd_bin_of looks at it: the fast path is
                                       // never trusted within delta of a bin edge, and d = r_min / r_max sit exactly on one
};
static thread_local int g_rdf_closed = 0;       // per host thread: set from the eval's spec right before that thread's launches (two evals with different specs on two threads must not race)
extern "C" int vmd_hip_set_rdf_closed(int on) { const int old = g_rdf_closed; g_rdf_closed = on ? 1 : 0; return old; }
static thread_local int g_rdf_raw = 0;          // k_rdf_brute: positions unwrapped, minimum image by rounding (DECISION D-WRAP flipped); per host thread like g_rdf_closed
extern "C" int vmd_hip_set_rdf_raw(int on) { const int old = g_rdf_raw; g_rdf_raw = on ? 1 : 0; return old; }
__host__ __device__ inline vmd_binning_t vmd_make_binning(float rmin, float rmax, int nbins, int closed = 0) {
    vmd_binning_t b;
    b.closed = closed;
    b.rmin = rmin; b.rmax = rmax; b.inv_range = 1.0f / (rmax - rmin); b.fnbins = (float)nbins; b.nbins = nbins;
    // error budget of t' = fma(v_sqrt(d2), k, c) against the spec value ((d - rmin) * inv_range) * nbins, in ulps of
    // T = rmax * inv_range * nbins (the largest intermediate): v_sqrt_f32 1 + fma 0.5 on our side, sqrtf 0.5 + subtraction
    // 0.5 + product 0.5 on the spec's (the factor nbins = 2^k is exact), the rounding of c = -rmin*k 0.5: 3.5 ulp
    // = 4.2e-7 * T.  delta = 6e-7 * T (5 ulp) + 2.5e-4 absolute head room.
    const float delta = 2.5e-4f + 6.0e-7f * rmax * b.inv_range * b.fnbins;
    b.fast_k = b.inv_range * b.fnbins;
    b.fast_c = -rmin * b.fast_k;
    b.fast_half = 0.5f - delta;
    b.fast_far = b.fnbins + 1.0f;                   // |t' - t| <= delta / 4 << 1: t' >= nbins + 1 implies d > r_max
    b.fast_c2 = delta; b.fast_2d = 2.0f * delta;
    if (!(delta < 0.25f)) { b.fast_half = -1.0f; b.fast_far = 3.0e38f; }      // degenerate range: nothing is ever "sure", exact path only
    if (!(delta < 0.25f) || rmin != 0.0f) b.fast_2d = 2.0f;                   // (v_fract_f32 < 1)
    return b;
}
__device__ __forceinline__ int vmd_bin_of(const vmd_binning_t& b, float d2) {
    const float d = sqrtf(d2);
    if (b.closed ? !(b.rmin <= d && d <= b.rmax) : !(b.rmin < d && d < b.rmax)) return -1;
    int bin = (int)(((d - b.rmin) * b.inv_range) * b.fnbins);
    bin = bin < 0 ? 0 : bin;
    return bin > b.nbins - 1 ? b.nbins - 1 : bin;
}

// ------------------------------------------------------------------------------------------------ K1: cell build

// A selection whose index list is periodic - atom(t) = first + (t / m) * period + off[t % m], m <= 4: "the O of every water" (m = 1, period 3),
// "the two H of every water" (m = 2), "every atom" - is not read from memory at all: the list costs 4 bytes per selected atom and frame (1.33 GB
// per 1 000 frames of the 1M-atom heavy-atom RDF, 4.8 % of the cell build's traffic) for something three integers say.  m = 0: read sel[t].
struct vmd_sel_pattern_t { int m, first, period; int off[4]; };
struct vmd_cells_params_t {
    const float* xyz; size_t frame_stride; size_t row_stride;
    const float* boxes; uint32_t pbc; const int32_t* sel; int nsel; int nsel_pad;
    vmd_grid_t grid;
    uint32_t* cell_count; uint32_t* rank; uint32_t* cell_start; float* sorted;
    float* aos;   // optional f32[B][nsel_pad][4] staging: scatter ONE 16-byte record per atom, k_cells_repack makes the SoA rows
    vmd_sel_pattern_t pat;
};

__device__ __forceinline__ int vmd_sel_atom(const vmd_cells_params_t& p, int t) {
    if (p.pat.m == 1) return p.pat.first + t * p.pat.period;
    if (p.pat.m > 1) { const int q = t / p.pat.m, r = t - q * p.pat.m; return p.pat.first + q * p.pat.period + p.pat.off[r]; }
    return p.sel ? p.sel[t] : t;
}

__device__ __forceinline__ uint32_t vmd_cell_of(const vmd_cells_params_t& p, int b, int t, float& xw, float& yw, float& zw) {
    const int a = vmd_sel_atom(p, t);
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const float* bq = p.boxes + (size_t)VMD_BOX_STRIDE * b;
    const float Lx = bq[0], Ly = bq[1], Lz = bq[2], iLx = bq[3], iLy = bq[4], iLz = bq[5];
    float ux, uy, uz;     // what the grid bins by: wrapped coordinate (periodic axis), offset from the batch's bounding box
                          // (open axis; L is then the box extent, slots 6..8 the origin), or s_k * L_k (triclinic cell)
    if (p.pbc & VMD_PBC_TRICLINIC) {
        vmd_box_t bx;
        bx.Lx = Lx; bx.Ly = Ly; bx.Lz = Lz; bx.iLx = iLx; bx.iLy = iLy; bx.iLz = iLz; bx.xy = bq[6]; bx.xz = bq[7]; bx.yz = bq[8];
        bx.px = bx.py = bx.pz = bx.tri = true;
        vmd_wrap_tri(bx, fx[a], fx[p.row_stride + a], fx[2 * p.row_stride + a], xw, yw, zw, ux, uy, uz);
    } else {
        const float x = fx[a], y = fx[p.row_stride + a], z = fx[2 * p.row_stride + a];
        if (p.pbc & 1u) { ux = xw = vmd_wrap(x, Lx, iLx); } else { xw = x; ux = x - bq[6]; }
        if (p.pbc & 2u) { uy = yw = vmd_wrap(y, Ly, iLy); } else { yw = y; uy = y - bq[7]; }
        if (p.pbc & 4u) { uz = zw = vmd_wrap(z, Lz, iLz); } else { zw = z; uz = z - bq[8]; }
    }
    const int cx = vmd_cell_coord(ux, (float)p.grid.nxf * iLx, p.grid.nxf);
    const int cy = vmd_cell_coord(uy, (float)p.grid.ny * iLy, p.grid.ny);
    const int cz = vmd_cell_coord(uz, (float)p.grid.nz * iLz, p.grid.nz);
    return (uint32_t)((cz * p.grid.ny + cy) * p.grid.nxf + cx);
}

// a scattered 4-byte store costs the L2 as much as a scattered 16-byte one: the sort writes ONE float4 record per atom and
// k_cells_repack turns the records into the SoA rows with coalesced traffic
typedef float vmd_f4a __attribute__((vector_size(16)));
__device__ __forceinline__ void vmd_store_aos(float* aos, size_t slot, float x, float y, float z) {
    const vmd_f4a v = {x, y, z, 0.0f};
    *(vmd_f4a*)(aos + 4 * slot) = v;
}
__global__ __launch_bounds__(256) void k_cells_repack(const float* __restrict__ aos, float* __restrict__ sorted, int nsel, int nsel_pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= nsel) return;
    const vmd_f4a v = *(const vmd_f4a*)(aos + 4 * ((size_t)b * nsel_pad + t));
    float* s = sorted + (size_t)b * 3 * nsel_pad;
    s[t] = v[0];
    s[nsel_pad + t] = v[1];
    s[2 * (size_t)nsel_pad + t] = v[2];
}

// atoms per thread.  Measured on the 333k-atom selection of config 3: 4 independent gathers per thread are ~10 % SLOWER than
// 1 (the kernels are bound by scattered 4-byte write / atomic transactions, not by latency), so this stays at 1.
#define VMD_CELLS_ILP 1
__global__ __launch_bounds__(256) void k_cells_count(vmd_cells_params_t p) {
    const int t0 = blockIdx.x * (256 * VMD_CELLS_ILP) + threadIdx.x;
    const int b = blockIdx.y;
    uint32_t c[VMD_CELLS_ILP];
    float xw, yw, zw;
#pragma unroll
    for (int u = 0; u < VMD_CELLS_ILP; ++u) {
        const int t = t0 + 256 * u;
        c[u] = t < p.nsel ? vmd_cell_of(p, b, t, xw, yw, zw) : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < VMD_CELLS_ILP; ++u) {
        const int t = t0 + 256 * u;
        if (c[u] != 0xffffffffu) p.rank[(size_t)b * p.nsel + t] = atomicAdd(&p.cell_count[(size_t)b * (p.grid.ncell + 1) + c[u]], 1u);
    }
}

// one 1024-thread block per frame: exclusive prefix over the cell populations.  Round 6: chunks of 4 096 cells, four CONSECUTIVE cells per
// thread - the previous version gave every thread its own contiguous stretch of ncell / 1024 cells, i.e. 64 lanes reading 64 different cache
// lines per load: 0.80 ms per dispatch for the 30 000 cells x 334 frames of config 5's solute class (profiles/r06z3_*), 2.4 ms of its step
__global__ __launch_bounds__(1024) void k_cells_scan(const uint32_t* __restrict__ cell_count, uint32_t* __restrict__ cell_start,
                                                     int ncell) {
    __shared__ uint32_t part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t* cnt = cell_count + (size_t)b * (ncell + 1);
    uint32_t* out = cell_start + (size_t)b * (ncell + 1);
    uint32_t carry = 0;
    for (int base = 0; base < ncell; base += 4096) {
        const int c0 = base + 4 * tid;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c0 + k < ncell ? cnt[c0 + k] : 0u;
        const uint32_t s = v[0] + v[1] + v[2] + v[3];
        part[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t u = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += u;
            __syncthreads();
        }
        uint32_t run = carry + part[tid] - s;   // exclusive
        const uint32_t total = part[1023];
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (c0 + k < ncell) out[c0 + k] = run; run += v[k]; }
        carry += total;
        __syncthreads();                         // part[] is rewritten by the next chunk
    }
    if (tid == 0) out[ncell] = carry;
}

__global__ __launch_bounds__(256) void k_cells_scatter(vmd_cells_params_t p) {
    const int t0 = blockIdx.x * (256 * VMD_CELLS_ILP) + threadIdx.x;
    const int b = blockIdx.y;
    float xw[VMD_CELLS_ILP], yw[VMD_CELLS_ILP], zw[VMD_CELLS_ILP];
    uint32_t pos[VMD_CELLS_ILP];
#pragma unroll
    for (int u = 0; u < VMD_CELLS_ILP; ++u) {
        const int t = t0 + 256 * u;
        pos[u] = 0xffffffffu;
        if (t < p.nsel) {
            const uint32_t c = vmd_cell_of(p, b, t, xw[u], yw[u], zw[u]);
            pos[u] = p.cell_start[(size_t)b * (p.grid.ncell + 1) + c] + p.rank[(size_t)b * p.nsel + t];
        }
    }
    float* s = p.sorted + (size_t)b * 3 * p.nsel_pad;
#pragma unroll
    for (int u = 0; u < VMD_CELLS_ILP; ++u) {
        if (pos[u] == 0xffffffffu) continue;
        if (p.aos) { vmd_store_aos(p.aos, (size_t)b * p.nsel_pad + pos[u], xw[u], yw[u], zw[u]); continue; }
        s[pos[u]] = xw[u];
        s[p.nsel_pad + pos[u]] = yw[u];
        s[2 * (size_t)p.nsel_pad + pos[u]] = zw[u];
    }
}

// Fused build for grids whose cell table fits in LDS: ONE 1024-thread block per frame does count (LDS atomics) ->
// exclusive scan (in LDS) -> scatter (LDS cursors).  No global atomics, no rank array, no separate scan launch.
// LDS: (ncell + 1) counters + 1024 scan partials.
__global__ __launch_bounds__(1024) void k_cells_fused(vmd_cells_params_t p) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    uint32_t* s_cnt = s_dyn;                       // [ncell + 1]
    uint32_t* s_part = s_dyn + p.grid.ncell + 1;   // [1024]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int ncell = p.grid.ncell;
    for (int c = tid; c <= ncell; c += 1024) s_cnt[c] = 0u;
    __syncthreads();
    float xw, yw, zw;
    for (int t = tid; t < p.nsel; t += 1024) atomicAdd(&s_cnt[vmd_cell_of(p, b, t, xw, yw, zw)], 1u);
    __syncthreads();
    // exclusive scan, same scheme as k_cells_scan
    const int per = (ncell + 1023) / 1024;
    const int beg = tid * per;
    const int end = beg + per < ncell ? beg + per : ncell;
    uint32_t sum = 0;
    for (int c = beg; c < end; ++c) sum += s_cnt[c];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    uint32_t* out = p.cell_start + (size_t)b * (ncell + 1);
    for (int c = beg; c < end; ++c) { const uint32_t n = s_cnt[c]; s_cnt[c] = run; out[c] = run; run += n; }
    if (tid == 1023) out[ncell] = s_part[1023];
    __syncthreads();
    float* srt = p.sorted + (size_t)b * 3 * p.nsel_pad;
    for (int t = tid; t < p.nsel; t += 1024) {
        const uint32_t c = vmd_cell_of(p, b, t, xw, yw, zw);
        const uint32_t pos = atomicAdd(&s_cnt[c], 1u);
        if (p.aos) { vmd_store_aos(p.aos, (size_t)b * p.nsel_pad + pos, xw, yw, zw); continue; }
        srt[pos] = xw;
        srt[p.nsel_pad + pos] = yw;
        srt[2 * (size_t)p.nsel_pad + pos] = zw;
    }
}

// Split build for selections too large for one block per frame: G blocks share a frame, each owns a contiguous slice of the
// selection.  count: LDS histogram of the slice -> global table [frame][g][cell]; scan: per frame, exclusive prefix over
// cells of the column sums + running offset over g (in place: the table becomes each block's first slot per cell);
// scatter: LDS cursors initialised from the table.  No global atomics, no rank array; the only scattered traffic is the
// 16-byte record store.  The order of atoms inside a cell differs from the other builds (it is arbitrary in all of them).
struct vmd_cells_split_t {
    vmd_cells_params_t c;
    uint32_t* table;     // [B][G][ncell]
    int G; int slice;    // atoms per block (multiple of 1024)
};

__global__ __launch_bounds__(1024) void k_cells_split_count(vmd_cells_split_t q) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ncell = q.c.grid.ncell;
    for (int c = tid; c < ncell; c += 1024) s_dyn[c] = 0u;
    __syncthreads();
    const int t1 = (g + 1) * q.slice < q.c.nsel ? (g + 1) * q.slice : q.c.nsel;
    float xw, yw, zw;
    for (int t = g * q.slice + tid; t < t1; t += 1024) atomicAdd(&s_dyn[vmd_cell_of(q.c, b, t, xw, yw, zw)], 1u);
    __syncthreads();
    uint32_t* out = q.table + ((size_t)b * q.G + g) * ncell;
    for (int c = tid; c < ncell; c += 1024) out[c] = s_dyn[c];
}

__global__ __launch_bounds__(1024) void k_cells_split_scan(uint32_t* __restrict__ table, uint32_t* __restrict__ cell_start, int ncell, int G) {
    __shared__ uint32_t part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    uint32_t* tb = table + (size_t)b * G * ncell;
    uint32_t* out = cell_start + (size_t)b * (ncell + 1);
    uint32_t carry = 0;
    // rows of 1024 consecutive cells: thread = cell, so every table access is coalesced
    for (int base = 0; base < ncell; base += 1024) {
        const int c = base + tid;
        uint32_t s = 0;
        if (c < ncell) for (int g = 0; g < G; ++g) s += tb[(size_t)g * ncell + c];
        part[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t v = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (c < ncell) {
            uint32_t run = carry + part[tid] - s;   // exclusive
            out[c] = run;
            for (int g = 0; g < G; ++g) { const uint32_t n = tb[(size_t)g * ncell + c]; tb[(size_t)g * ncell + c] = run; run += n; }
        }
        carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) out[ncell] = carry;
}

__global__ __launch_bounds__(1024) void k_cells_split_scatter(vmd_cells_split_t q) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ncell = q.c.grid.ncell;
    const uint32_t* first = q.table + ((size_t)b * q.G + g) * ncell;
    for (int c = tid; c < ncell; c += 1024) s_dyn[c] = first[c];
    __syncthreads();
    const int t1 = (g + 1) * q.slice < q.c.nsel ? (g + 1) * q.slice : q.c.nsel;
    float* srt = q.c.sorted + (size_t)b * 3 * q.c.nsel_pad;
    float xw, yw, zw;
    for (int t = g * q.slice + tid; t < t1; t += 1024) {
        const uint32_t c = vmd_cell_of(q.c, b, t, xw, yw, zw);
        const uint32_t pos = atomicAdd(&s_dyn[c], 1u);
        if (q.c.aos) { vmd_store_aos(q.c.aos, (size_t)b * q.c.nsel_pad + pos, xw, yw, zw); continue; }
        srt[pos] = xw;
        srt[q.c.nsel_pad + pos] = yw;
        srt[2 * (size_t)q.c.nsel_pad + pos] = zw;
    }
}

// Two-level build (the default): the frame is read ONCE and the sorted SoA rows are written once, with one 16-byte record per
// atom in between - against two reads of the frame, a scattered AoS write and a repack pass in the builds above.
//   level 1, k_cells_bin: a block takes a slice of the selection, wraps the atoms, counts them per PENCIL in LDS (a pencil =
//     one row of nxf fine cells; there are only ny*nz of them, so the table is tiny whatever nxf is), reserves room in every
//     pencil's bucket with one global atomic per (block, pencil) and writes the records {x, y, z, fine cell} there: a block's
//     atoms of one pencil land next to each other, so the scattered writes are short runs, not single records.
//   level 2, k_cells_pen_sort: one block per (frame, pencil) pulls the pencil's bucket through LDS (counting sort by fine cell)
//     and writes its stretch of the sorted rows and of cell_start fully coalesced.
// Buckets have a fixed capacity per pencil (pen_off, measured on a few frames by the host with k_cells_bin in counting mode, plus
// head room); an atom that finds its bucket full raises *overflow, every consumer of the sorted copy (k_rdf_pencil,
// k_hist_reduce) then does nothing, and the host re-measures and repeats the batch.  The order of atoms inside a cell depends on
// the order of the atomics (it is arbitrary in every build; the histograms do not depend on it).
struct vmd_bin_params_t {
    vmd_cells_params_t c;            // frame, boxes, selection, grid (cell_count / rank / aos unused)
    const uint32_t* pen_off;         // [npen + 1] exclusive prefix of the bucket capacities (records), NULL in counting mode
    uint32_t* pen_count;             // [B][npen], zeroed before the launch: atoms per pencil
    float* bucket;                   // [B][pen_off[npen]][4], NULL in counting mode
    uint32_t* overflow;              // [1]
    uint32_t overflow_bit;           // what an overflowing build ORs into it: one bit per selection, so that the host widens the buckets of THAT
                                     // selection only (c5, 1 000 frames: the wandering blob overflowed three times and took the water selections'
                                     // margins - and with them the two-level build - down with it: 49 ms of cell build per step instead of 21)
    int npen; int total_cap;
    int rec3;                        // 1: 12-byte records {x, y, z} (the fine cell follows from x alone: x-periodic, non-triclinic cells)
};
#define VMD_BIN_ILP 4          // atoms per thread: 4096-atom slices (8 per thread: measured slower, profiles/r02d_ab.txt)
__global__ __launch_bounds__(1024) void k_cells_bin(vmd_bin_params_t q) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    uint32_t* s_cnt = s_dyn;                 // [npen] atoms of this block per pencil, then the block's first slot in the bucket
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int npen = q.npen, nxf = q.c.grid.nxf;
    for (int c = tid; c < npen; c += 1024) s_cnt[c] = 0u;
    __syncthreads();
    float xw[VMD_BIN_ILP], yw[VMD_BIN_ILP], zw[VMD_BIN_ILP];
    uint32_t pen[VMD_BIN_ILP], cx[VMD_BIN_ILP], rank[VMD_BIN_ILP];
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) {
        const int t = (g * VMD_BIN_ILP + u) * 1024 + tid;
        pen[u] = 0xffffffffu;
        if (t < q.c.nsel) {
            const uint32_t cell = vmd_cell_of(q.c, b, t, xw[u], yw[u], zw[u]);
            pen[u] = cell / (uint32_t)nxf;
            cx[u] = cell - pen[u] * (uint32_t)nxf;
        }
    }
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) if (pen[u] != 0xffffffffu) rank[u] = atomicAdd(&s_cnt[pen[u]], 1u);
    __syncthreads();
    uint32_t* gcount = q.pen_count + (size_t)b * npen;
    for (int c = tid; c < npen; c += 1024) {
        const uint32_t n = s_cnt[c];
        if (n) s_cnt[c] = atomicAdd(&gcount[c], n);
    }
    if (!q.bucket) return;                   // counting mode: the host only wants the populations
    __syncthreads();
    float* bk = q.bucket + (size_t)b * q.total_cap * 4;
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) {
        if (pen[u] == 0xffffffffu) continue;
        const uint32_t off = q.pen_off[pen[u]], cap = q.pen_off[pen[u] + 1] - off;
        const uint32_t slot = s_cnt[pen[u]] + rank[u];
        if (slot < cap) {
            if (q.rec3) {
                float* r = bk + 3 * (size_t)(off + slot);
                r[0] = xw[u]; r[1] = yw[u]; r[2] = zw[u];
            } else {
                const vmd_f4a v = {xw[u], yw[u], zw[u], __int_as_float((int)cx[u])};
                *(vmd_f4a*)(bk + 4 * (size_t)(off + slot)) = v;
            }
        } else {
            atomicOr(q.overflow, q.overflow_bit);
        }
    }
}

// Level 1 with a block-local sort (the default): k_cells_bin stores every record from the lane that wrapped the atom, i.e. a wave's 64
// records go to ~64 different buckets - one partial-line write transaction each.  Here the block first orders its 4 096 records by
// pencil in LDS (rank inside the pencil from the LDS counter, the pencil's first slot from a block scan of the counters), then
// consecutive lanes write consecutive records: a (block, pencil) run of ~14 records leaves as one or two coalesced stores.
// LDS: 3 tables of npen words + 1 040 scan words + 4 096 records of 4 (12-byte output) or 5 (16-byte output) words.
__global__ __launch_bounds__(1024) void k_cells_bin_sorted(vmd_bin_params_t q) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    const int npen = q.npen, nxf = q.c.grid.nxf;
    const int rw = q.rec3 ? 4 : 5;             // words per LDS record: x, y, z, pencil (, fine cell)
    uint32_t* s_cnt = s_dyn;                   // [npen] atoms of this block per pencil
    uint32_t* s_off = s_dyn + npen;            // [npen] first LDS slot of the pencil's run
    uint32_t* s_gbase = s_dyn + 2 * npen;      // [npen] first slot of the run in the pencil's bucket
    uint32_t* s_p = s_dyn + 3 * npen;          // [1024 + 16] scan partials
    uint32_t* s_rec = s_p + 1040;              // [4096][rw]
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < npen; c += 1024) s_cnt[c] = 0u;
    __syncthreads();
    float xw[VMD_BIN_ILP], yw[VMD_BIN_ILP], zw[VMD_BIN_ILP];
    uint32_t pen[VMD_BIN_ILP], cx[VMD_BIN_ILP], rank[VMD_BIN_ILP];
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) {
        const int t = (g * VMD_BIN_ILP + u) * 1024 + tid;
        pen[u] = 0xffffffffu;
        if (t < q.c.nsel) {
            const uint32_t cell = vmd_cell_of(q.c, b, t, xw[u], yw[u], zw[u]);
            pen[u] = cell / (uint32_t)nxf;
            cx[u] = cell - pen[u] * (uint32_t)nxf;
        }
    }
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) if (pen[u] != 0xffffffffu) rank[u] = atomicAdd(&s_cnt[pen[u]], 1u);
    __syncthreads();
    // reserve the runs in the buckets, and scan the counters: thread t owns the counters [t * per, t * per + per)
    uint32_t* gcount = q.pen_count + (size_t)b * npen;
    const int per = (npen + 1023) / 1024;
    const int cb = tid * per, ce = cb + per < npen ? cb + per : npen;
    uint32_t sum = 0;
    for (int c = cb; c < ce; ++c) {
        const uint32_t n = s_cnt[c];
        s_gbase[c] = n ? atomicAdd(&gcount[c], n) : 0u;
        sum += n;
    }
    volatile uint32_t* vp = s_p;                             // lanes exchange through LDS between wave barriers: no caching in registers
    vp[tid] = sum;
    __builtin_amdgcn_wave_barrier();
    for (int o = 1; o < VMD_WAVE; o <<= 1) {                 // inclusive scan of the wave's 64 partials, through LDS
        const uint32_t v = lane >= o ? vp[tid - o] : 0u;
        __builtin_amdgcn_wave_barrier();
        vp[tid] = vp[tid] + v;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == VMD_WAVE - 1) vp[1024 + wave] = vp[tid];
    __syncthreads();
    if (wave == 0) {                                         // the 16 wave totals (every lane of the wave walks the barriers)
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t v = (lane < 16 && lane >= o) ? vp[1024 + lane - o] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) vp[1024 + lane] = vp[1024 + lane] + v;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    const uint32_t ntot = s_p[1024 + 15];
    uint32_t run = (s_p[tid] - sum) + (wave ? s_p[1024 + wave - 1] : 0u);
    for (int c = cb; c < ce; ++c) { s_off[c] = run; run += s_cnt[c]; }
    if (!q.bucket) return;                   // counting mode: the host only wants the populations
    __syncthreads();
#pragma unroll
    for (int u = 0; u < VMD_BIN_ILP; ++u) {
        if (pen[u] == 0xffffffffu) continue;
        uint32_t* r = s_rec + (size_t)rw * (s_off[pen[u]] + rank[u]);
        r[0] = (uint32_t)__float_as_int(xw[u]); r[1] = (uint32_t)__float_as_int(yw[u]); r[2] = (uint32_t)__float_as_int(zw[u]); r[3] = pen[u];
        if (!q.rec3) r[4] = cx[u];
    }
    __syncthreads();
    float* bk = q.bucket + (size_t)b * q.total_cap * 4;
    bool over = false;
    for (uint32_t t = tid; t < ntot; t += 1024) {
        const uint32_t* r = s_rec + (size_t)rw * t;
        const uint32_t pn = r[3];
        const uint32_t off = q.pen_off[pn], cap = q.pen_off[pn + 1] - off;
        const uint32_t slot = s_gbase[pn] + (t - s_off[pn]);
        if (slot < cap) {
            if (q.rec3) {
                float* d = bk + 3 * (size_t)(off + slot);
                d[0] = __int_as_float((int)r[0]); d[1] = __int_as_float((int)r[1]); d[2] = __int_as_float((int)r[2]);
            } else {
                const vmd_f4a v = {__int_as_float((int)r[0]), __int_as_float((int)r[1]), __int_as_float((int)r[2]), __int_as_float((int)r[4])};
                *(vmd_f4a*)(bk + 4 * (size_t)(off + slot)) = v;
            }
        } else {
            over = true;
        }
    }
    if (over) atomicOr(q.overflow, q.overflow_bit);
}

// per frame: exclusive prefix of the (capacity-clamped) pencil populations = first sorted slot of every pencil
__global__ __launch_bounds__(256) void k_cells_pen_scan(const uint32_t* __restrict__ pen_count, const uint32_t* __restrict__ pen_off,
                                                        uint32_t* __restrict__ pen_start, int npen) {
    __shared__ uint32_t part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t* cnt = pen_count + (size_t)b * npen;
    uint32_t* out = pen_start + (size_t)b * (npen + 1);
    const int per = (npen + 255) / 256;
    const int beg = tid * per, end = beg + per < npen ? beg + per : npen;
    uint32_t s = 0;
    for (int c = beg; c < end; ++c) { const uint32_t cap = pen_off[c + 1] - pen_off[c]; s += cnt[c] < cap ? cnt[c] : cap; }
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (int c = beg; c < end; ++c) { const uint32_t cap = pen_off[c + 1] - pen_off[c]; out[c] = run; run += cnt[c] < cap ? cnt[c] : cap; }
    if (tid == 255) out[npen] = part[255];
}

struct vmd_pensort_params_t {
    const float* bucket; const uint32_t* pen_off; const uint32_t* pen_count; const uint32_t* pen_start;
    uint32_t* cell_start; float* sorted;
    int npen, nxf, ncell, nsel_pad, total_cap, cap_max;
    const float* boxes; int rec3;      // rec3: 12-byte records, the fine cell is recomputed from x exactly as vmd_cell_of computed it
};
__global__ __launch_bounds__(256) void k_cells_pen_sort(vmd_pensort_params_t q) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    uint32_t* s_cnt = s_dyn;                                  // [nxf]
    uint32_t* s_part = s_dyn + q.nxf;                         // [256] scan partials (no static LDS: the dynamic part may take it all)
    float* s_xyz = (float*)(s_dyn + q.nxf + 256);             // [3][cap_max]
    const int pen = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nxf = q.nxf;
    const uint32_t off = q.pen_off[pen], cap = q.pen_off[pen + 1] - off;
    uint32_t n = q.pen_count[(size_t)b * q.npen + pen];
    n = n < cap ? n : cap;
    const uint32_t start = q.pen_start[(size_t)b * (q.npen + 1) + pen];
    const int rs = q.rec3 ? 3 : 4;                           // floats per record; a frame's buckets start at the same place either way
    const float* bk = q.bucket + 4 * (size_t)b * q.total_cap + (size_t)rs * off;
    const float inv_cx = (float)nxf * q.boxes[(size_t)VMD_BOX_STRIDE * b + 3];
    for (int c = tid; c < nxf; c += 256) s_cnt[c] = 0u;
    __syncthreads();
    for (uint32_t k = tid; k < n; k += 256) {
        const uint32_t c = q.rec3 ? (uint32_t)vmd_cell_coord(bk[3 * (size_t)k], inv_cx, nxf) : (uint32_t)__float_as_int(bk[4 * (size_t)k + 3]);
        atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    // exclusive scan over the fine cells of the pencil
    const int per = (nxf + 255) / 256;
    const int beg = tid * per, end = beg + per < nxf ? beg + per : nxf;
    uint32_t sum = 0;
    for (int c = beg; c < end; ++c) sum += s_cnt[c];
    // (a per-wave scan through LDS with one block barrier measured 7 % slower for this kernel than the plain block scan: profiles/r02y)
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    uint32_t* cs = q.cell_start + (size_t)b * (q.ncell + 1) + (size_t)pen * nxf;
    for (int c = beg; c < end; ++c) { const uint32_t m = s_cnt[c]; s_cnt[c] = run; cs[c] = start + run; run += m; }
    if (pen == q.npen - 1 && tid == 255) q.cell_start[(size_t)b * (q.ncell + 1) + q.ncell] = start + n;
    __syncthreads();
    float* sx = s_xyz; float* sy = s_xyz + q.cap_max; float* sz = s_xyz + 2 * (size_t)q.cap_max;
    for (uint32_t k = tid; k < n; k += 256) {
        float x, y, z; uint32_t c;
        if (q.rec3) {
            const float* r = bk + 3 * (size_t)k;
            x = r[0]; y = r[1]; z = r[2];
            c = (uint32_t)vmd_cell_coord(x, inv_cx, nxf);
        } else {
            const vmd_f4a v = *(const vmd_f4a*)(bk + 4 * (size_t)k);
            x = v[0]; y = v[1]; z = v[2]; c = (uint32_t)__float_as_int(v[3]);
        }
        const uint32_t pos = atomicAdd(&s_cnt[c], 1u);
        sx[pos] = x; sy[pos] = y; sz[pos] = z;
    }
    __syncthreads();
    float* srt = q.sorted + (size_t)b * 3 * q.nsel_pad + start;
    for (uint32_t k = tid; k < n; k += 256) {
        srt[k] = sx[k];
        srt[q.nsel_pad + k] = sy[k];
        srt[2 * (size_t)q.nsel_pad + k] = sz[k];
    }
}

// bounding box of all atoms of every frame (open axes: the pencil grid spans the box of the batch): out[b] = {min xyz, max xyz}
__global__ __launch_bounds__(1024) void k_bbox(const float* __restrict__ xyz, size_t frame_stride, size_t row_stride, int natoms,
                                               float* __restrict__ out) {
    __shared__ float s_lo[3][16], s_hi[3][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* fx = xyz + (size_t)b * frame_stride;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int a = tid; a < natoms; a += 1024)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float v = fx[k * row_stride + a]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = vmd_wave_min(lo[k]); hi[k] = vmd_wave_max(hi[k]);
        if ((tid & 63) == 0) { s_lo[k][tid >> 6] = lo[k]; s_hi[k][tid >> 6] = hi[k]; }
    }
    __syncthreads();
    if (tid < 3) {
        float l = s_lo[tid][0], h = s_hi[tid][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, s_lo[tid][w]); h = fmaxf(h, s_hi[tid][w]); }
        out[6 * b + tid] = l; out[6 * b + 3 + tid] = h;
    }
}

// ------------------------------------------------------------------------------------------------ K2: RDF, pencil grid

struct vmd_pair_params_t {
    const float* __restrict__ sref; const uint32_t* __restrict__ cs_ref; int nref_pad;
    const float* __restrict__ stgt; const uint32_t* __restrict__ cs_tgt; int ntgt_pad;
    const float* __restrict__ boxes; int B;
    vmd_grid_t grid;
    vmd_binning_t bin;
    float r2_up;        // conservative candidate filter (> rmax^2)
    float rpad;         // conservative range padding (> rmax)
    uint64_t* partial;  // [gridDim.x][nbins]: one row per block, written once at the end
    unsigned long long* counts;  // the accumulators: target of the (rare) overflow flush
    unsigned* work_counter;  // [8 * VMD_COUNTER_STRIDE], zeroed before launch: one dynamic work queue per XCD (frames f = q mod 8)
    int nsub;                // work items per pencil (i-chunks are dealt round-robin to the items)
    uint32_t pbc;            // bits 0..2: periodic axes (an open axis spans the batch's bounding box: boxes slots 6..8 = origin)
    const uint32_t* skip;    // device flag or NULL: non-zero = the sorted copies are incomplete (a bucket of the cell build overflowed), do nothing
    int ry, rz;              // neighbour reach in pencils per axis: 1 (cross-section >= rmax), 2 = split pencils (cross-section >= rmax/2)
    int nsplit;              // > 1 (small launches): a chunk's neighbour pencils are dealt to nsplit work items instead of one - a lone
                             // item is a dependent chain of cold scalar loads (170 us for a one-frame launch, profiles/r03aq)
    int pop;                 // host side only: variant 0 is launched with the folded pop (instantiation POP = 2); 0 unless r_min == 0 and the fast path is usable at all
    unsigned long long* cols_total;   // device counter or NULL: candidate columns of every launch since the host last reset it (one atomic per
                                      // wave at kernel end): bench.py's "candidate lanes per counted hit" is measured, not modelled
};
#define VMD_COUNTER_STRIDE 32    // one 128-byte line per queue counter

// per-wave state of the hit machinery.  POP is part of the TYPE (the helpers below deduce it): a run-time choice between the pops - three
// inlined at every drain site, +6 KB of code, 72 instead of 68 VGPRs - cost the kernel 8 % (profiles/r06q_pop_compile_time_ab.txt).
// 0 = vmd_pop_hot, 1 = vmd_pop_hot0 (folded delta) with a plain ds_read_b32, 2 = vmd_pop_hot0 reading the stack with ds_read_addtid_b32
template <int POP>
struct vmd_wave_acc_tt {
    static constexpr int pop = POP;
    unsigned* hist;       // LDS, nbins
    float* queue;         // LDS, VMD_QUEUE_CAP floats: the wave's hit stack
    unsigned qbase;       // LDS byte address of queue[0] (0 in the emulator build, where qtop is a plain offset)
    unsigned qlim;        // qbase + 4 * VMD_WAVE: the stack holds a full wave of hits when qtop reaches it
    unsigned hbase;       // LDS byte address of hist[0] (product build only)
    unsigned qtop;        // wave-uniform: LDS byte address of the top of the stack
    float fast_c;         // bn.fast_c held in a VGPR
    float fast_k, fast_far;   // bn.fast_k / bn.fast_far in VGPRs (an SGPR operand halves the issue rate of v_fma_f32)
    float fast_c2;            // bn.fast_c2 in a VGPR (vmd_pop_hot0)
    float* slow;          // LDS, 64 floats behind the stack: hits whose fast binning was not provably exact, waiting for
    unsigned nslow;       // (wave-uniform count) a full wave of them to go through the exact path together
    unsigned ncols;       // wave-uniform: candidate columns (<= 64 hits each) since the last flush
};

// Bin `d2` of every `active` lane into the LDS histogram.  Must be called by all lanes of the wave.
// Result is exactly vmd_bin_of(d2) (SPEC S4): the 1-ulp hardware sqrt is only used where it provably cannot change the
// bin or the open-interval test; the (rare) uncertain lanes re-do the computation with the correctly rounded sqrtf.
template <unsigned INC>
__device__ __forceinline__ void vmd_bin_add(const vmd_binning_t& bn, unsigned* hist, float d2, bool active) {
    const float t = fmaf(__builtin_amdgcn_sqrtf(d2), bn.fast_k, bn.fast_c);
    int bin = (int)t;                                   // truncation: floor for the t > 0 that can be "sure"
    const float fr = t - truncf(t);
    // sure <=> t is inside bin `bin` with margin delta on both sides and the bin exists (negative t wraps to a huge unsigned)
    const bool sure = fabsf(fr - 0.5f) < bn.fast_half && (unsigned)bin < (unsigned)bn.nbins;
    bool add = active && sure;
    if (VMD_BALLOT(active && !sure)) {
        if (active && !sure) {
            bin = vmd_bin_of(bn, d2);
            add = bin >= 0;
        }
    }
    if (add) atomicAdd(&hist[bin], INC);
}

// The exact path of vmd_bin_add costs ~25 VALU instructions and used to run for 1-2 lanes at a time in every fifth pop.
// The drain therefore parks uncertain hits on a second, 64-entry LDS stack and sends them through vmd_bin_of a full wave at
// a time.  Both functions must be called by all lanes of the wave.
// (Round 6: an out-of-line version - one copy behind a three-register call instead of one per drain site, 24 -> 17 KB of code - measured
// 0.7 % SLOWER on c3, profiles/r06s_pop_ab.txt; the copies stay.)
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_slow_flush(const vmd_binning_t& bn, W& w, int lane) {
    __builtin_amdgcn_wave_barrier();
    const float v = w.slow[lane];
    __builtin_amdgcn_wave_barrier();
    if ((unsigned)lane < w.nslow) {
        const int bin = vmd_bin_of(bn, v);
        if (bin >= 0) atomicAdd(&w.hist[bin], INC);
    }
    w.nslow = 0;
}
// parks the lanes of mask `m` (hits whose fast binning was not provably exact) on the slow stack
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_slow_park(const vmd_binning_t& bn, W& w, float d2, unsigned long long m, bool unsure, int lane) {
    const unsigned cnt = (unsigned)__popcll(m);
    if (w.nslow + cnt > VMD_WAVE) vmd_slow_flush<INC>(bn, w, lane);
    const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (unsure) w.slow[w.nslow + pre] = d2;
    w.nslow += cnt;
}
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_bin_add_deferred(const vmd_binning_t& bn, W& w, float d2, bool active, int lane) {
    const float t = fmaf(__builtin_amdgcn_sqrtf(d2), bn.fast_k, w.fast_c);
    const int bin = vmd_floor_to_int(t);                // floor: a negative t (d < rmin) can never pass the range test below
    const float fr = vmd_fract(t);                      // t - floor(t), exact
    const bool sure = fabsf(fr - 0.5f) < bn.fast_half && (unsigned)bin < (unsigned)bn.nbins;
    if (active && sure) atomicAdd(&w.hist[bin], INC);
    const bool unsure = active && !sure;
    const unsigned long long m = VMD_BALLOT(unsure);
    if (m) vmd_slow_park<INC>(bn, w, d2, m, unsure, lane);
}
// r_min = 0, delta folded into the constant (vmd_binning_t::fast_c2 / fast_2d): t2 = t' + delta.  sure <=> fract(t2) > 2 delta, i.e. t' lies
// in (n + delta, n + 1 - delta) with n = floor(t2): the spec's scaled distance, at most delta / 2 away from t', is strictly inside bin n - the
// argument of vmd_bin_add, one rounding (the fma's) as there.  No range test: t2 >= delta > 0, and a "sure" t2 >= nbins is a distance beyond
// r_max that the padded cutoff let through (t2 < nbins * sqrt(1.0001) + delta < nbins + 1): it lands in the spare bin nbins, which nobody reads.
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_bin_add_deferred0(const vmd_binning_t& bn, W& w, float d2, bool active, int lane) {
    const float t2 = fmaf(__builtin_amdgcn_sqrtf(d2), w.fast_k, w.fast_c2);
    const int bin = vmd_floor_to_int(t2);
    const bool sure = vmd_fract(t2) > bn.fast_2d;
    if (active && sure) atomicAdd(&w.hist[bin], INC);
    const bool unsure = active && !sure;
    const unsigned long long m = VMD_BALLOT(unsure);
    if (m) vmd_slow_park<INC>(bn, w, d2, m, unsure, lane);
}
#ifndef VMD_NO_INLINE_ASM
// vmd_pop_hot for r_min = 0 (every rdf() VIAMD's scripts write): 7 VALU instructions instead of 9 - v_add_f32 -0.5 and the |t| compare become
// one compare against 2 delta, the bin range compare goes (spare bin).  (k stays an SGPR operand of the fma: a VGPR would issue faster, and
// be the 73rd of a kernel that needs 72 for its seventh wave per SIMD.)
// ADDTID: the stack is read with ds_read_addtid_b32 (address = M0 + 4 * lane): the address add goes too, 6 VALU.
template <unsigned INC, bool ADDTID, class W>
__device__ __forceinline__ unsigned long long vmd_pop_hot0(const vmd_binning_t& bn, W& w, unsigned lane4, unsigned inc, float& d2) {
    unsigned long long m;
    float t;
    int b;
    if (ADDTID) {
        asm volatile(
            "s_mov_b32 m0, %[q]\n\t"
            "s_nop 0\n\t"                                   // SALU write of M0 -> LDS add-TID instruction: one wait state
            "ds_read_addtid_b32 %[d2]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_sqrt_f32 %[t], %[d2]\n\t"
            "s_nop 0\n\t"
            "v_fma_f32 %[t], %[k], %[t], %[c]\n\t"
            "v_cvt_flr_i32_f32 %[b], %[t]\n\t"
            "v_fract_f32 %[t], %[t]\n\t"
            "v_cmp_lt_f32 vcc, %[dd], %[t]\n\t"
            "v_lshl_add_u32 %[b], %[b], 2, %[hb]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "ds_add_u32 %[b], %[inc]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_not_b64 %[m], vcc\n\t"
            : [m] "=&s"(m), [t] "=&v"(t), [b] "=&v"(b), [d2] "=&v"(d2)
            : [q] "s"(w.qtop), [k] "s"(bn.fast_k), [c] "v"(w.fast_c2), [dd] "s"(bn.fast_2d), [hb] "s"(w.hbase), [inc] "v"(inc)
            : "vcc", "scc", "m0", "memory");
    } else {
        asm volatile(
            "v_add_u32 %[b], %[q], %[l4]\n\t"
            "ds_read_b32 %[d2], %[b]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_sqrt_f32 %[t], %[d2]\n\t"
            "s_nop 0\n\t"
            "v_fma_f32 %[t], %[k], %[t], %[c]\n\t"
            "v_cvt_flr_i32_f32 %[b], %[t]\n\t"
            "v_fract_f32 %[t], %[t]\n\t"
            "v_cmp_lt_f32 vcc, %[dd], %[t]\n\t"
            "v_lshl_add_u32 %[b], %[b], 2, %[hb]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "ds_add_u32 %[b], %[inc]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_not_b64 %[m], vcc\n\t"
            : [m] "=&s"(m), [t] "=&v"(t), [b] "=&v"(b), [d2] "=&v"(d2)
            : [q] "s"(w.qtop), [l4] "v"(lane4), [k] "s"(bn.fast_k), [c] "v"(w.fast_c2), [dd] "s"(bn.fast_2d), [hb] "s"(w.hbase), [inc] "v"(inc)
            : "vcc", "scc", "memory");
    }
    return m;
}
// The pop of the hot loop, hand-scheduled: read one full wave of hits from the top of the stack (w.qtop already lowered),
// fast-bin them (same arithmetic as vmd_bin_add_deferred) and ds_add under EXEC = sure mask; returns the mask of the lanes
// that must take the exact path.  9 VALU instructions; hipcc needs 14 for the same C++ (two address adds, a v_mov for the
// second uniform operand of the fma, trunc + sub instead of fract, and a v_cndmask + v_cmp round trip for the ballot).
// Requires EXEC = all lanes.
template <unsigned INC, class W>
__device__ __forceinline__ unsigned long long vmd_pop_hot(const vmd_binning_t& bn, W& w, unsigned lane4, unsigned inc, float& d2) {
    unsigned long long m;
    float t;
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
        "s_not_b64 %[m], vcc\n\t"
        : [m] "=&s"(m), [t] "=&v"(t), [b] "=&v"(b), [d2] "=&v"(d2)
        : [q] "s"(w.qtop), [l4] "v"(lane4), [k] "s"(bn.fast_k), [c] "v"(w.fast_c), [half] "s"(bn.fast_half),
          [nb] "s"(bn.nbins), [hb] "s"(w.hbase), [inc] "v"(inc)
        : "vcc", "scc", "memory");
    return m;
}
#endif

// variant 2: the same with the "certainly outside" test in front of the parking (a pair entry's partner is usually no hit)
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_bin_add_deferred_far(const vmd_binning_t& bn, W& w, float d2, bool active, int lane) {
    const float t = fmaf(__builtin_amdgcn_sqrtf(d2), bn.fast_k, w.fast_c);
    const int bin = vmd_floor_to_int(t);
    const float fr = vmd_fract(t);
    const bool sure = fabsf(fr - 0.5f) < bn.fast_half && (unsigned)bin < (unsigned)bn.nbins;
    if (active && sure) atomicAdd(&w.hist[bin], INC);
    const bool unsure = active && !sure && t < bn.fast_far;
    const unsigned long long m = VMD_BALLOT(unsure);
    if (m) vmd_slow_park<INC>(bn, w, d2, m, unsure, lane);
}
#ifndef VMD_NO_INLINE_ASM
// variant 2 pop: 64 pair entries = 128 values, two interleaved binning chains; returns the two park masks.  19 VALU.
template <unsigned INC, class W>
__device__ __forceinline__ void vmd_pop_hot2(const vmd_binning_t& bn, W& w, unsigned lane8 /* 4 * lane: entries are slots of 4 bytes */, unsigned inc, float& a, float& b,
                                             unsigned long long& ma, unsigned long long& mb) {
    unsigned long long sa;
    float ta, tb;
    int ba, bb;
    asm volatile(
        "v_add_u32 %[ba], %[q], %[l8]\n\t"
        "ds_read_b32 %[a], %[ba]\n\t"
        "ds_read_b32 %[b], %[ba] offset:512\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sqrt_f32 %[ta], %[a]\n\t"
        "v_sqrt_f32 %[tb], %[b]\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[ta], %[k], %[ta], %[c]\n\t"
        "v_fma_f32 %[tb], %[k], %[tb], %[c]\n\t"
        "v_cvt_flr_i32_f32 %[ba], %[ta]\n\t"
        "v_cvt_flr_i32_f32 %[bb], %[tb]\n\t"
        "v_cmp_gt_f32 %[ma], %[far], %[ta]\n\t"
        "v_cmp_gt_f32 %[mb], %[far], %[tb]\n\t"
        "v_fract_f32 %[ta], %[ta]\n\t"
        "v_fract_f32 %[tb], %[tb]\n\t"
        "v_add_f32 %[ta], -0.5, %[ta]\n\t"
        "v_add_f32 %[tb], -0.5, %[tb]\n\t"
        "v_cmp_lt_f32 %[sa], |%[ta]|, %[half]\n\t"
        "v_cmp_gt_u32 vcc, %[nb], %[ba]\n\t"
        "s_and_b64 vcc, vcc, %[sa]\n\t"
        "s_andn2_b64 %[ma], %[ma], vcc\n\t"
        "v_lshl_add_u32 %[ba], %[ba], 2, %[hb]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[ba], %[inc]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_cmp_lt_f32 %[sa], |%[tb]|, %[half]\n\t"
        "v_cmp_gt_u32 vcc, %[nb], %[bb]\n\t"
        "s_and_b64 vcc, vcc, %[sa]\n\t"
        "s_andn2_b64 %[mb], %[mb], vcc\n\t"
        "v_lshl_add_u32 %[bb], %[bb], 2, %[hb]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_add_u32 %[bb], %[inc]\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [ma] "=&s"(ma), [mb] "=&s"(mb), [sa] "=&s"(sa), [ta] "=&v"(ta), [tb] "=&v"(tb), [ba] "=&v"(ba), [bb] "=&v"(bb), [a] "=&v"(a), [b] "=&v"(b)
        : [q] "s"(w.qtop), [l8] "v"(lane8), [k] "v"(w.fast_k), [c] "v"(w.fast_c), [far] "v"(w.fast_far), [half] "s"(bn.fast_half),
          [nb] "s"(bn.nbins), [hb] "s"(w.hbase), [inc] "v"(inc)
        : "vcc", "scc", "memory");
}
#endif

// variant 2: TWO candidate columns share one compaction.  A lane whose smaller d2 of the pair is a candidate pushes both
// values as one entry: slot s of the stack holds the first value at queue[s] and the second at queue[VMD_PAIR_B + s] (two
// conflict-free 4-byte accesses per lane instead of one 8-byte access with a 2-way bank conflict); the pop bins both and drops the
// partner that is no hit with one compare.  Per pair of columns: v_min + v_cmp + 2 v_mbcnt + v_lshl_add + one ds_write_b64 instead of
// 2 x (v_cmp + 2 v_mbcnt + v_lshl_add + ds_write_b32): the integer ops of the prefix issue at 4 cycles per wave whatever their
// operands are (profiles/r02_valu_calibration.txt), so halving them is what counts; the pop handles ~1.85 values per hit.
template <class W>
__device__ __forceinline__ void vmd_push2(W& w, bool hit, float a, float b) {
    const unsigned long long mask = VMD_BALLOT(hit);
    if (mask) {
        const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (hit) {
            float* e = (float*)((char*)w.queue + ((w.qtop - w.qbase) + 4u * pre));
            e[0] = a; e[VMD_PAIR_B] = b;
        }
        w.qtop += 4u * (unsigned)__popcll(mask);
    }
}

// VARIANT 0: compact the hits of one candidate column onto the wave's LDS stack (order is irrelevant for a
// histogram, so LIFO: no head pointer, no wrap-around); vmd_drain_full pops full waves of 64 so that
// sqrt + binning + ds_add always run with every lane busy.
// VARIANT 1: bin the hits in place under the divergent mask (same arithmetic; A/B baseline and cross-check).
template <int VARIANT, unsigned INC, class W>
__device__ __forceinline__ void vmd_push(const vmd_binning_t& bn, W& w, bool hit, float d2) {
    if (VARIANT == 1) {
        vmd_bin_add<INC>(bn, w.hist, d2, hit);
        return;
    }
    if (VARIANT == 2) { vmd_push2(w, hit, d2, VMD_JUNK); return; }
    const unsigned long long mask = VMD_BALLOT(hit);
    if (mask) {
        const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (hit) *(float*)((char*)w.queue + ((w.qtop - w.qbase) + 4u * pre)) = d2;
        w.qtop += 4u * (unsigned)__popcll(mask);
    }
}

template <int VARIANT, unsigned INC, class W>
__device__ __forceinline__ void vmd_drain_full(const vmd_binning_t& bn, W& w, int lane) {
    if (VARIANT == 1) return;
    if (VARIANT == 2) {
        while (w.qtop - w.qbase >= 4u * VMD_WAVE) {
            w.qtop -= 4u * VMD_WAVE;
            float a, b;
#ifndef VMD_NO_INLINE_ASM
            unsigned long long ma, mb;
            vmd_pop_hot2<INC>(bn, w, 4u * (unsigned)lane, INC, a, b, ma, mb);
            if (ma) vmd_slow_park<INC>(bn, w, a, ma, (ma >> lane) & 1ull, lane);
            if (mb) vmd_slow_park<INC>(bn, w, b, mb, (mb >> lane) & 1ull, lane);
#else
            __builtin_amdgcn_wave_barrier();
            const float* e = (const float*)((const char*)w.queue + ((w.qtop - w.qbase) + 4u * (unsigned)lane));
            a = e[0]; b = e[VMD_PAIR_B];
            __builtin_amdgcn_wave_barrier();
            vmd_bin_add_deferred_far<INC>(bn, w, a, true, lane);
            vmd_bin_add_deferred_far<INC>(bn, w, b, true, lane);
#endif
        }
        return;
    }
    while (w.qtop >= w.qlim) {      // (against a kept limit: s_cmp + branch; `qtop - qbase >= 256` is one SALU instruction more per group of four
                                    // columns, and this loop feels every one of them - c3 -1.2 %, profiles/r06v_salu_ab.txt)
        w.qtop -= 4u * VMD_WAVE;
#ifndef VMD_NO_INLINE_ASM
        float v;
        const unsigned long long m = W::pop == 2 ? vmd_pop_hot0<INC, true>(bn, w, 4u * (unsigned)lane, INC, v)
                                   : W::pop == 1 ? vmd_pop_hot0<INC, false>(bn, w, 4u * (unsigned)lane, INC, v)
                                                 : vmd_pop_hot<INC>(bn, w, 4u * (unsigned)lane, INC, v);
        if (m) vmd_slow_park<INC>(bn, w, v, m, (m >> lane) & 1ull, lane);
#else
        __builtin_amdgcn_wave_barrier();
        const float v = *(const float*)((const char*)w.queue + ((w.qtop - w.qbase) + 4u * (unsigned)lane));
        __builtin_amdgcn_wave_barrier();
        if (W::pop) vmd_bin_add_deferred0<INC>(bn, w, v, true, lane);
        else vmd_bin_add_deferred<INC>(bn, w, v, true, lane);
#endif
    }
}

template <int VARIANT, unsigned INC, class W>
__device__ __forceinline__ void vmd_drain(const vmd_binning_t& bn, W& w, int lane) {
    if (VARIANT == 1) return;
    vmd_drain_full<VARIANT, INC>(bn, w, lane);
    if (VARIANT == 2) {
        const unsigned rem = (w.qtop - w.qbase) / 4u;
        __builtin_amdgcn_wave_barrier();
        const float a = w.queue[lane], b = w.queue[VMD_PAIR_B + lane];
        __builtin_amdgcn_wave_barrier();
        w.qtop = w.qbase;
        vmd_bin_add_deferred_far<INC>(bn, w, a, (unsigned)lane < rem, lane);
        vmd_bin_add_deferred_far<INC>(bn, w, b, (unsigned)lane < rem, lane);
        vmd_slow_flush<INC>(bn, w, lane);
        return;
    }
    const unsigned rem = (w.qtop - w.qbase) / 4u;
    __builtin_amdgcn_wave_barrier();
    const float v = w.queue[lane];
    __builtin_amdgcn_wave_barrier();
    w.qtop = w.qbase;
    if (W::pop) vmd_bin_add_deferred0<INC>(bn, w, v, (unsigned)lane < rem, lane);
    else vmd_bin_add_deferred<INC>(bn, w, v, (unsigned)lane < rem, lane);
    vmd_slow_flush<INC>(bn, w, lane);
}

// The push of the hot loop, hand-scheduled (hipcc spends 8 SALU + 5 VALU per column on the same thing): compare, and if
// any lane hit, prefix the hit lanes (v_mbcnt), store their d2 on the LDS stack under EXEC = hit mask, advance the stack.
// Requires EXEC = all 64 lanes on entry (true in the segment loops: padding lanes carry far-away coordinates instead of
// being masked off).  The SIMT emulator build (tests/emu) has no inline asm and takes the plain C++ vmd_push instead.
#ifndef VMD_NO_INLINE_ASM
#define VMD_PUSH_NOP "s_nop 0\n\t"      // (dropping it changes neither results nor speed on gfx950, profiles/r06v_salu_ab.txt; the VOP3P -> VALU wait state stays explicit)
#define VMD_LDS_ADDRESS(p) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))
template <class W>
__device__ __forceinline__ void vmd_push_hot(W& w, float d2, float r2) {
    unsigned t, n;
    // s_nop: d2 usually comes straight out of a v_pk_fma_f32; the hazard recogniser cannot see into this block and the
    // VOP3P result needs one wait state before a dependent VALU read on gfx940-class parts
    asm volatile(
        "s_nop 0\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
        "s_cbranch_vccz .Lvmd_nohit%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d2]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_nohit%=:"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n)
        : [d2] "v"(d2), [r2] "s"(r2)
        : "vcc", "scc", "memory");
}
// same with the own-pencil condition j > i folded in (j wave-uniform, i per lane)
template <class W>
__device__ __forceinline__ void vmd_push_hot_masked(W& w, float d2, float r2, unsigned j, unsigned i) {
    unsigned t, n;
    unsigned long long m;
    asm volatile(
        "s_nop 0\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
        "v_cmp_gt_u32 %[m], %[j], %[i]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_cbranch_scc0 .Lvmd_nohitm%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d2]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_nohitm%=:"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n), [m] "=&s"(m)
        : [d2] "v"(d2), [r2] "s"(r2), [j] "s"(j), [i] "v"(i)
        : "vcc", "scc", "memory");
}
// Four columns per asm block: consecutive blocks each drew a hazard nop from the compiler (which cannot see inside) on top
// of their own, and pinned the order of the packed chains around them; one block per s_load group needs a single wait
// state and lets hipcc interleave the two packed d2 chains in front of it.
template <class W>
__device__ __forceinline__ void vmd_push_hot4(W& w, float d0, float d1, float d2, float d3, float r2) {
    unsigned t, n;
    asm volatile(
        VMD_PUSH_NOP
        "v_cmp_gt_f32 vcc, %[r2], %[d0]\n\t"
        "s_cbranch_vccz .Lvmd_p4_0_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d0]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4_0_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d1]\n\t"
        "s_cbranch_vccz .Lvmd_p4_1_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d1]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4_1_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
        "s_cbranch_vccz .Lvmd_p4_2_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d2]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4_2_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d3]\n\t"
        "s_cbranch_vccz .Lvmd_p4_3_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d3]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4_3_%=:\n\t"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n)
        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [r2] "s"(r2)
        : "vcc", "scc", "memory");
}
// own pencil: column k counts only j + k > i.  ik = i - k as signed integers (indices stay far below 2^31), so the four
// conditions are v_cmp_gt_i32 j, ik with ONE uniform j.
template <class W>
__device__ __forceinline__ void vmd_push_hot4_masked(W& w, float d0, float d1, float d2, float d3, float r2,
                                                     int j, int i0, int i1, int i2, int i3) {
    unsigned t, n;
    unsigned long long m;
    asm volatile(
        VMD_PUSH_NOP
        "v_cmp_gt_f32 vcc, %[r2], %[d0]\n\t"
        "v_cmp_gt_i32 %[m], %[j], %[i0]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_cbranch_scc0 .Lvmd_p4m_0_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d0]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4m_0_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d1]\n\t"
        "v_cmp_gt_i32 %[m], %[j], %[i1]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_cbranch_scc0 .Lvmd_p4m_1_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d1]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4m_1_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d2]\n\t"
        "v_cmp_gt_i32 %[m], %[j], %[i2]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_cbranch_scc0 .Lvmd_p4m_2_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d2]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4m_2_%=:\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[d3]\n\t"
        "v_cmp_gt_i32 %[m], %[j], %[i3]\n\t"
        "s_and_b64 vcc, vcc, %[m]\n\t"
        "s_cbranch_scc0 .Lvmd_p4m_3_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[d3]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_p4m_3_%=:\n\t"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n), [m] "=&s"(m)
        : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [r2] "s"(r2), [j] "s"(j), [i0] "v"(i0), [i1] "v"(i1), [i2] "v"(i2), [i3] "v"(i3)
        : "vcc", "scc", "memory");
}
// variant 2: one pair-column (two candidate columns), a / b = its two d2 values.  The stack is drained after every pair-column, so it
// never holds more than 64 + 64 entries.
template <class W>
__device__ __forceinline__ void vmd_push_hot2p(W& w, float a, float b, float r2) {
    unsigned t, n;
    float m;
    asm volatile(
        "s_nop 0\n\t"
        "v_min_f32 %[m], %[a], %[b]\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[m]\n\t"
        "s_cbranch_vccz .Lvmd_pp_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[a]\n\t"
        "ds_write_b32 %[t], %[b] offset:512\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_pp_%=:\n\t"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n), [m] "=&v"(m)
        : [a] "v"(a), [b] "v"(b), [r2] "s"(r2)
        : "vcc", "scc", "memory");
}
// own pencil: the chunk against itself.  Lane L holds atom cbeg + L; the pair-column (j, j + 1) counts for the lanes L <= j - cbeg
// only (unordered pairs once; on lane L == j - cbeg the first value is the atom's distance to itself: the caller replaces it by
// VMD_JUNK).  lm: that lane mask, wave-uniform.
template <class W>
__device__ __forceinline__ void vmd_push_hot2p_masked(W& w, float a, float b, float r2, unsigned long long lm) {
    unsigned t, n;
    float m;
    asm volatile(
        "s_nop 0\n\t"
        "v_min_f32 %[m], %[a], %[b]\n\t"
        "v_cmp_gt_f32 vcc, %[r2], %[m]\n\t"
        "s_and_b64 vcc, vcc, %[lm]\n\t"
        "s_cbranch_scc0 .Lvmd_ppm_%=\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[q]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[t], %[a]\n\t"
        "ds_write_b32 %[t], %[b] offset:512\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_bcnt1_i32_b64 %[n], vcc\n\t"
        "s_lshl2_add_u32 %[q], %[n], %[q]\n"
        ".Lvmd_ppm_%=:\n\t"
        : [q] "+s"(w.qtop), [t] "=&v"(t), [n] "=&s"(n), [m] "=&v"(m)
        : [a] "v"(a), [b] "v"(b), [r2] "s"(r2), [lm] "s"(lm)
        : "vcc", "scc", "memory");
}
#else
#define VMD_LDS_ADDRESS(p) 0u
#endif

// one uniform j segment [ja, jb) against the wave's 64 i atoms.  MASKED: count only j > i (own pencil, same set).
// SHIFT: the segment is a periodic image, displaced by (sx,sy,sz) (SPEC S3: dx = fl(fl(xi-xj) - sx)).
template <int VARIANT, unsigned INC, bool MASKED, bool SHIFT, class W>
__device__ __forceinline__ void vmd_segment_loop(const vmd_pair_params_t& p, W& w, vmd_cf32* tx, vmd_cf32* ty, vmd_cf32* tz,
                                                 unsigned ja, unsigned jb, float sx, float sy, float sz,
                                                 float xi, float yi, float zi, unsigned i, int lane) {
    const float r2 = p.r2_up;
    vmd_cf32* px = tx + ja;
    vmd_cf32* py = ty + ja;
    vmd_cf32* pz = tz + ja;
    const unsigned n = jb - ja;
    // two j columns per VALU instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 are IEEE per element, so the
    // arithmetic is still SPEC S3 exactly; the j pair sits in an SGPR pair straight from s_load_dwordx4
    const vmd_f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
    const vmd_f2 sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz};
    // four columns (two packed pairs) from one s_load_dwordx4 per coordinate
    auto group = [&](const vmd_f4& xj, const vmd_f4& yj, const vmd_f4& zj, unsigned k0) {
        vmd_f2 d2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const vmd_f2 xj2 = {xj[2 * h], xj[2 * h + 1]}, yj2 = {yj[2 * h], yj[2 * h + 1]}, zj2 = {zj[2 * h], zj[2 * h + 1]};
            vmd_f2 dx = xi2 - xj2, dy = yi2 - yj2, dz = zi2 - zj2;
            if (SHIFT) { dx = dx - sx2; dy = dy - sy2; dz = dz - sz2; }
            d2[h] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        }
#ifndef VMD_NO_INLINE_ASM
        if (VARIANT == 0) {
            if (MASKED) vmd_push_hot4_masked(w, d2[0][0], d2[0][1], d2[1][0], d2[1][1], r2, (int)(ja + k0), (int)i, (int)i - 1, (int)i - 2, (int)i - 3);
            else vmd_push_hot4(w, d2[0][0], d2[0][1], d2[1][0], d2[1][1], r2);
            vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
            return;
        }
        if (VARIANT == 2) {
            if (MASKED) {
                // lanes 0 .. (j - cbeg) of the pair-column starting at j; the chunk starts at the atom of lane 0
                const int a = (int)(ja + k0) - __builtin_amdgcn_readfirstlane((int)i);
                const unsigned long long lm0 = a >= 63 ? ~0ull : ((2ull << a) - 1ull);
                const unsigned long long lm1 = a + 2 >= 63 ? ~0ull : ((2ull << (a + 2)) - 1ull);
                // the self pair (lane a of the first column, lane a + 2 of the third) never enters the stack
                const float s0 = lane == a ? VMD_JUNK : d2[0][0];
                const float s1 = lane == a + 2 ? VMD_JUNK : d2[1][0];
                vmd_push_hot2p_masked(w, s0, d2[0][1], r2, lm0);
                vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
                vmd_push_hot2p_masked(w, s1, d2[1][1], r2, lm1);
            } else {
                vmd_push_hot2p(w, d2[0][0], d2[0][1], r2);
                vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
                vmd_push_hot2p(w, d2[1][0], d2[1][1], r2);
            }
            vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
            return;
        }
#endif
        if (VARIANT == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float a = d2[h][0], b = d2[h][1];
                if (MASKED) { if (!(ja + k0 + 2 * h > i)) a = VMD_JUNK; if (!(ja + k0 + 2 * h + 1 > i)) b = VMD_JUNK; }
                vmd_push2(w, fminf(a, b) < r2, a, b);
                vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = d2[c >> 1][c & 1];
            bool hit = v < r2;
            if (MASKED) hit = hit && (ja + k0 + c > i);
            vmd_push<VARIANT, INC>(p.bin, w, hit, v);
        }
        vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
    };
    // software pipeline: the scalar loads of the next group are issued before the current group is processed (two register
    // sets, no copies).  Loads may run up to 16 bytes past the segment: the sorted rows carry that much slack.
    // The ragged end (n mod 8 columns) goes through the same packed group, its columns beyond the segment moved out of every cutoff
    // (x = -VMD_FAR: never a hit, for valid and padding lanes alike) - until round 6 a one-column loop with the plain C++ push and a
    // drain of its own: ~1 % of the columns at twice the price (c3: 71.9 -> 71.3 ms per 1 000 frames, profiles/r06r_tail_ab.txt).
    if (n == 0) return;
    vmd_f4 xa = vmd_uniform_load4(px, 0), ya = vmd_uniform_load4(py, 0), za = vmd_uniform_load4(pz, 0);
    // the loop counts BYTES of the j rows and runs against a kept last offset: add + compare + one backward branch per eight columns (counting
    // columns it was add, compare, two branches, a register copy and a shift; every scalar instruction of this loop shows, profiles/r06v_salu_ab.txt)
    unsigned off = 0;
    if (n >= 8) {
        const unsigned stop = 4u * n - 32u;
        do {
            const vmd_f4 xb = vmd_uniform_load4(px, off + 16u), yb = vmd_uniform_load4(py, off + 16u), zb = vmd_uniform_load4(pz, off + 16u);
            group(xa, ya, za, off >> 2);
            xa = vmd_uniform_load4(px, off + 32u); ya = vmd_uniform_load4(py, off + 32u); za = vmd_uniform_load4(pz, off + 32u);
            group(xb, yb, zb, (off >> 2) + 4u);
            off += 32u;
        } while (off <= stop);
    }
    unsigned k = off >> 2;
    while (k < n) {                                   // at most twice: xa / ya / za hold columns k .. k + 3
        const unsigned r = n - k;
        vmd_f4 xb = xa, yb = ya, zb = za;
        if (r > 4) { xb = vmd_uniform_load4(px, 4u * k + 16u); yb = vmd_uniform_load4(py, 4u * k + 16u); zb = vmd_uniform_load4(pz, 4u * k + 16u); }
        if (r < 4) {
            xa[3] = -VMD_FAR;
            if (r < 3) xa[2] = -VMD_FAR;
            if (r < 2) xa[1] = -VMD_FAR;
        }
        group(xa, ya, za, k);
        k += 4;
        xa = xb; ya = yb; za = zb;
    }
    vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
}

// VARIANT 3 (A/B): the j atoms of a segment are first tested, 64 at a time (one per lane), against the bounding box of the wave's
// i atoms; only those within reach of the box become candidate columns.  The segment's window is the box-shaped dilation of the
// chunk (whole neighbour pencils x an x range); ~30 % of its atoms are farther than r from every point of the chunk's box (the corners
// of the diagonal pencils, the ends of the x range) and would be columns without a single hit.  Survivors are taken off the ballot
// mask four at a time (s_ff1 / s_bitset0), their coordinates come through scalar loads as before, and the same packed filter and
// push run on them; a short last group is padded with far-away columns.  Conservative by construction: the box distance is a lower
// bound of every lane's distance, compared against the padded cutoff - results are bit-identical.
// Measured (profiles/r02r_ab_variant3.txt): 31 % fewer columns, but c3 runs at 8.1k instead of 12.5k frames/s: the bit scan and the
// per-survivor addressing cost ~9 SALU instructions per column on top of the push's 5, and an SALU instruction takes the same
// 4.2-cycle issue slot of the SIMD as the slow VALU classes (profiles/r02_valu_calibration.txt) - the loop turns SALU-issue bound;
// twelve one-dword scalar requests per group instead of three 16-byte ones do the rest.  Kept as a checked option, off.
struct vmd_bbox_t { float lox, hix, loy, hiy, loz, hiz, rp2; };

// lowest set bit of a wave-uniform mask (-1 when empty) / clear bit k (k & 63) - one SALU instruction each
#ifndef VMD_NO_INLINE_ASM
__device__ __forceinline__ int vmd_sff1(unsigned long long m) { int k; asm("s_ff1_i32_b64 %0, %1" : "=s"(k) : "s"(m)); return k; }
__device__ __forceinline__ unsigned long long vmd_sbitclr(unsigned long long m, int k) { asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(k)); return m; }
#else
__device__ __forceinline__ int vmd_sff1(unsigned long long m) { return m ? __builtin_ctzll(m) : -1; }
__device__ __forceinline__ unsigned long long vmd_sbitclr(unsigned long long m, int k) { return m & ~(1ull << (k & 63)); }
#endif
// one wave-uniform float at byte offset `off` (s_load_dword sdst, sbase, soffset)
__device__ __forceinline__ float vmd_uniform_load1(vmd_cf32* base, unsigned off) {
    return *(VMD_UNIFORM_AS const float*)((VMD_UNIFORM_AS const char*)base + off);
}

template <int VARIANT, unsigned INC, bool SHIFT, class W>
__device__ __forceinline__ void vmd_segment_pruned(const vmd_pair_params_t& p, W& w, vmd_cf32* tx, vmd_cf32* ty, vmd_cf32* tz,
                                                   unsigned ja, unsigned jb, float sx, float sy, float sz,
                                                   float xi, float yi, float zi, const vmd_bbox_t& bb, int lane) {
    const float r2 = p.r2_up;
    const vmd_f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
    const vmd_f2 sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz};
    // four survivor columns; have < 4: the group is short, the missing columns (which repeat a valid address) are moved out of reach
    auto group = [&](vmd_f4 xj, const vmd_f4& yj, const vmd_f4& zj, int have) {
        if (have < 4) {
            if (have < 2) xj[1] = VMD_FAR;
            if (have < 3) xj[2] = VMD_FAR;
            xj[3] = VMD_FAR;
        }
        vmd_f2 d2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const vmd_f2 xj2 = {xj[2 * h], xj[2 * h + 1]}, yj2 = {yj[2 * h], yj[2 * h + 1]}, zj2 = {zj[2 * h], zj[2 * h + 1]};
            vmd_f2 dx = xi2 - xj2, dy = yi2 - yj2, dz = zi2 - zj2;
            if (SHIFT) { dx = dx - sx2; dy = dy - sy2; dz = dz - sz2; }
            d2[h] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        }
#ifndef VMD_NO_INLINE_ASM
        vmd_push_hot4(w, d2[0][0], d2[0][1], d2[1][0], d2[1][1], r2);
#else
#pragma unroll
        for (int c = 0; c < 4; ++c) vmd_push<VARIANT, INC>(p.bin, w, d2[c >> 1][c & 1] < r2, d2[c >> 1][c & 1]);
#endif
        vmd_drain_full<VARIANT, INC>(p.bin, w, lane);
    };
    // takes up to four survivors off the mask and issues their scalar loads; returns how many were real
    auto fetch = [&](unsigned long long& m, vmd_cf32* px, vmd_cf32* py, vmd_cf32* pz, vmd_f4& xj, vmd_f4& yj, vmd_f4& zj) -> int {
        const int a0 = vmd_sff1(m); m = vmd_sbitclr(m, a0);
        int a1 = vmd_sff1(m); m = vmd_sbitclr(m, a1);
        int a2 = vmd_sff1(m); m = vmd_sbitclr(m, a2);
        int a3 = vmd_sff1(m); m = vmd_sbitclr(m, a3);
        const int have = 1 + (a1 >= 0) + (a2 >= 0) + (a3 >= 0);
        a1 = a1 < 0 ? a0 : a1; a2 = a2 < 0 ? a0 : a2; a3 = a3 < 0 ? a0 : a3;
        const unsigned o0 = 4u * (unsigned)a0, o1 = 4u * (unsigned)a1, o2 = 4u * (unsigned)a2, o3 = 4u * (unsigned)a3;
        xj = vmd_f4{vmd_uniform_load1(px, o0), vmd_uniform_load1(px, o1), vmd_uniform_load1(px, o2), vmd_uniform_load1(px, o3)};
        yj = vmd_f4{vmd_uniform_load1(py, o0), vmd_uniform_load1(py, o1), vmd_uniform_load1(py, o2), vmd_uniform_load1(py, o3)};
        zj = vmd_f4{vmd_uniform_load1(pz, o0), vmd_uniform_load1(pz, o1), vmd_uniform_load1(pz, o2), vmd_uniform_load1(pz, o3)};
        return have;
    };
    // the window after the current one is loaded while the current one is consumed
    auto load = [&](unsigned w0, float& qx, float& qy, float& qz) {
        const unsigned j = w0 + (unsigned)lane;
        const bool valid = j < jb;
        qx = valid ? tx[j] : VMD_FAR; qy = valid ? ty[j] : VMD_FAR; qz = valid ? tz[j] : VMD_FAR;
    };
    float cx, cy, cz;
    load(ja, cx, cy, cz);
    for (unsigned w0 = ja; w0 < jb; w0 += VMD_WAVE) {
        float nx = VMD_FAR, ny = VMD_FAR, nz = VMD_FAR;
        if (w0 + VMD_WAVE < jb) load(w0 + VMD_WAVE, nx, ny, nz);
        float qx = cx, qy = cy, qz = cz;
        if (SHIFT) { qx = qx + sx; qy = qy + sy; qz = qz + sz; }
        const float ex = fmaxf(fmaxf(bb.lox - qx, qx - bb.hix), 0.0f);
        const float ey = fmaxf(fmaxf(bb.loy - qy, qy - bb.hiy), 0.0f);
        const float ez = fmaxf(fmaxf(bb.loz - qz, qz - bb.hiz), 0.0f);
        unsigned long long m = VMD_BALLOT(vmd_d2(ex, ey, ez) <= bb.rp2);      // padding lanes sit at VMD_FAR: never within reach
        vmd_cf32* px = tx + w0;
        vmd_cf32* py = ty + w0;
        vmd_cf32* pz = tz + w0;
        if (m) {
            vmd_f4 xa, ya, za, xb, yb, zb;
            int ha = fetch(m, px, py, pz, xa, ya, za), hb = 0;
            for (;;) {
                const bool more_b = m != 0ull;
                if (more_b) hb = fetch(m, px, py, pz, xb, yb, zb);
                group(xa, ya, za, ha);
                if (!more_b) break;
                const bool more_a = m != 0ull;
                if (more_a) ha = fetch(m, px, py, pz, xa, ya, za);
                group(xb, yb, zb, hb);
                if (!more_a) break;
            }
        }
        cx = nx; cy = ny; cz = nz;
    }
}

template <int VARIANT, unsigned INC, bool MASKED, class W>
__device__ __forceinline__ void vmd_segment(const vmd_pair_params_t& p, W& w, vmd_cf32* st,
                                            unsigned ja, unsigned jb, float sx, float sy, float sz,
                                            float xi, float yi, float zi, unsigned i, int lane) {
    vmd_cf32* tx = st;
    vmd_cf32* ty = st + p.ntgt_pad;
    vmd_cf32* tz = st + 2 * (size_t)p.ntgt_pad;
    if (sx == 0.0f && sy == 0.0f && sz == 0.0f)
        vmd_segment_loop<VARIANT, INC, MASKED, false>(p, w, tx, ty, tz, ja, jb, sx, sy, sz, xi, yi, zi, i, lane);
    else
        vmd_segment_loop<VARIANT, INC, MASKED, true>(p, w, tx, ty, tz, ja, jb, sx, sy, sz, xi, yi, zi, i, lane);
}

// Work distribution: an item is (frame, pencil, sub): the i-chunks sub, sub + nsub, ... of one pencil.  Items are split into
// 8 queues by frame index modulo 8.  A wave first drains the queue of its "home" XCD (blockIdx % 8 — the observed block->XCD
// placement; affinity only, never correctness) so that all pencils of one frame are pulled through ONE XCD's L2, then steals
// from the other queues.  Splitting pencils (nsub > 1) shrinks the number of frames an XCD has in flight (waves / (npen *
// nsub)) so the j streams of the running items stay inside its 4 MiB L2.  Returns the global item id
// ((b * npen + pen) * nsub + sub) or -1 when every queue is empty.  Called by lane 0 only.
__device__ __forceinline__ int vmd_next_item(unsigned* counters, int& q, int& tries, int B, int nitem_frame) {
    while (tries < 8) {
        const int nframes_q = (B - q + 7) >> 3;                 // frames q, q+8, ... < B
        const unsigned t = atomicAdd(&counters[q * VMD_COUNTER_STRIDE], 1u);
        if (nframes_q > 0 && t < (unsigned)(nframes_q * nitem_frame)) {
            const int fq = (int)t / nitem_frame;
            return (q + 8 * fq) * nitem_frame + ((int)t - fq * nitem_frame);
        }
        q = (q + 1) & 7;
        tries += 1;
    }
    return -1;
}

// Register budget, rounds 1 - 5 (wave-private histograms): the kernel wants 106 SGPRs (6 waves/SIMD); at 96 a 7th wave fits and the extra
// spills land in the outer (per work item / per segment) loops: +1.7 % on c2 (profiles/r01s_ab.txt).  Round 6: "Occupancy" below.
// TRI: triclinic cell (SPEC S3t).  Pencils and fine cells live in the unsheared coordinates s_k * L_k; a neighbour pencil's
// periodic image is displaced by the lattice vector (kx, nb, nc), and because the Cartesian x of its atoms is
// s_x*Lx + xy*s_y + xz*s_z, the x window is widened by the range that offset takes over the pencil's cross-section.
// CELL = 2: orthorhombic with open axes (non-periodic systems, slabs): an open axis spans the bounding box of the batch (its
// origin sits in the tilt slot of the box record), has no images, and neighbour pencils end at the box.  CELL = 0 is the fully
// periodic orthorhombic cell and carries none of this.
// SHIST: ONE LDS histogram per block (ds_add is atomic across its four waves) instead of one per wave: 10 KB of LDS per block
// instead of 22.5, so 8 blocks = 8 waves per SIMD fit a CU instead of 7.
// Occupancy (round 6): with ONE histogram per block (SHIST: 10 KB of LDS instead of 22.5) eight blocks fit a CU, and the compiler meets the
// register budget of an eighth wave per SIMD (64 VGPRs, 80 SGPRs; 111 instead of 92 SGPRs spilled to VGPR lanes, nothing to scratch) when it
// is told to: c3 71.7 -> 70.4 ms per 1 000 frames, c2 7.35 -> 7.23 (profiles/r06u_eight_waves_ab.txt).  Round 2 had measured the shared
// histogram at seven waves (no gain) and an SGPR cap of 80 on the kernel of that time (no gain); the two together are the default now.
// Wave-private histograms keep the budget of seven waves (96 SGPRs, as capped since round 1: LDS admits no more).
#ifndef VMD_NO_INLINE_ASM
#define VMD_PENCIL_OCC(SH) __attribute__((amdgpu_waves_per_eu((SH) ? 8 : 7, (SH) ? 8 : 7)))      /* (amdgpu_num_sgpr takes no template-dependent value; seven waves = the cap of 96 SGPRs of rounds 1 - 5) */
#else
#define VMD_PENCIL_OCC(SH)
#endif
template <int VARIANT_, bool SAME, int CELL, bool SHIST, int POP = 0>
__global__ __launch_bounds__(256) VMD_PENCIL_OCC(SHIST) void k_rdf_pencil(vmd_pair_params_t p) {
    constexpr bool TRI = CELL == 1, OPEN = CELL == 2;
    constexpr bool PRUNE = VARIANT_ == 3;            // variant 3 = variant 0 + bounding-box pruning of the j windows
    constexpr int VARIANT = PRUNE ? 0 : VARIANT_;
    __shared__ unsigned s_hist[SHIST ? 1 : 4][VMD_MAX_BINS + 2];       // + the spare bin of vmd_pop_hot0 (index nbins; never read)
    // (looking at the stack once per EIGHT columns - a 640-float stack, which the one-histogram kernels have the LDS for - measured 1 % slower
    // than once per four: profiles/r06v_salu_ab.txt)
    __shared__ float s_queue[4][VMD_QUEUE_CAP];
    constexpr unsigned INC = SAME ? 2u : 1u;
    if (p.skip && *p.skip) return;       // set before this launch by the cell build; the host repeats the batch with larger buckets

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int nbins = p.bin.nbins;

    vmd_wave_acc_tt<POP> w;
    w.hist = s_hist[SHIST ? 0 : wave];
    w.queue = s_queue[wave];
    w.qbase = VMD_LDS_ADDRESS(s_queue[wave]);
    w.hbase = VMD_LDS_ADDRESS(s_hist[SHIST ? 0 : wave]);
    w.qtop = w.qbase;
    w.qlim = w.qbase + 4u * VMD_WAVE;
    w.slow = &s_queue[wave][VMD_QUEUE_CAP - VMD_WAVE];
    w.nslow = 0;
    w.fast_c = vmd_in_vgpr(p.bin.fast_c);
    w.fast_k = vmd_in_vgpr(p.bin.fast_k);
    w.fast_far = vmd_in_vgpr(p.bin.fast_far);
    w.fast_c2 = vmd_in_vgpr(p.bin.fast_c2);
    w.ncols = 0;
    if (SHIST) {
        for (int b = threadIdx.x; b < nbins; b += 256) s_hist[0][b] = 0u;
        __syncthreads();
    } else {
        for (int b = lane; b < nbins; b += VMD_WAVE) w.hist[b] = 0u;
        __builtin_amdgcn_wave_barrier();
    }

    const int nxf = p.grid.nxf, ny = p.grid.ny, nz = p.grid.nz;
    const int npen = ny * nz;

    unsigned long long cols_before_flush = 0ull;
    // work items are handed out dynamically (one returning atomic per item, fetched one item ahead)
    int q = blockIdx.x & 7, tries = 0;
    int item = -1, next_item = -1;
    const int nsub = p.nsub;
    const int nsplit = p.nsplit;
    const int nitem_frame = npen * nsub * nsplit;
    if (lane == 0) item = vmd_next_item(p.work_counter, q, tries, p.B, nitem_frame);
    item = __builtin_amdgcn_readfirstlane(item);
    for (; item >= 0; item = next_item) {
        if (lane == 0) next_item = vmd_next_item(p.work_counter, q, tries, p.B, nitem_frame);
        const int b = item / nitem_frame;
        const int rem0 = item - b * nitem_frame;
        const int rem = rem0 / nsplit;
        const int part = rem0 - rem * nsplit;               // which of the chunk's neighbour pencils this item walks (fastest: the parts of
        const int pen = rem / nsub;                         // a chunk go to consecutive waves, which share its i atoms and its neighbourhood)
        const int sub = rem - pen * nsub;
        const int pz = pen / ny;
        const int py = pen - pz * ny;
        vmd_cf32* boxes = (vmd_cf32*)p.boxes;
        const float Lx = boxes[VMD_BOX_STRIDE * b + 0], Ly = boxes[VMD_BOX_STRIDE * b + 1], Lz = boxes[VMD_BOX_STRIDE * b + 2];
        const float inv_cx = (float)nxf * boxes[VMD_BOX_STRIDE * b + 3];
        const float txy = TRI ? boxes[VMD_BOX_STRIDE * b + 6] : 0.0f, txz = TRI ? boxes[VMD_BOX_STRIDE * b + 7] : 0.0f,
                    tyz = TRI ? boxes[VMD_BOX_STRIDE * b + 8] : 0.0f;
        // open x axis: fine cells count from the bounding-box origin (0 on a periodic axis)
        const bool open_x = OPEN && !(p.pbc & 1u), open_y = OPEN && !(p.pbc & 2u), open_z = OPEN && !(p.pbc & 4u);
        const float orgx = open_x ? boxes[VMD_BOX_STRIDE * b + 6] : 0.0f;
        // open x axis: coordinates are not wrapped, so x - origin is rounded at the magnitude of the raw coordinate, in the
        // build and again here; widen the window by a few such ulps (a system far from the origin must not lose a boundary pair)
        const float pad_open = open_x ? 8.0e-7f * (fabsf(orgx) + Lx) : 0.0f;
        vmd_cu32* csr = (vmd_cu32*)p.cs_ref + (size_t)b * (p.grid.ncell + 1);
        vmd_cu32* cst = (vmd_cu32*)p.cs_tgt + (size_t)b * (p.grid.ncell + 1);
        const float* __restrict__ sr = p.sref + (size_t)b * 3 * p.nref_pad;
        vmd_cf32* st = (vmd_cf32*)p.stgt + (size_t)b * 3 * p.ntgt_pad;
        const unsigned pbeg = csr[pen * nxf];
        const unsigned pend = csr[(pen + 1) * nxf];

        for (unsigned cbeg = pbeg + (unsigned)sub * VMD_WAVE; cbeg < pend; cbeg += (unsigned)nsub * VMD_WAVE) {
            const unsigned i = cbeg + lane;
            const bool valid = i < pend;
            const float xi = valid ? sr[i] : VMD_FAR;
            const float yi = valid ? sr[p.nref_pad + i] : VMD_FAR;
            const float zi = valid ? sr[2 * (size_t)p.nref_pad + i] : VMD_FAR;
            const float xlo = vmd_uniform(vmd_wave_min(valid ? xi : 3.0e38f));
            const float xhi = vmd_uniform(vmd_wave_max(valid ? xi : -3.0e38f));
            vmd_bbox_t bb = {};
            if (PRUNE) {
                bb.lox = xlo; bb.hix = xhi;
                bb.loy = vmd_uniform(vmd_wave_min(valid ? yi : 3.0e38f)); bb.hiy = vmd_uniform(vmd_wave_max(valid ? yi : -3.0e38f));
                bb.loz = vmd_uniform(vmd_wave_min(valid ? zi : 3.0e38f)); bb.hiz = vmd_uniform(vmd_wave_max(valid ? zi : -3.0e38f));
                // reach of the box test: the padded cutoff + the rounding of coordinates of this magnitude (images included)
                const float mag = fmaxf(fmaxf(fmaxf(fabsf(bb.lox), fabsf(bb.hix)), fmaxf(fabsf(bb.loy), fabsf(bb.hiy))), fmaxf(fabsf(bb.loz), fabsf(bb.hiz)))
                                  + Lx + Ly + Lz + fabsf(txy) + fabsf(txz) + fabsf(tyz);
                const float rp = p.rpad + 4.0e-6f * mag;
                bb.rp2 = rp * rp;
            }

            for (int dz = SAME ? 0 : -p.rz; dz <= p.rz; ++dz) {
                int qz = pz + dz; float sz = 0.0f, nc = 0.0f;
                if (open_z && (qz < 0 || qz >= nz)) continue;       // nothing beyond the bounding box
                if (qz < 0) { qz += nz; sz = -Lz; nc = -1.0f; } else if (qz >= nz) { qz -= nz; sz = Lz; nc = 1.0f; }
                for (int dy = -p.ry; dy <= p.ry; ++dy) {
                    if (SAME && dz == 0 && dy < 0) continue;
                    if (nsplit > 1 && ((dz + p.rz) * (2 * p.ry + 1) + (dy + p.ry)) % nsplit != part) continue;
                    int qy = py + dy; float sy = 0.0f, nb = 0.0f;
                    if (open_y && (qy < 0 || qy >= ny)) continue;
                    if (qy < 0) { qy += ny; sy = -Ly; nb = -1.0f; } else if (qy >= ny) { qy -= ny; sy = Ly; nb = 1.0f; }
                    const bool own = SAME && dz == 0 && dy == 0;
                    const int q = qz * ny + qy;
                    float offmin = 0.0f, offmax = 0.0f;
                    // split pencils: a neighbour two pencils away is at least one pencil width off in that axis, so its x window shrinks
                    // (orthorhombic periodic cells only; the widths of the other cell kinds are not the box edge over the count)
                    float rpad = p.rpad;
                    if (CELL == 0 && (dy > 1 || dy < -1 || dz > 1 || dz < -1)) {
                        const float gy = (float)((dy < 0 ? -dy : dy) - 1) * (Ly / (float)ny), gz = (float)((dz < 0 ? -dz : dz) - 1) * (Lz / (float)nz);
                        const float gyy = gy > 0.0f ? gy : 0.0f, gzz = gz > 0.0f ? gz : 0.0f;
                        const float rr = p.rpad * p.rpad - 0.998f * (gyy * gyy + gzz * gzz);
                        if (rr <= 0.0f) continue;
                        rpad = sqrtf(rr) * 1.0001f;
                    }
                    if (TRI) {
                        // range of xy*s_y + xz*s_z over the cross-section of pencil q (+ head room for the roundings)
                        const float y0 = txy * ((float)qy / (float)ny), y1 = txy * ((float)(qy + 1) / (float)ny);
                        const float z0 = txz * ((float)qz / (float)nz), z1 = txz * ((float)(qz + 1) / (float)nz);
                        offmin = fminf(y0, y1) + fminf(z0, z1) - 1.0e-3f;
                        offmax = fmaxf(y0, y1) + fmaxf(z0, z1) + 1.0e-3f;
                    }
                    for (int kx = -1; kx <= 1; ++kx) {
                        if (open_x && kx != 0) continue;
                        float sx = (float)kx * Lx;
                        if (TRI) vmd_lattice_shift(Lx, Ly, Lz, txy, txz, tyz, (float)kx, nb, nc, sx, sy, sz);
                        const float lo = (xlo - rpad) - sx - offmax - orgx - pad_open;
                        const float hi = (xhi + rpad) - sx - offmin - orgx + pad_open;
                        if (hi < 0.0f || lo >= Lx) continue;
                        const int ca = lo <= 0.0f ? 0 : vmd_cell_coord(lo, inv_cx, nxf);
                        const int cb = hi >= Lx ? nxf - 1 : vmd_cell_coord(hi, inv_cx, nxf);
                        unsigned ja = cst[q * nxf + ca];
                        const unsigned jb = cst[q * nxf + cb + 1];
                        if (own) {
                            // unordered pairs once: j > i.  Inside the chunk the test is per lane, above it all lanes pass.
                            const unsigned cend = cbeg + VMD_WAVE;
                            const unsigned ma = ja > cbeg ? ja : cbeg;
                            const unsigned mb = jb < cend ? jb : cend;
                            if (ma < mb) {
                                w.ncols += mb - ma;
                                vmd_segment<VARIANT, INC, true>(p, w, st, ma, mb, sx, sy, sz, xi, yi, zi, i, lane);
                            }
                            ja = ja > cend ? ja : cend;
                        }
                        if (ja < jb) {
                            w.ncols += jb - ja;
                            if (PRUNE) {
                                vmd_cf32* tx = st; vmd_cf32* ty = st + p.ntgt_pad; vmd_cf32* tz = st + 2 * (size_t)p.ntgt_pad;
                                if (sx == 0.0f && sy == 0.0f && sz == 0.0f) vmd_segment_pruned<VARIANT, INC, false>(p, w, tx, ty, tz, ja, jb, sx, sy, sz, xi, yi, zi, bb, lane);
                                else vmd_segment_pruned<VARIANT, INC, true>(p, w, tx, ty, tz, ja, jb, sx, sy, sz, xi, yi, zi, bb, lane);
                            } else {
                                vmd_segment<VARIANT, INC, false>(p, w, st, ja, jb, sx, sy, sz, xi, yi, zi, i, lane);
                            }
                        }
                    }
                }
            }
            // u32 LDS counters: every candidate column adds at most 64*INC; flush long before 2^32 (rare: straight to the
            // device accumulators with atomics)
            // (a shared histogram receives the columns of four waves: a quarter of the budget each, and the flush takes the bins
            // with an atomic exchange - an increment of another wave lands either before it, and travels now, or after it, and stays)
            if (w.ncols >= (SHIST ? (1u << 21) : (1u << 23))) {
                cols_before_flush += w.ncols;
                vmd_drain<VARIANT, INC>(p.bin, w, lane);
                __builtin_amdgcn_wave_barrier();
                for (int bb = lane; bb < nbins; bb += VMD_WAVE) {
                    const unsigned v = SHIST ? atomicExch(&w.hist[bb], 0u) : w.hist[bb];
                    if (v) atomicAdd(&p.counts[bb], (unsigned long long)v);
                    if (!SHIST) w.hist[bb] = 0u;
                }
                __builtin_amdgcn_wave_barrier();
                w.ncols = 0;
            }
        }
        next_item = __builtin_amdgcn_readfirstlane(next_item);
    }
    vmd_drain<VARIANT, INC>(p.bin, w, lane);
    if (p.cols_total && lane == 0) atomicAdd(p.cols_total, cols_before_flush + (unsigned long long)w.ncols);
    // one row per block: the four wave histograms are summed through LDS and stored with plain, coalesced writes
    __syncthreads();
    uint64_t* __restrict__ prow = p.partial + (size_t)blockIdx.x * nbins;
    for (int bb = threadIdx.x; bb < nbins; bb += 256) {
        uint64_t v = s_hist[0][bb];
        if (!SHIST) v = v + s_hist[1][bb] + s_hist[2][bb] + s_hist[3][bb];
        prow[bb] = v;
    }
}

// sum the per-block partial rows into the u64 accumulators: block (x = 256 bins, y = slice of 32 rows), coalesced
// row reads, one atomicAdd(u64) per bin and slice
__global__ __launch_bounds__(256) void k_hist_reduce(const uint64_t* __restrict__ partial, int nrows, int nbins,
                                                     uint64_t* __restrict__ counts, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nbins) return;
    const int r0 = blockIdx.y * 32;
    const int r1 = r0 + 32 < nrows ? r0 + 32 : nrows;
    uint64_t s = 0;
    for (int r = r0; r < r1; ++r) s += partial[(size_t)r * nbins + b];
    if (s) atomicAdd((unsigned long long*)&counts[b], (unsigned long long)s);
}

// ------------------------------------------------------------------------------------------------ RDF, general (brute)

struct vmd_brute_params_t {
    const float* xyz; size_t frame_stride; size_t row_stride;
    const float* boxes; uint32_t pbc; int B;
    const int32_t* ref; int nref; const int32_t* tgt; int ntgt;
    vmd_binning_t bin;
    uint64_t* counts;
    int raw;            // DECISION(D-WRAP) flipped: positions as they are, minimum image by rounding (oracle: vo_set_spec("rdf_raw", 1))
};

__global__ __launch_bounds__(256) void k_rdf_brute(vmd_brute_params_t p) {
    __shared__ unsigned s_hist[VMD_MAX_BINS];
    __shared__ float s_t[3][256];
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const float* fy = fx + p.row_stride;
    const float* fz = fy + p.row_stride;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    for (int k = threadIdx.x; k < p.bin.nbins; k += 256) s_hist[k] = 0u;
    float xi = VMD_FAR, yi = VMD_FAR, zi = VMD_FAR;
    const bool valid = t < p.nref;
    if (valid) {
        const int a = p.ref ? p.ref[t] : t;
        if (p.raw) { xi = fx[a]; yi = fy[a]; zi = fz[a]; }
        else vmd_pair_coords(bx, fx[a], fy[a], fz[a], xi, yi, zi);
    }
    for (int j0 = 0; j0 < p.ntgt; j0 += 256) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        if (j < p.ntgt) {
            const int a = p.tgt ? p.tgt[j] : j;
            if (p.raw) { s_t[0][threadIdx.x] = fx[a]; s_t[1][threadIdx.x] = fy[a]; s_t[2][threadIdx.x] = fz[a]; }
            else vmd_pair_coords(bx, fx[a], fy[a], fz[a], s_t[0][threadIdx.x], s_t[1][threadIdx.x], s_t[2][threadIdx.x]);
        }
        __syncthreads();
        const int nj = p.ntgt - j0 < 256 ? p.ntgt - j0 : 256;
        if (valid) {
            for (int jj = 0; jj < nj; ++jj) {
                float d2;
                if (p.raw && !bx.tri) {
                    float dx = xi - s_t[0][jj], dy = yi - s_t[1][jj], dz = zi - s_t[2][jj];
                    dx = vmd_mi_rintf(dx, bx.Lx, bx.iLx, bx.px); dy = vmd_mi_rintf(dy, bx.Ly, bx.iLy, bx.py); dz = vmd_mi_rintf(dz, bx.Lz, bx.iLz, bx.pz);
                    d2 = vmd_d2(dx, dy, dz);
                } else d2 = vmd_pair_d2_general(bx, xi, yi, zi, s_t[0][jj], s_t[1][jj], s_t[2][jj]);      // S3t takes positions as they come
                const int bin = vmd_bin_of(p.bin, d2);
                if (bin >= 0) atomicAdd(&s_hist[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < p.bin.nbins; k += 256) {
        const unsigned v = s_hist[k];
        if (v) atomicAdd((unsigned long long*)&p.counts[k], (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------------ K3: SDF alignment (fp64)

// cyclic Jacobi on a symmetric 4x4 — identical operation order to oracle vo_jacobi4
__device__ void vmd_jacobi4(double A[4][4], double V[4][4]) {
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

__device__ void vmd_horn_rotation(const double S[3][3], double R[9]) {
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
    vmd_jacobi4(N, V);
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

// walks the unwrap chain of one structure; calls f(a, w, px, py, pz) for every atom in order
template <typename F>
__device__ __forceinline__ void vmd_unwrap_chain(const float* fx, const float* fy, const float* fz,
                                                 const int32_t* idx, const float* mass, int m, const vmd_box_t& bx, F f) {
    double qx = 0.0, qy = 0.0, qz = 0.0;
    for (int a = 0; a < m; ++a) {
        const int i = idx[a];
        double x = (double)fx[i], y = (double)fy[i], z = (double)fz[i];
        if (a > 0) {
            double dx = x - qx, dy = y - qy, dz = z - qz;
            vmd_mi3_rint(bx, dx, dy, dz);
            x = qx + dx; y = qy + dy; z = qz + dz;
        }
        qx = x; qy = y; qz = z;
        f(a, mass ? (double)mass[a] : 1.0, x, y, z);
    }
}

// DECISION(D-SDF-UNWRAP) as a switch: with the bonds of the system handed over (vmd_system_t::bonds) a structure is made whole along
// its bond tree - atom order[t] hangs on atom parent[order[t]] (local indices; parent < 0 = the root) - as mdlib's
// md_util_unwrap does (/root/reference/src/viamd.cpp:2257), instead of along the index order.  Positions go to `pos` (m x 3 doubles
// of scratch per thread: a parent is not the atom visited last); the callers then sum in INDEX order, as the chain walk does.
__device__ __forceinline__ void vmd_unwrap_tree(const float* fx, const float* fy, const float* fz, const int32_t* idx, int m, const vmd_box_t& bx,
                                                const int32_t* order, const int32_t* parent, double* pos) {
    for (int t = 0; t < m; ++t) {
        const int a = order[t], par = parent[a];
        const int i = idx[a];
        double x = (double)fx[i], y = (double)fy[i], z = (double)fz[i];
        if (par >= 0) {
            const double qx = pos[3 * par + 0], qy = pos[3 * par + 1], qz = pos[3 * par + 2];
            double dx = x - qx, dy = y - qy, dz = z - qz;
            vmd_mi3_rint(bx, dx, dy, dz);
            x = qx + dx; y = qy + dy; z = qz + dz;
        }
        pos[3 * a + 0] = x; pos[3 * a + 1] = y; pos[3 * a + 2] = z;
    }
}

struct vmd_align_params_t {
    const float* xyz; size_t frame_stride; size_t row_stride;
    const float* boxes; uint32_t pbc; int B;
    const int32_t* structs; const float* mass; int K; int m;
    const double* ref_pose;
    float* R32; float* c32; double* M64;
    const int32_t* tree_order; const int32_t* tree_parent;   // [K][m] each, or NULL: unwrap along the index order
    double* tree_pos;                                         // [B*K][m][3] scratch of the tree walk
};

__global__ __launch_bounds__(64) void k_sdf_align(vmd_align_params_t p) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= p.B * p.K) return;
    const int b = t / p.K, k = t - b * p.K;
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const float* fy = fx + p.row_stride;
    const float* fz = fy + p.row_stride;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const int32_t* idx = p.structs + (size_t)k * p.m;
    const float* mass = p.mass ? p.mass + (size_t)k * p.m : nullptr;

    double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const double* ref = p.ref_pose;
    double com0, com1, com2;
    auto add_com = [&](int, double w, double x, double y, double z) { sw = sw + w; sx = sx + w * x; sy = sy + w * y; sz = sz + w * z; };
    auto add_cov = [&](int a, double w, double x, double y, double z) {
        const double c0 = x - com0, c1 = y - com1, c2 = z - com2;
        const double r0 = ref[3 * a + 0], r1 = ref[3 * a + 1], r2 = ref[3 * a + 2];
        const double wc0 = w * c0, wc1 = w * c1, wc2 = w * c2;
        S[0][0] = S[0][0] + wc0 * r0; S[0][1] = S[0][1] + wc0 * r1; S[0][2] = S[0][2] + wc0 * r2;
        S[1][0] = S[1][0] + wc1 * r0; S[1][1] = S[1][1] + wc1 * r1; S[1][2] = S[1][2] + wc1 * r2;
        S[2][0] = S[2][0] + wc2 * r0; S[2][1] = S[2][1] + wc2 * r1; S[2][2] = S[2][2] + wc2 * r2;
    };
    if (p.tree_order) {
        double* pos = p.tree_pos + (size_t)t * 3 * p.m;
        vmd_unwrap_tree(fx, fy, fz, idx, p.m, bx, p.tree_order + (size_t)k * p.m, p.tree_parent + (size_t)k * p.m, pos);
        for (int a = 0; a < p.m; ++a) add_com(a, mass ? (double)mass[a] : 1.0, pos[3 * a + 0], pos[3 * a + 1], pos[3 * a + 2]);
        com0 = sx / sw; com1 = sy / sw; com2 = sz / sw;
        for (int a = 0; a < p.m; ++a) add_cov(a, mass ? (double)mass[a] : 1.0, pos[3 * a + 0], pos[3 * a + 1], pos[3 * a + 2]);
    } else {
        vmd_unwrap_chain(fx, fy, fz, idx, mass, p.m, bx, add_com);
        com0 = sx / sw; com1 = sy / sw; com2 = sz / sw;
        vmd_unwrap_chain(fx, fy, fz, idx, mass, p.m, bx, add_cov);
    }
    double R[9];
    vmd_horn_rotation(S, R);
    float* R32 = p.R32 + (size_t)t * 9;
    float* c32 = p.c32 + (size_t)t * 3;
    for (int i = 0; i < 9; ++i) R32[i] = (float)R[i];
    c32[0] = (float)com0; c32[1] = (float)com1; c32[2] = (float)com2;
    if (p.M64) {
        double* M = p.M64 + (size_t)t * 12;
        for (int r = 0; r < 3; ++r) {
            M[4 * r + 0] = R[3 * r + 0]; M[4 * r + 1] = R[3 * r + 1]; M[4 * r + 2] = R[3 * r + 2];
            M[4 * r + 3] = -(R[3 * r + 0] * com0 + R[3 * r + 1] * com1 + R[3 * r + 2] * com2);
        }
    }
}

__global__ __launch_bounds__(64) void k_sdf_ref_pose(const float* xyz, size_t row_stride, const float* box, uint32_t pbc,
                                                     const int32_t* idx, const float* mass, int m, double* ref_pose,
                                                     const int32_t* tree_order, const int32_t* tree_parent) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float* fx = xyz;
    const float* fy = fx + row_stride;
    const float* fz = fy + row_stride;
    const vmd_box_t bx = vmd_load_box(box, 0, pbc);
    double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    if (tree_order) {
        vmd_unwrap_tree(fx, fy, fz, idx, m, bx, tree_order, tree_parent, ref_pose);      // the pose array doubles as the walk's scratch
        for (int a = 0; a < m; ++a) {
            const double w = mass ? (double)mass[a] : 1.0;
            sw = sw + w; sx = sx + w * ref_pose[3 * a + 0]; sy = sy + w * ref_pose[3 * a + 1]; sz = sz + w * ref_pose[3 * a + 2];
        }
    } else {
        vmd_unwrap_chain(fx, fy, fz, idx, mass, m, bx,
                         [&](int a, double w, double x, double y, double z) {
                             ref_pose[3 * a + 0] = x; ref_pose[3 * a + 1] = y; ref_pose[3 * a + 2] = z;
                             sw = sw + w; sx = sx + w * x; sy = sy + w * y; sz = sz + w * z;
                         });
    }
    const double com0 = sx / sw, com1 = sy / sw, com2 = sz / sw;
    for (int a = 0; a < m; ++a) {
        ref_pose[3 * a + 0] = ref_pose[3 * a + 0] - com0;
        ref_pose[3 * a + 1] = ref_pose[3 * a + 1] - com1;
        ref_pose[3 * a + 2] = ref_pose[3 * a + 2] - com2;
    }
}

// Per frame: C = COM of structure 0 and R = max_k |mi(c_k - C)| (padded).  An atom farther than r + R from C is farther than r
// from every c_k (triangle inequality of the minimum-image metric), so the scatter can drop it with ONE test instead of K.
__global__ __launch_bounds__(64) void k_sdf_group(const float* __restrict__ c32, const float* __restrict__ boxes, uint32_t pbc,
                                                  int B, int K, float* __restrict__ group) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const vmd_box_t bx = vmd_load_box(boxes, b, pbc);
    const float* c0 = c32 + (size_t)b * K * 3;
    float r2max = 0.0f;
    for (int k = 1; k < K; ++k) {
        const float* c = c0 + 3 * k;
        float dx = c[0] - c0[0], dy = c[1] - c0[1], dz = c[2] - c0[2];
        vmd_mi3_rintf(bx, dx, dy, dz);
        r2max = fmaxf(r2max, vmd_d2(dx, dy, dz));
    }
    float* g = group + 4 * (size_t)b;
    g[0] = c0[0]; g[1] = c0[1]; g[2] = c0[2];
    g[3] = sqrtf(r2max) * 1.0005f + 1.0e-2f;
}

// ------------------------------------------------------------------------------------------------ K4: SDF scatter

#ifndef VMD_LOAD_NT
#define VMD_LOAD_NT(ptr) __builtin_nontemporal_load(ptr)
#endif
struct vmd_scatter_params_t {
    const float* __restrict__ xyz; size_t frame_stride; size_t row_stride;
    const float* __restrict__ boxes; uint32_t pbc; int B;
    const int32_t* __restrict__ structs; int K; int m;
    const float* __restrict__ R32; const float* __restrict__ c32;
    const int32_t* __restrict__ tgt; const int8_t* __restrict__ owner; int ntgt; float extent; int dim;
    unsigned long long* volume;
    const float* __restrict__ group;   // f32[B][4] from k_sdf_group, or NULL
    int tgt_first, tgt_stride;         // tgt_stride > 0: target t is atom tgt_first + t * tgt_stride (no index list to chase)
    int unowned;                       // 1: no target belongs to any structure (the exclusion rule never applies)
    int nt;                            // 1: the frame is read with non-temporal loads (a pure stream: nothing of it is read twice)
};

// one target atom against structure k of frame b (SPEC S5 scatter).  own_k: structure the atom belongs to (-1 none,
// -2 unknown: search the index list)
__device__ __forceinline__ void vmd_sdf_atom_k(const vmd_scatter_params_t& p, const vmd_box_t& bx, int b, int k, float x, float y, float z,
                                               int own_k, int i) {
    // SPEC D-SDF-EXCL: a target atom is skipped for the structure it belongs to
    if (own_k == k) return;
    if (own_k == -2) {
        const int32_t* sidx = p.structs + (size_t)k * p.m;
        bool own = false;
        for (int a = 0; a < p.m; ++a) own = own || (sidx[a] == i);
        if (own) return;
    }
    const float s = p.extent;
    const float* R = p.R32 + ((size_t)b * p.K + k) * 9;
    const float* c = p.c32 + ((size_t)b * p.K + k) * 3;
    float dx = x - c[0], dy = y - c[1], dz = z - c[2];
    vmd_mi3_rintf(bx, dx, dy, dz);
    const float qx = fmaf(R[2], dz, fmaf(R[1], dy, R[0] * dx));
    const float qy = fmaf(R[5], dz, fmaf(R[4], dy, R[3] * dx));
    const float qz = fmaf(R[8], dz, fmaf(R[7], dy, R[6] * dx));
    const float vscale = (float)p.dim / (2.0f * s);
    const float fdim = (float)p.dim;
    const float tx = (qx + s) * vscale;
    const float ty = (qy + s) * vscale;
    const float tz = (qz + s) * vscale;
    if (tx >= 0.0f && tx < fdim && ty >= 0.0f && ty < fdim && tz >= 0.0f && tz < fdim) {
        const int vx = (int)tx, vy = (int)ty, vz = (int)tz;
        atomicAdd(&p.volume[((size_t)vz * p.dim + vy) * p.dim + vx], 1ull);
    }
}

// group pre-filter (k_sdf_group): can this atom reach ANY structure's cube?  A voxel hit needs |q|_inf < s, hence
// |d| = |q| < sqrt(3) s; with the group sphere (centre g, radius g[3]) one distance test replaces K.
// The argument is the triangle inequality of the minimum-image metric.  Per-axis rounding (orthorhombic) IS that metric;
// rounding in fractional space (triclinic, S5) only agrees with it for vectors shorter than half the smallest cell width, so
// in a triclinic cell the filter is used only while the whole reach stays below that (else every atom goes to the K tests).
__device__ __forceinline__ bool vmd_sdf_near(const vmd_scatter_params_t& p, const vmd_box_t& bx, int b, float x, float y, float z) {
    if (!p.group) return true;
    const float* g = p.group + 4 * (size_t)b;
    const float reach = (1.7320508f * p.extent * 1.0005f + 1.0e-3f) + g[3];
    if (bx.tri) {
        const float ty = bx.yz * bx.iLz, tx1 = bx.xy * bx.iLy, tx2 = (bx.xy * bx.yz - bx.Ly * bx.xz) * (bx.iLy * bx.iLz);
        const float wy = bx.Ly / sqrtf(1.0f + ty * ty), wx = bx.Lx / sqrtf(1.0f + tx1 * tx1 + tx2 * tx2);
        const float wmin = fminf(bx.Lz, fminf(wx, wy));
        if (!(reach < 0.499f * wmin)) return true;
    }
    float dx = x - g[0], dy = y - g[1], dz = z - g[2];
    vmd_mi3_rintf(bx, dx, dy, dz);
    return vmd_d2(dx, dy, dz) <= reach * reach;
}

// Target atoms come in index order, i.e. spatially random: nearly every wave holds a few atoms near the structures, so
// running the K transforms under the divergent mask would cost every wave the full loop at ~5 % lane use.  Instead a block
// first compacts the atoms that pass the group test into LDS (1024 candidates per block, all gathers in flight at once),
// then spreads the (atom, structure) pairs evenly over its threads.
// ARITH: the target list is an arithmetic progression (every water oxygen of a regular solvent box: first + 3 t), so the atom index
// is computed instead of loaded and the coordinate gathers do not wait for an index load - one memory round trip per block
// instead of two (the kernel is bound by the latency of its loads under load: a wave used to live ~12 us).
// ILP atoms per thread: all their gathers are in flight at once.
#define VMD_SDF_CAP 512      // survivors of the group test a block holds in LDS at a time (~1 % of its atoms pass; more take another round)
template <int ILP, bool ARITH>
__global__ __launch_bounds__(256) void k_sdf_scatter(vmd_scatter_params_t p) {
    __shared__ float s_x[VMD_SDF_CAP], s_y[VMD_SDF_CAP], s_z[VMD_SDF_CAP];
    __shared__ int s_own[VMD_SDF_CAP], s_idx[VMD_SDF_CAP];
    __shared__ unsigned s_n;
    const int t0 = blockIdx.x * (256 * ILP) + threadIdx.x;
    const int b = blockIdx.y;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    int idx[ILP], own[ILP];
    float x[ILP], y[ILP], z[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        const int t = t0 + 256 * u;
        idx[u] = -1; own[u] = p.unowned ? -1 : -2;
        if (t < p.ntgt) {
            idx[u] = ARITH ? p.tgt_first + t * p.tgt_stride : (p.tgt ? p.tgt[t] : t);
            if (!p.unowned && p.owner) own[u] = (int)p.owner[t];
        }
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        x[u] = y[u] = z[u] = 0.0f;
        if (idx[u] >= 0) {
            if (p.nt) {
                x[u] = VMD_LOAD_NT(fx + idx[u]); y[u] = VMD_LOAD_NT(fx + p.row_stride + idx[u]); z[u] = VMD_LOAD_NT(fx + 2 * p.row_stride + idx[u]);
            } else {
                x[u] = fx[idx[u]]; y[u] = fx[p.row_stride + idx[u]]; z[u] = fx[2 * p.row_stride + idx[u]];
            }
        }
    }
    unsigned pending = 0u;
#pragma unroll
    for (int u = 0; u < ILP; ++u) if (idx[u] >= 0 && vmd_sdf_near(p, bx, b, x[u], y[u], z[u])) pending |= 1u << u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            if (pending & (1u << u)) {
                const unsigned slot = atomicAdd(&s_n, 1u);
                if (slot < VMD_SDF_CAP) {
                    s_x[slot] = x[u]; s_y[slot] = y[u]; s_z[slot] = z[u]; s_own[slot] = own[u]; s_idx[slot] = idx[u];
                    pending &= ~(1u << u);
                }
            }
        }
        __syncthreads();
        const unsigned total = s_n;
        const int nwork = (int)(total < VMD_SDF_CAP ? total : VMD_SDF_CAP) * p.K;
        for (int w = threadIdx.x; w < nwork; w += 256) {
            const int a = w / p.K, k = w - a * p.K;
            vmd_sdf_atom_k(p, bx, b, k, s_x[a], s_y[a], s_z[a], s_own[a], s_idx[a]);
        }
        if (total <= VMD_SDF_CAP) break;      // block-uniform: everybody read the same s_n
    }
}

// Wave-level variant (A/B, `sdf_wave`): the survivors of the group test are compacted per WAVE (ballot + prefix, 64 slots of LDS per
// wave and round) instead of per block, so the kernel has no block barrier at all - a wave that found nothing near the structures
// (most of them: ~1 % of the atoms pass) retires as soon as its loads are tested.
template <int ILP, bool ARITH>
__global__ __launch_bounds__(256) void k_sdf_scatter_wave(vmd_scatter_params_t p) {
    __shared__ float s_x[4][VMD_WAVE], s_y[4][VMD_WAVE], s_z[4][VMD_WAVE];
    __shared__ int s_own[4][VMD_WAVE], s_idx[4][VMD_WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t0 = blockIdx.x * (256 * ILP) + wave * (VMD_WAVE * ILP) + lane;      // a wave owns 64 * ILP consecutive targets
    const int b = blockIdx.y;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    int idx[ILP], own[ILP];
    float x[ILP], y[ILP], z[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        const int t = t0 + VMD_WAVE * u;
        idx[u] = -1; own[u] = p.unowned ? -1 : -2;
        if (t < p.ntgt) {
            idx[u] = ARITH ? p.tgt_first + t * p.tgt_stride : (p.tgt ? p.tgt[t] : t);
            if (!p.unowned && p.owner) own[u] = (int)p.owner[t];
        }
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        x[u] = y[u] = z[u] = 0.0f;
        if (idx[u] >= 0) { x[u] = fx[idx[u]]; y[u] = fx[p.row_stride + idx[u]]; z[u] = fx[2 * p.row_stride + idx[u]]; }
    }
    unsigned pending = 0u;
#pragma unroll
    for (int u = 0; u < ILP; ++u) if (idx[u] >= 0 && vmd_sdf_near(p, bx, b, x[u], y[u], z[u])) pending |= 1u << u;
    for (;;) {
        unsigned base = 0u;                                   // wave-uniform: survivors placed in this round
        bool left = false;
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const bool hit = (pending >> u) & 1u;
            const unsigned long long m = VMD_BALLOT(hit);
            if (m == 0ull) continue;
            const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned slot = base + pre;
            if (hit && slot < (unsigned)VMD_WAVE) {
                s_x[wave][slot] = x[u]; s_y[wave][slot] = y[u]; s_z[wave][slot] = z[u]; s_own[wave][slot] = own[u]; s_idx[wave][slot] = idx[u];
                pending &= ~(1u << u);
            }
            base += (unsigned)__popcll(m);
            if (base > (unsigned)VMD_WAVE) left = true;
        }
        if (base == 0u) break;
        __builtin_amdgcn_wave_barrier();
        const int nwork = (int)(base < (unsigned)VMD_WAVE ? base : (unsigned)VMD_WAVE) * p.K;
        for (int w = lane; w < nwork; w += VMD_WAVE) {
            const int a = w / p.K, k = w - a * p.K;
            vmd_sdf_atom_k(p, bx, b, k, s_x[wave][a], s_y[wave][a], s_z[wave][a], s_own[wave][a], s_idx[wave][a]);
        }
        __builtin_amdgcn_wave_barrier();
        if (!left) break;                                     // wave-uniform
    }
}

// Streaming variant (`sdf_wave` = 2): the one-shot kernels above retire a block after ~12 KB of gathers and pay the launch of its
// successor (kernel arguments, box and group record, index arithmetic - about a microsecond in which the wave slot has nothing in
// flight) once per 1 024 atoms.  Here the grid is persistent: a wave walks the (frame, 64 * ILP targets) tiles w, w + W, w + 2 W, ...
// (consecutive waves on consecutive tiles, so the grid reads one moving window of the trajectory) and issues the gathers of its NEXT
// tile before it tests the current one - two register sets, the wave always has a tile in flight.  Survivors of the group test
// are compacted per wave as in k_sdf_scatter_wave (no block barrier).  Same arithmetic, same result.
template <int ILP>
struct vmd_sdf_tile_t { int b; int idx[ILP], own[ILP]; float x[ILP], y[ILP], z[ILP]; };

template <int ILP, bool ARITH>
__device__ __forceinline__ void vmd_sdf_tile_load(const vmd_scatter_params_t& p, long long tile, int tiles_per_frame, int lane, vmd_sdf_tile_t<ILP>& T) {
    T.b = (int)(tile / tiles_per_frame);
    const int c = (int)(tile - (long long)T.b * tiles_per_frame);
    const float* fx = p.xyz + (size_t)T.b * p.frame_stride;
    const int t0 = c * (VMD_WAVE * ILP) + lane;
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        const int t = t0 + VMD_WAVE * u;
        T.idx[u] = -1; T.own[u] = p.unowned ? -1 : -2;
        if (t < p.ntgt) {
            T.idx[u] = ARITH ? p.tgt_first + t * p.tgt_stride : (p.tgt ? p.tgt[t] : t);
            if (!p.unowned && p.owner) T.own[u] = (int)p.owner[t];
        }
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
        T.x[u] = T.y[u] = T.z[u] = 0.0f;
        if (T.idx[u] >= 0) { T.x[u] = fx[T.idx[u]]; T.y[u] = fx[p.row_stride + T.idx[u]]; T.z[u] = fx[2 * p.row_stride + T.idx[u]]; }
    }
}

template <int ILP>
__device__ __forceinline__ void vmd_sdf_tile_scatter(const vmd_scatter_params_t& p, const vmd_sdf_tile_t<ILP>& T, int lane,
                                                     float* s_x, float* s_y, float* s_z, int* s_own, int* s_idx) {
    const int b = T.b;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    unsigned pending = 0u;
#pragma unroll
    for (int u = 0; u < ILP; ++u) if (T.idx[u] >= 0 && vmd_sdf_near(p, bx, b, T.x[u], T.y[u], T.z[u])) pending |= 1u << u;
    if (VMD_BALLOT(pending != 0u) == 0ull) return;          // ~99 % of the atoms fail the group test: most tiles end here
    for (;;) {
        unsigned base = 0u;                                   // wave-uniform: survivors placed in this round
        bool left = false;
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            const bool hit = (pending >> u) & 1u;
            const unsigned long long m = VMD_BALLOT(hit);
            if (m == 0ull) continue;
            const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned slot = base + pre;
            if (hit && slot < (unsigned)VMD_WAVE) {
                s_x[slot] = T.x[u]; s_y[slot] = T.y[u]; s_z[slot] = T.z[u]; s_own[slot] = T.own[u]; s_idx[slot] = T.idx[u];
                pending &= ~(1u << u);
            }
            base += (unsigned)__popcll(m);
            if (base > (unsigned)VMD_WAVE) left = true;
        }
        if (base == 0u) break;
        __builtin_amdgcn_wave_barrier();
        const int nwork = (int)(base < (unsigned)VMD_WAVE ? base : (unsigned)VMD_WAVE) * p.K;
        for (int w = lane; w < nwork; w += VMD_WAVE) {
            const int a = w / p.K, k = w - a * p.K;
            vmd_sdf_atom_k(p, bx, b, k, s_x[a], s_y[a], s_z[a], s_own[a], s_idx[a]);
        }
        __builtin_amdgcn_wave_barrier();
        if (!left) break;                                     // wave-uniform
    }
}

template <int ILP, bool ARITH>
__global__ __launch_bounds__(256) void k_sdf_scatter_stream(vmd_scatter_params_t p, int tiles_per_frame) {
    __shared__ float s_x[4][VMD_WAVE], s_y[4][VMD_WAVE], s_z[4][VMD_WAVE];
    __shared__ int s_own[4][VMD_WAVE], s_idx[4][VMD_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    const long long ntiles = (long long)tiles_per_frame * p.B;
    long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    vmd_sdf_tile_t<ILP> A, B2;
    vmd_sdf_tile_load<ILP, ARITH>(p, tile, tiles_per_frame, lane, A);
    for (;;) {
        const bool more1 = tile + W < ntiles;                 // wave-uniform
        if (more1) vmd_sdf_tile_load<ILP, ARITH>(p, tile + W, tiles_per_frame, lane, B2);
        vmd_sdf_tile_scatter<ILP>(p, A, lane, s_x[wave], s_y[wave], s_z[wave], s_own[wave], s_idx[wave]);
        if (!more1) break;
        tile += W;
        const bool more2 = tile + W < ntiles;
        if (more2) vmd_sdf_tile_load<ILP, ARITH>(p, tile + W, tiles_per_frame, lane, A);
        vmd_sdf_tile_scatter<ILP>(p, B2, lane, s_x[wave], s_y[wave], s_z[wave], s_own[wave], s_idx[wave]);
        if (!more2) break;
        tile += W;
    }
}

// Row-streaming variant for arithmetic-progression targets (first + t * stride, e.g. every water oxygen: stride 3): the lines of
// the x / y / z rows are needed in full anyway (a stride-3 selection touches every 32-byte sector), so a thread takes GPT groups of
// 4 consecutive ATOMS with 16-byte loads - three perfectly coalesced dwordx4 loads per group instead of twelve strided dword
// gathers - and tests only the atoms of its groups that are targets (one or two of four for stride 3).  Survivors of the group
// test go through the same bounded LDS compaction and K-structure loop as in k_sdf_scatter; same arithmetic, same result.
template <int GPT>
__global__ __launch_bounds__(256) void k_sdf_scatter_rows(vmd_scatter_params_t p, int g_first, int ngroups) {
    __shared__ float s_x[VMD_SDF_CAP], s_y[VMD_SDF_CAP], s_z[VMD_SDF_CAP];
    __shared__ int s_own[VMD_SDF_CAP], s_idx[VMD_SDF_CAP];
    __shared__ unsigned s_n;
    const int b = blockIdx.y;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const int s = p.tgt_stride, first = p.tgt_first;
    vmd_f4a x4[GPT], y4[GPT], z4[GPT];
    int a0[GPT];
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        const int gl = (blockIdx.x * GPT + u) * 256 + threadIdx.x;
        a0[u] = -1;
        if (gl < ngroups) {
            a0[u] = 4 * (g_first + gl);
            x4[u] = *(const vmd_f4a*)(fx + a0[u]);
            y4[u] = *(const vmd_f4a*)(fx + p.row_stride + a0[u]);
            z4[u] = *(const vmd_f4a*)(fx + 2 * p.row_stride + a0[u]);
        }
    }
    unsigned pending = 0u;          // bit 4u + k: atom k of group u is a target that passed the group test
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        if (a0[u] < 0) continue;
        // first target at or after atom a0: offset k0 in [0, s)
        const int d = a0[u] - first;
        int k0 = d >= 0 ? (s - d % s) % s : -d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k != k0) continue;
            const int t = (a0[u] + k - first) / s;
            if (t < p.ntgt && vmd_sdf_near(p, bx, b, x4[u][k], y4[u][k], z4[u][k])) pending |= 1u << (4 * u + k);
            k0 += s;
        }
    }
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (pending & (1u << (4 * u + k))) {
                    const unsigned slot = atomicAdd(&s_n, 1u);
                    if (slot < VMD_SDF_CAP) {
                        const int idx = a0[u] + k;
                        s_x[slot] = x4[u][k]; s_y[slot] = y4[u][k]; s_z[slot] = z4[u][k]; s_idx[slot] = idx;
                        s_own[slot] = p.unowned ? -1 : (p.owner ? (int)p.owner[(idx - first) / s] : -2);
                        pending &= ~(1u << (4 * u + k));
                    }
                }
            }
        }
        __syncthreads();
        const unsigned total = s_n;
        const int nwork = (int)(total < VMD_SDF_CAP ? total : VMD_SDF_CAP) * p.K;
        for (int w = threadIdx.x; w < nwork; w += 256) {
            const int a = w / p.K, k = w - a * p.K;
            vmd_sdf_atom_k(p, bx, b, k, s_x[a], s_y[a], s_z[a], s_own[a], s_idx[a]);
        }
        if (total <= VMD_SDF_CAP) break;
    }
}

// dense targets (a sizeable fraction of all atoms, e.g. every water oxygen): stream the WHOLE frame with 16-byte loads per
// lane and pick the targets by a per-atom tag byte (255 = not a target, 254 = target, k <= 253 = target owned by structure k)
// instead of gathering 4 bytes per lane through an index list.  Same arithmetic, same result.
__global__ __launch_bounds__(256) void k_sdf_scatter_dense(vmd_scatter_params_t p, const uint8_t* __restrict__ tag, int natoms4) {
    const int t4 = blockIdx.x * 256 + threadIdx.x;       // group of 4 consecutive atoms
    const int b = blockIdx.y;
    if (t4 >= natoms4) return;
    const uint32_t tg = ((const uint32_t*)tag)[t4];
    if (tg == 0xffffffffu) return;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const vmd_f4a x4 = *(const vmd_f4a*)(fx + 4 * (size_t)t4);
    const vmd_f4a y4 = *(const vmd_f4a*)(fx + p.row_stride + 4 * (size_t)t4);
    const vmd_f4a z4 = *(const vmd_f4a*)(fx + 2 * p.row_stride + 4 * (size_t)t4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tu = (int)((tg >> (8 * u)) & 255u);
        if (tu == 255 || !vmd_sdf_near(p, bx, b, x4[u], y4[u], z4[u])) continue;
        for (int k = 0; k < p.K; ++k) vmd_sdf_atom_k(p, bx, b, k, x4[u], y4[u], z4[u], tu == 254 ? -1 : tu, 4 * t4 + u);
    }
}

// ------------------------------------------------------------------------------------------------ K5: distances

struct vmd_dist_params_t {
    const float* xyz; size_t frame_stride; size_t row_stride;
    const float* boxes; uint32_t pbc; int B;
    // population of P contexts (`... in residue(:)`): context c uses a[aoff[c]..aoff[c+1]) and b[boff[c]..boff[c+1])
    const int32_t* a; const float* mass_a; const int32_t* aoff; const int32_t* b; const float* mass_b; const int32_t* boff; int P;
    int per;      // values per context: 1, or |a_c|*|b_c| for distance_pair (equal for all contexts)
    float* out;   // [B][P*per]
};

__device__ void vmd_set_com(const float* fx, const float* fy, const float* fz, const int32_t* idx, const float* mass, int n,
                            const vmd_box_t& bx, float out[3]) {
    double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    double p0x = 0.0, p0y = 0.0, p0z = 0.0;
    for (int a = 0; a < n; ++a) {
        const int i = idx[a];
        double x = (double)fx[i], y = (double)fy[i], z = (double)fz[i];
        if (a == 0) { p0x = x; p0y = y; p0z = z; }
        else {
            double dx = x - p0x, dy = y - p0y, dz = z - p0z;
            vmd_mi3_rint(bx, dx, dy, dz);
            x = p0x + dx; y = p0y + dy; z = p0z + dz;
        }
        const double w = mass ? (double)mass[a] : 1.0;
        sw = sw + w; sx = sx + w * x; sy = sy + w * y; sz = sz + w * z;
    }
    out[0] = (float)(sx / sw); out[1] = (float)(sy / sw); out[2] = (float)(sz / sw);
}

__global__ __launch_bounds__(64) void k_distance_com(vmd_dist_params_t p) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= p.B * p.P) return;
    const int b = t / p.P, c = t - b * p.P;
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const float* fy = fx + p.row_stride;
    const float* fz = fy + p.row_stride;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    const int a0 = p.aoff[c], na = p.aoff[c + 1] - a0, b0 = p.boff[c], nb = p.boff[c + 1] - b0;
    float ca[3], cb[3];
    vmd_set_com(fx, fy, fz, p.a + a0, p.mass_a ? p.mass_a + a0 : nullptr, na, bx, ca);
    vmd_set_com(fx, fy, fz, p.b + b0, p.mass_b ? p.mass_b + b0 : nullptr, nb, bx, cb);
    float dx = ca[0] - cb[0], dy = ca[1] - cb[1], dz = ca[2] - cb[2];
    vmd_mi3_rintf(bx, dx, dy, dz);
    p.out[t] = sqrtf(vmd_d2(dx, dy, dz));
}

__device__ __forceinline__ float vmd_pair_d2(const vmd_dist_params_t& p, int b, int i, int j) {
    const float* fx = p.xyz + (size_t)b * p.frame_stride;
    const float* fy = fx + p.row_stride;
    const float* fz = fy + p.row_stride;
    const vmd_box_t bx = vmd_load_box(p.boxes, b, p.pbc);
    float xi, yi, zi, xj, yj, zj;
    vmd_pair_coords(bx, fx[i], fy[i], fz[i], xi, yi, zi);
    vmd_pair_coords(bx, fx[j], fy[j], fz[j], xj, yj, zj);
    return vmd_pair_d2_general(bx, xi, yi, zi, xj, yj, zj);
}

// one block per (frame, context); MAXI = false -> min, true -> max
template <bool MAXI>
__global__ __launch_bounds__(256) void k_distance_minmax(vmd_dist_params_t p) {
    __shared__ float s_red[256];
    const int b = blockIdx.x / p.P, c = blockIdx.x - b * p.P;
    const int a0 = p.aoff[c], na = p.aoff[c + 1] - a0, b0 = p.boff[c], nb = p.boff[c + 1] - b0;
    const long long npairs = (long long)na * nb;
    float best = MAXI ? 0.0f : 3.4028235e38f;
    for (long long k = threadIdx.x; k < npairs; k += 256) {
        const int ia = (int)(k / nb), ib = (int)(k - (long long)ia * nb);
        const float d2 = vmd_pair_d2(p, b, p.a[a0 + ia], p.b[b0 + ib]);
        best = MAXI ? fmaxf(best, d2) : fminf(best, d2);
    }
    s_red[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v = s_red[threadIdx.x + o];
            s_red[threadIdx.x] = MAXI ? fmaxf(s_red[threadIdx.x], v) : fminf(s_red[threadIdx.x], v);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) p.out[blockIdx.x] = sqrtf(s_red[0]);
}

// grid (B*P, ceil(per/256)): all |a_c| x |b_c| pairs of context c, row-major.  The (frame, context) index sits on grid.x: a
// population of >= 64 contexts over a 1 024-frame batch exceeds the 65 535 limit of grid.y
__global__ __launch_bounds__(256) void k_distance_pair(vmd_dist_params_t p) {
    const long long k = (long long)blockIdx.y * 256 + threadIdx.x;
    const int b = blockIdx.x / p.P, c = blockIdx.x - b * p.P;
    if (k >= p.per) return;
    const int a0 = p.aoff[c], b0 = p.boff[c], nb = p.boff[c + 1] - b0;
    const int ia = (int)(k / nb), ib = (int)(k - (long long)ia * nb);
    p.out[((size_t)b * p.P + c) * p.per + k] = sqrtf(vmd_pair_d2(p, b, p.a[a0 + ia], p.b[b0 + ib]));
}

// ------------------------------------------------------------------------------------------------ misc

__global__ __launch_bounds__(256) void k_counts_to_float(const uint64_t* __restrict__ counts, size_t n, float* __restrict__ values,
                                                         unsigned* __restrict__ max_bits, float scale) {
    // 8 voxels per thread, one atomicMax per WAVE: a filled 128^3 volume used to issue ~10^6 same-address atomics (0.38 ms per call,
    // more than the conversion's 25 MB of traffic costs)
    const size_t i0 = (size_t)blockIdx.x * 2048 + threadIdx.x;
    float vmax = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t i = i0 + 256 * (size_t)u;
        if (i < n) { const float v = (float)counts[i] * scale; values[i] = v; vmax = fmaxf(vmax, v); }   // scale = 1: raw counts (SPEC S5)
    }
    if (max_bits) {
        vmax = vmd_wave_max(vmax);
        if ((threadIdx.x & 63) == 0 && vmax > 0.0f) atomicMax(max_bits, (unsigned)__float_as_int(vmax));
    }
}

__device__ __forceinline__ uint64_t vmd_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float vmd_synth_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t atom) {
    const uint64_t key = vmd_mix64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)stream);
    const uint64_t h = vmd_mix64(key ^ (((uint64_t)frame << 32) | (uint64_t)atom));
    return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}

struct vmd_synth_params_t {
    float* xyz; size_t frame_stride; size_t row_stride; int B; uint32_t frame0;
    uint64_t seed; uint32_t n_atoms; uint32_t n_blob; float L; float sigma;
};

// oracle S9 twin: integer RNG, explicit rounding steps -> bit-identical to vo_synth_frame
__global__ __launch_bounds__(256) void k_synth(vmd_synth_params_t p) {
    const uint32_t i = p.n_blob + blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= p.n_atoms) return;
    const uint32_t frame = p.frame0 + (uint32_t)b;
    const float sig = (float)((double)p.sigma * sqrt((double)frame));
    const uint32_t w = i - p.n_blob;
    const uint32_t mol = w / 3u, site = w % 3u;
    float* f = p.xyz + (size_t)b * p.frame_stride;
    for (uint32_t c = 0; c < 3; ++c) {
        float p0 = vmd_synth_uniform(p.seed, 1u + c, 0u, mol) * p.L;
        if (site) {
            const float off = (vmd_synth_uniform(p.seed, 4u + 3u * (site - 1u) + c, 0u, mol) - 0.5f) * 1.1f;
            p0 = p0 + off;
        }
        const float g = (((vmd_synth_uniform(p.seed, 10u + c, frame, i) + vmd_synth_uniform(p.seed, 13u + c, frame, i)) +
                          (vmd_synth_uniform(p.seed, 16u + c, frame, i) + vmd_synth_uniform(p.seed, 19u + c, frame, i))) - 2.0f) * 1.7320508f;
        const float t = sig * g;
        f[(size_t)c * p.row_stride + i] = vmd_wrap(p0 + t, p.L, 1.0f / p.L);
    }
}

// ================================================================================================ C ABI

#define VMD_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

static int g_cells_fused = 1;
extern "C" int vmd_hip_set_cells_fused(int on) { const int old = g_cells_fused; g_cells_fused = on; return old; }
static int g_cells_split = 1;
extern "C" int vmd_hip_set_cells_split(int on) { const int old = g_cells_split; g_cells_split = on; return old; }
// the fused single-block-per-frame build needs the cell table in LDS (<= 96 KB of counters)
// ... and pays off while one block per frame still has enough parallelism (measured: 33k atoms/frame 1.5x faster, 333k slower)
extern "C" int vmd_hip_cells_fused_ok(vmd_grid_t grid, int nsel) { return g_cells_fused && g_cells_split < 2 && grid.ncell + 1 <= 24576 && nsel <= 65536; }
// larger selections: G blocks per frame with LDS tables (k_cells_split_*); 0 = not applicable
// (cells_split = 2 forces this path with 2048-atom slices whatever the selection size, >= 1024 forces it with that slice
// size: test / tuning hook)
static int vmd_split_slice(void) { return g_cells_split == 2 ? 2048 : g_cells_split >= 1024 ? (g_cells_split & ~1023) : 32768; }
extern "C" int vmd_hip_cells_split_blocks(vmd_grid_t grid, int nsel) {
    if (!g_cells_fused || !g_cells_split || grid.ncell + 1 > 24576 || (nsel <= 65536 && g_cells_split < 2)) return 0;
    return (nsel + vmd_split_slice() - 1) / vmd_split_slice();
}
// u32 words of `rank` scratch per frame that vmd_hip_cells_build needs for this grid and selection
extern "C" size_t vmd_hip_cells_scratch_words(vmd_grid_t grid, int nsel) {
    const size_t split = (size_t)vmd_hip_cells_split_blocks(grid, nsel) * (size_t)grid.ncell;
    return split > (size_t)nsel ? split : (size_t)nsel;
}

static int vmd_lds_opt_in(const void* kernel) {
    return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
}

// per host thread, set by the evaluator before the builds of one selection and cleared after them (like the overflow bit below): the
// periodic form of that selection's index list, or m = 0
static thread_local vmd_sel_pattern_t g_cells_sel_pattern = {0, 0, 0, {0, 0, 0, 0}};
extern "C" void vmd_hip_set_cells_sel_pattern(int m, int first, int period, const int* off) {
    vmd_sel_pattern_t p = {0, 0, 0, {0, 0, 0, 0}};
    if (m >= 1 && m <= 4 && period > 0) {
        p.m = m; p.first = first; p.period = period;
        for (int k = 0; k < m; ++k) p.off[k] = off ? off[k] : 0;
    }
    g_cells_sel_pattern = p;
}

extern "C" int vmd_hip_cells_build(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                                   const float* boxes, uint32_t pbc_flags, int B, const int32_t* sel, int nsel, int nsel_pad,
                                   vmd_grid_t grid, uint32_t* cell_count, uint32_t* rank, uint32_t* cell_start, float* sorted,
                                   float* aos) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || nsel <= 0) return 0;
    const dim3 grp((nsel + 255) / 256, B);
    vmd_cells_params_t p{xyz, frame_stride, row_stride, boxes, pbc_flags, sel, nsel, nsel_pad, grid, cell_count, rank, cell_start, sorted, aos, g_cells_sel_pattern};
    if (vmd_hip_cells_fused_ok(grid, nsel)) {
        const size_t shm = sizeof(uint32_t) * ((size_t)grid.ncell + 1 + 1024);
        int ea = vmd_lds_opt_in((const void*)k_cells_fused);
        if (ea) return ea;
        hipLaunchKernelGGL(k_cells_fused, dim3(B), dim3(1024), shm, s, p);
        VMD_LAUNCH_CHECK();
    } else if (const int G = vmd_hip_cells_split_blocks(grid, nsel)) {
        vmd_cells_split_t q{p, rank, G, vmd_split_slice()};
        const size_t shm = sizeof(uint32_t) * (size_t)grid.ncell;
        int ea = vmd_lds_opt_in((const void*)k_cells_split_count);
        if (!ea) ea = vmd_lds_opt_in((const void*)k_cells_split_scatter);
        if (ea) return ea;
        hipLaunchKernelGGL(k_cells_split_count, dim3(G, B), dim3(1024), shm, s, q);
        VMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cells_split_scan, dim3(B), dim3(1024), 0, s, rank, cell_start, (int)grid.ncell, G);
        VMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cells_split_scatter, dim3(G, B), dim3(1024), shm, s, q);
        VMD_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemsetAsync(cell_count, 0, sizeof(uint32_t) * (size_t)B * (grid.ncell + 1), s);
        if (e != hipSuccess) return (int)e;
        const dim3 g((nsel + 256 * VMD_CELLS_ILP - 1) / (256 * VMD_CELLS_ILP), B);
        hipLaunchKernelGGL(k_cells_count, g, dim3(256), 0, s, p);
        VMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cells_scan, dim3(B), dim3(1024), 0, s, (const uint32_t*)cell_count, cell_start, (int)grid.ncell);
        VMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_cells_scatter, g, dim3(256), 0, s, p);
        VMD_LAUNCH_CHECK();
    }
    if (aos) { hipLaunchKernelGGL(k_cells_repack, grp, dim3(256), 0, s, (const float*)aos, sorted, nsel, nsel_pad); VMD_LAUNCH_CHECK(); }
    return 0;
}

// which pop the hit stack of k_rdf_pencil (variant 0) is drained with when r_min == 0: 0 = vmd_pop_hot (9 VALU instructions), 1 = vmd_pop_hot0
// (delta folded into the constant, a spare bin instead of the range compare, the stack read with ds_read_addtid_b32: 6).  A/B and cross-check
// switch; measured on c3: 73.3 -> 71.9 ms per 1 000 frames (profiles/r06q_pop_compile_time_ab.txt; the variant with a plain ds_read, 7
// instructions, sits between them: profiles/r06p_pop_ab.txt)
static int g_rdf_pop = 1;
extern "C" int vmd_hip_set_rdf_pop(int mode) { const int old = g_rdf_pop; if (mode >= 0 && mode <= 1) g_rdf_pop = mode; return old; }
static int g_rdf_nsub = 0;   // 0 = automatic: about one i-chunk per item
extern "C" int vmd_hip_set_rdf_nsub(int n) { const int old = g_rdf_nsub; if (n >= 0 && n <= 64) g_rdf_nsub = n; return old; }
static int g_rdf_nsub_pct = 100;   // automatic nsub = mean chunks per pencil x this / 100
extern "C" int vmd_hip_set_rdf_nsub_pct(int n) { const int old = g_rdf_nsub_pct; if (n >= 25 && n <= 800) g_rdf_nsub_pct = n; return old; }
static int g_rdf_shist = 1;       // one LDS histogram per block instead of one per wave: 8 instead of 7 waves per SIMD (default since round 6, see VMD_PENCIL_OCC)
extern "C" int vmd_hip_set_rdf_shared_hist(int on) { const int old = g_rdf_shist; g_rdf_shist = on ? 1 : 0; return old; }
static thread_local int g_pen_ry = 1, g_pen_rz = 1;   // neighbour reach of the pencil walk; the grid handed to the build and to the walk must be cut to match.  Per host thread like g_rdf_closed: the evaluator sets it (choose_grid) right before that thread's launches, two evals on two threads must not race (TSan: profiles/r04r_tsan.txt)
extern "C" void vmd_hip_set_pencil_reach(int ry, int rz) { g_pen_ry = ry < 1 ? 1 : (ry > 4 ? 4 : ry); g_pen_rz = rz < 1 ? 1 : (rz > 4 ? 4 : rz); }
// candidate columns walked by k_rdf_pencil (all launches of this process on the current device) since the last reset
static unsigned long long* g_cols_dev[64] = {};
static unsigned long long* vmd_cols_counter() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_cols_dev[dev]) {
        if (hipMalloc((void**)&g_cols_dev[dev], sizeof(unsigned long long)) != hipSuccess) { g_cols_dev[dev] = nullptr; return nullptr; }
        (void)hipMemset(g_cols_dev[dev], 0, sizeof(unsigned long long));
    }
    return g_cols_dev[dev];
}
extern "C" uint64_t vmd_hip_rdf_columns(int reset) {
    unsigned long long* c = vmd_cols_counter();
    unsigned long long v = 0;
    if (!c || hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, c, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (reset) (void)hipMemset(c, 0, sizeof(v));
    return (uint64_t)v;
}
static int g_rdf_nsplit = -1;     // small launches split a chunk's neighbour pencils over work items: -1 automatic (5 / 9), 0 never, n > 0 = n parts
extern "C" int vmd_hip_set_rdf_nsplit(int n) { const int old = g_rdf_nsplit; g_rdf_nsplit = n < -1 ? -1 : (n > 25 ? 25 : n); return old; }
static int g_rdf_blocks = 2048;   // 8 blocks x 4 waves per CU: all resident with the shared histogram (7 per CU with wave-private ones)
extern "C" int vmd_hip_rdf_num_blocks(void) { return 2048; }   // capacity of the partial-row scratch
extern "C" int vmd_hip_set_rdf_blocks(int n) { const int old = g_rdf_blocks; if (n >= 8 && n <= 2048) g_rdf_blocks = n; return old; }
extern "C" size_t vmd_hip_rdf_partial_words(void) {
    return (size_t)vmd_hip_rdf_num_blocks() * VMD_MAX_BINS + 8 * VMD_COUNTER_STRIDE * sizeof(unsigned) / sizeof(uint64_t);
}

extern "C" int vmd_hip_rdf_pencil(void* stream, const float* sorted_ref, const uint32_t* cell_start_ref, int nref, int nref_pad,
                                  const float* sorted_tgt, const uint32_t* cell_start_tgt, int ntgt, int ntgt_pad,
                                  const float* boxes, int B, vmd_grid_t grid, float rmin, float rmax, int nbins,
                                  int same_set, int variant, uint32_t pbc_flags, uint64_t* partial, uint64_t* counts,
                                  const uint32_t* skip_flag) {
    hipStream_t s = (hipStream_t)stream;
    if (nbins <= 0 || nbins > VMD_MAX_BINS) return (int)hipErrorInvalidValue;
    if (B <= 0 || nref <= 0 || ntgt <= 0) return 0;
    // the work counter lives behind the partial rows (see vmd_hip_rdf_partial_words)
    unsigned* work_counter = (unsigned*)(partial + (size_t)vmd_hip_rdf_num_blocks() * VMD_MAX_BINS);
    {
        hipError_t e = hipMemsetAsync(work_counter, 0, 8 * VMD_COUNTER_STRIDE * sizeof(unsigned), s);
        if (e != hipSuccess) return (int)e;
    }
    vmd_pair_params_t p;
    p.work_counter = work_counter;
    p.sref = sorted_ref; p.cs_ref = cell_start_ref; p.nref_pad = nref_pad;
    p.stgt = sorted_tgt; p.cs_tgt = cell_start_tgt; p.ntgt_pad = ntgt_pad;
    p.boxes = boxes; p.B = B; p.grid = grid;
    p.bin = vmd_make_binning(rmin, rmax, nbins, g_rdf_closed);
    p.r2_up = nextafterf(rmax * rmax, 3.0e38f) * 1.0001f;
    p.rpad = rmax * 1.0001f + 1.0e-4f;
    p.pop = p.bin.fast_2d < 1.0f ? g_rdf_pop : 0;       // the folded test needs r_min == 0 and a usable fast path (vmd_make_binning)
    p.partial = partial;
    p.counts = (unsigned long long*)counts;
    // items per pencil: the mean number of 64-atom i-chunks in a pencil, so that an XCD's waves in flight cover as few frames
    // as possible (profiles/r01t_ab.txt: c2 +11 %, c3 +8.5 % over whole-pencil items)
    p.nsub = g_rdf_nsub;
    if (p.nsub == 0) {
        const long long per_chunk = (long long)grid.ny * grid.nz * VMD_WAVE;
        p.nsub = (int)(((nref + per_chunk - 1) / per_chunk) * g_rdf_nsub_pct / 100);
        if (p.nsub < 1) p.nsub = 1;
        if (p.nsub > 64) p.nsub = 64;
    }
    // small launches: fewer items than the grid has waves.  Every item is then a lone, latency-bound chain; dealing a chunk's neighbour
    // pencils (5 in the half shell, 9 otherwise) to separate items shortens the chains and fills the idle waves
    p.nsplit = 1;
    if (g_rdf_nsplit != 0 && (long long)B * grid.ny * grid.nz * p.nsub < 4ll * g_rdf_blocks) p.nsplit = g_rdf_nsplit > 0 ? g_rdf_nsplit : (same_set ? 5 : 9);
    const int nitems = B * grid.ny * grid.nz * p.nsub * p.nsplit;
    int nblocks = (nitems + 3) / 4;
    if (nblocks < 8) nblocks = 8;
    if (nblocks > g_rdf_blocks) nblocks = g_rdf_blocks;
    const dim3 g(nblocks), blk(256);
    p.pbc = pbc_flags;
    p.skip = skip_flag;
    p.ry = g_pen_ry; p.rz = g_pen_rz;
    p.cols_total = vmd_cols_counter();
    const int cell = (pbc_flags & VMD_PBC_TRICLINIC) ? 1 : ((pbc_flags & 7u) != 7u ? 2 : 0);
    const int which = (variant == 1 ? 6 : variant == 2 ? 12 : variant == 3 ? 18 : 0) + (same_set ? 3 : 0) + cell;
#define VMD_PENCIL_CASE(n, V, S, C) case n: if (g_rdf_shist) hipLaunchKernelGGL((k_rdf_pencil<V, S, C, true>), g, blk, 0, s, p); \
                                            else hipLaunchKernelGGL((k_rdf_pencil<V, S, C, false>), g, blk, 0, s, p); break;
    // variant 0 (the default): the pop is part of the instantiation (vmd_wave_acc_tt)
#define VMD_PENCIL_CASE0(n, S, C) case n: if (p.pop) { if (g_rdf_shist) hipLaunchKernelGGL((k_rdf_pencil<0, S, C, true, 2>), g, blk, 0, s, p); \
                                                       else hipLaunchKernelGGL((k_rdf_pencil<0, S, C, false, 2>), g, blk, 0, s, p); } \
                                          else if (g_rdf_shist) hipLaunchKernelGGL((k_rdf_pencil<0, S, C, true, 0>), g, blk, 0, s, p); \
                                          else hipLaunchKernelGGL((k_rdf_pencil<0, S, C, false, 0>), g, blk, 0, s, p); break;
    switch (which) {
    VMD_PENCIL_CASE0(0, false, 0) VMD_PENCIL_CASE0(1, false, 1) VMD_PENCIL_CASE0(2, false, 2)
    VMD_PENCIL_CASE0(3, true, 0) VMD_PENCIL_CASE0(4, true, 1) VMD_PENCIL_CASE0(5, true, 2)
    VMD_PENCIL_CASE(6, 1, false, 0) VMD_PENCIL_CASE(7, 1, false, 1) VMD_PENCIL_CASE(8, 1, false, 2)
    VMD_PENCIL_CASE(9, 1, true, 0) VMD_PENCIL_CASE(10, 1, true, 1) VMD_PENCIL_CASE(11, 1, true, 2)
    VMD_PENCIL_CASE(12, 2, false, 0) VMD_PENCIL_CASE(13, 2, false, 1) VMD_PENCIL_CASE(14, 2, false, 2)
    VMD_PENCIL_CASE(15, 2, true, 0) VMD_PENCIL_CASE(16, 2, true, 1) VMD_PENCIL_CASE(17, 2, true, 2)
    case 18: hipLaunchKernelGGL((k_rdf_pencil<3, false, 0, false>), g, blk, 0, s, p); break;
    case 19: hipLaunchKernelGGL((k_rdf_pencil<3, false, 1, false>), g, blk, 0, s, p); break;
    case 20: hipLaunchKernelGGL((k_rdf_pencil<3, false, 2, false>), g, blk, 0, s, p); break;
    case 21: hipLaunchKernelGGL((k_rdf_pencil<3, true, 0, false>), g, blk, 0, s, p); break;
    case 22: hipLaunchKernelGGL((k_rdf_pencil<3, true, 1, false>), g, blk, 0, s, p); break;
    case 23: hipLaunchKernelGGL((k_rdf_pencil<3, true, 2, false>), g, blk, 0, s, p); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef VMD_PENCIL_CASE
#undef VMD_PENCIL_CASE0
    VMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hist_reduce, dim3((nbins + 255) / 256, (nblocks + 31) / 32), dim3(256), 0, s, (const uint64_t*)partial, nblocks, nbins, counts, skip_flag);
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_rdf_brute(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                                 const float* boxes, uint32_t pbc_flags, int B,
                                 const int32_t* ref, int nref, const int32_t* tgt, int ntgt,
                                 float rmin, float rmax, int nbins, uint64_t* counts) {
    hipStream_t s = (hipStream_t)stream;
    if (nbins <= 0 || nbins > VMD_MAX_BINS) return (int)hipErrorInvalidValue;
    if (B <= 0 || nref <= 0 || ntgt <= 0) return 0;
    vmd_brute_params_t p{xyz, frame_stride, row_stride, boxes, pbc_flags, B, ref, nref, tgt, ntgt, {}, counts, g_rdf_raw};
    p.bin = vmd_make_binning(rmin, rmax, nbins, g_rdf_closed);
    hipLaunchKernelGGL(k_rdf_brute, dim3((nref + 255) / 256, B), dim3(256), 0, s, p);
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_sdf_align(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                                 const float* boxes, uint32_t pbc_flags, int B,
                                 const int32_t* structs, const float* mass, int K, int m, const double* ref_pose,
                                 float* R32, float* c32, double* M64, float* group,
                                 const int32_t* tree_order, const int32_t* tree_parent, double* tree_pos) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || K <= 0 || m <= 0) return 0;
    if ((tree_order != nullptr) != (tree_parent != nullptr) || (tree_order && !tree_pos)) return (int)hipErrorInvalidValue;
    vmd_align_params_t p{xyz, frame_stride, row_stride, boxes, pbc_flags, B, structs, mass, K, m, ref_pose, R32, c32, M64, tree_order, tree_parent, tree_pos};
    hipLaunchKernelGGL(k_sdf_align, dim3((B * K + 63) / 64), dim3(64), 0, s, p);
    VMD_LAUNCH_CHECK();
    if (group) {
        hipLaunchKernelGGL(k_sdf_group, dim3((B + 63) / 64), dim3(64), 0, s, (const float*)c32, boxes, pbc_flags, B, K, group);
        VMD_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int vmd_hip_sdf_ref_pose(void* stream, const float* xyz, size_t row_stride, const float* box, uint32_t pbc_flags,
                                    const int32_t* struct0, const float* mass0, int m, double* ref_pose,
                                    const int32_t* tree_order, const int32_t* tree_parent) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sdf_ref_pose, dim3(1), dim3(64), 0, s, xyz, row_stride, box, pbc_flags, struct0, mass0, m, ref_pose, tree_order, tree_parent);
    VMD_LAUNCH_CHECK();
    return 0;
}

static int g_sdf_nt = 0;        // the scatter's gathers as non-temporal loads (A/B: profiles/r03i_ab_c4.txt)
extern "C" int vmd_hip_set_sdf_nt(int on) { const int old = g_sdf_nt; g_sdf_nt = on ? 1 : 0; return old; }
static int g_sdf_rows = 0;      // row-streaming scatter for arithmetic-progression targets: groups of 4 atoms per thread (0 = off, 1 / 2 / 4)
extern "C" int vmd_hip_set_sdf_rows(int n) { const int old = g_sdf_rows; if (n == 0 || n == 1 || n == 2 || n == 4) g_sdf_rows = n; return old; }
static int g_sdf_wave = 0;      // 1: per-wave instead of per-block compaction of the group test's survivors (no block barrier);
                                // 2: the persistent streaming kernel (k_sdf_scatter_stream) on 2 048 blocks, n >= 16: on n blocks
extern "C" int vmd_hip_set_sdf_wave(int on) { const int old = g_sdf_wave; g_sdf_wave = on < 0 ? 0 : (on > 65535 ? 65535 : on); return old; }
static int g_sdf_ilp = 4;
extern "C" int vmd_hip_set_sdf_ilp(int n) { const int old = g_sdf_ilp; if (n == 4 || n == 8 || n == 16) g_sdf_ilp = n; return old; }
extern "C" int vmd_hip_sdf_scatter(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                                   const float* boxes, uint32_t pbc_flags, int B,
                                   const int32_t* structs, int K, int m, const float* R32, const float* c32,
                                   const int32_t* tgt, const int8_t* owner, int ntgt, float extent, int dim, uint64_t* volume,
                                   const float* group, const uint8_t* atom_tag, int tgt_first, int tgt_stride, int unowned) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || K <= 0 || ntgt <= 0) return 0;
    vmd_scatter_params_t p{xyz, frame_stride, row_stride, boxes, pbc_flags, B, structs, K, m, R32, c32, tgt, owner, ntgt, extent, dim,
                           (unsigned long long*)volume, group, tgt_first, tgt_stride, unowned, g_sdf_nt};
    if (atom_tag) {
        const int natoms4 = (int)(row_stride / 4);       // rows are padded to a multiple of 64 floats; the tag array covers the padding
        hipLaunchKernelGGL(k_sdf_scatter_dense, dim3((natoms4 + 255) / 256, B), dim3(256), 0, s, p, atom_tag, natoms4);
        VMD_LAUNCH_CHECK();
        return 0;
    }
    const bool arith = tgt_stride > 0;
    if (arith && g_sdf_rows && tgt_stride <= 4 && ((uintptr_t)xyz & 15u) == 0 && row_stride % 4 == 0 && frame_stride % 4 == 0) {
        // the rows are read in full anyway: stream them with 16-byte loads (stride <= 4: every group of 4 atoms holds a target)
        const int g_first = tgt_first / 4;
        const int g_last = (int)(((long long)tgt_first + (long long)(ntgt - 1) * tgt_stride) / 4);
        const int ngroups = g_last - g_first + 1;
        if (g_sdf_rows == 2) hipLaunchKernelGGL((k_sdf_scatter_rows<2>), dim3((ngroups + 511) / 512, B), dim3(256), 0, s, p, g_first, ngroups);
        else if (g_sdf_rows == 4) hipLaunchKernelGGL((k_sdf_scatter_rows<4>), dim3((ngroups + 1023) / 1024, B), dim3(256), 0, s, p, g_first, ngroups);
        else hipLaunchKernelGGL((k_sdf_scatter_rows<1>), dim3((ngroups + 255) / 256, B), dim3(256), 0, s, p, g_first, ngroups);
        VMD_LAUNCH_CHECK();
        return 0;
    }
    if (g_sdf_wave >= 2) {
        const int ilp = g_sdf_ilp == 8 ? 8 : 4;
        const int tiles_per_frame = (ntgt + VMD_WAVE * ilp - 1) / (VMD_WAVE * ilp);
        const long long ntiles = (long long)tiles_per_frame * B;
        long long nblocks = g_sdf_wave >= 16 ? g_sdf_wave : 2048;
        if (nblocks > (ntiles + 3) / 4) nblocks = (ntiles + 3) / 4;
        const dim3 g((unsigned)nblocks);
        if (ilp == 8) {
            if (arith) hipLaunchKernelGGL((k_sdf_scatter_stream<8, true>), g, dim3(256), 0, s, p, tiles_per_frame);
            else hipLaunchKernelGGL((k_sdf_scatter_stream<8, false>), g, dim3(256), 0, s, p, tiles_per_frame);
        } else {
            if (arith) hipLaunchKernelGGL((k_sdf_scatter_stream<4, true>), g, dim3(256), 0, s, p, tiles_per_frame);
            else hipLaunchKernelGGL((k_sdf_scatter_stream<4, false>), g, dim3(256), 0, s, p, tiles_per_frame);
        }
        VMD_LAUNCH_CHECK();
        return 0;
    }
    if (g_sdf_wave) {
        if (g_sdf_ilp == 8) {
            const dim3 g((ntgt + 256 * 8 - 1) / (256 * 8), B);
            if (arith) hipLaunchKernelGGL((k_sdf_scatter_wave<8, true>), g, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((k_sdf_scatter_wave<8, false>), g, dim3(256), 0, s, p);
        } else {
            const dim3 g((ntgt + 256 * 4 - 1) / (256 * 4), B);
            if (arith) hipLaunchKernelGGL((k_sdf_scatter_wave<4, true>), g, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((k_sdf_scatter_wave<4, false>), g, dim3(256), 0, s, p);
        }
        VMD_LAUNCH_CHECK();
        return 0;
    }
    if (g_sdf_ilp == 16) {
        const dim3 g((ntgt + 256 * 16 - 1) / (256 * 16), B);
        if (arith) hipLaunchKernelGGL((k_sdf_scatter<16, true>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_sdf_scatter<16, false>), g, dim3(256), 0, s, p);
    } else if (g_sdf_ilp == 8) {
        const dim3 g((ntgt + 256 * 8 - 1) / (256 * 8), B);
        if (arith) hipLaunchKernelGGL((k_sdf_scatter<8, true>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_sdf_scatter<8, false>), g, dim3(256), 0, s, p);
    } else {
        const dim3 g((ntgt + 256 * 4 - 1) / (256 * 4), B);
        if (arith) hipLaunchKernelGGL((k_sdf_scatter<4, true>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_sdf_scatter<4, false>), g, dim3(256), 0, s, p);
    }
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_distance(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                                const float* boxes, uint32_t pbc_flags, int B, int kind, int P, int per,
                                const int32_t* a, const float* mass_a, const int32_t* aoff,
                                const int32_t* b, const float* mass_b, const int32_t* boff, float* out) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || P <= 0 || per <= 0) return 0;
    vmd_dist_params_t p{xyz, frame_stride, row_stride, boxes, pbc_flags, B, a, mass_a, aoff, b, mass_b, boff, P, per, out};
    switch (kind) {
    case 0: hipLaunchKernelGGL(k_distance_com, dim3((B * P + 63) / 64), dim3(64), 0, s, p); break;
    case 1: hipLaunchKernelGGL((k_distance_minmax<false>), dim3(B * P), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL((k_distance_minmax<true>), dim3(B * P), dim3(256), 0, s, p); break;
    case 3:
        if ((per + 255) / 256 > 65535) return (int)hipErrorInvalidValue;      // > 16.7M pairs per context: not a distance_pair population
        hipLaunchKernelGGL(k_distance_pair, dim3((unsigned)(B * P), (unsigned)((per + 255) / 256)), dim3(256), 0, s, p);
        break;
    default: return (int)hipErrorInvalidValue;
    }
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_bbox(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, int B, int natoms, float* out) {
    if (B <= 0 || natoms <= 0) return 0;
    hipLaunchKernelGGL(k_bbox, dim3(B), dim3(1024), 0, (hipStream_t)stream, xyz, frame_stride, row_stride, natoms, out);
    VMD_LAUNCH_CHECK();
    return 0;
}

// ---- two-level cell build (k_cells_bin / k_cells_pen_scan / k_cells_pen_sort) --------------------------------------------------
#define VMD_PEN_MAX 4096            // pencils per frame the LDS table of k_cells_bin holds
#define VMD_PEN_CAP_MAX 8192        // atoms of one pencil k_cells_pen_sort stages through LDS (96 KB)
static int g_cells_pencil = 1;
extern "C" int vmd_hip_set_cells_pencil(int on) { const int old = g_cells_pencil; g_cells_pencil = on; return old; }
extern "C" int vmd_hip_cells_pencil_ok(vmd_grid_t grid) {
    return g_cells_pencil && (long long)grid.ny * grid.nz <= VMD_PEN_MAX && grid.nxf <= 8192;
}
extern "C" int vmd_hip_cells_pencil_cap_max(void) { return VMD_PEN_CAP_MAX; }

extern "C" int vmd_hip_cells_pencil_count(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, const float* boxes,
                                          uint32_t pbc_flags, int S, const int32_t* sel, int nsel, vmd_grid_t grid, uint32_t* counts) {
    hipStream_t s = (hipStream_t)stream;
    if (S <= 0 || nsel <= 0) return 0;
    const int npen = grid.ny * grid.nz;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * (size_t)S * npen, s);
    if (e != hipSuccess) return (int)e;
    vmd_bin_params_t q{{xyz, frame_stride, row_stride, boxes, pbc_flags, sel, nsel, 0, grid, nullptr, nullptr, nullptr, nullptr, nullptr, g_cells_sel_pattern},
                       nullptr, counts, nullptr, nullptr, 1u, npen, 0, 0};
    hipLaunchKernelGGL(k_cells_bin, dim3((nsel + 1024 * VMD_BIN_ILP - 1) / (1024 * VMD_BIN_ILP), S), dim3(1024), sizeof(uint32_t) * npen, s, q);
    VMD_LAUNCH_CHECK();
    return 0;
}

static int g_cells_bin_lds = 1;   // level 1 orders a block's records by pencil in LDS before writing them (A/B switch)
extern "C" int vmd_hip_set_cells_bin_lds(int on) { const int old = g_cells_bin_lds; g_cells_bin_lds = on ? 1 : 0; return old; }
static thread_local uint32_t g_cells_overflow_bit = 1u;    // per host thread, set by the evaluator before each selection's build (vmd_hip_set_cells_overflow_bit)
extern "C" uint32_t vmd_hip_set_cells_overflow_bit(uint32_t bit) { const uint32_t old = g_cells_overflow_bit; g_cells_overflow_bit = bit ? bit : 1u; return old; }
static int g_cells_rec3 = 1;      // 12-byte bucket records where the cell kind allows it (A/B switch)
extern "C" int vmd_hip_set_cells_rec3(int on) { const int old = g_cells_rec3; g_cells_rec3 = on ? 1 : 0; return old; }
extern "C" int vmd_hip_cells_build_pencil(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, const float* boxes,
                                          uint32_t pbc_flags, int B, const int32_t* sel, int nsel, int nsel_pad, vmd_grid_t grid,
                                          const uint32_t* pen_off, int total_cap, int cap_max, uint32_t* pen_count, uint32_t* pen_start,
                                          float* bucket, uint32_t* overflow, uint32_t* cell_start, float* sorted) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || nsel <= 0) return 0;
    const int npen = grid.ny * grid.nz;
    if (!vmd_hip_cells_pencil_ok(grid) || cap_max > VMD_PEN_CAP_MAX) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(pen_count, 0, sizeof(uint32_t) * (size_t)B * npen, s);
    if (e != hipSuccess) return (int)e;
    // x-periodic, non-triclinic cells: the grid bins the wrapped x itself, so the record need not carry the fine cell
    const int rec3 = (g_cells_rec3 && (pbc_flags & 1u) && !(pbc_flags & VMD_PBC_TRICLINIC)) ? 1 : 0;
    vmd_bin_params_t q{{xyz, frame_stride, row_stride, boxes, pbc_flags, sel, nsel, nsel_pad, grid, nullptr, nullptr, nullptr, nullptr, nullptr, g_cells_sel_pattern},
                       pen_off, pen_count, bucket, overflow, g_cells_overflow_bit ? g_cells_overflow_bit : 1u, npen, total_cap, rec3};
    const size_t shm_sorted = sizeof(uint32_t) * (3 * (size_t)npen + 1040 + (size_t)1024 * VMD_BIN_ILP * (rec3 ? 4 : 5));
    if (g_cells_bin_lds && shm_sorted <= 160 * 1024 - 64) {
        int eb = vmd_lds_opt_in((const void*)k_cells_bin_sorted);
        if (eb) return eb;
        hipLaunchKernelGGL(k_cells_bin_sorted, dim3((nsel + 1024 * VMD_BIN_ILP - 1) / (1024 * VMD_BIN_ILP), B), dim3(1024), shm_sorted, s, q);
    } else {
        hipLaunchKernelGGL(k_cells_bin, dim3((nsel + 1024 * VMD_BIN_ILP - 1) / (1024 * VMD_BIN_ILP), B), dim3(1024), sizeof(uint32_t) * npen, s, q);
    }
    VMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cells_pen_scan, dim3(B), dim3(256), 0, s, (const uint32_t*)pen_count, pen_off, pen_start, npen);
    VMD_LAUNCH_CHECK();
    vmd_pensort_params_t ps{bucket, pen_off, pen_count, pen_start, cell_start, sorted, npen, grid.nxf, grid.ncell, nsel_pad, total_cap, cap_max, boxes, rec3};
    const size_t shm = sizeof(uint32_t) * ((size_t)grid.nxf + 256 + 3 * (size_t)cap_max);
    if (shm > 160 * 1024 - 64) return (int)hipErrorInvalidValue;
    int ea = vmd_lds_opt_in((const void*)k_cells_pen_sort);
    if (ea) return ea;
    // grid.y = B <= 65535 (frame batches are far smaller); pencils on grid.x
    hipLaunchKernelGGL(k_cells_pen_sort, dim3(npen, B), dim3(256), shm, s, ps);
    VMD_LAUNCH_CHECK();
    return 0;
}

// dst[i] += mult * src[i] (u64): one pair pass feeding several histograms (class decomposition of co-evaluated RDFs)
__global__ __launch_bounds__(256) void k_axpy_u64(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, size_t n, uint64_t mult,
                                                  const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const uint64_t v = src[i]; if (v) dst[i] += mult * v; }
}
extern "C" int vmd_hip_axpy_u64(void* stream, uint64_t* dst, const uint64_t* src, size_t n, uint64_t mult, const uint32_t* skip_flag) {
    if (n == 0 || mult == 0) return 0;
    hipLaunchKernelGGL(k_axpy_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, mult, skip_flag);
    VMD_LAUNCH_CHECK();
    return 0;
}

// dst[i] += src[i]: merges one frame block's partial accumulator into another (block partials, filtered evaluation)
__global__ __launch_bounds__(256) void k_add_u64(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const uint64_t v = src[i]; if (v) dst[i] += v; }
}
extern "C" int vmd_hip_add_u64(void* stream, uint64_t* dst, const uint64_t* src, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_add_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    VMD_LAUNCH_CHECK();
    return 0;
}

// counts[index] += value: the self pairs (d = 0) a half-shell pass never visits, when the closed interval counts them
// An empty kernel with a name of its own: bench.py launches it where its timed region begins, so that a counter collection of the same
// command (scripts/pmc_traffic.py) can tell the steady-state dispatches from the warm-up's (capacity sampling, first-touch, overflow repeats)
__global__ void k_marker_timed_region() {}
extern "C" int vmd_hip_marker(void* stream) {
    hipLaunchKernelGGL(k_marker_timed_region, dim3(1), dim3(64), 0, (hipStream_t)stream);
    VMD_LAUNCH_CHECK();
    return 0;
}
__global__ void k_bump_u64(uint64_t* p, uint64_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += v; }
extern "C" int vmd_hip_bump_u64(void* stream, uint64_t* p, uint64_t v) {
    hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(64), 0, (hipStream_t)stream, p, v);
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_counts_to_float(void* stream, const uint64_t* counts, size_t n, float* values, float* max_out, float scale) {
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return 0;
    if (max_out) {
        hipError_t e = hipMemsetAsync(max_out, 0, sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_counts_to_float, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, counts, n, values, (unsigned*)max_out, scale);
    VMD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vmd_hip_synth_frames(void* stream, float* xyz, size_t frame_stride, size_t row_stride, int B, uint32_t frame0,
                                    uint64_t seed, uint32_t n_atoms, uint32_t n_blob, float L, float sigma) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || n_atoms <= n_blob) return 0;
    vmd_synth_params_t p{xyz, frame_stride, row_stride, B, frame0, seed, n_atoms, n_blob, L, sigma};
    hipLaunchKernelGGL(k_synth, dim3((n_atoms - n_blob + 255) / 256, B), dim3(256), 0, s, p);
    VMD_LAUNCH_CHECK();
    return 0;
}
