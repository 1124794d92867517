// XTC coordinate blocks decompressed on the GPU (VIAMD: md_xtc_attach_from_file, /root/reference/src/loader.cpp:147-148).
//
// Why on the device: with the trajectory on disk the hot path is bounded by frame decode + PCIe, not by the kernels
// (c2 from pinned memory 43.8k frames/s against 112k resident).  An XTC frame is 0.42 of its float size,
// so moving the COMPRESSED bytes over the bus and decoding them next to the kernels lifts both limits at once, and takes
// the decode off the host cores that one process per GPU has to share.
//
// Parallelism: a frame's bit stream is strictly sequential (every field's position depends on all flags before it), but the
// frames of a staged batch are independent, so k_xtc_decode runs ONE THREAD PER FRAME: a batch of 1 000 frames is 16 waves.
// Lanes of a wave walk different frames through the same code; the data-dependent branches (run flag, run length) diverge for
// a few instructions only.  The 16 waves leave the rest of the chip to the pair kernel of the previous batch, which is what
// the staging pipeline overlaps them with.
//
// The arithmetic is the host reader's (vmd_xdr.cpp: xtc_decode), so both produce the same floats: a packed triple is rebuilt as
// one integer from its little-endian wire bytes and split by two independent reciprocal multiplications in fp64 (exact below
// 2^52 after a +-1 fix-up; MI355X runs fp64 at full VALU rate), 53..64-bit numbers by integer division; a number that does not
// fit 64 bits (three ranges just below 2^24 each: never seen in practice) is reported back (status 2) and the batch falls back
// to the host reader, which carries 128-bit arithmetic for it.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include "vmd_hip.h"

namespace {

#define XTC_FIRSTIDX 9
#define XTC_LASTIDX 73
__device__ const int kXtcMagic[XTC_LASTIDX] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64,
    80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290,
    1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003,
    16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031,
    131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561,
    832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021,
    4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216};

// MSB-first bit reader over 8-byte words held in registers: `hi` is the big-endian word that contains the next bit, `lo` the one
// behind it, `nxt` the one after that - already in flight since the previous word boundary, so the stream's memory latency is
// hidden behind ~two atoms of arithmetic and a field costs a handful of ALU instructions, not two dependent loads.
struct Bits {
    const uint64_t* words;       // 64-byte aligned stream start, >= 32 readable bytes behind the stream
    uint64_t pos;                // next bit
    uint64_t hi, lo, nxt;
    uint64_t nwords;             // readable words: a damaged stream keeps asking for more, the prefetch index is clamped
};
__device__ __forceinline__ void xtc_open(Bits& b, const unsigned char* base, uint64_t nbytes) {
    b.words = (const uint64_t*)base;
    b.nwords = (nbytes + 32ull) >> 3;
    b.pos = 0;
    b.hi = __builtin_bswap64(b.words[0]);
    b.lo = __builtin_bswap64(b.words[1]);
    b.nxt = __builtin_bswap64(b.words[2]);
}
// `bits` in [1, 56]
__device__ __forceinline__ uint64_t xtc_get(Bits& b, int bits) {
    const unsigned sh = (unsigned)(b.pos & 63ull);
    const uint64_t w = sh ? ((b.hi << sh) | (b.lo >> (64u - sh))) : b.hi;
    const uint64_t word = b.pos >> 6;
    b.pos += (uint64_t)bits;
    if ((b.pos >> 6) != word) {                      // at most one boundary per field (bits < 64)
        b.hi = b.lo;
        b.lo = b.nxt;
        const uint64_t k = word + 3;
        b.nxt = __builtin_bswap64(b.words[k < b.nwords ? k : b.nwords - 1]);
    }
    return w >> (64 - bits);
}

__device__ __forceinline__ int xtc_bit_length(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }

struct Radix {
    uint32_t s1, s2;
    uint64_t s12;
    double inv2, inv12;
};
__device__ __forceinline__ void xtc_radix(Radix& r, uint32_t s1, uint32_t s2) {
    r.s1 = s1; r.s2 = s2;
    r.s12 = (uint64_t)s1 * s2;
    r.inv2 = 1.0 / (double)s2;
    r.inv12 = 1.0 / (double)r.s12;
}
__device__ __forceinline__ uint64_t xtc_div(uint64_t w, uint64_t d, double inv) {
    uint64_t q = (uint64_t)((double)w * inv);
    const int64_t r = (int64_t)(w - q * d);
    if (r < 0) --q;
    else if ((uint64_t)r >= d) ++q;
    return q;
}

// one packed triple of `bits` bits: little-endian bytes on the wire, the partial top byte last.  Numbers up to 64 bits are
// handled here; false = the number has bits set above 2^64 (possible for bits > 64 only): the caller reports status 2.
__device__ __forceinline__ bool xtc_triple(Bits& b, int bits, const Radix& rx, int out[3]) {
    const int q = (bits - 1) >> 3, r = bits - 8 * q;            // q full bytes, then r in [1, 8] bits
    uint64_t w;
    bool ok = true;
    if (bits <= 56) {
        const uint64_t raw = xtc_get(b, bits);
        const uint64_t top = raw >> r, low = raw & ((1ull << r) - 1ull);
        w = (q ? (__builtin_bswap64(top) >> (64 - 8 * q)) : 0ull) | (low << (8 * q));
    } else {
        w = 0;
        for (int j = 0; j < q && j < 8; ++j) w |= xtc_get(b, 8) << (8 * j);
        const uint64_t last = xtc_get(b, r);
        if (q < 8) w |= last << (8 * q);
        else ok = last == 0;                                     // byte 8 of the number
    }
    uint64_t qa, qb;
    if (bits <= 52) {
        qa = xtc_div(w, rx.s2, rx.inv2);
        qb = xtc_div(w, rx.s12, rx.inv12);
    } else {
        qa = w / rx.s2;
        qb = qa / rx.s1;
    }
    out[2] = (int)(w - qa * rx.s2);
    out[1] = (int)(qa - qb * rx.s1);
    out[0] = (int)qb;
    return ok;
}

__global__ __launch_bounds__(64) void k_xtc_decode(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info,
                                                   int B, int natoms, float* __restrict__ xyz, size_t frame_stride,
                                                   size_t row_stride, uint32_t* __restrict__ status) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= B) return;
    const vmd_xtc_frame_t fi = info[f];
    float* x = xyz + (size_t)f * frame_stride;
    float* y = x + row_stride;
    float* z = y + row_stride;
    uint32_t st = 0;
    do {
        if (!(fi.precision > 0.0f)) { st = 1; break; }
        const float invp = 1.0f / fi.precision;
        int smallidx = fi.smallidx;
        if (smallidx < XTC_FIRSTIDX || smallidx >= XTC_LASTIDX) { st = 1; break; }
        uint32_t sizeint[3];
        bool bad = false;
        for (int k = 0; k < 3; ++k) {
            const int64_t s = (int64_t)fi.maxint[k] - (int64_t)fi.minint[k] + 1;
            if (s <= 0 || s > 0xffffffffll) bad = true;
            sizeint[k] = (uint32_t)s;
        }
        if (bad) { st = 1; break; }
        int bitsizeint[3] = {0, 0, 0}, bitsize;
        if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
            for (int k = 0; k < 3; ++k) bitsizeint[k] = xtc_bit_length(sizeint[k]);
            bitsize = 0;
        } else {
            // bit length of the product of the three ranges (each < 2^24): 128-bit product through two 64-bit halves
            const uint64_t p01 = (uint64_t)sizeint[0] * sizeint[1];                    // < 2^48
            const uint64_t lo = (p01 & 0xffffffffull) * sizeint[2], hi = (p01 >> 32) * sizeint[2];
            const uint64_t top = hi + (lo >> 32);                                      // product >> 32
            bitsize = top ? 32 + xtc_bit_length(top) : xtc_bit_length(lo);
        }
        int smaller = kXtcMagic[smallidx - 1 > XTC_FIRSTIDX ? smallidx - 1 : XTC_FIRSTIDX] / 2;
        int smallnum = kXtcMagic[smallidx] / 2;
        Radix large, small;
        xtc_radix(large, sizeint[1], sizeint[2]);
        xtc_radix(small, (uint32_t)kXtcMagic[smallidx], (uint32_t)kXtcMagic[smallidx]);

        Bits br;
        xtc_open(br, raw + fi.offset, fi.nbytes);
        const uint64_t nbits = 8ull * fi.nbytes;
        int i = 0, run = 0;
        while (i < natoms) {
            int cur[3], prev[3];
            if (bitsize == 0) {
                for (int k = 0; k < 3; ++k) {
                    const int nb = bitsizeint[k];
                    cur[k] = (int)(uint32_t)(nb > 24 ? ((xtc_get(br, nb - 24) << 24) | xtc_get(br, 24)) : xtc_get(br, nb));
                }
            } else {
                if (!xtc_triple(br, bitsize, large, cur)) { st = 2; break; }
            }
            for (int k = 0; k < 3; ++k) { cur[k] += fi.minint[k]; prev[k] = cur[k]; }
            int is_smaller = 0;
            if (xtc_get(br, 1)) {
                run = (int)xtc_get(br, 5);
                is_smaller = run % 3;
                run -= is_smaller;
                is_smaller--;
            }
            if (run > 0) {
                if (i + 1 + run / 3 > natoms) { st = 1; break; }
                for (int k = 0; k < run; k += 3) {
                    int d[3], nxt[3];
                    if (!xtc_triple(br, smallidx, small, d)) st = 2;
                    for (int c = 0; c < 3; ++c) nxt[c] = d[c] + prev[c] - smallnum;
                    x[i] = ((float)nxt[0] * invp) * 10.0f;
                    y[i] = ((float)nxt[1] * invp) * 10.0f;
                    z[i] = ((float)nxt[2] * invp) * 10.0f;
                    ++i;
                    if (k == 0) {            // the large triple in front of the run is the SECOND atom of the pair
                        x[i] = ((float)cur[0] * invp) * 10.0f;
                        y[i] = ((float)cur[1] * invp) * 10.0f;
                        z[i] = ((float)cur[2] * invp) * 10.0f;
                        ++i;
                    }
                    for (int c = 0; c < 3; ++c) prev[c] = nxt[c];
                }
                if (st) break;
            } else {
                x[i] = ((float)cur[0] * invp) * 10.0f;
                y[i] = ((float)cur[1] * invp) * 10.0f;
                z[i] = ((float)cur[2] * invp) * 10.0f;
                ++i;
            }
            if (is_smaller) {
                smallidx += is_smaller;
                if (smallidx < XTC_FIRSTIDX || smallidx >= XTC_LASTIDX) { st = 1; break; }
                if (is_smaller < 0) {
                    smallnum = smaller;
                    smaller = smallidx > XTC_FIRSTIDX ? kXtcMagic[smallidx - 1] / 2 : 0;
                } else {
                    smaller = smallnum;
                    smallnum = kXtcMagic[smallidx] / 2;
                }
                xtc_radix(small, (uint32_t)kXtcMagic[smallidx], (uint32_t)kXtcMagic[smallidx]);
            }
            if (br.pos > nbits) { st = 1; break; }
        }
    } while (false);
    status[f] = st;
}

}  // namespace

extern "C" int vmd_hip_xtc_decode(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                  float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_xtc_decode, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, raw, info, B, natoms, xyz,
                       frame_stride, row_stride, status);
    return (int)hipGetLastError();
}
