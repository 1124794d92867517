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
__device__ __forceinline__ void xtc_open(Bits& b, const unsigned char* base, uint64_t nbytes, uint64_t pos) {
    b.words = (const uint64_t*)base;
    b.nwords = (nbytes + 32ull) >> 3;
    b.pos = pos;
    const uint64_t w = pos >> 6;
    b.hi = __builtin_bswap64(b.words[w < b.nwords ? w : b.nwords - 1]);
    b.lo = __builtin_bswap64(b.words[w + 1 < b.nwords ? w + 1 : b.nwords - 1]);
    b.nxt = __builtin_bswap64(b.words[w + 2 < b.nwords ? w + 2 : b.nwords - 1]);
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

// The reader of variant 3: a lane decodes ONE group, so its first four words (>= 193 bits behind the group's first bit: every
// group of up to 128 bits, i.e. a large triple and two or three small ones) are requested together up front and a field is picked
// out of registers; longer groups (runs of 4+ atoms) fetch the words they need.  No load sits between two fields of a normal group.
struct BitsG {
    const uint64_t* words;
    uint64_t nwords;
    uint64_t first;              // index of w0
    uint64_t pos;
    uint64_t w0, w1, w2, w3;     // four named registers, not an array: an array indexed by the word number ends up in scratch memory
};
__device__ __forceinline__ uint64_t xtc_word(const BitsG& b, uint64_t i) { return __builtin_bswap64(b.words[i < b.nwords ? i : b.nwords - 1]); }
__device__ __forceinline__ void xtc_open(BitsG& b, const unsigned char* base, uint64_t nbytes, uint64_t pos) {
    b.words = (const uint64_t*)base;
    b.nwords = (nbytes + 32ull) >> 3;
    b.pos = pos;
    b.first = pos >> 6;
    b.w0 = xtc_word(b, b.first);
    b.w1 = xtc_word(b, b.first + 1);
    b.w2 = xtc_word(b, b.first + 2);
    b.w3 = xtc_word(b, b.first + 3);
}
__device__ __forceinline__ uint64_t xtc_get(BitsG& b, int bits) {
    const unsigned sh = (unsigned)(b.pos & 63ull);
    const uint64_t k = (b.pos >> 6) - b.first;
    uint64_t hi, lo;
    if (k <= 2) {
        // masks, not a chain of selects between the fields: the compiler folds "select of two loaded fields" into a load through a
        // selected ADDRESS, which pins the whole reader in scratch memory
        const uint64_t m0 = k == 0 ? ~0ull : 0ull, m1 = k == 1 ? ~0ull : 0ull, m2 = k == 2 ? ~0ull : 0ull;
        hi = (b.w0 & m0) | (b.w1 & m1) | (b.w2 & m2);
        lo = (b.w1 & m0) | (b.w2 & m1) | (b.w3 & m2);
    } else {
        hi = xtc_word(b, b.first + k);
        lo = xtc_word(b, b.first + k + 1);
    }
    const uint64_t w = sh ? ((hi << sh) | (lo >> (64u - sh))) : hi;
    b.pos += (uint64_t)bits;
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
template <class BR>
__device__ __forceinline__ bool xtc_triple(BR& b, int bits, const Radix& rx, int out[3]) {
    const int q = (bits - 1) >> 3, r = bits - 8 * q;            // q full bytes, then r in [1, 8] bits
    if (bits <= 32) {
        // the small triples of a run (and the large one of a small system): the number fits 32 bits, so the byte order, both
        // quotients (u32 -> fp64 is exact, one v_cvt each way) and the remainders are 32-bit work - a third of the instructions
        // of the 64-bit path below, for two of the three triples of a water molecule
        const uint32_t raw = (uint32_t)xtc_get(b, bits);
        const uint32_t top = raw >> r, low = raw & ((1u << r) - 1u);
        const uint32_t w32 = (q ? (__builtin_bswap32(top) >> (32 - 8 * q)) : 0u) | (low << (8 * q));
        uint32_t qa = (uint32_t)((double)w32 * rx.inv2);
        const int32_t ra = (int32_t)(w32 - qa * rx.s2);          // true remainder in (-s2, 2 s2), s2 < 2^31: exact modulo 2^32
        if (ra < 0) --qa;
        else if ((uint32_t)ra >= rx.s2) ++qa;
        uint32_t qb = 0u;
        if (rx.s12 <= (uint64_t)w32) {                           // otherwise the quotient is 0 (s12 may not even fit 32 bits)
            const uint32_t d12 = (uint32_t)rx.s12;
            qb = (uint32_t)((double)w32 * rx.inv12);
            const int64_t rb = (int64_t)w32 - (int64_t)((uint64_t)qb * d12);
            if (rb < 0) --qb;
            else if (rb >= (int64_t)d12) ++qb;
        }
        out[2] = (int)(w32 - qa * rx.s2);
        out[1] = (int)(qa - qb * rx.s1);
        out[0] = (int)qb;
        return true;
    }
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
    // quotient through the reciprocal in fp64: w and 1/d carry a relative error of 2^-53 each, the product one more, so the estimate
    // is within 1 of floor(w / d) - which xtc_div repairs - as long as the quotient itself stays below 2^50: always for numbers of up
    // to 52 bits, and for wider ones when the divisor has the bits to spare (bits - 50 <= floor(log2 d): the large triple of a
    // 216 A box at precision 1000 is a 54-bit number over divisors of 18 and 36 bits; the integer division it used to take is two
    // ~150-instruction sequences per atom)
    if (bits <= 52 || (bits <= 64 && (rx.s2 >> (bits - 50)) != 0u)) {
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

// what is fixed for a whole frame
struct FrameSetup {
    float invp;
    uint32_t sizeint[3];
    int bitsizeint[3], bitsize;          // bitsize == 0: three separate fields
    Radix large;
};
// 0 ok, 1 corrupt header
__device__ __forceinline__ uint32_t xtc_setup(const vmd_xtc_frame_t& fi, FrameSetup& fs) {
    if (!(fi.precision > 0.0f)) return 1;
    fs.invp = 1.0f / fi.precision;
    if (fi.smallidx < XTC_FIRSTIDX || fi.smallidx >= XTC_LASTIDX) return 1;
    bool bad = false;
    for (int k = 0; k < 3; ++k) {
        const int64_t s = (int64_t)fi.maxint[k] - (int64_t)fi.minint[k] + 1;
        if (s <= 0 || s > 0xffffffffll) bad = true;
        fs.sizeint[k] = (uint32_t)s;
    }
    if (bad) return 1;
    fs.bitsizeint[0] = fs.bitsizeint[1] = fs.bitsizeint[2] = 0;
    if ((fs.sizeint[0] | fs.sizeint[1] | fs.sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) fs.bitsizeint[k] = xtc_bit_length(fs.sizeint[k]);
        fs.bitsize = 0;
    } else {
        // bit length of the product of the three ranges (each < 2^24): 128-bit product through two 64-bit halves
        const uint64_t p01 = (uint64_t)fs.sizeint[0] * fs.sizeint[1];                  // < 2^48
        const uint64_t lo = (p01 & 0xffffffffull) * fs.sizeint[2], hi = (p01 >> 32) * fs.sizeint[2];
        const uint64_t top = hi + (lo >> 32);                                          // product >> 32
        fs.bitsize = top ? 32 + xtc_bit_length(top) : xtc_bit_length(lo);
    }
    xtc_radix(fs.large, fs.sizeint[1], fs.sizeint[2]);
    return 0;
}

// One atom group at the reader's position: the large triple, the run flag, the run's small triples.  The decoder state at a group
// boundary is exactly (pos, i, smallidx, run); `small` is the radix record of `smallidx` on entry.  Advances i / smallidx / run
// (the caller refreshes `small` when smallidx moved).  Returns the status: 0 ok, 1 corrupt, 2 a number above 2^64.
template <class BR>
__device__ __forceinline__ uint32_t xtc_group(BR& br, const vmd_xtc_frame_t& fi, const FrameSetup& fs, int natoms, int& i, int& smallidx,
                                              int& run, const Radix& small, float* __restrict__ x, float* __restrict__ y,
                                              float* __restrict__ z) {
    const float invp = fs.invp;
    const int smallnum = (int)(small.s1 / 2u);             // small.s1 == kXtcMagic[smallidx]
    uint32_t st = 0;
    int cur[3], prev[3];
    if (fs.bitsize == 0) {
        for (int k = 0; k < 3; ++k) {
            const int nb = fs.bitsizeint[k];
            if (nb > 24) {      // two reads from the stream: sequenced explicitly (operands of | have no evaluation order)
                const uint64_t hi = xtc_get(br, nb - 24);
                const uint64_t lo = xtc_get(br, 24);
                cur[k] = (int)(uint32_t)((hi << 24) | lo);
            } else {
                cur[k] = (int)(uint32_t)xtc_get(br, nb);
            }
        }
    } else {
        if (!xtc_triple(br, fs.bitsize, fs.large, cur)) return 2;
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
        if (i + 1 + run / 3 > natoms) return 1;
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
        if (st) return st;
    } else {
        x[i] = ((float)cur[0] * invp) * 10.0f;
        y[i] = ((float)cur[1] * invp) * 10.0f;
        z[i] = ((float)cur[2] * invp) * 10.0f;
        ++i;
    }
    if (is_smaller) {
        smallidx += is_smaller;
        if (smallidx <= XTC_FIRSTIDX - 1 || smallidx >= XTC_LASTIDX) return 1;
    }
    return 0;
}

// Decode the atom groups that start at bit `pos` with atom `i`, state (smallidx, run), up to atom `end_i` (a group boundary or
// natoms).  Returns the status.
__device__ __forceinline__ uint32_t xtc_decode_range(const unsigned char* stream, const vmd_xtc_frame_t& fi, const FrameSetup& fs,
                                                     int natoms, uint64_t pos, int i, int smallidx, int run, int end_i,
                                                     float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
    Radix small;
    xtc_radix(small, (uint32_t)kXtcMagic[smallidx], (uint32_t)kXtcMagic[smallidx]);
    Bits br;
    xtc_open(br, stream, fi.nbytes, pos);
    const uint64_t nbits = 8ull * fi.nbytes;
    while (i < end_i) {
        const int before = smallidx;
        const uint32_t st = xtc_group(br, fi, fs, natoms, i, smallidx, run, small, x, y, z);
        if (st) return st;
        if (smallidx != before) xtc_radix(small, (uint32_t)kXtcMagic[smallidx], (uint32_t)kXtcMagic[smallidx]);
        if (br.pos > nbits) return 1;
    }
    return 0;
}

// ---- variant 1: one thread per frame, the whole stream
__global__ __launch_bounds__(64) void k_xtc_decode(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info,
                                                   int B, int natoms, float* __restrict__ xyz, size_t frame_stride,
                                                   size_t row_stride, uint32_t* __restrict__ status) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= B) return;
    const vmd_xtc_frame_t fi = info[f];
    float* x = xyz + (size_t)f * frame_stride;
    FrameSetup fs;
    uint32_t st = xtc_setup(fi, fs);
    if (!st) st = xtc_decode_range(raw + fi.offset, fi, fs, natoms, 0, 0, fi.smallidx, 0, natoms, x, x + row_stride, x + 2 * row_stride);
    status[f] = st;
}

// ---- variant 2: two passes.  A stream can only be ENTERED at an atom-group boundary whose bit position is known, and positions
// depend on every flag before them - but finding them needs no arithmetic on the coordinates: k_xtc_index (one thread per
// frame) follows only the flag bit, the 5-bit run code and the field widths, ~20 ALU instructions per group, and drops a
// checkpoint (bit position, atom index, smallidx, run length) at the first group boundary at or after every `chunk` atoms.
// k_xtc_chunks then decodes all chunks of all frames in parallel, one thread per (frame, chunk): thousands of waves instead of
// sixteen, and the expensive part (mixed-radix splits, float conversion, stores) is the parallel one.
struct Peek {                       // random-access MSB-first reads of up to 8 bits
    const uint64_t* words;
    uint64_t nwords;
    uint64_t word;                  // index of the cached pair
    uint64_t hi, lo;
};
__device__ __forceinline__ uint64_t xtc_peek(Peek& p, uint64_t pos, int bits) {
    const uint64_t w = pos >> 6;
    if (w != p.word) {
        const uint64_t w1 = w + 1;
        p.hi = __builtin_bswap64(p.words[w < p.nwords ? w : p.nwords - 1]);
        p.lo = __builtin_bswap64(p.words[w1 < p.nwords ? w1 : p.nwords - 1]);
        p.word = w;
    }
    const unsigned sh = (unsigned)(pos & 63ull);
    const uint64_t v = sh ? ((p.hi << sh) | (p.lo >> (64u - sh))) : p.hi;
    return v >> (64 - bits);
}

__global__ __launch_bounds__(64) void k_xtc_index(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info,
                                                  int B, int natoms, int chunk, int maxck, uint64_t* __restrict__ ck_pos,
                                                  uint32_t* __restrict__ ck_atom, uint32_t* __restrict__ ck_state,
                                                  uint32_t* __restrict__ nck, uint32_t* __restrict__ status) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= B) return;
    const vmd_xtc_frame_t fi = info[f];
    FrameSetup fs;
    uint32_t st = xtc_setup(fi, fs);
    uint32_t n = 0;
    if (!st) {
        const int large_bits = fs.bitsize ? fs.bitsize : fs.bitsizeint[0] + fs.bitsizeint[1] + fs.bitsizeint[2];
        Peek pk;
        pk.words = (const uint64_t*)(raw + fi.offset);
        pk.nwords = (fi.nbytes + 32ull) >> 3;
        pk.word = ~0ull;
        const uint64_t nbits = 8ull * fi.nbytes;
        uint64_t pos = 0;
        int i = 0, smallidx = fi.smallidx, run = 0;
        while (i < natoms) {
            if (i >= (int)n * chunk && (int)n < maxck) {
                const size_t o = (size_t)f * maxck + n;
                ck_pos[o] = pos;
                ck_atom[o] = (uint32_t)i;
                ck_state[o] = (uint32_t)smallidx | ((uint32_t)run << 8);
                ++n;
            }
            pos += (uint64_t)large_bits;
            int is_smaller = 0;
            const uint64_t code = xtc_peek(pk, pos, 6);          // flag + run code in one read
            if (code & 32u) {
                run = (int)(code & 31u);
                is_smaller = run % 3;
                run -= is_smaller;
                is_smaller--;
                pos += 6;
            } else {
                pos += 1;
            }
            if (run > 0) {
                if (i + 1 + run / 3 > natoms) { st = 1; break; }
                pos += (uint64_t)(run / 3) * (uint64_t)smallidx;
                i += 1 + run / 3;
            } else {
                i += 1;
            }
            smallidx += is_smaller;
            if (smallidx <= XTC_FIRSTIDX - 1 || smallidx >= XTC_LASTIDX) { st = 1; break; }
            if (pos > nbits) { st = 1; break; }
        }
    }
    nck[f] = n;
    status[f] = st;
}

__global__ __launch_bounds__(64) void k_xtc_chunks(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info,
                                                   int B, int natoms, int maxck, const uint64_t* __restrict__ ck_pos,
                                                   const uint32_t* __restrict__ ck_atom, const uint32_t* __restrict__ ck_state,
                                                   const uint32_t* __restrict__ nck, float* __restrict__ xyz, size_t frame_stride,
                                                   size_t row_stride, uint32_t* __restrict__ status) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int f = blockIdx.y;
    if (f >= B || status[f] == 1u || c >= (int)nck[f]) return;
    const vmd_xtc_frame_t fi = info[f];
    FrameSetup fs;
    if (xtc_setup(fi, fs)) return;
    const size_t o = (size_t)f * maxck + c;
    const int end_i = c + 1 < (int)nck[f] ? (int)ck_atom[o + 1] : natoms;
    const uint32_t state = ck_state[o];
    float* x = xyz + (size_t)f * frame_stride;
    const uint32_t st = xtc_decode_range(raw + fi.offset, fi, fs, natoms, ck_pos[o], (int)ck_atom[o], (int)(state & 255u), (int)(state >> 8),
                                         end_i, x, x + row_stride, x + 2 * row_stride);
    if (st) atomicMax(&status[f], st);
}

// ---- variant 3: one WAVE per frame.  The serial part of a stream is only the walk from group to group (flag bit, 5-bit run code,
// field widths); everything else - mixed-radix splits, conversions, stores - is independent per group.  So a wave walks its frame
// and decodes it 64 groups at a time:
//   * the stream lives in VGPRs, never in LDS (the pair kernel next door owns the CU's LDS): two banks of XTC_BANK registers, one
//     64-dword block per register.  While the walk is inside one bank (an "epoch") the other bank is in flight: its coalesced
//     256-byte loads are issued at the start of the epoch and nobody touches those registers until the walk gets there.  The loop body exists twice, once per bank role, so that no register is ever moved - a move of a
//     loaded value is a use, and a use waits (the first version rotated three registers and stalled a full memory latency per
//     block: 3.9 of 6.1 ms per 100k-atom frame);
//   * speculative walk: a group whose flag bit is 0 inherits (run, smallidx) and has the same length L as the one before it, so
//     lane k looks at the flag bit of "group k from here, if every flag before it is 0" (ds_bpermute out of the two blocks of the
//     window, picked by a uniform switch); a ballot finds the first flag that is set (or leaves the window / the frame), every
//     group before it is confirmed at once, the flagged group is stepped over with scalar arithmetic on its 6-bit code, and the
//     walk speculates again from there.  ~60 instructions confirm 1 + (zero-flag run) groups: 12 on average for the synthetic
//     water box, 40 - 64 for rigid water;
//   * every confirmed group gets a lane: (bit position, atom index, smallidx, run) - the complete decoder state at a group boundary.
//     When 64 are known (or the frame ends) all lanes decode their group with the code of variant 1 (xtc_group), reading the
//     stream through L1 (the bank loads touched those lines a moment ago).  The reciprocals of the 64 possible small radices are
//     computed once per wave, one per lane, and fetched with ds_bpermute instead of two fp64 divisions per group;
//   * gridDim.y waves share a frame: every one of them walks the whole stream (the walk needs no communication) and decodes every
//     gridDim.y-th tile, so a small batch still fills the chip.
#ifndef VMD_SHFL_U32
#define VMD_XTC_SETPRIO() __builtin_amdgcn_s_setprio(3)
#define VMD_SHFL_U32(v, src) ((uint32_t)__shfl((int)(v), (int)(src)))
#define VMD_READLANE_U32(v, lane) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(lane)))
#define VMD_XTC_BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)
#endif

__device__ __forceinline__ double xtc_shfl_f64(double v, int src) {
    uint64_t b;
    memcpy(&b, &v, 8);
    const uint32_t lo = VMD_SHFL_U32((uint32_t)b, src), hi = VMD_SHFL_U32((uint32_t)(b >> 32), src);
    b = ((uint64_t)hi << 32) | lo;
    double r;
    memcpy(&r, &b, 8);
    return r;
}

#define XTC_BANK 4                          // blocks per bank: 4 x 256 bytes in flight behind the walk (8 cost 8 more VGPRs for nothing measurable)
struct XtcBank { uint32_t r[XTC_BANK]; };   // 64-dword blocks of the stream, one dword per lane, as loaded (little-endian)

// at most 80 VGPRs: one of these waves then fits on a SIMD next to six waves of the pair kernel (72 VGPRs each) - the decode of batch
// k + 1 is meant to run UNDER the pair kernel of batch k, not after it
// Checkpoints (vmd_xtc_ck_t: the decoder state at a tile boundary).  A stream can only be entered where that state is known, and
// the first pass over a frame has to walk it from bit 0 - but it can leave breadcrumbs: with `ck_out` the wave that walks the frame
// drops the state at every ck_tiles-th tile boundary.  A later pass over the same frame (VIAMD re-evaluates a loaded trajectory
// after every script edit; mdlib keeps a frame-offset cache per file for the same reason) is given them as `ck_in` and splits the
// frame into SECTIONS: wave y walks and decodes sections y, y + gridDim.y, ... - no walk is repeated, and a frame occupies as many
// SIMDs as it has sections instead of one.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_xtc_wave(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info,
                                                 int B, int natoms, float* __restrict__ xyz, size_t frame_stride, size_t row_stride,
                                                 uint32_t* __restrict__ status, const vmd_xtc_ck_t* __restrict__ ck_in,
                                                 vmd_xtc_ck_t* __restrict__ ck_out, uint32_t* __restrict__ nck, int ck_max, int ck_tiles,
                                                 uint16_t* __restrict__ rec_out, uint32_t* __restrict__ nrec_out, uint32_t rec_stride) {
    // a serial walk issues one dependent instruction every few cycles: next to the VALU-bound waves of the pair kernel it would get a
    // seventh of the SIMD's issue slots and crawl.  At the highest wave priority it takes the slots it can use (a fifth of them) first.
    VMD_XTC_SETPRIO();
    const int f = blockIdx.x;
    const bool sectioned = ck_in != nullptr;
    const int nshare = sectioned ? 1 : (int)gridDim.y;
    int turn = sectioned ? 0 : (int)blockIdx.y;                   // tiles until this wave's next one
    const int lane = (int)threadIdx.x;
    if (f >= B) return;
    const int nsec = sectioned ? (int)nck[f] : 1;
    if (sectioned && ((int)blockIdx.y >= nsec || nsec > ck_max)) {
        if (nsec < 1 || nsec > ck_max) { if (lane == 0) atomicMax(&status[f], 1u); }      // a frame without checkpoints: the caller's bug
        return;
    }
    const bool emit = ck_out != nullptr && !sectioned && blockIdx.y == 0;
    uint32_t tile_no = 0, emitted = 0, next_ck_tile = 0;
    uint32_t ngroups = 0;                                         // groups decoded so far (rec_out: the record count of the frame)
    bool rec_ok = true;
    const vmd_xtc_frame_t fi = info[f];
    FrameSetup fs;
    uint32_t st = xtc_setup(fi, fs);
    if (!st && fi.nbytes >= (1ull << 27)) st = 2;                 // bit positions are kept in 32 bits here
    if (st) {
        if (lane == 0) atomicMax(&status[f], st);
        return;
    }
    const unsigned char* stream = raw + fi.offset;
    const uint32_t* dw = (const uint32_t*)stream;
    // a stream DMA'd straight out of the mapped file starts where the file has it: on a 4-byte boundary (XDR).  The walk reads
    // dwords; the per-lane reader works on 8-byte words and enters 4 bytes early, all its bit positions shifted by 32.
    const uint64_t phase = (uint64_t)((uintptr_t)stream & 4u);
    const uint32_t ndw = (uint32_t)((fi.nbytes + 32ull) >> 2);   // readable dwords (the pad behind the stream included)
    const uint32_t nbits = (uint32_t)(8ull * fi.nbytes);
    const uint32_t large_bits = (uint32_t)(fs.bitsize ? fs.bitsize : fs.bitsizeint[0] + fs.bitsizeint[1] + fs.bitsizeint[2]);
    float* x = xyz + (size_t)f * frame_stride;
    float* y = x + row_stride;
    float* z = y + row_stride;

    auto load_block = [&](uint32_t first) {
        uint32_t k = first + (uint32_t)lane;
        k = k < ndw ? k : ndw - 1;
        return dw[k];
    };
    uint32_t epoch = 0;                                           // first dword of the bank the walk is in
    uint32_t pos = 0;                                             // uniform walk state: next group's first bit, its first atom, ...
    int i = 0, smallidx = fi.smallidx, run = 0;
    int end_i = natoms;                                           // where this wave's walk ends (the next section's first atom)
    int g = 0;                                                    // groups waiting in the lanes
    uint32_t vpos = 0, vstate = (uint32_t)XTC_FIRSTIDX;           // per lane: the group this lane will decode
    int vatom = 0;
    bool done = false;

    // walks while the window stays inside `cur` (dwords [epoch, epoch + 64 * XTC_BANK)); `nxt` receives the blocks behind it
    auto run_epoch = [&](const XtcBank& cur, XtcBank& nxt) {
#pragma unroll
        for (int k = 0; k < XTC_BANK; ++k) nxt.r[k] = load_block(epoch + 64u * (uint32_t)XTC_BANK + 64u * (uint32_t)k);
        for (;;) {
            const bool finished = i >= end_i || st != 0;
            if (g == 64 || (finished && g > 0)) {
                tile_no += 1;
                const bool mine = turn == 0;
                turn = mine ? nshare - 1 : turn - 1;
                if (rec_out) {
                    // group records (vmd_hip.h): what the NEXT decode of this frame needs to place a group without walking to it - the
                    // bits it spans, how it moves smallidx, the run it leaves behind.  A lane's successor holds its after-state; the
                    // last group's is the walk's own.  Every sharing wave sees the same lanes; the tile's owner writes.
                    const int nl = lane + 1 < g ? lane + 1 : lane;
                    uint32_t npos = VMD_SHFL_U32(vpos, nl), nstate = VMD_SHFL_U32(vstate, nl);
                    if (lane == g - 1) { npos = pos; nstate = (uint32_t)smallidx | ((uint32_t)run << 8); }
                    const uint32_t len = npos - vpos;
                    const int dsm = (int)(nstate & 255u) - (int)(vstate & 255u) + 1;
                    const uint32_t rq = (nstate >> 8) / 3u;
                    const uint32_t slot = (tile_no - 1u) * 64u + (uint32_t)lane;
                    const bool bad = lane < g && (len > 1023u || dsm < 0 || dsm > 2 || rq > 15u || slot >= rec_stride);
                    if (VMD_XTC_BALLOT(bad)) rec_ok = false;
                    else if (mine && lane < g) rec_out[(size_t)f * rec_stride + slot] = (uint16_t)(len | ((uint32_t)dsm << 10) | (rq << 12));
                    ngroups += (uint32_t)g;
                }
                if (mine) {
                    // lane l computes small radix XTC_FIRSTIDX + l and its reciprocals (64 lanes = the 64 legal values of smallidx) - here,
                    // once per tile, not up front: five registers less across the walk, where the kernel sits at its 80-VGPR limit
                    const uint32_t my_magic = (uint32_t)kXtcMagic[XTC_FIRSTIDX + lane];
                    const double t_inv2 = 1.0 / (double)my_magic, t_inv12 = 1.0 / (double)((uint64_t)my_magic * my_magic);
                    const int sidx = (int)(vstate & 255u);
                    const int tl = (sidx - XTC_FIRSTIDX) & 63;
                    Radix small;
                    const uint32_t m = VMD_SHFL_U32(my_magic, tl);
                    small.s1 = small.s2 = m;
                    small.s12 = (uint64_t)m * m;
                    small.inv2 = xtc_shfl_f64(t_inv2, tl);
                    small.inv12 = xtc_shfl_f64(t_inv12, tl);
                    uint32_t lst = 0;
                    if (lane < g) {
                        BitsG br;
                        xtc_open(br, stream - phase, fi.nbytes + phase, (uint64_t)vpos + 8ull * phase);
                        int gi = vatom, gs = sidx, gr = (int)(vstate >> 8);
                        lst = xtc_group(br, fi, fs, natoms, gi, gs, gr, small, x, y, z);
                        if (!lst && br.pos > (uint64_t)nbits + 8ull * phase) lst = 1;
                    }
                    if (VMD_XTC_BALLOT(lst == 1u)) st = 1;
                    else if (VMD_XTC_BALLOT(lst == 2u) && !st) st = 2;
                }
                g = 0;
            }
            if (i >= end_i || st != 0) { done = true; return; }
            if (emit && g == 0 && tile_no == next_ck_tile && emitted < (uint32_t)ck_max) {
                // a tile starts here: (pos, i, smallidx, run) is everything a decoder needs to enter the stream at this bit
                if (lane == 0) {
                    vmd_xtc_ck_t c;
                    c.pos = pos; c.atom = (uint32_t)i; c.state = (uint32_t)smallidx | ((uint32_t)run << 8); c.reserved = 0;
                    ck_out[(size_t)f * ck_max + emitted] = c;
                }
                emitted += 1;
                next_ck_tile += (uint32_t)ck_tiles;
            }
            const uint32_t blk = ((pos >> 5) - epoch) >> 6;       // the window is blocks blk, blk + 1 of this epoch
            if (blk >= (uint32_t)XTC_BANK) return;                // the walk has left the bank
            const uint32_t base = epoch + 64u * blk;
            const int per = 1 + run / 3;                          // atoms and bits of a group that inherits (run, smallidx)
            const uint32_t L = large_bits + 1u + (uint32_t)(run / 3) * (uint32_t)smallidx;
            const uint32_t P = pos + large_bits + (uint32_t)lane * L;    // this lane's flag bit
            const uint32_t q = (P >> 5) - base, q1 = q + 1u;      // dwords q, q + 1 of the window hold it and the run code
            uint32_t a0, b0, a1, b1;
#define XTC_WIN(A, Bk) a0 = VMD_SHFL_U32(A, q & 63u); b0 = VMD_SHFL_U32(Bk, q & 63u); a1 = VMD_SHFL_U32(A, q1 & 63u); b1 = VMD_SHFL_U32(Bk, q1 & 63u)
            switch (blk) {
                case 0: XTC_WIN(cur.r[0], cur.r[1]); break;
                case 1: XTC_WIN(cur.r[1], cur.r[2]); break;
                case 2: XTC_WIN(cur.r[2], cur.r[3]); break;
#if XTC_BANK == 8
                case 3: XTC_WIN(cur.r[3], cur.r[4]); break;
                case 4: XTC_WIN(cur.r[4], cur.r[5]); break;
                case 5: XTC_WIN(cur.r[5], cur.r[6]); break;
                case 6: XTC_WIN(cur.r[6], cur.r[7]); break;
#endif
                default: XTC_WIN(cur.r[XTC_BANK - 1], nxt.r[0]); break;
            }
#undef XTC_WIN
            const uint32_t d0 = __builtin_bswap32((q & 64u) ? b0 : a0), d1 = __builtin_bswap32((q1 & 64u) ? b1 : a1);
            const uint32_t sh = P & 31u;
            const uint32_t code = (uint32_t)(((((uint64_t)d0 << 32) | (uint64_t)d1) << sh) >> 58);   // flag + run code
            // why the speculation ends at this lane: 3 = past the last atom, 2 = outside the window, 1 = flag set
            const uint32_t reason = (i + lane * per >= end_i) ? 3u : (q1 >= 128u) ? 2u : (code >> 5);
            const unsigned long long stop = VMD_XTC_BALLOT(reason != 0u);
            const int n0 = stop ? __builtin_ctzll(stop) : 64;     // groups with flag 0 in front of it
            const int room = 64 - g;
            const int take0 = n0 < room ? n0 : room;
            uint32_t why = 0, code_n0 = 0;
            if (n0 < 64) {
                why = VMD_READLANE_U32(reason, n0);
                code_n0 = VMD_READLANE_U32(code, n0);
            }
            const bool flagged = n0 < room && why == 1u;          // the group with the set flag goes along
            const int ntake = take0 + (flagged ? 1 : 0);
            const int rel = lane - g;
            if (rel >= 0 && rel < ntake) {
                vpos = pos + (uint32_t)rel * L;
                vatom = i + rel * per;
                vstate = (uint32_t)smallidx | ((uint32_t)run << 8);
            }
            g += ntake;
            pos += (uint32_t)take0 * L;
            i += take0 * per;
            if (flagged) {
                int nrun = (int)(code_n0 & 31u);
                const int is_smaller = nrun % 3;
                nrun -= is_smaller;
                pos += large_bits + 6u + (uint32_t)(nrun / 3) * (uint32_t)smallidx;
                i += 1 + nrun / 3;
                run = nrun;
                smallidx += is_smaller - 1;
                if (smallidx <= XTC_FIRSTIDX - 1 || smallidx >= XTC_LASTIDX) st = 1;
            }
            if (i > end_i || pos > nbits || ntake == 0) st = 1;   // a group past the last atom (or into the next section) / the last bit
        }
    };

    XtcBank bank_a, bank_b;
    // sectioned: wave y owns the consecutive sections [y * per_wave, (y + 1) * per_wave) and walks them as ONE stretch - entering the
    // stream costs a bank of loads, and a frame has up to 64 checkpoints however few waves the batch leaves it
    const int per_wave = sectioned ? (nsec + (int)gridDim.y - 1) / (int)gridDim.y : 1;
    for (int sec = sectioned ? (int)blockIdx.y * per_wave : 0; sec < nsec; sec = nsec) {
        if (sectioned) {
            const int sec_end = sec + per_wave < nsec ? sec + per_wave : nsec;
            const vmd_xtc_ck_t c = ck_in[(size_t)f * ck_max + sec];
            pos = c.pos; i = (int)c.atom; smallidx = (int)(c.state & 255u); run = (int)(c.state >> 8);
            end_i = sec_end < nsec ? (int)ck_in[(size_t)f * ck_max + sec_end].atom : natoms;
            // a checkpoint table that does not belong to this frame must not take the walk anywhere it cannot go
            if (pos > nbits || i < 0 || i >= natoms || end_i <= i || end_i > natoms || smallidx < XTC_FIRSTIDX || smallidx >= XTC_LASTIDX || run > 30 || run % 3) { st = 1; break; }
            g = 0; done = false;
        }
        epoch = ((pos >> 5) / (64u * (uint32_t)XTC_BANK)) * (64u * (uint32_t)XTC_BANK);
#pragma unroll
        for (int k = 0; k < XTC_BANK; ++k) { bank_a.r[k] = load_block(epoch + 64u * (uint32_t)k); bank_b.r[k] = 0; }
        for (;;) {
            run_epoch(bank_a, bank_b);
            if (done) break;
            epoch += 64u * (uint32_t)XTC_BANK;
            run_epoch(bank_b, bank_a);
            if (done) break;
            epoch += 64u * (uint32_t)XTC_BANK;
        }
        if (st) break;
    }
    if (emit && lane == 0) nck[f] = st ? 0u : emitted;
    if (emit && rec_out && nrec_out && lane == 0) nrec_out[f] = (st || !rec_ok) ? 0u : ngroups;
    if (st && lane == 0) atomicMax(&status[f], st);
}

// ---- decode from group records: no walk at all.  A first pass (k_xtc_wave with rec_out) left one 16-bit record per group - bits
// it spans (10), smallidx step + 1 (2), atoms of its run (4) - and the decoder state at every ck_tiles-th tile boundary.  A section
// (the groups between two checkpoints) is decoded tile by tile: 64 records, three wave-wide prefix sums (bit position, smallidx,
// first atom), one group per lane with the arithmetic of the host reader.  What was a dependent chain of ~10k rounds per c2 frame is
// 520 independent tiles; the kernel is bound by the frame's bytes (0.5 MB in, 1.2 MB out).
__device__ __forceinline__ uint32_t xtc_scan_incl(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = VMD_SHFL_U32(v, (lane - d) & 63);
        if (lane >= d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_xtc_records(const unsigned char* __restrict__ raw, const vmd_xtc_frame_t* __restrict__ info, int B, int natoms,
                                                    float* __restrict__ xyz, size_t frame_stride, size_t row_stride, uint32_t* __restrict__ status,
                                                    const vmd_xtc_ck_t* __restrict__ ck, const uint32_t* __restrict__ nck, int ck_max, int ck_tiles,
                                                    const uint16_t* __restrict__ rec, const uint32_t* __restrict__ nrec, uint32_t rec_stride) {
    VMD_XTC_SETPRIO();                                            // like k_xtc_wave: it runs in the wave slots the pair kernel leaves (80 VGPRs)
    const int f = blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (f >= B) return;
    const int nsec = (int)nck[f];
    const uint32_t ng = nrec[f];
    if (nsec < 1 || nsec > ck_max || ng == 0u || ng > rec_stride) {            // no (valid) records for this frame: the caller's bug
        if (lane == 0 && blockIdx.y == 0) atomicMax(&status[f], 1u);
        return;
    }
    const vmd_xtc_frame_t fi = info[f];
    FrameSetup fs;
    uint32_t st = xtc_setup(fi, fs);
    if (!st && fi.nbytes >= (1ull << 27)) st = 2;
    if (st) {
        if (lane == 0) atomicMax(&status[f], st);
        return;
    }
    const unsigned char* stream = raw + fi.offset;
    const uint64_t phase = (uint64_t)((uintptr_t)stream & 4u);
    const uint32_t nbits = (uint32_t)(8ull * fi.nbytes);
    float* x = xyz + (size_t)f * frame_stride;
    float* y = x + row_stride;
    float* z = y + row_stride;
    const uint32_t my_magic = (uint32_t)kXtcMagic[XTC_FIRSTIDX + lane];
    const double t_inv2 = 1.0 / (double)my_magic, t_inv12 = 1.0 / (double)((uint64_t)my_magic * my_magic);
    const uint16_t* frec = rec + (size_t)f * rec_stride;
    const int per_wave = (nsec + (int)gridDim.y - 1) / (int)gridDim.y;           // consecutive sections of one wave: one running state
    for (int sec = (int)blockIdx.y * per_wave; sec < nsec; sec = nsec) {
        const int sec_end = sec + per_wave < nsec ? sec + per_wave : nsec;
        const vmd_xtc_ck_t c = ck[(size_t)f * ck_max + sec];
        uint32_t pos = c.pos;
        int atom = (int)c.atom, sidx = (int)(c.state & 255u), run = (int)(c.state >> 8);
        const uint32_t q0 = (uint32_t)sec * (uint32_t)ck_tiles * 64u;
        const uint32_t q1 = sec_end < nsec ? (uint32_t)sec_end * (uint32_t)ck_tiles * 64u : ng;
        if (q0 >= q1 || q1 > ng || pos > nbits || atom < 0 || atom >= natoms || sidx < XTC_FIRSTIDX || sidx >= XTC_LASTIDX || run > 30 || run % 3) { st = 1; break; }
        for (uint32_t q = q0; q < q1 && !st; q += 64u) {
            const int n = (int)(q1 - q < 64u ? q1 - q : 64u);
            const uint32_t r = lane < n ? (uint32_t)frec[q + (uint32_t)lane] : 0u;
            const uint32_t len = r & 1023u, rq = r >> 12;
            const uint32_t dsm1 = lane < n ? ((r >> 10) & 3u) : 1u;               // smallidx step + 1
            const uint32_t nat = lane < n ? 1u + rq : 0u;
            const uint32_t s_len = xtc_scan_incl(len, lane), s_d = xtc_scan_incl(dsm1, lane), s_at = xtc_scan_incl(nat, lane);
            const uint32_t vpos = pos + s_len - len;
            const int vsidx = sidx + (int)(s_d - dsm1) - lane;                      // sum of (step + 1) over the lanes before, minus their count
            const int vatom = atom + (int)(s_at - nat);
            const uint32_t prq = VMD_SHFL_U32(rq, (lane - 1) & 63);
            const int vrun = lane == 0 ? run : 3 * (int)prq;
            const int tl = (vsidx - XTC_FIRSTIDX) & 63;
            Radix small;
            const uint32_t m = VMD_SHFL_U32(my_magic, tl);
            small.s1 = small.s2 = m;
            small.s12 = (uint64_t)m * m;
            small.inv2 = xtc_shfl_f64(t_inv2, tl);
            small.inv12 = xtc_shfl_f64(t_inv12, tl);
            uint32_t lst = 0;
            if (lane < n) {
                if (vsidx < XTC_FIRSTIDX || vsidx >= XTC_LASTIDX || vpos > nbits || vatom + (int)nat > natoms || dsm1 > 2u || rq > 10u) lst = 1;
                else {
                    BitsG br;
                    xtc_open(br, stream - phase, fi.nbytes + phase, (uint64_t)vpos + 8ull * phase);
                    int gi = vatom, gs = vsidx, gr = vrun;
                    lst = xtc_group(br, fi, fs, natoms, gi, gs, gr, small, x, y, z);
                    // the group must be the one the record describes: it ends where the next begins and leaves the recorded state
                    if (!lst && (br.pos != (uint64_t)vpos + len + 8ull * phase || gi != vatom + (int)nat || gs != vsidx + (int)dsm1 - 1 || gr != 3 * (int)rq)) lst = 1;
                }
            }
            if (VMD_XTC_BALLOT(lst == 1u)) st = 1;
            else if (VMD_XTC_BALLOT(lst == 2u)) st = 2;
            pos += VMD_READLANE_U32(s_len, 63);
            sidx += (int)VMD_READLANE_U32(s_d, 63) - n;
            atom += (int)VMD_READLANE_U32(s_at, 63);
            run = 3 * (int)VMD_READLANE_U32(rq, n - 1);
        }
        if (st) break;
        // a section ends where the next checkpoint begins (the last one: behind the last atom)
        if (sec_end < nsec) {
            const vmd_xtc_ck_t e = ck[(size_t)f * ck_max + sec_end];
            if (e.pos != pos || (int)e.atom != atom || e.state != ((uint32_t)sidx | ((uint32_t)run << 8))) st = 1;
        } else if (atom != natoms || pos > nbits) st = 1;
    }
    if (st && lane == 0) atomicMax(&status[f], st);
}

}  // namespace

extern "C" size_t vmd_hip_xtc_scratch_bytes(int B, int natoms, int chunk) {
    if (chunk < 64) chunk = 64;
    const size_t maxck = (size_t)(natoms + chunk - 1) / (size_t)chunk + 1;
    return (size_t)B * maxck * 16 + (size_t)B * 4 + 64;
}

// ---- plain floats (TRR / DCD frames DMA'd out of the mapped file): one thread per atom, one block row per frame
__global__ __launch_bounds__(256) void k_raw_f32(const unsigned char* __restrict__ raw, const vmd_f32_frame_t* __restrict__ info, int natoms,
                                                 float* __restrict__ xyz, size_t frame_stride, size_t row_stride) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= natoms) return;
    const vmd_f32_frame_t fi = info[f];
    float* out = xyz + (size_t)f * frame_stride + (size_t)i;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t w = *(const uint32_t*)(raw + fi.offset[c] + 4ull * fi.stride * (uint64_t)i);
        if (fi.flags & 1u) w = __builtin_bswap32(w);
        float v;
        __builtin_memcpy(&v, &w, 4);
        if (fi.scale != 1.0f) v = v * fi.scale;
        out[(size_t)c * row_stride] = v;
    }
}

extern "C" int vmd_hip_raw_f32_decode(void* stream, const unsigned char* raw, const vmd_f32_frame_t* info, int B, int natoms,
                                      float* xyz, size_t frame_stride, size_t row_stride) {
    if (B <= 0 || natoms <= 0) return 0;
    hipLaunchKernelGGL(k_raw_f32, dim3((unsigned)((natoms + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, raw, info, natoms, xyz,
                       frame_stride, row_stride);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int vmd_hip_xtc_decode_chunked(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                          float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int chunk,
                                          void* scratch) {
    if (B <= 0) return 0;
    if (chunk < 64) chunk = 64;
    const int maxck = (natoms + chunk - 1) / chunk + 1;
    uint64_t* ck_pos = (uint64_t*)scratch;
    uint32_t* ck_atom = (uint32_t*)(ck_pos + (size_t)B * maxck);
    uint32_t* ck_state = ck_atom + (size_t)B * maxck;
    uint32_t* nck = ck_state + (size_t)B * maxck;
    hipLaunchKernelGGL(k_xtc_index, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, raw, info, B, natoms, chunk,
                       maxck, ck_pos, ck_atom, ck_state, nck, status);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    hipLaunchKernelGGL(k_xtc_chunks, dim3((unsigned)((maxck + 63) / 64), (unsigned)B), dim3(64), 0, (hipStream_t)stream, raw, info, B,
                       natoms, maxck, (const uint64_t*)ck_pos, (const uint32_t*)ck_atom, (const uint32_t*)ck_state,
                       (const uint32_t*)nck, xyz, frame_stride, row_stride, status);
    return (int)hipGetLastError();
}

extern "C" int vmd_hip_xtc_decode(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                  float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_xtc_decode, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, raw, info, B, natoms, xyz,
                       frame_stride, row_stride, status);
    return (int)hipGetLastError();
}

static int g_xtc_waves = 0;     // waves per frame of k_xtc_wave; 0 = automatic (see vmd_hip_xtc_decode_wave)
extern "C" int vmd_hip_set_xtc_waves(int n) { const int old = g_xtc_waves; g_xtc_waves = n < 0 ? 0 : (n > 64 ? 64 : n); return old; }

// mode 0: plain; 1: emit checkpoints while decoding (ck, nck: out); 2: decode in sections from checkpoints (ck, nck: in);
// 3: walk only and emit checkpoints (xyz may be NULL)
static int xtc_launch_wave(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms, float* xyz,
                           size_t frame_stride, size_t row_stride, uint32_t* status, int mode, vmd_xtc_ck_t* ck, uint32_t* nck,
                           uint16_t* rec = nullptr, uint32_t* nrec = nullptr, size_t rec_stride = 0) {
    if (B <= 0) return 0;
    if (mode == 2 && rec && nrec) {
        // group records exist: nothing is walked, a section is a run of independent tiles
        int share = (8192 + B - 1) / B;
        if (share > VMD_XTC_CK_MAX) share = VMD_XTC_CK_MAX;
        if (share < 1) share = 1;
        if (hipMemsetAsync(status, 0, (size_t)B * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) return 1;
        const int ck_tiles_r = (natoms / 64 + 1 + VMD_XTC_CK_MAX - 1) / VMD_XTC_CK_MAX;
        hipLaunchKernelGGL(k_xtc_records, dim3((unsigned)B, (unsigned)share), dim3(64), 0, (hipStream_t)stream, raw, info, B, natoms, xyz, frame_stride,
                           row_stride, status, (const vmd_xtc_ck_t*)ck, (const uint32_t*)nck, VMD_XTC_CK_MAX, ck_tiles_r < 1 ? 1 : ck_tiles_r,
                           (const uint16_t*)rec, (const uint32_t*)nrec, (uint32_t)rec_stride);
        return (int)hipGetLastError();
    }
    int share = g_xtc_waves;
    if (mode == 2) {
        // sections are independent: up to VMD_XTC_CK_MAX waves per frame, as many as fill the chip a few times over
        share = (8192 + B - 1) / B;
        if (share > VMD_XTC_CK_MAX) share = VMD_XTC_CK_MAX;
        if (share < 1) share = 1;
    } else if (share <= 0) {
        // measured (profiles/r03_xtc_device_decode.txt): every sharing wave repeats the walk, so sharing pays only while SIMDs would
        // otherwise idle - 8 waves per frame for a batch of 128, 2 for 1 024, 1 from 2 048 frames on (2 waves per SIMD of the chip)
        share = (2048 + B / 2) / B;
        const int tiles = natoms / 128 + 1;                        // a frame has at most natoms groups; a wave should own a few tiles
        if (share > tiles) share = tiles;
        if (share > 8) share = 8;
        if (share < 1) share = 1;
    }
    if (hipMemsetAsync(status, 0, (size_t)B * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) return 1;
    // a frame has at most natoms / 64 + 1 tiles of 64 groups: a checkpoint every ck_tiles-th of them gives <= VMD_XTC_CK_MAX sections
    const int ck_tiles = (natoms / 64 + 1 + VMD_XTC_CK_MAX - 1) / VMD_XTC_CK_MAX;
    hipLaunchKernelGGL(k_xtc_wave, dim3((unsigned)B, (unsigned)share), dim3(64), 0, (hipStream_t)stream, raw, info, B, natoms, xyz, frame_stride,
                       row_stride, status, mode == 2 ? (const vmd_xtc_ck_t*)ck : (const vmd_xtc_ck_t*)nullptr,
                       (mode == 1 || mode == 3) ? ck : (vmd_xtc_ck_t*)nullptr, nck, VMD_XTC_CK_MAX, ck_tiles < 1 ? 1 : ck_tiles,
                       (mode == 1 || mode == 3) ? rec : (uint16_t*)nullptr, nrec, (uint32_t)rec_stride);
    return (int)hipGetLastError();
}

extern "C" int vmd_hip_xtc_decode_wave_rec(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                           float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int use,
                                           vmd_xtc_ck_t* ck, uint32_t* nck, uint16_t* rec, uint32_t* nrec, size_t rec_stride) {
    if (!ck || !nck || !rec || !nrec || rec_stride < (size_t)natoms) return (int)hipErrorInvalidValue;
    return xtc_launch_wave(stream, raw, info, B, natoms, xyz, frame_stride, row_stride, status, use ? 2 : 1, ck, nck, rec, nrec, rec_stride);
}

extern "C" int vmd_hip_xtc_decode_wave(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                       float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status) {
    return xtc_launch_wave(stream, raw, info, B, natoms, xyz, frame_stride, row_stride, status, 0, nullptr, nullptr);
}

extern "C" int vmd_hip_xtc_decode_wave_ck(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                          float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int use,
                                          vmd_xtc_ck_t* ck, uint32_t* nck) {
    if (!ck || !nck) return (int)hipErrorInvalidValue;
    return xtc_launch_wave(stream, raw, info, B, natoms, xyz, frame_stride, row_stride, status, use ? 2 : 1, ck, nck);
}
