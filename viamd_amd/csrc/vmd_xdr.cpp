// GROMACS XTC (compressed) and TRR (full precision) trajectory readers behind vmd_trajectory_i, plus a writer for both
// (the step before the hot path).
//
// VIAMD attaches these files through mdlib (`md_xtc_attach_from_file` / `md_trr_attach_from_file`,
// /root/reference/src/loader.cpp:147-150) and the evaluator pulls frames with md_trajectory_load_frame
// (src/viamd.cpp:465-467).  Same role here: a frame-offset index is built once on open (one small pread per frame, the
// job of mdlib's offset cache), after that load_frame is random access and re-entrant (pread + thread-local scratch), so
// the frames of one staged batch are decompressed on several host threads while the previous batch is in the kernels.
//
// Formats (XDR = big-endian 4-byte units; mdlib's sources are absent, so this follows the published xdrfile layout):
//   XTC frame : int magic (1995, or 2023 with a 64-bit byte count) | int natoms | int step | float time | float box[3][3]
//               | int natoms | natoms <= 9 ? float xyz[natoms][3]
//               : float precision | int minint[3] | int maxint[3] | int smallidx | int nbytes | bytes padded to 4
//     The byte stream is an MSB-first bit stream.  Every atom is either a "large" triple (three integers relative to
//     minint, packed as ONE mixed-radix number of bitlength(prod(sizeint)) bits, or three fixed-width fields when a
//     range exceeds 24 bits) optionally followed by a run of up to 8 "small" triples (differences to the previous
//     atom + smallnum, mixed-radix in base magicints[smallidx], smallidx bits per triple); a 1-bit flag + 5-bit code after
//     each large triple changes the run length and moves smallidx by -1/0/+1.  The first small triple of a run is
//     swapped with the large one in front of it (water: O is stored relative to H).
//     A mixed-radix number V is transmitted as its little-endian bytes: full bytes first, the top (partial) one last.
//   TRR frame : int magic (1993) | int 13 | string "GMX_trn_file" | int ir,e,box,vir,pres,top,sym,x,v,f sizes | int natoms
//               | int step | int nre | real t | real lambda | box[9] vir[9] pres[9] x[natoms][3] v[..] f[..]
//     real = float or double, derived from box_size / 9 (or x_size / 3 natoms).
//   Lengths are nm, the evaluator works in Angstrom: coordinates and box are multiplied by 10 in float
//   (XTC: fl(fl(int * fl(1/precision)) * 10), the xdrfile float followed by the unit conversion).
//   Box rows are the lattice vectors a = (x,0,0), b = (xy,y,0), c = (xz,yz,z) of md_unitcell_t.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

namespace {

typedef unsigned __int128 u128;

constexpr int kXtcMagic = 1995, kXtcMagicBig = 2023, kTrrMagic = 1993;
constexpr int FIRSTIDX = 9;
// round(2^(i/3)) with the historical irregularities of the format (5060, 524287, 8388607) kept: both sides must agree
constexpr int kMagicInts[] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64,
    80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290,
    1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003,
    16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031,
    131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561,
    832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021,
    4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216};
constexpr int LASTIDX = (int)(sizeof(kMagicInts) / sizeof(kMagicInts[0]));

inline uint32_t be32(const unsigned char* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return __builtin_bswap32(v);
}
inline uint64_t be64(const unsigned char* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}
inline float be_f32(const unsigned char* p) {
    const uint32_t u = be32(p);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline double be_f64(const unsigned char* p) {
    const uint64_t u = be64(p);
    double d;
    memcpy(&d, &u, 8);
    return d;
}

inline int bit_length(u128 v) {
    int n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}
// bits of the mixed-radix number that holds three digits of these bases (xdrfile: sizeofints)
inline int bits_of_product(const uint32_t s[3]) { return bit_length((u128)s[0] * s[1] * s[2]); }

bool fail(const char* fmt, const std::string& path) {
    char buf[600];
    snprintf(buf, sizeof(buf), fmt, path.c_str());
    vmd_set_last_error(buf);
    return false;
}

// ------------------------------------------------------------------ bit stream, MSB first
// one atom group reads at most 96 + 6 + 8 * 72 bits before the overrun check: the pad keeps a corrupt stream inside the buffer
constexpr size_t kStreamPad = 128;
struct BitReader {
    const unsigned char* base;   // the buffer carries kStreamPad zero bytes behind the payload
    uint64_t pos = 0;            // next bit
    uint64_t nbits;              // payload bits
    BitReader(const unsigned char* b, size_t nbytes) : base(b), nbits(8ull * nbytes) {}
    inline uint64_t get(int bits) {          // bits in [0, 56]: one unaligned big-endian 64-bit load covers any such field
        if (bits == 0) return 0;
        uint64_t w;
        memcpy(&w, base + (pos >> 3), 8);
        w = __builtin_bswap64(w) << (pos & 7);
        pos += (uint64_t)bits;
        return w >> (64 - bits);
    }
    bool overrun() const { return pos > nbits; }         // consumed bits beyond the payload
};

// one mixed-radix number of `bits` bits: little-endian bytes on the wire, the partial top byte last
inline u128 get_packed(BitReader& br, int bits) {
    if (bits <= 0) return 0;
    const int q = (bits - 1) >> 3, r = bits - 8 * q;         // q full bytes, then r in [1,8] bits
    if (bits <= 56) {
        const uint64_t w = br.get(bits);
        const uint64_t top = w >> r, low = w & ((1ull << r) - 1);
        const uint64_t le = q ? (__builtin_bswap64(top) >> (64 - 8 * q)) : 0ull;
        return (u128)(le | (low << (8 * q)));
    }
    u128 v = 0;
    for (int j = 0; j < q; ++j) v |= (u128)br.get(8) << (8 * j);
    v |= (u128)br.get(r) << (8 * q);
    return v;
}

// Three mixed-radix digits of one number.  The bases are fixed for a whole frame (large triples) or per smallidx (small
// triples), so the two divisions become multiplications by precomputed reciprocals: q = trunc(w * (1/d)) in double is
// within one of the true quotient for w < 2^52 and is corrected with the exact remainder; both quotients are taken from
// w itself so the two chains are independent.  Larger numbers take the plain integer path.
struct Radix {
    uint32_t s[3];
    uint64_t s12;
    double inv2, inv12;
    void set(const uint32_t z[3]) {
        s[0] = z[0]; s[1] = z[1]; s[2] = z[2];
        s12 = (uint64_t)z[1] * z[2];
        inv2 = 1.0 / (double)z[2];
        inv12 = 1.0 / (double)s12;
    }
};

inline uint64_t div_recip(uint64_t w, uint64_t d, double inv) {
    uint64_t q = (uint64_t)((double)w * inv);
    const int64_t r = (int64_t)(w - q * d);
    if (r < 0) --q;
    else if ((uint64_t)r >= d) ++q;
    return q;
}

inline void unpack3(u128 v, const Radix& rx, int out[3]) {
    if ((v >> 64) == 0) {
        const uint64_t w = (uint64_t)v;
        const uint64_t a = w / rx.s[2];
        out[2] = (int)(w - a * rx.s[2]);
        const uint64_t b = a / rx.s[1];
        out[1] = (int)(a - b * rx.s[1]);
        out[0] = (int)b;
    } else {
        const u128 a = v / rx.s[2];
        out[2] = (int)(uint64_t)(v - a * rx.s[2]);
        const u128 b = a / rx.s[1];
        out[1] = (int)(uint64_t)(a - b * rx.s[1]);
        out[0] = (int)(uint64_t)b;
    }
}

// read one triple of `bits` bits and split it; everything below 53 bits (every realistic frame) stays in 64-bit registers
inline void get_triple(BitReader& br, int bits, const Radix& rx, int out[3]) {
    if (bits <= 52 && bits > 0) {
        const int q = (bits - 1) >> 3, r = bits - 8 * q;
        const uint64_t raw = br.get(bits);
        const uint64_t top = raw >> r, low = raw & ((1ull << r) - 1);
        const uint64_t w = (q ? (__builtin_bswap64(top) >> (64 - 8 * q)) : 0ull) | (low << (8 * q));
        const uint64_t a = div_recip(w, rx.s[2], rx.inv2);       // w / s2
        const uint64_t b = div_recip(w, rx.s12, rx.inv12);       // w / (s1 s2)
        out[2] = (int)(w - a * rx.s[2]);
        out[1] = (int)(a - b * rx.s[1]);
        out[0] = (int)b;
    } else {
        unpack3(get_packed(br, bits), rx, out);
    }
}

struct BitWriter {
    std::vector<unsigned char> out;
    uint64_t acc = 0;
    int n = 0;
    inline void put(int bits, uint64_t v) {  // bits in [0, 56]
        if (!bits) return;
        acc = (acc << bits) | (v & ((~0ull) >> (64 - bits)));
        n += bits;
        while (n >= 8) { n -= 8; out.push_back((unsigned char)(acc >> n)); }
    }
    void finish() {
        if (n) { out.push_back((unsigned char)(acc << (8 - n))); n = 0; }
    }
};

inline void put_packed(BitWriter& bw, int bits, u128 v) {
    const int q = (bits - 1) >> 3, r = bits - 8 * q;
    for (int j = 0; j < q; ++j) bw.put(8, (uint64_t)(v >> (8 * j)) & 0xff);
    bw.put(r, (uint64_t)(v >> (8 * q)) & 0xff);
}

inline u128 pack3(const uint32_t s[3], const uint32_t d[3]) { return ((u128)d[0] * s[1] + d[1]) * s[2] + d[2]; }

// ------------------------------------------------------------------ file + index
enum Kind { KIND_XTC = 0, KIND_TRR = 1 };

struct FrameRec {
    uint64_t off;        // first byte of the frame
    uint64_t bytes;      // XTC: payload bytes (compressed stream, or 12 * natoms raw floats when `raw`); TRR: unused
    bool raw;            // XTC: <= 9 atoms are stored as plain floats
    uint32_t head;       // bytes from `off` to the payload (XTC) / to the box block (TRR)
    // TRR block sizes
    uint32_t box_size, skip_size, x_size, real_size;
    int32_t step;
    double time;
};

struct Xdr {
    int fd = -1;
    Kind kind = KIND_XTC;
    size_t num_atoms = 0;
    std::vector<FrameRec> frames;
    // XTC: decoder parameters and cell of every compressed frame, parsed once while indexing (load_raw's info-only pass - one call
    // per frame of every staged batch - then needs no system call)
    std::vector<vmd_raw_frame_t> raw_info;
    std::vector<vmd_unitcell_t> raw_cell;
    vmd_trajectory_i iface;
    std::string path;
    // the file mapped read-only on first request (raw_mapped_view): the evaluator DMAs compressed frames straight out of it
    std::mutex map_mtx;
    std::atomic<const unsigned char*> map{nullptr};
    size_t map_bytes = 0;
    bool map_failed = false;
    std::vector<uint64_t> stream_off;
};

bool read_at(int fd, void* dst, size_t n, uint64_t off) {
    char* p = (char*)dst;
    while (n) {
        const ssize_t r = pread(fd, p, n, (off_t)off);
        if (r <= 0) return false;
        p += r; off += (uint64_t)r; n -= (size_t)r;
    }
    return true;
}

vmd_unitcell_t cell_from_box_nm(const float b[9]) {
    vmd_unitcell_t c;
    memset(&c, 0, sizeof(c));
    c.x = b[0] * 10.0f; c.y = b[4] * 10.0f; c.z = b[8] * 10.0f;
    c.xy = b[3] * 10.0f; c.xz = b[6] * 10.0f; c.yz = b[7] * 10.0f;
    if (c.x > 0.0f && c.y > 0.0f && c.z > 0.0f) c.flags = VMD_UNITCELL_PBC_ALL;
    else memset(&c, 0, sizeof(c));
    return c;
}

// ------------------------------------------------------------------ XTC
// Decompress one coordinate block.  `hdr` points at precision (36 bytes: precision, minint, maxint, smallidx), `data` at
// the bit stream (padded).  Writes Angstrom to x/y/z (any may be NULL).
const char* xtc_decode(const unsigned char* hdr, const unsigned char* data, size_t nbytes, size_t natoms,
                       float* x, float* y, float* z) {
    const float precision = be_f32(hdr);
    if (!(precision > 0.0f) || !std::isfinite(precision)) return "XTC '%s': bad precision";
    const float invp = 1.0f / precision;
    int minint[3], maxint[3];
    for (int k = 0; k < 3; ++k) { minint[k] = (int)be32(hdr + 4 + 4 * k); maxint[k] = (int)be32(hdr + 16 + 4 * k); }
    int smallidx = (int)be32(hdr + 28);
    if (smallidx < FIRSTIDX || smallidx >= LASTIDX) return "XTC '%s': corrupt frame (smallidx)";
    uint32_t sizeint[3];
    int bitsizeint[3] = {0, 0, 0}, bitsize;
    for (int k = 0; k < 3; ++k) {
        const int64_t s = (int64_t)maxint[k] - (int64_t)minint[k] + 1;
        if (s <= 0 || s > 0xffffffffll) return "XTC '%s': corrupt frame (integer range)";
        sizeint[k] = (uint32_t)s;
    }
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) bitsizeint[k] = bit_length(sizeint[k]);
        bitsize = 0;
    } else {
        bitsize = bits_of_product(sizeint);
    }
    int smaller = kMagicInts[std::max(FIRSTIDX, smallidx - 1)] / 2;
    int smallnum = kMagicInts[smallidx] / 2;
    Radix large, small;
    large.set(sizeint);
    auto set_small = [&](int idx) {
        const uint32_t m[3] = {(uint32_t)kMagicInts[idx], (uint32_t)kMagicInts[idx], (uint32_t)kMagicInts[idx]};
        small.set(m);
    };
    set_small(smallidx);

    BitReader br(data, nbytes);
    auto emit = [&](size_t o, const int c[3]) {
        if (x) x[o] = ((float)c[0] * invp) * 10.0f;
        if (y) y[o] = ((float)c[1] * invp) * 10.0f;
        if (z) z[o] = ((float)c[2] * invp) * 10.0f;
    };
    size_t i = 0;
    int run = 0;
    while (i < natoms) {
        int cur[3], prev[3];
        if (bitsize == 0) {
            for (int k = 0; k < 3; ++k) {
                // a range above 24 bits: three plain MSB-first fields of up to 32 bits
                const int b = bitsizeint[k];
                if (b > 24) {       // two reads from the stream: sequenced explicitly (operands of | have no evaluation order)
                    const uint64_t hi = br.get(b - 24);
                    const uint64_t lo = br.get(24);
                    cur[k] = (int)(uint32_t)((hi << 24) | lo);
                } else {
                    cur[k] = (int)(uint32_t)br.get(b);
                }
            }
        } else {
            get_triple(br, bitsize, large, cur);
        }
        for (int k = 0; k < 3; ++k) { cur[k] += minint[k]; prev[k] = cur[k]; }
        int is_smaller = 0;
        if (br.get(1)) {
            run = (int)br.get(5);
            is_smaller = run % 3;
            run -= is_smaller;
            is_smaller--;
        }
        if (run > 0) {
            if (i + 1 + (size_t)(run / 3) > natoms) return "XTC '%s': corrupt frame (run past the last atom)";
            for (int k = 0; k < run; k += 3) {
                int d[3], nxt[3];
                get_triple(br, smallidx, small, d);
                for (int c = 0; c < 3; ++c) nxt[c] = d[c] + prev[c] - smallnum;
                if (k == 0) {
                    // the large triple in front of the run is the SECOND atom of the pair
                    emit(i, nxt);
                    emit(i + 1, cur);
                    i += 2;
                } else {
                    emit(i, nxt);
                    i += 1;
                }
                for (int c = 0; c < 3; ++c) prev[c] = nxt[c];
            }
        } else {
            emit(i, cur);
            i += 1;
        }
        smallidx += is_smaller;
        if (is_smaller < 0) {
            if (smallidx < FIRSTIDX) return "XTC '%s': corrupt frame (smallidx underflow)";
            smallnum = smaller;
            smaller = smallidx > FIRSTIDX ? kMagicInts[smallidx - 1] / 2 : 0;
        } else if (is_smaller > 0) {
            if (smallidx >= LASTIDX) return "XTC '%s': corrupt frame (smallidx overflow)";
            smaller = smallnum;
            smallnum = kMagicInts[smallidx] / 2;
        }
        if (is_smaller) set_small(smallidx);
        if (br.overrun()) return "XTC '%s': corrupt frame (bit stream ends early)";
    }
    return nullptr;
}

bool xtc_index(Xdr* d, uint64_t file_bytes) {
    uint64_t off = 0;
    while (off + 56 <= file_bytes) {
        unsigned char h[100];
        const size_t want = (size_t)std::min<uint64_t>(sizeof(h), file_bytes - off);
        if (!read_at(d->fd, h, want, off)) break;
        const int magic = (int)be32(h);
        if (magic != kXtcMagic && magic != kXtcMagicBig) {
            if (off == 0) return fail("XTC '%s': not an XTC file (bad magic number)", d->path);
            break;                                               // trailing garbage: keep the frames before it
        }
        const uint32_t natoms = be32(h + 4);
        if (natoms == 0 || be32(h + 52) != natoms) {
            if (off == 0) return fail("XTC '%s': corrupt first frame header", d->path);
            break;
        }
        if (d->frames.empty()) d->num_atoms = natoms;
        else if (natoms != d->num_atoms) return fail("XTC '%s': the atom count changes between frames", d->path);
        FrameRec r;
        memset(&r, 0, sizeof(r));
        r.off = off;
        r.step = (int32_t)be32(h + 8);
        r.time = be_f32(h + 12);
        uint64_t total;
        if (natoms <= 9) {
            r.head = 56; r.bytes = 12ull * natoms; r.raw = true;
            total = 56 + 12ull * natoms;
        } else {
            const bool big = magic == kXtcMagicBig;
            if (want < (size_t)(big ? 100 : 96)) break;
            r.bytes = big ? be64(h + 88) : be32(h + 88);
            r.head = big ? 96 : 92;
            // the byte count comes from the file (64 bits for magic 2023): bound it by what the file can hold BEFORE any
            // arithmetic on it, so that neither the 4-byte padding nor off + total can wrap; an empty compressed stream is corrupt
            if (r.bytes == 0 || r.bytes > file_bytes - off - r.head) {
                if (off == 0) return fail("XTC '%s': corrupt first frame (payload size)", d->path);
                break;
            }
            total = r.head + ((r.bytes + 3) & ~3ull);
        }
        if (total > file_bytes - off) {                          // truncated last frame (only the padding can be missing here)
            if (!(natoms > 9 && r.head + r.bytes <= file_bytes - off)) break;
            total = file_bytes - off;
        }
        d->frames.push_back(r);
        vmd_raw_frame_t info;
        memset(&info, 0, sizeof(info));
        vmd_unitcell_t cell;
        memset(&cell, 0, sizeof(cell));
        if (!r.raw) {
            info.codec = VMD_RAW_CODEC_XTC;
            info.precision = be_f32(h + 56);
            for (int k = 0; k < 3; ++k) { info.minint[k] = (int32_t)be32(h + 60 + 4 * k); info.maxint[k] = (int32_t)be32(h + 72 + 4 * k); }
            info.smallidx = (int32_t)be32(h + 84);
            info.nbytes = r.bytes;
            float box[9];
            for (int k = 0; k < 9; ++k) box[k] = be_f32(h + 16 + 4 * k);
            cell = cell_from_box_nm(box);
        }
        d->raw_info.push_back(info);
        d->raw_cell.push_back(cell);
        off += total;
    }
    if (d->frames.empty()) return fail("XTC '%s': no complete frame", d->path);
    return true;
}

bool xtc_load(Xdr* d, const FrameRec& r, vmd_unitcell_t* cell, float* x, float* y, float* z) {
    static thread_local std::vector<unsigned char> buf;
    const size_t payload = (size_t)r.bytes;
    buf.resize(r.head + payload + kStreamPad);
    if (!read_at(d->fd, buf.data(), r.head + payload, r.off)) return fail("XTC '%s': truncated frame", d->path);
    memset(buf.data() + r.head + payload, 0, kStreamPad);
    float box[9];
    for (int k = 0; k < 9; ++k) box[k] = be_f32(buf.data() + 16 + 4 * k);
    *cell = cell_from_box_nm(box);
    if (r.raw) {
        const unsigned char* p = buf.data() + r.head;
        for (size_t i = 0; i < d->num_atoms; ++i, p += 12) {
            if (x) x[i] = be_f32(p) * 10.0f;
            if (y) y[i] = be_f32(p + 4) * 10.0f;
            if (z) z[i] = be_f32(p + 8) * 10.0f;
        }
        return true;
    }
    const char* err = xtc_decode(buf.data() + 56, buf.data() + r.head, payload, d->num_atoms, x, y, z);
    return err ? fail(err, d->path) : true;
}

// ------------------------------------------------------------------ TRR
bool trr_index(Xdr* d, uint64_t file_bytes) {
    uint64_t off = 0;
    while (off + 84 <= file_bytes) {
        unsigned char h[96];
        const size_t want = (size_t)std::min<uint64_t>(sizeof(h), file_bytes - off);
        if (!read_at(d->fd, h, want, off)) break;
        if ((int)be32(h) != kTrrMagic) {
            if (off == 0) return fail("TRR '%s': not a TRR file (bad magic number)", d->path);
            break;
        }
        const uint32_t slen = be32(h + 4), sl2 = be32(h + 8);
        if (slen != sl2 + 1 || sl2 > 64) return fail("TRR '%s': corrupt version string", d->path);
        const size_t p0 = 12 + ((sl2 + 3) & ~3u);                // first of the 13 header ints
        if (p0 + 52 + 8 > want) break;
        uint32_t v[13];
        for (int k = 0; k < 13; ++k) v[k] = be32(h + p0 + 4 * k);
        // ir e box vir pres top sym x v f natoms step nre
        const uint32_t box_size = v[2], vir = v[3], pres = v[4], x_size = v[7], v_size = v[8], f_size = v[9], natoms = v[10];
        if (natoms == 0) return fail("TRR '%s': frame without atoms", d->path);
        uint32_t real_size = 0;
        if (box_size) real_size = box_size / 9;
        else if (x_size) real_size = x_size / (3 * natoms);
        else if (v_size) real_size = v_size / (3 * natoms);
        else if (f_size) real_size = f_size / (3 * natoms);
        if (real_size != 4 && real_size != 8) return fail("TRR '%s': cannot determine the precision of a frame", d->path);
        const size_t head = p0 + 52 + 2 * real_size;
        if (head > want) break;
        const uint64_t total = head + (uint64_t)v[0] + v[1] + box_size + vir + pres + v[5] + v[6] + x_size + v_size + f_size;
        if (off + total > file_bytes) break;
        if (x_size) {                                            // frames without positions (velocity-only output) are skipped
            if (x_size != 3ull * natoms * real_size) return fail("TRR '%s': position block has the wrong size", d->path);
            if (d->frames.empty()) d->num_atoms = natoms;
            else if (natoms != d->num_atoms) return fail("TRR '%s': the atom count changes between frames", d->path);
            FrameRec r;
            memset(&r, 0, sizeof(r));
            r.off = off;
            r.head = (uint32_t)(head + v[0] + v[1]);
            r.box_size = box_size;
            r.skip_size = vir + pres + v[5] + v[6];
            r.x_size = x_size;
            r.real_size = real_size;
            r.step = (int32_t)v[11];
            r.time = real_size == 4 ? (double)be_f32(h + p0 + 52) : be_f64(h + p0 + 52);
            d->frames.push_back(r);
        }
        off += total;
    }
    if (d->frames.empty()) return fail("TRR '%s': no complete frame with positions", d->path);
    return true;
}

bool trr_load(Xdr* d, const FrameRec& r, vmd_unitcell_t* cell, float* x, float* y, float* z) {
    static thread_local std::vector<unsigned char> buf;
    const size_t n = (size_t)r.box_size + r.skip_size + r.x_size;
    buf.resize(n);
    if (!read_at(d->fd, buf.data(), n, r.off + r.head)) return fail("TRR '%s': truncated frame", d->path);
    memset(cell, 0, sizeof(*cell));
    if (r.box_size) {
        float box[9];
        for (int k = 0; k < 9; ++k) box[k] = r.real_size == 4 ? be_f32(buf.data() + 4 * k) : (float)be_f64(buf.data() + 8 * k);
        *cell = cell_from_box_nm(box);
    }
    const unsigned char* p = buf.data() + r.box_size + r.skip_size;
    if (r.real_size == 4) {
        for (size_t i = 0; i < d->num_atoms; ++i, p += 12) {
            if (x) x[i] = be_f32(p) * 10.0f;
            if (y) y[i] = be_f32(p + 4) * 10.0f;
            if (z) z[i] = be_f32(p + 8) * 10.0f;
        }
    } else {
        for (size_t i = 0; i < d->num_atoms; ++i, p += 24) {
            if (x) x[i] = (float)(be_f64(p) * 10.0);
            if (y) y[i] = (float)(be_f64(p + 8) * 10.0);
            if (z) z[i] = (float)(be_f64(p + 16) * 10.0);
        }
    }
    return true;
}

size_t xdr_num_frames(void* inst) { return ((Xdr*)inst)->frames.size(); }
size_t xdr_num_atoms(void* inst) { return ((Xdr*)inst)->num_atoms; }

bool xdr_load_frame(void* inst, int64_t idx, vmd_frame_header_t* hdr, float* x, float* y, float* z) {
    Xdr* d = (Xdr*)inst;
    if (idx < 0 || (size_t)idx >= d->frames.size()) return fail("trajectory '%s': frame index out of range", d->path);
    const FrameRec& r = d->frames[(size_t)idx];
    vmd_unitcell_t cell;
    const bool ok = d->kind == KIND_XTC ? xtc_load(d, r, &cell, x, y, z) : trr_load(d, r, &cell, x, y, z);
    if (!ok) return false;
    if (hdr) {
        memset(hdr, 0, sizeof(*hdr));
        hdr->num_atoms = d->num_atoms;
        hdr->index = idx;
        hdr->timestamp = r.time;
        hdr->unitcell = cell;
    }
    return true;
}

// The frame as stored, for the device decoder (vmd_trajectory_i::load_raw): decoder parameters in host byte order, the bit
// stream copied verbatim.  false = not available raw (TRR, frames of <= 9 atoms stored as floats): the caller uses load_frame.
// TRR: the position block as it lies in the file (big-endian nm, xyz interleaved), for the evaluator's mapped-file path; `info` only
bool trr_load_raw(Xdr* d, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst) {
    if (idx < 0 || (size_t)idx >= d->frames.size() || !info || dst) return false;
    const FrameRec& r = d->frames[(size_t)idx];
    if (r.real_size != 4) return false;                       // double precision files go through load_frame
    memset(info, 0, sizeof(*info));
    info->codec = VMD_RAW_CODEC_F32;
    const uint64_t x0 = (uint64_t)r.box_size + r.skip_size;
    for (int a = 0; a < 3; ++a) info->f32_offset[a] = x0 + 4u * (unsigned)a;
    info->f32_stride = 3;
    info->f32_flags = VMD_RAW_F32_BIG_ENDIAN;
    info->f32_scale = 10.0f;
    info->nbytes = x0 + r.x_size;
    if (hdr) {
        memset(hdr, 0, sizeof(*hdr));
        hdr->num_atoms = d->num_atoms;
        hdr->index = idx;
        hdr->timestamp = r.time;
        if (r.box_size) {
            unsigned char b[36];
            const unsigned char* mb = d->map.load(std::memory_order_acquire);
            if (mb) memcpy(b, mb + r.off + r.head, sizeof(b));
            else if (!read_at(d->fd, b, sizeof(b), r.off + r.head)) return false;
            float box[9];
            for (int k = 0; k < 9; ++k) box[k] = be_f32(b + 4 * k);
            hdr->unitcell = cell_from_box_nm(box);
        }
    }
    return true;
}

bool xdr_load_raw(void* inst, int64_t idx, vmd_frame_header_t* hdr, vmd_raw_frame_t* info, void* dst, size_t cap) {
    Xdr* d = (Xdr*)inst;
    if (d->kind == KIND_TRR) return trr_load_raw(d, idx, hdr, info, dst);
    if (d->kind != KIND_XTC || idx < 0 || (size_t)idx >= d->frames.size() || !info) return false;
    const FrameRec& r = d->frames[(size_t)idx];
    if (r.raw) return false;
    *info = d->raw_info[(size_t)idx];
    if (hdr) {
        memset(hdr, 0, sizeof(*hdr));
        hdr->num_atoms = d->num_atoms;
        hdr->index = idx;
        hdr->timestamp = r.time;
        hdr->unitcell = d->raw_cell[(size_t)idx];
    }
    if (dst) {
        if (cap < r.bytes) return fail("XTC '%s': raw frame buffer too small", d->path);
        if (!read_at(d->fd, dst, (size_t)r.bytes, r.off + r.head)) return fail("XTC '%s': truncated frame", d->path);
    }
    return true;
}

// The file mapped for the evaluator's DMA (vmd_trajectory_i::raw_mapped_view).  Mapped once, on first request; frames this reader
// cannot hand over raw make the whole file unmappable (load_raw refuses them one by one, the evaluator then decodes on the host).
bool xdr_raw_mapped_view(void* inst, vmd_raw_mapped_view_t* out) {
    Xdr* d = (Xdr*)inst;
    if (!out || d->frames.empty()) return false;
    std::lock_guard<std::mutex> lk(d->map_mtx);
    if (d->map_failed) return false;
    if (!d->map.load()) {
        struct stat sb;
        if (fstat(d->fd, &sb) != 0 || sb.st_size <= 0) { d->map_failed = true; return false; }
        const FrameRec& last = d->frames.back();
        const uint64_t last_end = last.off + last.head + (d->kind == KIND_XTC ? last.bytes : (uint64_t)last.box_size + last.skip_size + last.x_size);
        if (last_end > (uint64_t)sb.st_size) { d->map_failed = true; return false; }   // truncated since it was indexed
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, d->fd, 0);
        if (m == MAP_FAILED) { d->map_failed = true; return false; }
        (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        d->stream_off.resize(d->frames.size());
        for (size_t i = 0; i < d->frames.size(); ++i) {
            if (d->kind == KIND_XTC ? d->frames[i].raw : d->frames[i].real_size != 4) { munmap(m, (size_t)sb.st_size); d->map_failed = true; return false; }
            d->stream_off[i] = d->frames[i].off + d->frames[i].head;
        }
        d->map_bytes = (size_t)sb.st_size;
        d->map.store((const unsigned char*)m, std::memory_order_release);
    }
    out->base = d->map.load();
    out->bytes = d->map_bytes;
    out->stream_offset = d->stream_off.data();
    out->codec = d->kind == KIND_XTC ? VMD_RAW_CODEC_XTC : VMD_RAW_CODEC_F32;
    return true;
}

// ------------------------------------------------------------------ writer
struct Writer {
    FILE* fh = nullptr;
    Kind kind = KIND_XTC;
    size_t num_atoms = 0;
    float precision = 1000.0f;
    std::string path;
    std::vector<int> ints;
    std::vector<unsigned char> rec;
};

inline void put_be32(std::vector<unsigned char>& o, uint32_t v) {
    o.push_back((unsigned char)(v >> 24)); o.push_back((unsigned char)(v >> 16));
    o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v);
}
inline void put_f32(std::vector<unsigned char>& o, float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    put_be32(o, u);
}

void box_nm(const vmd_unitcell_t* c, float b[9]) {
    for (int k = 0; k < 9; ++k) b[k] = 0.0f;
    if (!c) return;
    b[0] = c->x * 0.1f; b[4] = c->y * 0.1f; b[8] = c->z * 0.1f;
    b[3] = c->xy * 0.1f; b[6] = c->xz * 0.1f; b[7] = c->yz * 0.1f;
}

// The compression side of the format (xdrfile: xdr3dfcoord, write branch).  `ints` holds natoms triples and is permuted in
// place by the water swap.
bool xtc_encode(Writer* w, std::vector<unsigned char>& o) {
    const size_t natoms = w->num_atoms;
    int* ip = w->ints.data();
    int minint[3] = {INT_MAX, INT_MAX, INT_MAX}, maxint[3] = {INT_MIN, INT_MIN, INT_MIN};
    int64_t mindiff = INT_MAX;
    for (size_t i = 0; i < natoms; ++i) {
        for (int k = 0; k < 3; ++k) {
            minint[k] = std::min(minint[k], ip[3 * i + k]);
            maxint[k] = std::max(maxint[k], ip[3 * i + k]);
        }
        if (i) {
            int64_t diff = 0;
            for (int k = 0; k < 3; ++k) diff += std::llabs((int64_t)ip[3 * i + k] - ip[3 * i - 3 + k]);
            if (diff < mindiff) mindiff = diff;
        }
    }
    uint32_t sizeint[3];
    for (int k = 0; k < 3; ++k) {
        if ((int64_t)maxint[k] - minint[k] >= INT_MAX - 2) return fail("XTC '%s': coordinate range too large to compress", w->path);
        sizeint[k] = (uint32_t)(maxint[k] - minint[k] + 1);
    }
    int bitsizeint[3] = {0, 0, 0}, bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) bitsizeint[k] = bit_length(sizeint[k]);
        bitsize = 0;
    } else {
        bitsize = bits_of_product(sizeint);
    }
    int smallidx = FIRSTIDX;
    while (smallidx < LASTIDX - 1 && kMagicInts[smallidx] < mindiff) smallidx++;
    for (int k = 0; k < 3; ++k) put_be32(o, (uint32_t)minint[k]);
    for (int k = 0; k < 3; ++k) put_be32(o, (uint32_t)maxint[k]);
    put_be32(o, (uint32_t)smallidx);

    const int maxidx = std::min(LASTIDX - 1, smallidx + 8);
    const int minidx = maxidx - 8;
    int smaller = kMagicInts[std::max(FIRSTIDX, smallidx - 1)] / 2;
    int smallnum = kMagicInts[smallidx] / 2;
    uint32_t sizesmall[3] = {(uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx]};
    const int larger = kMagicInts[maxidx] / 2;

    BitWriter bw;
    bw.out.reserve(natoms * 5);
    size_t i = 0;
    int prevrun = -1;
    int prev[3] = {0, 0, 0};
    auto near = [](const int* a, const int* b, int64_t lim) {
        return std::llabs((int64_t)a[0] - b[0]) < lim && std::llabs((int64_t)a[1] - b[1]) < lim && std::llabs((int64_t)a[2] - b[2]) < lim;
    };
    while (i < natoms) {
        int* cur = ip + 3 * i;
        int is_small = 0, is_smaller;
        if (smallidx < maxidx && i >= 1 && near(cur, prev, larger)) is_smaller = 1;
        else if (smallidx > minidx) is_smaller = -1;
        else is_smaller = 0;
        if (i + 1 < natoms && near(cur, cur + 3, smallnum)) {
            for (int k = 0; k < 3; ++k) std::swap(cur[k], cur[3 + k]);        // water: H first, O relative to it
            is_small = 1;
        }
        uint32_t t[3];
        for (int k = 0; k < 3; ++k) t[k] = (uint32_t)(cur[k] - minint[k]);
        if (bitsize == 0) {
            for (int k = 0; k < 3; ++k) {
                const int b = bitsizeint[k];
                if (b > 24) { bw.put(b - 24, t[k] >> 24); bw.put(24, t[k] & 0xffffffu); }
                else bw.put(b, t[k]);
            }
        } else {
            put_packed(bw, bitsize, pack3(sizeint, t));
        }
        for (int k = 0; k < 3; ++k) prev[k] = cur[k];
        cur += 3;
        ++i;
        int run = 0;
        uint32_t tmp[24];
        if (is_small == 0 && is_smaller == -1) is_smaller = 0;
        while (is_small && run < 8 * 3) {
            if (is_smaller == -1) {
                int64_t s = 0;
                for (int k = 0; k < 3; ++k) { const int64_t dlt = (int64_t)cur[k] - prev[k]; s += dlt * dlt; }
                if (s >= (int64_t)smaller * smaller) is_smaller = 0;
            }
            for (int k = 0; k < 3; ++k) tmp[run++] = (uint32_t)(cur[k] - prev[k] + smallnum);
            for (int k = 0; k < 3; ++k) prev[k] = cur[k];
            ++i;
            cur += 3;
            is_small = (i < natoms && near(cur, prev, smallnum)) ? 1 : 0;
        }
        if (run != prevrun || is_smaller != 0) {
            prevrun = run;
            bw.put(1, 1);
            bw.put(5, (uint64_t)(run + is_smaller + 1));
        } else {
            bw.put(1, 0);
        }
        for (int k = 0; k < run; k += 3) put_packed(bw, smallidx, pack3(sizesmall, tmp + k));
        if (is_smaller != 0) {
            smallidx += is_smaller;
            if (is_smaller < 0) {
                smallnum = smaller;
                smaller = kMagicInts[smallidx - 1] / 2;
            } else {
                smaller = smallnum;
                smallnum = kMagicInts[smallidx] / 2;
            }
            sizesmall[0] = sizesmall[1] = sizesmall[2] = (uint32_t)kMagicInts[smallidx];
        }
    }
    bw.finish();
    put_be32(o, (uint32_t)bw.out.size());
    o.insert(o.end(), bw.out.begin(), bw.out.end());
    while (o.size() & 3) o.push_back(0);
    return true;
}

}  // namespace

struct vmd_xdrtraj_t { Xdr d; };
struct vmd_xdrwriter_t { Writer w; };

extern "C" vmd_xdrtraj_t* vmd_xdrtraj_open(const char* path) {
    if (!path) { vmd_set_last_error("vmd_xdrtraj_open: path is NULL"); return nullptr; }
    vmd_xdrtraj_t* t = new vmd_xdrtraj_t();
    Xdr& d = t->d;
    d.path = path;
    auto bail = [&](const char* fmt) -> vmd_xdrtraj_t* {
        if (fmt) fail(fmt, d.path);
        if (d.fd >= 0) close(d.fd);
        delete t;
        return nullptr;
    };
    d.fd = open(path, O_RDONLY);
    if (d.fd < 0) return bail("trajectory '%s': cannot open");
    struct stat sb;
    if (fstat(d.fd, &sb) != 0) return bail("trajectory '%s': cannot stat");
    unsigned char m[4];
    if (!read_at(d.fd, m, 4, 0)) return bail("trajectory '%s': empty file");
    const int magic = (int)be32(m);
    if (magic == kXtcMagic || magic == kXtcMagicBig) d.kind = KIND_XTC;
    else if (magic == kTrrMagic) d.kind = KIND_TRR;
    else return bail("trajectory '%s': neither an XTC nor a TRR file (bad magic number)");
    const bool ok = d.kind == KIND_XTC ? xtc_index(&d, (uint64_t)sb.st_size) : trr_index(&d, (uint64_t)sb.st_size);
    if (!ok) return bail(nullptr);
    d.iface.inst = &t->d;
    d.iface.num_frames = xdr_num_frames;
    d.iface.num_atoms = xdr_num_atoms;
    d.iface.load_frame = xdr_load_frame;
    d.iface.device_view = nullptr;
    d.iface.host_view = nullptr;
    d.iface.load_raw = xdr_load_raw;
    d.iface.raw_device_view = nullptr;
    d.iface.raw_mapped_view = xdr_raw_mapped_view;
    return t;
}

extern "C" void vmd_ckcache_drop(const void* inst);      // vmd_eval_traj.cpp: the decoder checkpoints kept for this trajectory
extern "C" void vmd_mapreg_drop(const void* base);       // vmd_eval_traj.cpp: the pinned windows of this mapping
extern "C" void vmd_xdrtraj_close(vmd_xdrtraj_t* t) {
    if (!t) return;
    vmd_ckcache_drop(&t->d);
    if (const unsigned char* m = t->d.map.load()) {
        vmd_mapreg_drop(m);
        munmap((void*)m, t->d.map_bytes);
    }
    if (t->d.fd >= 0) close(t->d.fd);
    delete t;
}

extern "C" vmd_trajectory_i* vmd_xdrtraj_interface(vmd_xdrtraj_t* t) { return t ? &t->d.iface : nullptr; }
extern "C" int vmd_xdrtraj_kind(const vmd_xdrtraj_t* t) { return t ? (int)t->d.kind : -1; }
extern "C" int64_t vmd_xdrtraj_frame_step(const vmd_xdrtraj_t* t, size_t frame) {
    return (t && frame < t->d.frames.size()) ? (int64_t)t->d.frames[frame].step : -1;
}

extern "C" vmd_xdrwriter_t* vmd_xdrwriter_open(const char* path, int kind, size_t num_atoms, float precision) {
    if (!path || num_atoms == 0 || (kind != KIND_XTC && kind != KIND_TRR) || num_atoms > 0x7fffffffu / 12) {
        vmd_set_last_error("vmd_xdrwriter_open: bad arguments");
        return nullptr;
    }
    vmd_xdrwriter_t* h = new vmd_xdrwriter_t();
    Writer& w = h->w;
    w.path = path;
    w.kind = (Kind)kind;
    w.num_atoms = num_atoms;
    w.precision = precision > 0.0f ? precision : 1000.0f;
    w.fh = fopen(path, "wb");
    if (!w.fh) { fail("trajectory '%s': cannot create", w.path); delete h; return nullptr; }
    return h;
}

extern "C" bool vmd_xdrwriter_write_frame(vmd_xdrwriter_t* h, int64_t step, float time_ps, const vmd_unitcell_t* cell,
                                          const float* x, const float* y, const float* z) {
    if (!h || !h->w.fh || !x || !y || !z) { vmd_set_last_error("vmd_xdrwriter_write_frame: bad arguments"); return false; }
    Writer& w = h->w;
    const size_t n = w.num_atoms;
    std::vector<unsigned char>& o = w.rec;
    o.clear();
    float box[9];
    box_nm(cell, box);
    if (w.kind == KIND_TRR) {
        put_be32(o, (uint32_t)kTrrMagic);
        put_be32(o, 13); put_be32(o, 12);
        const char* ver = "GMX_trn_file";
        o.insert(o.end(), ver, ver + 12);
        const uint32_t sizes[13] = {0, 0, 36, 0, 0, 0, 0, (uint32_t)(12 * n), 0, 0, (uint32_t)n, (uint32_t)step, 0};
        for (uint32_t s : sizes) put_be32(o, s);
        put_f32(o, time_ps); put_f32(o, 0.0f);
        for (int k = 0; k < 9; ++k) put_f32(o, box[k]);
        for (size_t i = 0; i < n; ++i) { put_f32(o, x[i] * 0.1f); put_f32(o, y[i] * 0.1f); put_f32(o, z[i] * 0.1f); }
    } else {
        put_be32(o, (uint32_t)kXtcMagic);
        put_be32(o, (uint32_t)n); put_be32(o, (uint32_t)step); put_f32(o, time_ps);
        for (int k = 0; k < 9; ++k) put_f32(o, box[k]);
        put_be32(o, (uint32_t)n);
        if (n <= 9) {
            for (size_t i = 0; i < n; ++i) { put_f32(o, x[i] * 0.1f); put_f32(o, y[i] * 0.1f); put_f32(o, z[i] * 0.1f); }
        } else {
            put_f32(o, w.precision);
            w.ints.resize(3 * n);
            const float* src[3] = {x, y, z};
            for (size_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) {
                    const float nm = src[k][i] * 0.1f;
                    const float lf = nm >= 0.0f ? nm * w.precision + 0.5f : nm * w.precision - 0.5f;
                    if (!(std::fabs(lf) < (float)(INT_MAX - 2))) return fail("XTC '%s': coordinate too large for the precision", w.path);
                    w.ints[3 * i + k] = (int)lf;
                }
            if (!xtc_encode(&w, o)) return false;
        }
    }
    if (fwrite(o.data(), 1, o.size(), w.fh) != o.size()) return fail("trajectory '%s': write failed", w.path);
    return true;
}

extern "C" bool vmd_xdrwriter_close(vmd_xdrwriter_t* h) {
    if (!h) return true;
    bool ok = true;
    if (h->w.fh) ok = fclose(h->w.fh) == 0;
    delete h;
    return ok;
}
