"""Test-only second implementation of the XTC coordinate compression (pure Python, byte-wise like the published xdrfile
routines: sendbits / sendints / receivebits / receiveints work on byte arrays, no wide integers).  It is written
independently of viamd_amd/csrc/vmd_xdr.cpp (which uses 64/128-bit mixed-radix arithmetic and a word-wise bit reader), so
the two can check each other in both directions: files must be byte-identical and decode to the same integers.

mdlib's own reader is absent (ext/mdlib is empty) and no XTC fixture exists in /root/reference: parity of the format is
pinned to this restatement of the published algorithm, not to a file written by GROMACS.
"""
import struct

import numpy as np

MAGICINTS = [
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64,
    80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290,
    1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003,
    16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031,
    131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561,
    832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021,
    4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216]
FIRSTIDX = 9
LASTIDX = len(MAGICINTS)


class Bits:
    """xdrfile's buf[]: byte counter, pending bit count, pending bits; bytes appended / consumed one at a time."""

    def __init__(self, data=b""):
        self.bytes = bytearray(data)
        self.cnt = 0
        self.lastbits = 0
        self.lastbyte = 0

    # --- write side
    def send(self, nbits, num):
        num &= (1 << nbits) - 1 if nbits < 64 else num
        while nbits >= 8:
            self.lastbyte = ((self.lastbyte << 8) | ((num >> (nbits - 8)) & 0xff)) & 0xffffffff
            self.bytes.append((self.lastbyte >> self.lastbits) & 0xff)
            nbits -= 8
        if nbits > 0:
            self.lastbyte = ((self.lastbyte << nbits) | (num & ((1 << nbits) - 1))) & 0xffffffff
            self.lastbits += nbits
            if self.lastbits >= 8:
                self.lastbits -= 8
                self.bytes.append((self.lastbyte >> self.lastbits) & 0xff)

    def flush(self):
        if self.lastbits:
            self.bytes.append((self.lastbyte << (8 - self.lastbits)) & 0xff)
            self.lastbits = 0
        return bytes(self.bytes)

    # --- read side
    def _next(self):
        b = self.bytes[self.cnt] if self.cnt < len(self.bytes) else 0
        self.cnt += 1
        return b

    def receive(self, nbits):
        mask = (1 << nbits) - 1
        num = 0
        while nbits >= 8:
            self.lastbyte = ((self.lastbyte << 8) | self._next()) & 0xffffffff
            num |= ((self.lastbyte >> self.lastbits) & 0xff) << (nbits - 8)
            nbits -= 8
        if nbits > 0:
            if self.lastbits < nbits:
                self.lastbits += 8
                self.lastbyte = ((self.lastbyte << 8) | self._next()) & 0xffffffff
            self.lastbits -= nbits
            num |= (self.lastbyte >> self.lastbits) & ((1 << nbits) - 1)
        return num & mask


def sizeofint(size):
    num, bits = 1, 0
    while size >= num and bits < 32:
        bits += 1
        num <<= 1
    return bits


def sizeofints(sizes):
    nbytes, b = 1, [1] + [0] * 31
    for s in sizes:
        tmp, cnt = 0, 0
        while cnt < nbytes:
            tmp = b[cnt] * s + tmp
            b[cnt] = tmp & 0xff
            tmp >>= 8
            cnt += 1
        while tmp:
            b[cnt] = tmp & 0xff
            tmp >>= 8
            cnt += 1
        nbytes = cnt
    num, bits = 1, 0
    nbytes -= 1
    while b[nbytes] >= num:
        bits += 1
        num *= 2
    return bits + nbytes * 8


def sendints(buf, nbits, sizes, nums):
    b = []
    tmp = nums[0]
    while True:
        b.append(tmp & 0xff)
        tmp >>= 8
        if not tmp:
            break
    for i in (1, 2):
        assert nums[i] < sizes[i]
        tmp = nums[i]
        for c in range(len(b)):
            tmp = b[c] * sizes[i] + tmp
            b[c] = tmp & 0xff
            tmp >>= 8
        while tmp:
            b.append(tmp & 0xff)
            tmp >>= 8
    n = len(b)
    if nbits >= n * 8:
        for v in b:
            buf.send(8, v)
        rest = nbits - n * 8
        while rest > 0:                      # zero padding, chunked: the wire bits are the same
            buf.send(min(8, rest), 0)
            rest -= 8
    else:
        for v in b[:-1]:
            buf.send(8, v)
        buf.send(nbits - (n - 1) * 8, b[-1])


def receiveints(buf, nbits, sizes):
    b = []
    while nbits > 8:
        b.append(buf.receive(8))
        nbits -= 8
    if nbits > 0:
        b.append(buf.receive(nbits))
    b += [0] * (4 - len(b)) if len(b) < 4 else []
    nums = [0, 0, 0]
    for i in (2, 1):
        num = 0
        for j in range(len(b) - 1, -1, -1):
            num = (num << 8) | b[j]
            p = num // sizes[i]
            b[j] = p
            num -= p * sizes[i]
        nums[i] = num
    nums[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)
    return nums


def to_ints(coords_nm, precision):
    """xdrfile's rounding in float arithmetic: (int)(x * precision +- 0.5f)."""
    c = np.asarray(coords_nm, np.float32)
    p = np.float32(precision)
    lf = np.where(c >= 0, c * p + np.float32(0.5), c * p - np.float32(0.5)).astype(np.float32)
    return np.trunc(lf).astype(np.int64)


def compress(ints, stats=None):
    """ints [N, 3] -> (minint, maxint, smallidx, payload bytes); N > 9.  stats (dict) collects which branches ran."""
    st = stats if stats is not None else {}
    for key in ("swaps", "up", "down", "flag0", "flag1", "three_field"):
        st.setdefault(key, 0)
    st.setdefault("runs", {})
    ip = [list(map(int, r)) for r in ints]
    n = len(ip)
    minint = [min(r[k] for r in ip) for k in range(3)]
    maxint = [max(r[k] for r in ip) for k in range(3)]
    mindiff = 2 ** 31 - 1
    for i in range(1, n):
        d = sum(abs(ip[i][k] - ip[i - 1][k]) for k in range(3))
        mindiff = min(mindiff, d)
    sizeint = [maxint[k] - minint[k] + 1 for k in range(3)]
    if (sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff:
        bitsizeint, bitsize = [sizeofint(s) for s in sizeint], 0
    else:
        bitsizeint, bitsize = None, sizeofints(sizeint)
    smallidx = FIRSTIDX
    while smallidx < LASTIDX - 1 and MAGICINTS[smallidx] < mindiff:
        smallidx += 1
    first_smallidx = smallidx
    maxidx = min(LASTIDX - 1, smallidx + 8)
    minidx = maxidx - 8
    smaller = MAGICINTS[max(FIRSTIDX, smallidx - 1)] // 2
    smallnum = MAGICINTS[smallidx] // 2
    sizesmall = [MAGICINTS[smallidx]] * 3
    larger = MAGICINTS[maxidx] // 2
    buf = Bits()
    i, prevrun, prev = 0, -1, [0, 0, 0]
    while i < n:
        cur = ip[i]
        is_small = 0
        if smallidx < maxidx and i >= 1 and all(abs(cur[k] - prev[k]) < larger for k in range(3)):
            is_smaller = 1
        elif smallidx > minidx:
            is_smaller = -1
        else:
            is_smaller = 0
        if i + 1 < n and all(abs(cur[k] - ip[i + 1][k]) < smallnum for k in range(3)):
            ip[i], ip[i + 1] = ip[i + 1], ip[i]
            cur = ip[i]
            is_small = 1
            st["swaps"] += 1
        tmp = [cur[k] - minint[k] for k in range(3)]
        if bitsize == 0:
            for k in range(3):
                buf.send(bitsizeint[k], tmp[k])
            st["three_field"] += 1
        else:
            sendints(buf, bitsize, sizeint, tmp)
        prev = list(cur)
        i += 1
        run = []
        if is_small == 0 and is_smaller == -1:
            is_smaller = 0
        while is_small and len(run) < 24:
            cur = ip[i]
            if is_smaller == -1 and sum((cur[k] - prev[k]) ** 2 for k in range(3)) >= smaller * smaller:
                is_smaller = 0
            run += [cur[k] - prev[k] + smallnum for k in range(3)]
            prev = list(cur)
            i += 1
            is_small = 1 if i < n and all(abs(ip[i][k] - prev[k]) < smallnum for k in range(3)) else 0
        if len(run) != prevrun or is_smaller != 0:
            prevrun = len(run)
            st["flag1"] += 1
            buf.send(1, 1)
            buf.send(5, len(run) + is_smaller + 1)
        else:
            st["flag0"] += 1
            buf.send(1, 0)
        st["runs"][len(run) // 3] = st["runs"].get(len(run) // 3, 0) + 1
        st["up"] += is_smaller > 0
        st["down"] += is_smaller < 0
        for k in range(0, len(run), 3):
            sendints(buf, smallidx, sizesmall, run[k:k + 3])
        if is_smaller != 0:
            smallidx += is_smaller
            if is_smaller < 0:
                smallnum = smaller
                smaller = MAGICINTS[smallidx - 1] // 2
            else:
                smaller = smallnum
                smallnum = MAGICINTS[smallidx] // 2
            sizesmall = [MAGICINTS[smallidx]] * 3
    return minint, maxint, first_smallidx, buf.flush()


def decompress(n, minint, maxint, smallidx, payload):
    """-> int64 [N, 3]"""
    sizeint = [maxint[k] - minint[k] + 1 for k in range(3)]
    if (sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff:
        bitsizeint, bitsize = [sizeofint(s) for s in sizeint], 0
    else:
        bitsizeint, bitsize = None, sizeofints(sizeint)
    smaller = MAGICINTS[max(FIRSTIDX, smallidx - 1)] // 2
    smallnum = MAGICINTS[smallidx] // 2
    sizesmall = [MAGICINTS[smallidx]] * 3
    buf = Bits(payload)
    out = []
    run = 0
    while len(out) < n:
        if bitsize == 0:
            cur = [buf.receive(bitsizeint[k]) for k in range(3)]
        else:
            cur = receiveints(buf, bitsize, sizeint)
        cur = [cur[k] + minint[k] for k in range(3)]
        prev = list(cur)
        is_smaller = 0
        if buf.receive(1) == 1:
            run = buf.receive(5)
            is_smaller = run % 3
            run -= is_smaller
            is_smaller -= 1
        if run > 0:
            for k in range(0, run, 3):
                d = receiveints(buf, smallidx, sizesmall)
                this = [d[c] + prev[c] - smallnum for c in range(3)]
                if k == 0:
                    this, prev = prev, this
                    out.append(prev)
                else:
                    prev = list(this)
                out.append(this)
        else:
            out.append(cur)
        smallidx += is_smaller
        if is_smaller < 0:
            smallnum = smaller
            smaller = MAGICINTS[smallidx - 1] // 2 if smallidx > FIRSTIDX else 0
        elif is_smaller > 0:
            smaller = smallnum
            smallnum = MAGICINTS[smallidx] // 2
        sizesmall = [MAGICINTS[smallidx]] * 3
    assert len(out) == n
    return np.array(out, np.int64)


def frame_bytes(coords_A, box_A, step, time_ps, precision=1000.0):
    """One XTC frame.  coords_A [3, N] Angstrom, box_A = 3x3 rows a, b, c in Angstrom (or None)."""
    xyz_nm = (np.asarray(coords_A, np.float32) * np.float32(0.1)).T          # [N, 3]
    n = xyz_nm.shape[0]
    box = np.zeros(9, np.float32) if box_A is None else (np.asarray(box_A, np.float32).ravel() * np.float32(0.1))
    out = struct.pack(">iiif", 1995, n, step, time_ps) + struct.pack(">9f", *box) + struct.pack(">i", n)
    if n <= 9:
        return out + xyz_nm.astype(">f4").tobytes()
    ints = to_ints(xyz_nm, precision)
    minint, maxint, smallidx, payload = compress(ints)
    out += struct.pack(">f3i3ii", precision, *minint, *maxint, smallidx)
    out += struct.pack(">i", len(payload)) + payload + b"\0" * (-len(payload) % 4)
    return out


def parse_frames(data):
    """-> list of dict(step, time, box_nm [9], ints [N,3] or None, xyz_nm [N,3] float32, precision)"""
    off, frames = 0, []
    while off < len(data):
        magic, n, step, time = struct.unpack_from(">iiif", data, off)
        assert magic == 1995
        box = np.array(struct.unpack_from(">9f", data, off + 16), np.float32)
        assert struct.unpack_from(">i", data, off + 52)[0] == n
        off += 56
        if n <= 9:
            xyz = np.frombuffer(data, ">f4", 3 * n, off).astype(np.float32).reshape(n, 3)
            off += 12 * n
            frames.append(dict(step=step, time=time, box_nm=box, ints=None, xyz_nm=xyz, precision=None))
            continue
        precision, = struct.unpack_from(">f", data, off)
        mm = struct.unpack_from(">7i", data, off + 4)
        nbytes, = struct.unpack_from(">i", data, off + 32)
        payload = data[off + 36: off + 36 + nbytes]
        off += 36 + ((nbytes + 3) & ~3)
        ints = decompress(n, list(mm[0:3]), list(mm[3:6]), mm[6], payload)
        invp = np.float32(1.0) / np.float32(precision)
        xyz = ints.astype(np.float32) * invp
        frames.append(dict(step=step, time=time, box_nm=box, ints=ints, xyz_nm=xyz, precision=precision))
    return frames
