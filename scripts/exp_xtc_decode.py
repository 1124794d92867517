"""Device XTC decoders timed in isolation (round 3): the compressed frames of a batch already sit in HBM, each
variant (1 = thread per frame, 2 = index + chunks, 3 = wave per frame) decodes them; hipEvent time per batch, frames/s, and the
floats compared with the host reader.  Two data sets of 100 002 atoms: the synthetic c2 box (loose O,H,H triplets: ~96k groups per
frame, 9 % of the flags set) and a water box with real molecular geometry (33k groups of three atoms).
usage (GPU box): python scripts/exp_xtc_decode.py [out.txt]"""
import ctypes as C
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import viamd_amd as V
from viamd_amd import _lib as L
from viamd_amd import synth

SHARES = (1, 4, 0)
QUICK = "--quick" in sys.argv
lib = V.default_lib()
lib.vmd_set_device(0)
out = open(sys.argv[1], "w") if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else sys.stdout


def say(*a):
    print(*a, file=out, flush=True)
    if out is not sys.stdout:
        print(*a, flush=True)


def real_water(n_mol, box, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(0, box, (n_mol, 3))
    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    a = unit(rng.normal(size=(n_mol, 3)))
    b = unit(np.cross(a, rng.normal(size=(n_mol, 3))))
    half = np.deg2rad(104.52 / 2)
    h1 = o + 0.9572 * (np.cos(half) * a + np.sin(half) * b)
    h2 = o + 0.9572 * (np.cos(half) * a - np.sin(half) * b)
    xyz = np.empty((3 * n_mol, 3))
    xyz[0::3], xyz[1::3], xyz[2::3] = o, h1, h2
    return np.ascontiguousarray(xyz.T, np.float32)


def parse(blob, natoms):
    off, infos = 0, []
    while off < len(blob):
        assert struct.unpack_from(">i", blob, off + 4)[0] == natoms
        precision, = struct.unpack_from(">f", blob, off + 56)
        mm = struct.unpack_from(">7i", blob, off + 60)
        nbytes, = struct.unpack_from(">i", blob, off + 88)
        infos.append((precision, mm[0:3], mm[3:6], mm[6], off + 92, nbytes))
        off += 92 + ((nbytes + 3) & ~3)
    return infos


def run(name, path, natoms, B):
    blob = open(path, "rb").read()
    infos = parse(blob, natoms)
    F = len(infos)
    arr = (L.XtcFrame * B)()
    raw = bytearray()
    for b in range(B):
        precision, mi, ma, sidx, o, nbytes = infos[b % F]
        arr[b].precision = precision
        arr[b].minint[:] = mi
        arr[b].maxint[:] = ma
        arr[b].smallidx = sidx
        arr[b].offset = len(raw)
        arr[b].nbytes = nbytes
        raw += blob[o:o + nbytes] + b"\0" * ((-nbytes) % 64 + 64)
    d_raw = torch.frombuffer(raw, dtype=torch.uint8).cuda()
    d_info = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    npad = (natoms + 63) & ~63
    d_xyz = torch.zeros((B, 3, npad), dtype=torch.float32, device="cuda")
    d_status = torch.zeros(B, dtype=torch.int32, device="cuda")
    t = V.XdrTrajectory(path)
    ref = [t.load_frame(f)[0] for f in range(min(F, 3))]
    say(f"== {name}: {natoms} atoms, {F} distinct frames, batch of {B}, {len(raw) / B / 1e6:.3f} MB compressed per frame")
    chunk = 256
    d_scratch = torch.zeros(lib.vmd_hip_xtc_scratch_bytes(B, natoms, chunk) // 8 + 1, dtype=torch.int64, device="cuda")
    variants = {
        "3 wave per frame": lambda: lib.vmd_hip_xtc_decode_wave(None, d_raw.data_ptr(), d_info.data_ptr(), B, natoms, d_xyz.data_ptr(), 3 * npad, npad, d_status.data_ptr()),
        "2 index + chunks": lambda: lib.vmd_hip_xtc_decode_chunked(None, d_raw.data_ptr(), d_info.data_ptr(), B, natoms, d_xyz.data_ptr(), 3 * npad, npad, d_status.data_ptr(), chunk, d_scratch.data_ptr()),
        "1 thread per frame": lambda: lib.vmd_hip_xtc_decode(None, d_raw.data_ptr(), d_info.data_ptr(), B, natoms, d_xyz.data_ptr(), 3 * npad, npad, d_status.data_ptr()),
    }
    for share in SHARES:
        old = lib.vmd_set_option(b"xtc_waves", share)
        ms = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert variants["3 wave per frame"]() == 0
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        lib.vmd_set_option(b"xtc_waves", old)
        got = d_xyz[:len(ref), :, :natoms].cpu().numpy()
        same = all(np.array_equal(got[f], ref[f]) for f in range(len(ref)))
        best = min(ms[1:])
        say(f"  variant 3, {share:2d} waves per frame (0 = auto) {best:9.3f} ms per batch = {B / best * 1e3:10.0f} frames/s   status ok: {bool((d_status.cpu().numpy() == 0).all())}  floats == host reader: {same}")
    # checkpoints: first pass (walk + emit), then sectioned passes
    d_ck = torch.zeros(B * 16 * 4, dtype=torch.int32, device="cuda")
    d_nck = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d_xyz.zero_()
        torch.cuda.synchronize()
        e0.record()
        assert lib.vmd_hip_xtc_decode_wave_ck(None, d_raw.data_ptr(), d_info.data_ptr(), B, natoms, d_xyz.data_ptr(), 3 * npad, npad, d_status.data_ptr(),
                                              0 if r == 0 else 1, d_ck.data_ptr(), d_nck.data_ptr()) == 0
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    got = d_xyz[:len(ref), :, :natoms].cpu().numpy()
    same = all(np.array_equal(got[f], ref[f]) for f in range(len(ref)))
    say(f"  variant 3 with checkpoints: first pass (emits) {ms[0]:8.3f} ms, sectioned passes {min(ms[1:]):8.3f} ms per batch = {B / min(ms[1:]) * 1e3:10.0f} frames/s"
        f"   sections per frame {d_nck.float().mean().item():.1f}  status ok: {bool((d_status.cpu().numpy() == 0).all())}  floats == host reader: {same}")
    for vname, fn in variants.items():
        reps = 3 if vname.startswith("3") else 1
        if not vname.startswith("3") and (B > 1024 or QUICK):
            continue
        d_xyz.zero_()
        torch.cuda.synchronize()
        ms = []
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn() == 0
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        st = d_status.cpu().numpy()
        got = d_xyz[:len(ref), :, :natoms].cpu().numpy()
        same = all(np.array_equal(got[f], ref[f]) for f in range(len(ref)))
        best = min(ms[1:])
        say(f"  variant {vname:20s} {best:9.3f} ms per batch = {B / best * 1e3:10.0f} frames/s   (runs: {' '.join(f'{m:.2f}' for m in ms)})  status ok: {bool((st == 0).all())}  floats == host reader: {same}")


N = 100002
tmp = tempfile.gettempdir()
cell = V.make_unitcell(100.0)
# synthetic c2
F = 64
dev = synth.make_device_trajectory(V, 2, N, 100.0, F)
host = V.PinnedHostTrajectory(F, N)
host.copy_from_device(dev)
dev.close()
p1 = os.path.join(tmp, "exp_c2.xtc")
t0 = time.time()
V.write_xtc(p1, host, cell)
say(f"wrote {p1}: {os.path.getsize(p1) / F / 1e6:.3f} MB/frame in {time.time() - t0:.1f} s")
p2 = os.path.join(tmp, "exp_water.xtc")
frames = np.stack([real_water(N // 3, 100.0, 100 + f) for f in range(16)])
V.write_xtc(p2, frames, cell)
say(f"wrote {p2}: {os.path.getsize(p2) / 16 / 1e6:.3f} MB/frame")
for B in (128, 512, 1024, 4096):
    run("synthetic c2", p1, N, B)
    run("real water geometry", p2, N, B)
