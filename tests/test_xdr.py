"""GROMACS XTC / TRR readers (viamd_amd/csrc/vmd_xdr.cpp; VIAMD: md_xtc_attach_from_file / md_trr_attach_from_file,
/root/reference/src/loader.cpp:147-150).  No GROMACS-written fixture exists in /root/reference, so the format is pinned to
tests/xtc_ref.py, an independent byte-wise restatement of the published compression: files of the two writers must be
byte-identical and both decoders must return the same integers."""
import struct

import numpy as np
import pytest

import viamd_amd as V
import xtc_ref


def _systems():
    rng = np.random.default_rng(11)
    out = {}
    # water (rigid TIP3P geometry, random orientations) -> runs of 2, the water swap, smallidx adaptation
    n_w = 7 * 8 * 8                                                          # liquid-like: no two molecules overlap
    o = np.stack(np.meshgrid(np.arange(7), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 3.1
    o = o + rng.uniform(-0.3, 0.3, o.shape) + 1.0
    u = rng.normal(0, 1, (n_w, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)          # bisector
    v = np.cross(u, rng.normal(0, 1, (n_w, 3))); v /= np.linalg.norm(v, axis=1, keepdims=True)  # in-plane normal to it
    half = np.radians(104.52 / 2)
    h = o[:, None, :] + 0.9572 * np.stack([np.cos(half) * u + np.sin(half) * v, np.cos(half) * u - np.sin(half) * v], axis=1)
    w = np.concatenate([o[:, None, :], h], axis=1).reshape(-1, 3)
    out["water"] = w.T.astype(np.float32)
    # ideal gas with negative coordinates: hardly any run
    out["gas"] = rng.uniform(-40.0, 55.0, (3, 257)).astype(np.float32)
    # a chain: long runs, smallidx walks down and up
    steps = rng.normal(0, 0.3, (900, 3))
    steps[300:600] *= 8.0
    out["chain"] = np.cumsum(steps, axis=0).T.astype(np.float32)
    # a range above 2^24 grid steps on x: the three-field branch (bitsize == 0)
    wide = rng.uniform(0, 50.0, (3, 64)).astype(np.float32)
    wide[0] *= 5000.0
    out["wide"] = wide
    # every axis just below 2^24 grid steps: one packed number of 68 bits per atom (the 128-bit path of the native reader),
    # with a few close pairs so that small triples follow
    huge = rng.uniform(0, 60000.0, (3, 48)).astype(np.float32)
    huge[:, 1::4] = huge[:, 0::4] + rng.normal(0, 0.5, (3, 12)).astype(np.float32)
    out["huge"] = huge
    # 2 500 A along every axis: 250 000 grid steps each, one packed number of 54 bits per atom - wider than the 52 bits fp64 holds
    # exactly, divided through the reciprocal all the same because the divisors (18 and 36 bits) keep the quotients below 2^50
    vast = rng.uniform(0, 2500.0, (3, 96)).astype(np.float32)
    vast[:, 1::3] = vast[:, 0::3] + rng.normal(0, 0.4, (3, 32)).astype(np.float32)
    out["vast"] = vast
    # exactly on a lattice: many identical differences, zero differences
    g = np.stack(np.meshgrid(np.arange(6), np.arange(5), np.arange(4), indexing="ij"), -1).reshape(-1, 3) * 1.5
    out["lattice"] = g.T.astype(np.float32)
    # tight clusters of 2..9 atoms far from each other: every run length 1..8
    cl = []
    for g_i in range(64):
        centre = rng.uniform(0, 60.0, 3)
        cl.append(centre + rng.normal(0, 0.05, (2 + g_i % 8, 3)))
    out["clusters"] = np.concatenate(cl).T.astype(np.float32)
    out["tiny"] = rng.uniform(0, 10.0, (3, 7)).astype(np.float32)          # <= 9 atoms: stored as plain floats
    out["ten"] = rng.uniform(0, 10.0, (3, 10)).astype(np.float32)          # smallest compressed frame
    return out


def _expected_xtc(coords_A, precision):
    """What the reader must return: fl(fl(int * fl(1/precision)) * 10) of the integers the format stores."""
    nm = (np.asarray(coords_A, np.float32) * np.float32(0.1)).T
    ints = xtc_ref.to_ints(nm, precision)
    invp = np.float32(1.0) / np.float32(precision)
    return ((ints.astype(np.float32) * invp) * np.float32(10.0)).T, ints


@pytest.mark.parametrize("precision", [1000.0, 100.0, 12345.0])
def test_xtc_reader_and_writer_against_the_bytewise_restatement(tmp_path, host_lib, precision):
    emu_lib = host_lib
    systems = _systems()
    box = np.array([[31.0, 0, 0], [4.0, 29.5, 0], [-3.0, 5.5, 33.25]], np.float32)
    cell = V.make_unitcell((31.0, 29.5, 33.25), tilt=(4.0, -3.0, 5.5))
    for name, xyz in systems.items():
        if name in ("wide", "huge") and precision > 1000.0:
            continue                                                       # beyond the integer range of the format
        F = 3
        frames = [xyz + np.float32(0.37 * f) for f in range(F)]
        blob = b"".join(xtc_ref.frame_bytes(frames[f], box, 10 * f, 0.5 * f, precision) for f in range(F))
        p_ref = tmp_path / f"{name}_ref.xtc"
        p_ref.write_bytes(blob)
        # 1. the native writer produces the same bytes as the byte-wise restatement
        p_nat = tmp_path / f"{name}_nat.xtc"
        w = emu_lib.vmd_xdrwriter_open(str(p_nat).encode(), 0, xyz.shape[1], precision)
        assert w
        import ctypes as C
        from viamd_amd import _lib as L
        for f in range(F):
            a = np.ascontiguousarray(frames[f], np.float32)
            assert emu_lib.vmd_xdrwriter_write_frame(w, 10 * f, 0.5 * f, C.byref(cell), a[0].ctypes.data_as(L.c_float_p),
                                                     a[1].ctypes.data_as(L.c_float_p), a[2].ctypes.data_as(L.c_float_p))
        assert emu_lib.vmd_xdrwriter_close(w)
        assert p_nat.read_bytes() == blob, name
        # 2. the native reader returns the stored integers (through the float formula), the box, step and time
        t = V.XdrTrajectory(p_ref, lib=emu_lib)
        assert (t.kind, t.num_frames(), t.num_atoms()) == ("xtc", F, xyz.shape[1])
        for f in (F - 1, 0, 1):
            got, c, hdr = t.load_frame(f, with_header=True)
            if xyz.shape[1] <= 9:
                want = ((frames[f] * np.float32(0.1)) * np.float32(10.0)).astype(np.float32)
            else:
                want, ints = _expected_xtc(frames[f], precision)
            np.testing.assert_array_equal(got, want, err_msg=name)
            assert abs(got - frames[f]).max() <= 10.0 * 0.5 / precision * 1.001 + 4e-7 * abs(frames[f]).max()   # half a grid step + float rounding
            np.testing.assert_allclose([c.x, c.y, c.z, c.xy, c.xz, c.yz], [31.0, 29.5, 33.25, 4.0, -3.0, 5.5], rtol=1e-6)
            assert c.flags == 7 and hdr.timestamp == 0.5 * f and t.frame_step(f) == 10 * f
        t.close()
        # 3. the byte-wise decoder reads the natively written file back to the same integers
        if xyz.shape[1] > 9:
            for f, fr in enumerate(xtc_ref.parse_frames(p_nat.read_bytes())):
                np.testing.assert_array_equal(fr["ints"], _expected_xtc(frames[f], precision)[1])


def test_xtc_fixtures_reach_every_branch_of_the_format():
    """Guards the fixtures: the water swap, runs of every length up to 8, unchanged and changed run flags, smallidx moving
    in both directions and the three-field branch all occur, and water compresses the way the format is meant to."""
    s = _systems()
    tot = {}
    for name in ("water", "chain", "wide", "gas", "lattice", "clusters"):
        st = {}
        ints = xtc_ref.to_ints((s[name] * np.float32(0.1)).T, 1000.0)
        mi, ma, sidx, payload = xtc_ref.compress(ints, st)
        if name == "water":
            size = [ma[k] - mi[k] + 1 for k in range(3)]
            assert len(payload) * 8 < 0.9 * len(ints) * xtc_ref.sizeofints(size)
            assert st["swaps"] > 300 and st["runs"].get(2, 0) > 300
        for k, v in st.items():
            if k == "runs":
                for r, c in v.items():
                    tot.setdefault("runs", {})[r] = tot.get("runs", {}).get(r, 0) + c
            else:
                tot[k] = tot.get(k, 0) + v
    assert all(tot[k] > 0 for k in ("swaps", "up", "down", "flag0", "flag1", "three_field")), tot
    assert set(tot["runs"]) >= set(range(0, 9)), tot["runs"]


def _trr_frame(xyz_A, box_A, step, time, double=False, with_x=True, natoms=None):
    n = xyz_A.shape[1] if natoms is None else natoms
    rs = 8 if double else 4
    e = ">d" if double else ">f"
    xyz_nm = (xyz_A.astype(np.float64) * 0.1 if double else (xyz_A * np.float32(0.1))).T
    sizes = [0, 0, 9 * rs, 9 * rs, 0, 0, 0, 3 * n * rs if with_x else 0, 3 * n * rs, 0, n, step, 0]
    out = struct.pack(">i", 1993) + struct.pack(">ii", 13, 12) + b"GMX_trn_file" + struct.pack(">13i", *sizes)
    out += struct.pack(e, time) + struct.pack(e, 0.0)
    box = np.asarray(box_A, np.float64).ravel() * 0.1
    out += box.astype(e).tobytes()
    out += np.zeros(9).astype(e).tobytes()                                   # virial block
    if with_x:
        out += xyz_nm.astype(e).tobytes()
    out += np.full(3 * n, 0.25).astype(e).tobytes()                          # velocities
    return out


def test_trr_reader_single_and_double_precision(tmp_path, host_lib):
    emu_lib = host_lib
    rng = np.random.default_rng(3)
    F, N = 4, 53
    coords = rng.normal(0, 25, (F, 3, N)).astype(np.float32)
    box = [[40.0, 0, 0], [0, 42.5, 0], [0, 0, 38.0]]
    for double in (False, True):
        blob = b""
        for f in range(F):
            blob += _trr_frame(coords[f], box, 100 * f, 2.0 * f, double)
            blob += _trr_frame(coords[f], box, 100 * f + 50, 2.0 * f + 1.0, double, with_x=False)   # velocity-only frame: skipped
        p = tmp_path / f"d{int(double)}.trr"
        p.write_bytes(blob + b"\0\0\0")                                     # trailing garbage is ignored
        t = V.XdrTrajectory(p, lib=emu_lib)
        assert (t.kind, t.num_frames(), t.num_atoms()) == ("trr", F, N)
        for f in (2, 0, F - 1):
            got, c, hdr = t.load_frame(f, with_header=True)
            if double:
                want = ((coords[f].astype(np.float64) * 0.1) * 10.0).astype(np.float32)
            else:
                want = (coords[f] * np.float32(0.1)) * np.float32(10.0)
            np.testing.assert_array_equal(got, want)
            assert (c.x, c.y, c.z, c.xy, c.xz, c.yz, c.flags) == (40.0, 42.5, 38.0, 0.0, 0.0, 0.0, 7)
            assert hdr.timestamp == 2.0 * f and t.frame_step(f) == 100 * f
    # the native TRR writer is read back bit for bit (x * 0.1f * 10.0f)
    q = tmp_path / "w.trr"
    cell = V.make_unitcell((40.0, 42.5, 38.0), tilt=(3.0, -2.0, 1.5))
    V.write_trr(q, coords, cell, dt=2.0, lib=emu_lib)
    t = V.XdrTrajectory(q, lib=emu_lib)
    assert t.num_frames() == F
    got, c = t.load_frame(1)
    np.testing.assert_array_equal(got, (coords[1] * np.float32(0.1)) * np.float32(10.0))
    np.testing.assert_allclose([c.x, c.y, c.z, c.xy, c.xz, c.yz], [40.0, 42.5, 38.0, 3.0, -2.0, 1.5], rtol=1e-6)


def test_xdr_error_paths_and_truncation(tmp_path, host_lib):
    emu_lib = host_lib
    xyz = _systems()["water"]
    p = tmp_path / "w.xtc"
    V.write_xtc(p, np.stack([xyz, xyz + 1, xyz + 2]), V.make_unitcell(31.0), lib=emu_lib)
    data = p.read_bytes()
    bad = tmp_path / "bad.xtc"
    bad.write_bytes(b"\0\0\0\1 this is not a trajectory ..........................................")
    with pytest.raises(V.VmdError, match="neither an XTC nor a TRR"):
        V.XdrTrajectory(bad, lib=emu_lib)
    with pytest.raises(V.VmdError, match="cannot open"):
        V.XdrTrajectory(tmp_path / "missing.xtc", lib=emu_lib)
    cut = tmp_path / "cut.xtc"
    cut.write_bytes(data[:len(data) - 40])                                  # last frame incomplete -> dropped from the index
    t = V.XdrTrajectory(cut, lib=emu_lib)
    assert t.num_frames() == 2
    with pytest.raises(V.VmdError, match="out of range"):
        t.load_frame(2)
    # corrupt payloads must be rejected or decoded to garbage, never crash: flip bytes all over the first frame
    rng = np.random.default_rng(0)
    first = len(xtc_ref.frame_bytes(xyz, np.diag([31.0] * 3), 0, 0.0))
    assert data[:first] == xtc_ref.frame_bytes(xyz, np.diag([31.0] * 3), 0, 0.0)
    rejected = 0
    for trial in range(300):
        b = bytearray(data)
        for _ in range(rng.integers(1, 6)):
            b[rng.integers(56, first)] = rng.integers(0, 256)
        f = tmp_path / "fuzz.xtc"
        f.write_bytes(bytes(b))
        try:
            tt = V.XdrTrajectory(f, lib=emu_lib)
            if tt.num_frames():
                tt.load_frame(0)
            tt.close()
        except V.VmdError:
            rejected += 1
    assert rejected > 0


def test_xtc_trajectory_through_the_evaluator_on_emulator(tmp_path, emu_lib, oracle):
    """An XTC file staged through load_frame (frames decompressed on several host threads) gives the same histogram as the
    decoded frames handed over from memory, and the oracle agrees on those coordinates."""
    import cases
    box, F, N = 36.0, 10, 1200
    coords = cases.water_box(oracle, 31, N, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "w.xtc"
    V.write_xtc(p, coords, cell, lib=emu_lib)
    t = V.XdrTrajectory(p, lib=emu_lib)
    decoded = np.stack([t.load_frame(f)[0] for f in range(F)])
    assert abs(decoded - coords).max() < 0.0051
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    res = []
    for threads in (1, 8):
        old_t = emu_lib.vmd_set_option(b"load_threads", threads)
        old_b = emu_lib.vmd_set_option(b"batch_frames", 4)
        try:
            for traj in (V.XdrTrajectory(p, lib=emu_lib), V.HostTrajectory(decoded, cell)):
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, traj, 0, F)
                res.append(ev.property_data("g").counts.copy())
        finally:
            emu_lib.vmd_set_option(b"load_threads", old_t)
            emu_lib.vmd_set_option(b"batch_frames", old_b)
    for r in res[1:]:
        np.testing.assert_array_equal(r, res[0])
    cases.check_rdf(emu_lib, oracle, decoded, box, [("g", o, o, 0.0, 9.0)])
    assert res[0].sum() > 0


def _checkpoint_sidecar_case(tmp_path, lib, oracle, n_water=1200, box=36.0, F=10):
    """vmd_ckcache_save / vmd_ckcache_load: the decoder checkpoints a first pass leaves survive the process (here: the trajectory
    object) in a sidecar file.  A trajectory opened later decodes in sections right away; a table that belongs to other bytes
    (signature) costs a first pass and nothing else; a table that lies about the streams it claims (signatures intact,
    checkpoints damaged) is rejected by the sectioned decode, the batch falls back to the host reader, the frames walk again."""
    import cases
    coords = cases.water_box(oracle, 37, n_water, box, F)
    cell = V.make_unitcell(box)
    p, p2 = tmp_path / "a.xtc", tmp_path / "b.xtc"
    V.write_xtc(p, coords, cell, lib=lib)
    V.write_xtc(p2, coords + np.float32(0.5), cell, lib=lib)                  # same frame and atom count, other bytes
    o = cases.oxygen(n_water)
    ir = V.ScriptIR(lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(n_water, unitcell=cell)
    old = (lib.vmd_set_option(b"xtc_device_decode", 3), lib.vmd_set_option(b"batch_frames", 4))
    try:
        a = V.XdrTrajectory(p, lib=lib)
        with pytest.raises(V.VmdError):
            a.save_checkpoints(tmp_path / "none.ck")                            # nothing decoded yet
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, a, 0, F) and ev.frames_device_decoded() == F and ev.frames_section_decoded() == 0
        want = ev.property_data("g").counts.copy()
        assert want.sum() > 0
        ck = tmp_path / "a.xtc.vmdck"
        a.save_checkpoints(ck)
        assert 0 < ck.stat().st_size < 2 * (F * 64 * 16 + F * 13 + 64)
        a.close()
        # a later "session": a new trajectory object of the same file
        b = V.XdrTrajectory(p, lib=lib)
        assert b.load_checkpoints(ck) == F
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, b, 0, F) and ev.frames_device_decoded() == F and ev.frames_section_decoded() == F
        np.testing.assert_array_equal(ev.property_data("g").counts, want)
        # the table of another file: every signature differs -> an ordinary first pass, then sections from ITS OWN checkpoints
        c = V.XdrTrajectory(p2, lib=lib)
        assert c.load_checkpoints(ck) == F
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, c, 0, F) and ev.frames_device_decoded() == F and ev.frames_section_decoded() == 0
        want2 = ev.property_data("g").counts.copy()
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, c, 0, F) and ev.frames_section_decoded() == F
        np.testing.assert_array_equal(ev.property_data("g").counts, want2)
        # a damaged table behind intact signatures: rejected, host fallback for those batches, right answer; the next pass walks again
        blob = bytearray(ck.read_bytes())
        hdr = 8 + 4 + 4 + 8 + 8
        ck_off = hdr + F + 8 * F + 4 * F
        for f in range(F):
            for k in range(1, 3):                                             # checkpoints 1 and 2 of every frame: bit position + 1000
                off = ck_off + (f * 64 + k) * 16
                pos = int.from_bytes(blob[off:off + 4], "little")
                blob[off:off + 4] = ((pos + 1000) & 0xffffffff).to_bytes(4, "little")
        bad = tmp_path / "bad.vmdck"
        bad.write_bytes(bytes(blob))
        d = V.XdrTrajectory(p, lib=lib)
        assert d.load_checkpoints(bad) == F
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, d, 0, F)
        np.testing.assert_array_equal(ev.property_data("g").counts, want)
        assert ev.frames_section_decoded() < F                                 # frames with a single section have no second checkpoint to trip over
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, d, 0, F)
        np.testing.assert_array_equal(ev.property_data("g").counts, want)
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, d, 0, F) and ev.frames_section_decoded() == F   # ... and by now every frame has good ones again
        np.testing.assert_array_equal(ev.property_data("g").counts, want)
        # not this trajectory / not a checkpoint file
        small = tmp_path / "s.xtc"
        V.write_xtc(small, coords[:, :, :300], cell, lib=lib)
        assert V.XdrTrajectory(small, lib=lib).load_checkpoints(ck) == 0
        with pytest.raises(V.VmdError):
            d.load_checkpoints(p)
        with pytest.raises(V.VmdError):
            d.load_checkpoints(tmp_path / "missing.vmdck")
        trunc = tmp_path / "t.vmdck"
        trunc.write_bytes(ck.read_bytes()[:200])
        with pytest.raises(V.VmdError):
            d.load_checkpoints(trunc)
    finally:
        lib.vmd_set_option(b"xtc_device_decode", old[0]); lib.vmd_set_option(b"batch_frames", old[1])


def test_decoder_checkpoints_survive_in_a_sidecar_file(tmp_path, emu_lib, oracle):
    _checkpoint_sidecar_case(tmp_path, emu_lib, oracle)


@pytest.mark.gpu
def test_decoder_checkpoints_survive_in_a_sidecar_file_on_the_gpu(tmp_path, gpu_lib, oracle):
    _checkpoint_sidecar_case(tmp_path, gpu_lib, oracle, n_water=9000, box=66.0, F=12)


def _device_decode(lib, blob, natoms, chunk=0, gpu=False):
    """Run vmd_hip_xtc_decode (emulator build: "device" memory is host memory) on every frame of an XTC byte string."""
    import ctypes as C
    from viamd_amd import _lib as L
    # chunk -3 / -4: k_xtc_wave (-1) / with checkpoints (-2) on the streams WHERE THE FILE HAS THEM (what the evaluator DMAs out of the
    # mapped file): 4-byte aligned starts of both phases modulo 8, the next frame's header as the readable bytes behind a stream
    # chunk -5 / -6: like -2 / -4 with GROUP RECORDS next to the checkpoints - the second pass places every group from them (k_xtc_records)
    with_rec = chunk in (-5, -6)
    if with_rec:
        chunk += 3
    file_layout = chunk in (-3, -4)
    if file_layout:
        chunk += 2
    off, infos, streams, starts = 0, [], [], []
    while off < len(blob):
        n = struct.unpack_from(">i", blob, off + 4)[0]
        precision, = struct.unpack_from(">f", blob, off + 56)
        mm = struct.unpack_from(">7i", blob, off + 60)
        nbytes, = struct.unpack_from(">i", blob, off + 88)
        streams.append(blob[off + 92: off + 92 + nbytes])
        starts.append(off + 92)
        infos.append((precision, mm[0:3], mm[3:6], mm[6], nbytes))
        off += 92 + ((nbytes + 3) & ~3)
        assert n == natoms
    B = len(infos)
    raw = bytearray()
    arr = (L.XtcFrame * B)()
    for b, (precision, mi, ma, sidx, nbytes) in enumerate(infos):
        arr[b].precision = precision
        arr[b].minint[:] = mi
        arr[b].maxint[:] = ma
        arr[b].smallidx = sidx
        arr[b].offset = 8 + starts[b] if file_layout else len(raw)
        arr[b].nbytes = nbytes
        if not file_layout:
            raw += streams[b] + b"\0" * ((-len(streams[b])) % 64 + 64)
    if file_layout:
        raw = bytearray(b"\xa5" * 8 + blob + b"\xa5" * 64)
        assert len({int(a.offset) % 8 for a in arr}) == 2 or B < 4, "fixture without both phases"
    npad = (natoms + 63) & ~63
    if gpu:                    # the product library on a real GPU: torch owns the device memory (hipMalloc: 256-byte aligned)
        import torch
        d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        d_info = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
        d_out = torch.full((B, 3, npad), float("nan"), dtype=torch.float32, device="cuda")
        d_status = torch.full((B,), 99, dtype=torch.int32, device="cuda")
        raw_p, info_p, out_p, status_p = d_raw.data_ptr(), d_info.data_ptr(), d_out.data_ptr(), d_status.data_ptr()
        if chunk > 0:
            d_scratch = torch.zeros(lib.vmd_hip_xtc_scratch_bytes(B, natoms, chunk) // 8 + 1, dtype=torch.int64, device="cuda")
            scratch_p = d_scratch.data_ptr()
        torch.cuda.synchronize()
    else:
        # 64-byte alignment of the stream starts: copy into an aligned numpy buffer
        store = np.zeros(len(raw) + 64, np.uint8)
        base = (-store.ctypes.data) % 64
        store[base:base + len(raw)] = np.frombuffer(raw, np.uint8)
        out = np.full((B, 3, npad), np.nan, np.float32)
        status = np.full(B, 99, np.uint32)
        raw_p, info_p, out_p, status_p = store.ctypes.data + base, C.addressof(arr), out.ctypes.data, status.ctypes.data
        if chunk > 0:
            scratch = np.zeros(lib.vmd_hip_xtc_scratch_bytes(B, natoms, chunk) // 8 + 1, np.uint64)
            scratch_p = scratch.ctypes.data
    if chunk == -2:            # k_xtc_wave twice: the first pass leaves checkpoints, the second decodes the frames in sections from them
        CK = 64                # VMD_XTC_CK_MAX (include/vmd_hip.h)
        if gpu:
            d_ck = torch.zeros(B * CK * 4, dtype=torch.int32, device="cuda")
            d_nck = torch.zeros(B, dtype=torch.int32, device="cuda")
            ck_p, nck_p = d_ck.data_ptr(), d_nck.data_ptr()
        else:
            ck = np.zeros(B * CK * 4, np.uint32)
            nck = np.zeros(B, np.uint32)
            ck_p, nck_p = ck.ctypes.data, nck.ctypes.data
        stride = npad
        if with_rec:
            if gpu:
                d_rec = torch.zeros(B * stride, dtype=torch.int16, device="cuda")
                d_nrec = torch.zeros(B, dtype=torch.int32, device="cuda")
                rec_p, nrec_p = d_rec.data_ptr(), d_nrec.data_ptr()
            else:
                rec = np.zeros(B * stride, np.uint16)
                nrec = np.zeros(B, np.uint32)
                rec_p, nrec_p = rec.ctypes.data, nrec.ctypes.data
            two_pass = lambda use: lib.vmd_hip_xtc_decode_wave_rec(None, raw_p, info_p, B, natoms, out_p, 3 * npad, npad, status_p, use, ck_p, nck_p, rec_p, nrec_p, stride)
        else:
            two_pass = lambda use: lib.vmd_hip_xtc_decode_wave_ck(None, raw_p, info_p, B, natoms, out_p, 3 * npad, npad, status_p, use, ck_p, nck_p)
        rc = two_pass(0)
        assert rc == 0
        if gpu:
            torch.cuda.synchronize()
            first, st1 = d_out.cpu().numpy().copy(), d_status.cpu().numpy().copy()
            d_out.fill_(float("nan"))
            counts = d_nck.cpu().numpy()
        else:
            first, st1 = out.copy(), status.copy()
            out[:] = np.nan
            counts = nck
        if (st1 == 0).all():
            assert (counts >= 1).all() and (counts <= CK).all(), counts
            if with_rec:
                ng = d_nrec.cpu().numpy() if gpu else nrec
                assert (ng >= 1).all() and (ng <= natoms).all(), ng                  # every frame got its records: one per group
            rc = two_pass(1)
            if gpu:
                torch.cuda.synchronize()
                second = d_out.cpu().numpy()
            else:
                second = out
            np.testing.assert_array_equal(second[:, :, :natoms], first[:, :, :natoms])      # sections == one walk, bit for bit
    elif chunk == -1:          # one wave per frame (k_xtc_wave)
        rc = lib.vmd_hip_xtc_decode_wave(None, raw_p, info_p, B, natoms, out_p, 3 * npad, npad, status_p)
    elif chunk:                # two passes: index (one thread per frame) + chunks (one thread per chunk)
        rc = lib.vmd_hip_xtc_decode_chunked(None, raw_p, info_p, B, natoms, out_p, 3 * npad, npad, status_p, chunk, scratch_p)
    else:
        rc = lib.vmd_hip_xtc_decode(None, raw_p, info_p, B, natoms, out_p, 3 * npad, npad, status_p)
    if gpu:
        torch.cuda.synchronize()
        out, status = d_out.cpu().numpy(), d_status.cpu().numpy().astype(np.uint32)
    assert rc == 0, f"launch of the device decoder (chunk {chunk}) returned {rc}"
    return out[:, :, :natoms], status


def _device_decoder_against_host_reader(tmp_path, lib, chunk, gpu, trials=25):
    systems = _systems()
    for name, xyz in systems.items():
        if xyz.shape[1] <= 9:
            continue
        F = 3
        frames = [xyz + np.float32(0.37 * f) for f in range(F)]
        blob = b"".join(xtc_ref.frame_bytes(frames[f], np.diag([30.0, 30.0, 30.0]), f, 0.0, 1000.0) for f in range(F))
        got, status = _device_decode(lib, blob, xyz.shape[1], chunk, gpu)
        if name == "huge":
            assert (status == 2).all()
            continue
        assert (status == 0).all(), (name, status)
        p = tmp_path / f"{name}.xtc"
        p.write_bytes(blob)
        t = V.XdrTrajectory(p, lib=lib)
        for f in range(F):
            np.testing.assert_array_equal(got[f], t.load_frame(f)[0], err_msg=name)
    # damaged streams: never out of bounds (the emulator build runs under ASan in scripts/sanitize_emu.sh), some are rejected
    rng = np.random.default_rng(1)
    xyz = systems["water"]
    blob = xtc_ref.frame_bytes(xyz, np.diag([30.0, 30.0, 30.0]), 0, 0.0, 1000.0)
    rejected = 0
    for trial in range(trials):
        b = bytearray(blob)
        for _ in range(rng.integers(1, 5)):
            b[rng.integers(92, len(b))] = rng.integers(0, 256)
        _, status = _device_decode(lib, bytes(b), xyz.shape[1], chunk, gpu)
        rejected += int(status[0] != 0)
    assert rejected > 0


@pytest.mark.parametrize("chunk", [0, 64, 300, -1, -2, -3, -4, -5, -6])
def test_device_xtc_decoder_matches_the_host_reader(tmp_path, emu_lib, chunk):
    """k_xtc_decode (one GPU thread per frame), the two-pass k_xtc_index + k_xtc_chunks (one thread per chunk of `chunk` atoms) and
    k_xtc_wave (chunk -1: one wave per frame, speculative group walk; all here on the SIMT emulator) against the host reader on
    every fixture: the same floats bit for bit; a 68-bit packed triple is reported as unsupported (status 2), a damaged stream as
    corrupt (status 1) or decoded without leaving the frame's buffers."""
    _device_decoder_against_host_reader(tmp_path, emu_lib, chunk, False, trials=15)     # damaged streams: 15 here, 10 on the GPU, hundreds in scripts/fuzz_xtc.py


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [-1, -2, -3, -4, -5, -6, 0, 128])
def test_device_xtc_decoder_matches_the_host_reader_on_the_gpu(tmp_path, gpu_lib, chunk):
    """The same fixtures through the hipcc-built kernels on the MI355X (device memory from torch)."""
    _device_decoder_against_host_reader(tmp_path, gpu_lib, chunk, True, trials=10)


def test_xtc_batches_decoded_on_the_device_through_the_evaluator(tmp_path, emu_lib, oracle):
    """vmd_set_option("xtc_device_decode", 1): the evaluator pulls the compressed frames (load_raw), moves the bit streams to
    the device and decompresses the batch there; same histogram as host decoding, every frame counted as device-decoded; a TRR
    file (nothing to decompress) and a damaged XTC file fall back to load_frame."""
    import cases
    box, F, N = 36.0, 10, 1200
    coords = cases.water_box(oracle, 31, N, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "w.xtc"
    V.write_xtc(p, coords, cell, lib=emu_lib)
    q = tmp_path / "w.trr"
    V.write_trr(q, coords, cell, lib=emu_lib)
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    old_b = emu_lib.vmd_set_option(b"batch_frames", 4)
    res = {}
    try:
        for mode in (0, 1, 2, 3):
            old = emu_lib.vmd_set_option(b"xtc_device_decode", mode)
            old_c = emu_lib.vmd_set_option(b"xtc_chunk", 64)
            try:
                ev = V.ScriptEval(F, ir)
                xt = V.XdrTrajectory(p, lib=emu_lib)
                assert ev.frame_range(sysm, xt, 0, F)
                res[mode] = ev.property_data("g").counts.copy()
                assert ev.frames_device_decoded() == (F if mode else 0)
                if mode == 3:
                    # the same eval, the same file again (VIAMD re-evaluates after every script edit): the first pass left decoder
                    # checkpoints, this one decodes every frame in sections from them - and must count the same pairs
                    ev.clear_data()
                    assert ev.frame_range(sysm, xt, 0, 5) and ev.frame_range(sysm, xt, 5, F)
                    np.testing.assert_array_equal(ev.property_data("g").counts, res[mode])
                    assert ev.frames_device_decoded() == F and ev.frames_section_decoded() == F
                    # VIAMD creates a NEW eval for every script edit (src/main.cpp:966-972): the checkpoints belong to the trajectory,
                    # not to the eval, so a fresh eval decodes in sections right away; a different file (another trajectory object,
                    # whatever its address) starts from bit 0 again
                    ev3 = V.ScriptEval(F, ir)
                    assert ev3.frame_range(sysm, xt, 0, F) and ev3.frames_section_decoded() == F
                    np.testing.assert_array_equal(ev3.property_data("g").counts, res[mode])
                    ev4 = V.ScriptEval(F, ir)
                    assert ev4.frame_range(sysm, V.XdrTrajectory(p, lib=emu_lib), 0, F) and ev4.frames_section_decoded() == 0
                if mode:
                    ev2 = V.ScriptEval(F, ir)
                    assert ev2.frame_range(sysm, V.XdrTrajectory(q, lib=emu_lib), 0, F)
                    # nothing to decompress in a TRR file: its floats leave the mapped file by DMA and k_raw_f32 lays them out
                    assert ev2.frames_device_decoded() == F and ev2.frames_section_decoded() == 0 and ev2.property_data("g").counts.sum() > 0
            finally:
                emu_lib.vmd_set_option(b"xtc_device_decode", old)
                emu_lib.vmd_set_option(b"xtc_chunk", old_c)
    finally:
        emu_lib.vmd_set_option(b"batch_frames", old_b)
    np.testing.assert_array_equal(res[1], res[0])
    np.testing.assert_array_equal(res[2], res[0])
    np.testing.assert_array_equal(res[3], res[0])
    assert res[0].sum() > 0
    # the same file kept compressed in device memory (vmd_rawtraj_*): batches are decoded from the resident copy whatever the option
    # says, two evaluations in a row, several frame ranges; a TRR file has no compressed form
    old_b = emu_lib.vmd_set_option(b"batch_frames", 3)
    try:
        for records in (1, 2):              # 2: the resident object also keeps group records (a later pass walks nothing)
            old_r = emu_lib.vmd_set_option(b"xtc_records", records)
            try:
                ct = V.CompressedDeviceTrajectory(V.XdrTrajectory(p, lib=emu_lib))
            finally:
                emu_lib.vmd_set_option(b"xtc_records", old_r)
            assert 0 < ct.device_bytes() < coords.nbytes
            for rep in range(2):
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, ct, 0, 4) and ev.frame_range(sysm, ct, 4, F)
                np.testing.assert_array_equal(ev.property_data("g").counts, res[0])
                assert ev.frames_device_decoded() == F and ev.frames_section_decoded() == (F if rep else 0)
        with pytest.raises(V.VmdError, match="compressed"):
            V.CompressedDeviceTrajectory(V.XdrTrajectory(q, lib=emu_lib))
    finally:
        emu_lib.vmd_set_option(b"batch_frames", old_b)


def test_xtc_batches_leave_the_mapped_file_by_dma(tmp_path, emu_lib, oracle):
    """The native XTC reader maps its file (raw_mapped_view); the evaluator pins the mapping and the copy engine takes every batch's
    span of it as it lies in the file - streams on 4-byte boundaries, frame headers in between.  Same histogram as the load_raw copy
    into a pinned block and as host decoding; the copy stays the fallback when the mapping cannot be pinned or the option is off."""
    import os
    import cases
    box, F, N = 33.0, 11, 1101            # frame sizes of every residue modulo 8: both phases of the 8-byte reader
    coords = cases.water_box(oracle, 17, N, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "m.xtc"
    V.write_xtc(p, coords, cell, lib=emu_lib)
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 8.0)
    sysm = V.MolSystem(N, unitcell=cell)
    old_b = emu_lib.vmd_set_option(b"batch_frames", 4)
    old_d = emu_lib.vmd_set_option(b"xtc_device_decode", 0)
    try:
        ev = V.ScriptEval(F, ir)
        assert ev.frame_range(sysm, V.XdrTrajectory(p, lib=emu_lib), 0, F)
        want = ev.property_data("g").counts.copy()
        assert want.sum() > 0 and ev.frames_mapped() == 0
        emu_lib.vmd_set_option(b"xtc_device_decode", 3)
        xt = V.XdrTrajectory(p, lib=emu_lib)
        for rep in range(2):               # the second evaluation enters every frame at its checkpoints, through the mapping again
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, xt, 0, 3) and ev.frame_range(sysm, xt, 3, F)
            np.testing.assert_array_equal(ev.property_data("g").counts, want)
            assert ev.frames_device_decoded() == F and ev.frames_mapped() == F
            assert ev.frames_section_decoded() == (F if rep else 0)
        old_m = emu_lib.vmd_set_option(b"xtc_mapped", 0)
        try:
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, xt, 0, F)
            np.testing.assert_array_equal(ev.property_data("g").counts, want)
            assert ev.frames_device_decoded() == F and ev.frames_mapped() == 0
        finally:
            emu_lib.vmd_set_option(b"xtc_mapped", old_m)
        # the evaluator's own batch plan (no batch_frames): a first pass keeps several walks in flight on their own streams (batches of
        # 2 x stage_frames, four staged ahead), the re-evaluation runs batches of stage_frames one ahead
        emu_lib.vmd_set_option(b"batch_frames", 0)
        old_s = emu_lib.vmd_set_option(b"stage_frames", 1)
        try:
            xt3 = V.XdrTrajectory(p, lib=emu_lib)
            for rep in range(2):
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, xt3, 0, F)
                np.testing.assert_array_equal(ev.property_data("g").counts, want)
                assert ev.frames_mapped() == F and ev.frames_section_decoded() == (F if rep else 0)
        finally:
            emu_lib.vmd_set_option(b"stage_frames", old_s)
            emu_lib.vmd_set_option(b"batch_frames", 4)
        # the driver refuses to pin the mapping (emulator switch): the window is marked refused, batches take the copy
        os.environ["VIAMD_EMU_NO_HOST_REGISTER"] = "1"
        try:
            xt2 = V.XdrTrajectory(p, lib=emu_lib)
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, xt2, 0, F)
            np.testing.assert_array_equal(ev.property_data("g").counts, want)
            assert ev.frames_device_decoded() == F and ev.frames_mapped() == 0
        finally:
            del os.environ["VIAMD_EMU_NO_HOST_REGISTER"]
    finally:
        emu_lib.vmd_set_option(b"xtc_device_decode", old_d)
        emu_lib.vmd_set_option(b"batch_frames", old_b)


def test_plain_float_files_leave_the_mapped_file_by_dma(tmp_path, emu_lib, oracle):
    """TRR (big-endian nm, xyz interleaved) and DCD (a block per component, either byte order, with or without a cell record): the
    copy engine takes every batch's span of the mapped file, k_raw_f32 swaps / scales / transposes it into the frame layout.  Same
    integers as load_frame on host threads (raw_f32_device = 0), which stays the path when the mapping cannot be pinned."""
    import os
    import cases
    box, F, N = 31.0, 9, 1001
    coords = cases.water_box(oracle, 41, N, box, F)
    cell = V.make_unitcell(box)
    o = cases.oxygen(N)
    ir = V.ScriptIR(emu_lib); ir.add_rdf("g", o, o, 8.0)
    sysm = V.MolSystem(N, unitcell=cell)
    files = {}
    q = tmp_path / "f.trr"; V.write_trr(q, coords, cell, lib=emu_lib); files["trr"] = lambda: V.XdrTrajectory(q, lib=emu_lib)
    for tag, kw in (("dcd", {}), ("dcd_be", {"big_endian": True})):
        d = tmp_path / (tag + ".dcd"); V.write_dcd(d, coords, cell, **kw)
        files[tag] = (lambda d=d: V.DcdTrajectory(d, lib=emu_lib))
    dn = tmp_path / "nocell.dcd"; V.write_dcd(dn, coords, None)
    old_b = emu_lib.vmd_set_option(b"batch_frames", 4)
    try:
        for tag, opener in files.items():
            old = emu_lib.vmd_set_option(b"raw_f32_device", 0)
            try:
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, opener(), 0, F)
                want = ev.property_data("g").counts.copy()
                assert want.sum() > 0 and ev.frames_mapped() == 0 and ev.frames_device_decoded() == 0
            finally:
                emu_lib.vmd_set_option(b"raw_f32_device", old)
            t = opener()
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, t, 0, 2) and ev.frame_range(sysm, t, 2, F)
            np.testing.assert_array_equal(ev.property_data("g").counts, want, err_msg=tag)
            assert ev.frames_mapped() == F and ev.frames_device_decoded() == F, tag
            os.environ["VIAMD_EMU_NO_HOST_REGISTER"] = "1"        # the mapping cannot be pinned: load_frame on host threads
            try:
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sysm, opener(), 0, F)
                np.testing.assert_array_equal(ev.property_data("g").counts, want, err_msg=tag)
                assert ev.frames_mapped() == 0 and ev.frames_device_decoded() == 0
            finally:
                del os.environ["VIAMD_EMU_NO_HOST_REGISTER"]
        # a DCD file without cell records: open boundaries from the system, same floats either way
        sys_open = V.MolSystem(N)
        got = []
        for dev in (0, 1):
            old = emu_lib.vmd_set_option(b"raw_f32_device", dev)
            try:
                ev = V.ScriptEval(F, ir)
                assert ev.frame_range(sys_open, V.DcdTrajectory(dn, lib=emu_lib), 0, F)
                got.append(ev.property_data("g").counts.copy())
                assert ev.frames_mapped() == (F if dev else 0)
            finally:
                emu_lib.vmd_set_option(b"raw_f32_device", old)
        np.testing.assert_array_equal(got[0], got[1])
        assert got[0].sum() > 0
    finally:
        emu_lib.vmd_set_option(b"batch_frames", old_b)
