"""XTC / TRR files staged through the multi-threaded host decode into the kernels, and XTC batches decompressed on the device
(k_xtc_decode, k_xtc_index + k_xtc_chunks).  Written after the last GPU session of round 1: the file name sorts last so that
`pytest -x` reaches the never-run-on-hardware device decoder only after everything else has passed."""
import numpy as np
import pytest

import cases
import viamd_amd as V


def _xdr_through_the_evaluator(lib, oracle, tmp_path, box, F, N, device_check):
    coords = cases.water_box(oracle, 31, N, box, F)
    cell = V.make_unitcell(box)
    o = cases.oxygen(N)
    ir = V.ScriptIR(lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    res = {}
    for fmt, write in (("xtc", V.write_xtc), ("trr", V.write_trr)):
        p = tmp_path / f"w.{fmt}"
        write(p, coords, cell, lib=lib)
        t = V.XdrTrajectory(p, lib=lib)
        decoded = np.stack([t.load_frame(f)[0] for f in range(F)])
        assert abs(decoded - coords).max() < (0.0051 if fmt == "xtc" else 1e-4)
        old = lib.vmd_set_option(b"batch_frames", max(4, F // 3))          # several staged batches, decoded on several threads
        try:
            ev = V.ScriptEval(F, ir)
            assert ev.frame_range(sysm, V.XdrTrajectory(p, lib=lib), 0, F)
        finally:
            lib.vmd_set_option(b"batch_frames", old)
        got = ev.property_data("g").counts.copy()
        counts, _ = cases.oracle_rdf(oracle, decoded, oracle.make_cell(box), o, o, 0.0, 9.0)
        np.testing.assert_array_equal(got, counts, err_msg=fmt)            # bit-exact on the coordinates the file holds
        res[fmt] = got
        if fmt == "xtc":
            # the same file with the batch decompressed on the device: one thread per frame (1), index pass + one thread per
            # chunk (2), one wave per frame (3); identical integers, every frame counted as device-decoded
            for mode in (1, 2, 3):
                old_d = lib.vmd_set_option(b"xtc_device_decode", mode)
                old_c = lib.vmd_set_option(b"xtc_chunk", 256)
                old = lib.vmd_set_option(b"batch_frames", max(4, F // 3))
                try:
                    ev = V.ScriptEval(F, ir)
                    assert ev.frame_range(sysm, V.XdrTrajectory(p, lib=lib), 0, F)
                finally:
                    lib.vmd_set_option(b"batch_frames", old)
                    lib.vmd_set_option(b"xtc_chunk", old_c)
                    lib.vmd_set_option(b"xtc_device_decode", old_d)
                assert ev.frames_device_decoded() == F
                np.testing.assert_array_equal(ev.property_data("g").counts, counts, err_msg=f"xtc, device decode variant {mode}")
    assert res["xtc"].sum() > 0 and abs(int(res["xtc"].sum()) - int(res["trr"].sum())) < 0.01 * res["trr"].sum()


def test_xdr_files_through_the_evaluator_on_emulator(emu_lib, oracle, tmp_path):
    _xdr_through_the_evaluator(emu_lib, oracle, tmp_path, 30.0, 6, 600, False)


@pytest.mark.gpu
def test_xdr_files_through_the_evaluator(gpu_lib, oracle, tmp_path):
    _xdr_through_the_evaluator(gpu_lib, oracle, tmp_path, 70.0, 24, 30000, True)


@pytest.mark.gpu
def test_xtc_batches_leave_the_mapped_file_by_dma_on_the_gpu(gpu_lib, oracle, tmp_path):
    """hipHostRegister on the reader's file mapping, hipMemcpyAsync straight out of it, k_xtc_wave on streams that start on 4-byte
    boundaries: identical histogram to the pinned-block copy and to host decoding, first pass and re-evaluation from checkpoints."""
    box, F, N = 60.0, 40, 21003
    coords = cases.water_box(oracle, 23, N, box, F)
    cell = V.make_unitcell(box)
    p = tmp_path / "m.xtc"
    V.write_xtc(p, coords, cell, lib=gpu_lib)
    o = cases.oxygen(N)
    ir = V.ScriptIR(gpu_lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    res = {}
    old_b = gpu_lib.vmd_set_option(b"batch_frames", 16)
    try:
        for name, decode, mapped in (("host", 0, 0), ("copy", 3, 0), ("mapped", 3, 1)):
            old_d = gpu_lib.vmd_set_option(b"xtc_device_decode", decode)
            old_m = gpu_lib.vmd_set_option(b"xtc_mapped", mapped)
            try:
                xt = V.XdrTrajectory(p, lib=gpu_lib)
                for rep in range(2):
                    ev = V.ScriptEval(F, ir)
                    assert ev.frame_range(sysm, xt, 0, 7) and ev.frame_range(sysm, xt, 7, F)
                    got = ev.property_data("g").counts.copy()
                    assert ev.frames_device_decoded() == (F if decode else 0) and ev.frames_mapped() == (F if mapped else 0)
                    if decode:
                        assert ev.frames_section_decoded() == (F if rep else 0)
                    np.testing.assert_array_equal(got, res.setdefault(name, got))
            finally:
                gpu_lib.vmd_set_option(b"xtc_mapped", old_m)
                gpu_lib.vmd_set_option(b"xtc_device_decode", old_d)
    finally:
        gpu_lib.vmd_set_option(b"batch_frames", old_b)
    np.testing.assert_array_equal(res["copy"], res["host"])
    np.testing.assert_array_equal(res["mapped"], res["host"])
    assert res["host"].sum() > 0


@pytest.mark.gpu
def test_plain_float_files_leave_the_mapped_file_by_dma_on_the_gpu(gpu_lib, oracle, tmp_path):
    """TRR and DCD frames: DMA out of the pinned file mapping + k_raw_f32 against load_frame on host threads - the same floats, so the
    same integers; every frame counted as mapped and device-decoded."""
    box, F, N = 60.0, 36, 21003
    coords = cases.water_box(oracle, 29, N, box, F)
    cell = V.make_unitcell(box)
    o = cases.oxygen(N)
    ir = V.ScriptIR(gpu_lib); ir.add_rdf("g", o, o, 9.0)
    sysm = V.MolSystem(N, unitcell=cell)
    q = tmp_path / "f.trr"; V.write_trr(q, coords, cell, lib=gpu_lib)
    d1 = tmp_path / "le.dcd"; V.write_dcd(d1, coords, cell)
    d2 = tmp_path / "be.dcd"; V.write_dcd(d2, coords, cell, big_endian=True)
    openers = {"trr": lambda: V.XdrTrajectory(q, lib=gpu_lib), "dcd": lambda: V.DcdTrajectory(d1, lib=gpu_lib), "dcd_be": lambda: V.DcdTrajectory(d2, lib=gpu_lib)}
    old_b = gpu_lib.vmd_set_option(b"batch_frames", 8)
    try:
        for tag, opener in openers.items():
            res = []
            for dev in (0, 1):
                old = gpu_lib.vmd_set_option(b"raw_f32_device", dev)
                try:
                    t = opener()
                    ev = V.ScriptEval(F, ir)
                    assert ev.frame_range(sysm, t, 0, 5) and ev.frame_range(sysm, t, 5, F)
                    res.append(ev.property_data("g").counts.copy())
                    assert ev.frames_mapped() == (F if dev else 0) and ev.frames_device_decoded() == (F if dev else 0), tag
                finally:
                    gpu_lib.vmd_set_option(b"raw_f32_device", old)
            np.testing.assert_array_equal(res[0], res[1], err_msg=tag)
            assert res[0].sum() > 0
    finally:
        gpu_lib.vmd_set_option(b"batch_frames", old_b)


def _pool_from_files(lib, oracle, tmp_path, box, F, N, nthreads):
    """VIAMD's call pattern on trajectories that come out of FILES: pool threads ask for one frame each, read-ahead evaluates regions of
    frame blocks - whose frames are staged / decompressed on the device / parsed by the native readers - and every file type must leave what
    one call over the same file leaves, bit for bit (XTC: the device decoder and its checkpoints under regions of changing size)."""
    import threading
    from viamd_amd import pdb
    from viamd_amd.script import Topology
    coords = cases.water_box(oracle, 37, N, box, F)
    cell = V.make_unitcell(box)
    o = cases.oxygen(N)
    ir = V.ScriptIR(lib); ir.add_rdf("g", o, o, 9.0); ir.add_distance("d", np.arange(3, dtype=np.int32), np.arange(30, 33, dtype=np.int32), 1)
    sysm = V.MolSystem(N, unitcell=cell)
    topo = Topology((["O", "H", "H"] * N)[:N])
    files = {}
    for fmt, write in (("xtc", V.write_xtc), ("trr", V.write_trr), ("dcd", None), ("pdb", None)):
        p = tmp_path / f"pool.{fmt}"
        if fmt == "dcd":
            V.write_dcd(p, coords, cell)
        elif fmt == "pdb":
            pdb.write_pdb(p, coords, topo, box=box)
        else:
            write(p, coords, cell, lib=lib)
        files[fmt] = p
    opts = [(b"readahead_block", 4), (b"readahead_frames", 8), (b"readahead_growth", 2), (b"readahead_company_us", 200000)]
    old = [(k, lib.vmd_set_option(k, v)) for k, v in opts]
    try:
        for fmt, p in files.items():
            opener = {"xtc": V.XdrTrajectory, "trr": V.XdrTrajectory, "dcd": V.DcdTrajectory, "pdb": V.TextTrajectory}[fmt]
            one = V.ScriptEval(F, ir)
            old_ra = lib.vmd_set_option(b"readahead", 0)
            try:
                assert one.frame_range(sysm, opener(p, lib=lib), 0, F)
            finally:
                lib.vmd_set_option(b"readahead", old_ra)
            traj = opener(p, lib=lib)
            ev = V.ScriptEval(F, ir)
            nxt = [0]; lock = threading.Lock(); res = []
            gate = threading.Barrier(nthreads)
            def work():
                gate.wait()
                while True:
                    with lock:
                        f = nxt[0]; nxt[0] += 1
                    if f >= F:
                        return
                    res.append(ev.frame_range(sysm, traj, f, f + 1))
            ths = [threading.Thread(target=work) for _ in range(nthreads)]
            [t.start() for t in ths]; [t.join() for t in ths]
            assert all(res) and ev.frames_done() == F and ev.frame_mask().all(), fmt
            st = ev.readahead_stats()
            assert st["engaged"] == 1 and st["regions"] >= 2, (fmt, st)
            np.testing.assert_array_equal(ev.property_data("g").counts, one.property_data("g").counts, err_msg=fmt)
            np.testing.assert_array_equal(ev.property_data("d").values, one.property_data("d").values, err_msg=fmt)
            np.testing.assert_array_equal(ev.property_data("g").weights, one.property_data("g").weights, err_msg=fmt)
            assert one.property_data("g").counts.sum() > 0
            ev.close(); one.close()
    finally:
        for k, v in old:
            lib.vmd_set_option(k, v)


def test_pool_pattern_on_file_trajectories_on_emulator(emu_lib, oracle, tmp_path):
    _pool_from_files(emu_lib, oracle, tmp_path, 30.0, 24, 600, 5)


@pytest.mark.gpu
def test_pool_pattern_on_file_trajectories(gpu_lib, oracle, tmp_path):
    _pool_from_files(gpu_lib, oracle, tmp_path, 70.0, 96, 30000, 12)
