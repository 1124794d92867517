"""N>1 path on CPU: world_size-2 gloo job, frames block-sharded, one all-reduce at the end.
Compute runs on the SIMT emulator build (tests only); the merge code is the product's viamd_amd/dist.py."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir, mode="host", on_gpu=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import cases
    import conftest
    import viamd_amd as V
    from viamd_amd import _lib as L
    from viamd_amd.dist import reduce_eval, shard_frames
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode.endswith("+measure"):
        # the path config 4 takes at N >= 2: the static bound cannot prove the 32-bit sum, the ranks measure their largest voxel, the sums of
        # the maxima travel in the packed all-reduce FIRST and decide (here: forced, on a system of test size)
        os.environ["VIAMD_AMD_REDUCE_MEASURE"] = "1"
        mode = mode[: -len("+measure")]
    defer = mode.endswith("+defer")
    if defer:
        mode = mode[: -len("+defer")]
    if on_gpu:
        # real device memory, real kernels, every process on GPU 0; the merge's all-reduces are staged through host memory (gloo)
        # because RCCL refuses two ranks on one device.  torch first: its HIP runtime must be the one that finds the GPU (conftest.gpu_lib)
        import torch
        torch.cuda.set_device(0)
        os.environ["VIAMD_AMD_STAGED_COLLECTIVE"] = "1"
        lib = V.default_lib()
        assert lib.vmd_device_count() > 0
        lib.vmd_set_device(0)
    else:
        lib = V.VmdLib(conftest.EMU_LIB)
    F = 5
    coords, structures, mass = cases.sdf_system(O, 21, 900, 36.0, F)
    N = coords.shape[2]
    n_s = structures.size
    o = np.arange(n_s, N, 3, dtype=np.int32)
    ir = V.ScriptIR(lib)
    ir.add_rdf("goo", o, o, 12.0)
    ir.add_sdf("v", structures, o, 8.0)
    ir.add_distance("d", structures[0], structures[1], L.DIST_COM)
    ir.add_distance("dp", structures[0][:2], structures[2][:3], L.DIST_PAIR)
    ev = V.ScriptEval(F, ir)
    if defer:
        ev.defer_volume_views(True)          # what bench.py does at N > 1: a rank's partial volume gets no float view of its own
    vcell = V.make_unitcell(36.0)
    beg, end = shard_frames(F, rank, world)
    F = 11 if mode == "shard_pool" else 5
    if mode == "shard_pool":
        coords, structures, mass = cases.sdf_system(O, 21, 900, 36.0, F)
        ev.close()
        ev = V.ScriptEval(F, ir)
        beg, end = shard_frames(F, rank, world)
    if mode == "host":
        traj = V.HostTrajectory(coords, vcell)
    else:
        # one rank's shard of a device-resident trajectory: global frame indices, only [beg, end) (+ frame 0: SDF reference
        # pose) resident; frames outside the shard must be refused
        traj = V.DeviceTrajectory(F, N, lib=lib, shard=(beg, end))
        assert not on_gpu or bool(traj.device_ptr()[0])
        for f in sorted(set(range(beg, end)) | {0}):
            traj.upload_frame(f, vcell, coords[f, 0], coords[f, 1], coords[f, 2])
        if rank == world - 1 and mode == "shard":       # world 4: the EMPTY shard [5, 5) - nothing is resident there, it is not "the whole trajectory"
            import pytest
            with pytest.raises(V.VmdError, match="not resident"):
                ev.frame_range(V.MolSystem(N, mass=mass, unitcell=vcell), traj, 0, F)
            ev.clear_data()
    if mode == "shard_pool":
        # the rank's pool threads walk its shard one frame per call (VIAMD's pattern on every GPU of the node): read-ahead evaluates regions of
        # whole blocks INSIDE the shard, the blocks that straddle its ends are evaluated frame by frame
        import threading
        old = [(k, lib.vmd_set_option(k, v)) for k, v in ((b"readahead_block", 2), (b"readahead_frames", 2), (b"readahead_company_us", 200000))]
        sysm = V.MolSystem(N, mass=mass, unitcell=vcell)
        nxt = [beg]; lock = threading.Lock(); res = []
        def work():
            while True:
                with lock:
                    f = nxt[0]; nxt[0] += 1
                if f >= end:
                    return
                res.append(ev.frame_range(sysm, traj, f, f + 1))
        ths = [threading.Thread(target=work) for _ in range(3)]
        [t.start() for t in ths]; [t.join() for t in ths]
        for k, v in old:
            lib.vmd_set_option(k, v)
        assert all(res)
    else:
        assert ev.frame_range(V.MolSystem(N, mass=mass, unitcell=vcell), traj, beg, end)
    assert ev.frames_done() == end - beg
    if defer:
        # nothing travelled for the volume, everything else is current
        assert not ev.property_data("v").values.any() and (end == beg or ev.property_data("goo").values.sum() > 0)
    reduce_eval(ev)
    if defer:
        pv = ev.property_data("v")
        assert pv.values.sum() > 0 and np.array_equal(pv.values, pv.counts.astype(np.float32)) and pv.max_value == pv.values.max()
    from viamd_amd.dist import reduce_stats
    st = reduce_stats(ev)
    if os.environ.get("VIAMD_AMD_REDUCE_MEASURE") == "1":
        assert st["volumes_as_u32"] == 1 and st["allreduce_calls"] >= 3, st      # measured: a few hundred hits per voxel at most - the volume travelled as u32
    assert ev.frame_mask().all() and ev.frames_done() == F
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), goo=ev.property_data("goo").counts, w=ev.property_data("goo").weights64,
             v=ev.property_data("v").counts, d=ev.property_data("d").values, dp=ev.property_data("dp").values,
             gv=ev.property_data("goo").values)
    dist.destroy_process_group()


def test_shard_frames_covers_everything():
    from viamd_amd.dist import shard_frames
    for F in (1, 5, 8, 1000, 1001):
        for G in (1, 2, 3, 8):
            got = []
            for r in range(G):
                b, e = shard_frames(F, r, G)
                got += list(range(b, e))
            assert got == list(range(F))


import pytest


@pytest.mark.parametrize("mode,world", [("host", 2), ("shard", 2), ("host", 4), ("shard", 4), ("shard_pool", 2), ("shard+measure", 2), ("host+measure", 3), ("shard+defer", 2)])
def test_two_rank_gloo_merge_matches_oracle(oracle, emu_lib, tmp_path, mode, world):
    """every rank evaluates its block of frames, ONE vmd_eval_reduce (C++, behind the ABI) merges; `shard`: each rank holds
    only its block of a device trajectory; 4 ranks on 5 frames: blocks of 2, 2, 1 and an EMPTY block (a rank without frames still
    takes part in the merge)"""
    import cases
    from viamd_amd import _lib as L
    port = 29500 + (os.getpid() % 2000) + (7 if mode.startswith("shard") else 0) + (11 if mode == "shard_pool" else 0) + (17 if mode.endswith("measure") else 0) + (23 if mode.endswith("defer") else 0) + 13 * (world - 2)
    # ("shard", 4): rank 3 owns no frame; its device view must refuse every range (it used to read as "unsharded")
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    F = 11 if mode == "shard_pool" else 5
    coords, structures, mass = cases.sdf_system(oracle, 21, 900, 36.0, F)
    N = coords.shape[2]
    o = np.arange(structures.size, N, 3, dtype=np.int32)
    ocell = oracle.make_cell(36.0)
    counts, weights = cases.oracle_rdf(oracle, coords, ocell, o, o, 0.0, 12.0)
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, structures, mass, o, 8.0)
    d = cases.oracle_distance(oracle, coords, ocell, mass, structures[0], structures[1], L.DIST_COM)
    dp = cases.oracle_distance(oracle, coords, ocell, mass, structures[0][:2], structures[2][:3], L.DIST_PAIR)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(z["goo"], counts)           # bit-identical on every rank
        np.testing.assert_array_equal(z["gv"], counts.astype(np.float32))
        np.testing.assert_allclose(z["w"], weights, rtol=1e-12)
        np.testing.assert_array_equal(z["v"], vol)
        np.testing.assert_array_equal(z["d"].reshape(F, -1), d)
        np.testing.assert_array_equal(z["dp"].reshape(F, -1), dp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "shard"), (3, "shard"), (2, "shard+measure")])
def test_processes_sharing_one_gpu_merge_matches_oracle(oracle, gpu_lib, tmp_path, world, mode):
    """VERDICT r03 #5 / SURVEY 8e on the hardware there is: `world` PROCESSES on GPU 0, each holding only its shard of a device
    trajectory (vmd_devtraj_create_shard) and running the real kernels on it, ONE vmd_eval_reduce per rank whose collective is a
    host-staged vmd_collective_i (D2H, gloo all-reduce, H2D).  Covers shard residency, per-process device state and the merge's
    pack / narrow / unpack on real device memory; the RCCL transport itself needs one GPU per rank (driver's SCALE run)."""
    import cases
    from viamd_amd import _lib as L
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port + (5 if mode != "shard" else 0), str(tmp_path), mode, True), nprocs=world, join=True)
    F = 5
    coords, structures, mass = cases.sdf_system(oracle, 21, 900, 36.0, F)
    N = coords.shape[2]
    o = np.arange(structures.size, N, 3, dtype=np.int32)
    ocell = oracle.make_cell(36.0)
    counts, weights = cases.oracle_rdf(oracle, coords, ocell, o, o, 0.0, 12.0)
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, structures, mass, o, 8.0)
    d = cases.oracle_distance(oracle, coords, ocell, mass, structures[0], structures[1], L.DIST_COM)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(z["goo"], counts)
        np.testing.assert_allclose(z["w"], weights, rtol=1e-12)
        np.testing.assert_array_equal(z["v"], vol)
        np.testing.assert_array_equal(z["d"].reshape(F, -1), d)
