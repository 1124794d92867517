#!/usr/bin/env python
"""bench.py — throughput of the RDF / SDF / distance hot path on MI355X, one JSON line (contract: the task brief).

A *step* is one pass of the hot path over one batch of synthetic input: a full evaluation (clear_data ->
frame_range over the whole resident trajectory -> merge) of the workload's script.  Inputs are resident in HBM
before the timed region (trajectories are pre-staged).  N > 1: one process per GPU
(torch.distributed / RCCL), every rank owns its own block of frames (weak scaling), one all-reduce per step
merges the integer accumulators.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c3d|c4|c5] [--frames F] [--scaling weak|strong]

Default (no flags): N = 1, workload c3 = BASELINE.json configs[2], the configuration the metric and north_star are quoted on
(1M-atom box, 1k frames, all-heavy-atom RDF r_cut 12 A; 12 GB resident).  At N = 1 the same JSON line carries a `secondary`
block with short runs of the other single-GPU configurations (c2 = configs[1], c4 = configs[3], c5 = configs[4], c3d = SURVEY 8d's dense
variant of configs[2], c4_1250 = one rank's share of configs[3] at 8 GPUs): value,
ms_per_step, dominant-kernel time, kernel-level and step-level HBM fraction each.
--scaling strong: the workload's frames are block-sharded over the ranks (configs[3]: 10 000 frames / N) instead of every
rank owning its own frames; either way ONE vmd_eval_reduce (RCCL, C++ behind the ABI) merges per step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU ceiling: the guide's 2 cycles per wave64 instruction per SIMD (/opt/skills/guides/MI355X_MICROARCH.md) - the one rate nobody
# calibrated here.  Only plain fp32 add / mul / fma on VGPR operands reach it (profiles/r02_valu_calibration.txt: 840 - 970 G/s with the
# clock pulled down); v_cmp, v_mbcnt, v_lshl_add, DPP and SGPR-fed ops issue at ~4 cycles (the same file, profiles/r05a_subwave_calibration.txt),
# so a kernel made of them tops out near 0.5 of this ceiling.  `roofline.valu.frac` is priced against it and nothing else; how busy the
# vector ALU actually is comes from the counters (`busy` = SQ_ACTIVE_INST_VALU / SIMD cycles of the kernel: a ratio of two counters).
VALU_CLOCK_HZ = 2.4e9
VALU_PEAK_2CYCLE_WINST_S = 256 * 4 * VALU_CLOCK_HZ / 2.0

WORKLOADS = {
    # SURVEY.md 8d.  frames = frames resident per GPU and evaluated per step.
    "c2": dict(atoms=100002, blob=0, box=100.0, frames=1000, seed=2, steps=20, sec_steps=10, kernel="rdf_pencil",
               script="g = rdf(element('O'), element('O'), 12.0);",
               desc="BASELINE configs[1]: synthetic 100002-atom periodic water box, 1000 frames, O-O RDF r_cut=12 A, 1024 bins"),
    "c3": dict(atoms=1000002, blob=0, box=215.443, frames=1000, seed=3, steps=3, kernel="rdf_pencil",
               script="g = rdf(not element('H'), not element('H'), 12.0);",
               desc="BASELINE configs[2]: synthetic 1000002-atom box, 1000 frames, all-heavy-atom RDF r_cut=12 A, 1024 bins"),
    "c3d": dict(atoms=1000002, blob=0, box=215.443, frames=200, seed=3, steps=3, kernel="rdf_pencil",
                script="g = rdf(all, all, 12.0);",
                desc="SURVEY 8d C3-dense: synthetic 1000002-atom box, every atom counted as heavy (7.24e8 ordered pairs per frame), RDF r_cut=12 A"),
    "c4": dict(atoms=100001, blob=2000, box=100.0, frames=10000, seed=4, steps=5, sec_steps=3, kernel="sdf_scatter",
               script="s = residue(5:11); v = sdf(s, element('O') and water, 10.0);",
               desc="BASELINE configs[3]: 100001-atom solvated protein-like blob, 10000 frames, SDF 128^3 around 7 residues + reference-frame tracking"),
    "c5": dict(atoms=1001999, blob=2000, box=215.443, frames=1000, seed=5, steps=3, sec_steps=2, kernel="rdf_pencil",
               script=("goo = rdf(element('O') and water, element('O') and water, 12.0);"
                       "goh = rdf(element('O') and water, element('H') and water, 12.0);"
                       "ghv = rdf(not element('H'), not element('H'), 12.0);"
                       "s = residue(5:11); v = sdf(s, element('O') and water, 10.0);"
                       "d1 = distance(1, 1990); d2 = distance(residue(1), residue(200));"
                       "d3 = distance_min(residue(3), residue(150)); d4 = distance_max(residue(10), residue(20));"),
               desc="BASELINE configs[4]: 1M-atom box + blob, 3x RDF + 1x SDF + 4x distance co-evaluated per frame"),
}


def host_cpus():
    """(logical CPUs this process may run on, physical cores among them, cgroup CPU quota in cores or None)."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical, quota = logical, None
    try:
        allowed = os.sched_getaffinity(0)
        cores, cpu, phys = set(), None, "0"
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "physical id":
                phys = v
            elif k == "core id" and cpu in allowed:
                cores.add((phys, v))
        if cores:
            physical = len(cores)
    except Exception:
        pass
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                          # cgroup v2: "max 100000" or "<quota> <period>"
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
        else:                                                                 # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
    except Exception:
        pass
    return logical, physical, quota


def node_physical_cores():
    """physical cores of the whole node (/proc/cpuinfo), whatever slice of it this process may use"""
    try:
        cores, phys = set(), "0"
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "physical id":
                phys = v
            elif k == "core id":
                cores.add((phys, v))
        return len(cores) or None
    except Exception:
        return None


def cpu_baseline(name, w, topo, info):
    """The CPU column, driven like VIAMD drives mdlib: one thread per physical core this process is granted (src/main.cpp:494-495),
    frames handed out dynamically with grain 1 (src/task_system.cpp:73-81), on a bounded sample of the same workload.  ext/mdlib is
    empty, so mdlib itself cannot be timed; two stand-ins are:
      scalar - the oracle (oracle/vmd_oracle.c): r_c cells, full 27-cell shell, scalar binning - the checker, not a tuned code;
      simd   - oracle/vmd_cpu_fast.c: half shell, AVX-512 filter + hit compaction, the same integers (tests/test_oracle.py) - what a
               careful CPU implementation reaches; RDF only, the SDF / distance part of a script stays with the oracle.
    `value` is the FASTER of the two.  A single-thread run is timed as well, so the line shows how many cores' worth of CPU the box
    really delivered, and `extrapolated_to_node` scales the value to every physical core of the node (frames are independent)."""
    from oracle import oracle as O
    from viamd_amd import synth
    logical, physical, quota = host_cpus()
    cores = max(1, min(physical, int(np.ceil(quota))) if quota else physical)
    cell = O.make_cell(w["box"])
    budget = 9.0                        # seconds of wall time aimed at, per leg
    mass = topo.mass

    def run(traj, nthreads, fast):
        hits = 0
        cells = [cell] * len(traj)
        for nm, d in info.items():
            if d["kind"] == "rdf":
                r = O.rdf_run_fast(traj, cells, d["ref"], d["target"], d["rmin"], d["rmax"], nthreads=nthreads) if fast else None
                hits += r[1] if r is not None else O.rdf_run(traj, cells, d["ref"], d["target"], d["rmin"], d["rmax"], nthreads=nthreads)[2]
            elif d["kind"] == "sdf":
                O.sdf_run(traj, cells, d["structures"], mass[d["structures"]], d["target"], d["cutoff"], nthreads=nthreads)
            # the distance family is O(|a||b|) on a handful of atoms: below timer resolution next to rdf/sdf
        return hits

    def frames_for(n):
        # host copy of the first n frames, generated by the oracle's generator on a thread pool (ctypes drops the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(cores, 64)) as ex:
            fr = list(ex.map(lambda f: O.synth_frame(w["seed"], w["atoms"], w["box"], 0.05, f, n_blob=w["blob"]), range(n)))
        out = np.stack(fr)
        if w["blob"]:
            for b0, xyz in synth.blob_trajectory(w["seed"], w["blob"], w["box"], n):
                out[b0:b0 + xyz.shape[0], :, :w["blob"]] = xyz
        return out

    cap = max(1, int(6e9 // (12 * w["atoms"])))                    # host copy below ~6 GB

    def leg(fast):
        # calibrate on one frame per core (all cores busy), then spend the rest of the budget on a larger sample
        nfr = min(cores, cap)
        traj = frames_for(nfr)
        n1 = 1 if w["atoms"] > 500000 else min(2, nfr)             # the same code on ONE thread
        t = time.perf_counter()
        run(traj[:n1], 1, fast)
        single = n1 / (time.perf_counter() - t)
        t = time.perf_counter()
        hits = run(traj, cores, fast)
        dt = time.perf_counter() - t
        if dt < 0.4 * budget:
            n2 = int(min(cap, 1000, nfr * 0.8 * budget / max(dt, 1e-3)))
            if n2 > 2 * nfr:
                nfr = n2
                traj = frames_for(nfr)
                t = time.perf_counter()
                hits = run(traj, cores, fast)
                dt = time.perf_counter() - t
        v = nfr / dt
        return {"value": v, "unit": "frames/s", "frames": nfr, "seconds": dt, "pairs_per_s": hits / dt,
                "single_thread": {"value": single, "unit": "frames/s", "frames": n1}, "parallel_speedup": v / single}

    scalar = leg(False)
    has_rdf = any(d["kind"] == "rdf" for d in info.values())
    simd = leg(True) if has_rdf else None
    if simd is not None:
        simd["avx512"] = bool(O.have_avx512())
    best = simd if (simd is not None and simd["value"] > scalar["value"]) else scalar
    # VERDICT r05 next #4c: the CPU leg moves by +-2 % from run to run (16 cores of a shared 128-core node) while the GPU side moves by 0.2 %: the
    # reported value is the MEDIAN of three samples of the winning leg on the same frames, the spread is printed
    fast = best is simd
    traj3 = frames_for(best["frames"])
    samples = [best["value"]]
    for _ in range(2):
        t = time.perf_counter()
        run(traj3, cores, fast)
        samples.append(best["frames"] / (time.perf_counter() - t))
    del traj3
    median = sorted(samples)[1]
    best = dict(best, value=median, seconds=best["frames"] / median, pairs_per_s=best["pairs_per_s"] * median / samples[0])
    node = node_physical_cores()
    return {"value": best["value"], "unit": "frames/s", "cores": cores, "kind": "port",
            "samples": samples, "spread": (max(samples) - min(samples)) / median, "statistic": "median of three samples of the same frames",
            "sample": f"{best['frames']} frames of {name} ({w['atoms']} atoms), {'tuned CPU code (half shell, AVX-512; oracle for SDF)' if fast else 'oracle (cell-list RDF / SDF align+scatter)'}, "
                      f"{cores} OpenMP threads, dynamic grain 1 over frames, {best['seconds']:.1f} s",
            "pairs_per_s": best["pairs_per_s"],
            "scalar": scalar, "simd": simd,
            "single_thread": best["single_thread"],
            "parallel_speedup": best["parallel_speedup"],       # far below `cores` = the box delivers fewer CPUs than it lists
            # frames are independent: a node-wide run of the same code scales with the cores (memory bandwidth permitting)
            "extrapolated_to_node": {"physical_cores": node, "value": best["value"] * node / cores if node else None, "unit": "frames/s",
                                     "note": "value x node cores / cores used: the lease grants a slice of the node's CPUs"},
            "host": {"logical_cpus": logical, "physical_cores": physical, "cgroup_quota_cores": quota}}


def c1_system():
    """SURVEY 8d's C1 stand-in (datasets/1ALA-500.pdb is a missing blob): a 112-atom capped deca-alanine-like chain - ACE (6 atoms), 10 x ALA
    (N H CA HA CB HB1 HB2 HB3 C O), NME (6 atoms) - no periodic cell, 500 frames of a helix-like frame 0 + Gaussian jitter (sigma 0.3 A, seed 1).
    -> (elements, resnames, residue_index, coords [F][3][N])"""
    rng = np.random.default_rng(1)
    ala = ["N", "H", "C", "H", "C", "H", "H", "H", "C", "O"]
    elements = ["C", "H", "H", "H", "C", "O"] + ala * 10 + ["N", "H", "C", "H", "H", "H"]
    resnames = ["ACE"] * 6 + ["ALA"] * 100 + ["NME"] * 6
    residue_index = [0] * 6 + [1 + i // 10 for i in range(100)] + [11] * 6
    n = len(elements)
    assert n == 112
    base = np.zeros((3, n), np.float32)
    for i in range(n):
        r = residue_index[i]
        ang = 1.745 * r                                    # 100 degrees per residue, 1.5 A rise: an alpha helix's backbone trace
        centre = np.array([2.3 * np.cos(ang), 2.3 * np.sin(ang), 1.5 * r])
        base[:, i] = centre + rng.normal(0.0, 0.9, 3)
    F = 500
    coords = (base[None] + rng.normal(0.0, 0.3, (F, 3, n))).astype(np.float32)
    return elements, resnames, np.asarray(residue_index, np.int32), coords


C1_SCRIPT = "s1 = resname(\"ALA\")[2:8]; d1 = distance(10,30); r = rdf(element('C'), element('H'), 10.0); v = sdf(s1, element('H'), 10.0);"


def run_c1(ctx, dry):
    """secondary.c1: the size class of VIAMD's default dataset with the hot-path statements of VIAMD's default script
    (src/main.cpp:528: d1, r, v), once through the boundary the way VIAMD calls it - clear_data, then 16 pool threads pulling ranges of ONE frame
    off [0, 500) (src/main.cpp:990-997) - from a host-memory trajectory (VIAMD's frame cache), and once through the oracle on the same host
    cores with the same dynamic grain-1 hand-out.  The ratio is what include/vmd_md_script_shim.h's work threshold is set from."""
    import torch
    from oracle import oracle as O
    from viamd_amd import _lib as L
    V, lib, script = ctx["V"], ctx["lib"], ctx["script"]
    elements, resnames, residue_index, coords = c1_system()
    F, _, N = coords.shape
    if dry:
        F = 16
        coords = coords[:F]
    from viamd_amd import synth
    topo = script.Topology(elements=elements, resnames=resnames, residue_index=residue_index,
                           mass=np.array([synth.MASS.get(e, 12.0) for e in elements], np.float32))
    ir, info = script.compile_script(C1_SCRIPT, topo)
    cell = V.make_unitcell(None)                               # no periodic cell
    traj = V.PinnedHostTrajectory(F, N)
    traj.upload(coords, cell)
    sysm = V.MolSystem(N, mass=topo.mass, unitcell=cell)
    ev = V.ScriptEval(F, ir)
    threads = 16

    def gpu_once(pooled):
        ev.clear_data()
        t = time.perf_counter()
        ok = ev.frame_range_pooled(sysm, traj, 0, F, threads, 1) if pooled else ev.frame_range(sysm, traj, 0, F)
        assert ok
        return time.perf_counter() - t

    for _ in range(3):
        gpu_once(True)
    n = 3 if dry else 30
    pooled = sorted(gpu_once(True) for _ in range(n))[n // 2]
    one_call = sorted(gpu_once(False) for _ in range(n))[n // 2]
    hits_r = int(ev.property_data("r").counts.sum())
    vox_v = int(ev.property_data("v").counts.sum())
    # the same three statements on the CPU: one thread per core this process is granted, frames handed out dynamically (grain 1)
    logical, physical, quota = host_cpus()
    cores = max(1, min(physical, int(np.ceil(quota))) if quota else physical)
    ocell = O.make_cell(None)
    cells = [ocell] * F
    mass = topo.mass

    def cpu_once(nthreads):
        t = time.perf_counter()
        for nm, d in info.items():
            if d["kind"] == "rdf":
                O.rdf_run(coords, cells, d["ref"], d["target"], d["rmin"], d["rmax"], nthreads=nthreads)
            elif d["kind"] == "sdf":
                O.sdf_run(coords, cells, d["structures"], mass[d["structures"]], d["target"], d["cutoff"], nthreads=nthreads)
        return time.perf_counter() - t

    cpu_once(cores)
    m = 3 if dry else 9
    cpu_pool = sorted(cpu_once(cores) for _ in range(m))[m // 2]
    cpu_single = sorted(cpu_once(1) for _ in range(3))[1]
    work = int(lib.vmd_ir_work_per_frame(ir.h)) * F
    ev.close(); traj.close()
    return {"workload": f"SURVEY 8d C1 stand-in: {N}-atom capped deca-alanine chain, no cell, {F} frames, host-memory trajectory; hot-path statements of "
                        f"VIAMD's default script (src/main.cpp:528): d1 = distance, r = rdf(C, H, 10), v = sdf(7 residues, H, 10)",
            "script": C1_SCRIPT, "frames_per_step": F,
            "gpu_ms": {"pool_threads_16_grain_1": pooled * 1e3, "one_call": one_call * 1e3,
                       "note": "clear_data excluded, vmd_eval_frame_range[_pooled] over [0, F) incl. the 8.4 MB float view of v; median of 30"},
            "cpu_ms": {"cores": cores, "pool": cpu_pool * 1e3, "single_thread": cpu_single * 1e3,
                       "note": "oracle (scalar restatement; kind port) rdf + sdf over the same frames, dynamic grain 1; median of 9 / 3"},
            "gpu_over_cpu": cpu_pool / pooled, "rdf_hits": hits_r, "voxel_hits": vox_v,
            "work_pairs_times_frames": work,
            "shim_min_work_default": 1000000,
            "note": "work = vmd_ir_work_per_frame x frames, the quantity include/vmd_md_script_shim.h compares with vmd_shim_set_min_work: below the "
                    "threshold md_script_eval_create leaves the whole script with the evaluator behind the shim"}


def run_workload(name, args, ctx, steps, warmup, frames=None, with_cpu=False, opts=(), defer_views=False):
    """One workload end to end: synthesise the (rank's) trajectory in HBM, compile the script, W untimed + K timed steps
    bracketed by barrier + device sync, max over ranks.  Returns the result dict (rank 0: complete)."""
    import torch
    V, lib, script, synth, reduce_eval, dist = ctx["V"], ctx["lib"], ctx["script"], ctx["synth"], ctx["reduce_eval"], ctx["dist"]
    world, rank = ctx["world"], ctx["rank"]
    w = dict(WORKLOADS[name])
    F = frames or w["frames"]
    strong = args.scaling == "strong"
    lib.vmd_set_option(b"rdf_variant", args.variant)
    for kv in opts:
        k, v = kv.split("=")
        if lib.vmd_set_option(k.encode(), int(v)) == -1:
            raise SystemExit(f"--opt {k}: unknown option")
    if args.batch:
        lib.vmd_set_option(b"batch_frames", args.batch)

    # weak: every rank owns F frames of its own (own seed); strong: the F frames of ONE trajectory are block-sharded
    t0 = time.perf_counter()
    if strong:
        from viamd_amd.dist import shard_frames
        beg, end = shard_frames(F, rank, world)
        traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"], shard=(beg, end))
    else:
        beg, end = 0, F
        traj = synth.make_device_trajectory(V, w["seed"] + 1000 * rank, w["atoms"], w["box"], F, w["blob"])
    topo = synth.water_box_topology(w["atoms"], w["blob"])
    cell = V.make_unitcell(w["box"])
    if args.tilt:
        cell = V.make_unitcell(w["box"], tilt=tuple(float(t) for t in args.tilt.split(",")))
        traj.set_cell(cell, beg, end)
        if beg > 0:
            traj.set_cell(cell, 0, 1)           # a shard keeps frame 0 for the SDF reference pose: it lives in the same sheared cell
        w["desc"] += f", sheared cell (tilt {args.tilt})"
    if strong and world > 1 and args.traj not in ("device", "pinned"):
        # a file holds the WHOLE trajectory, a strong-scaling shard only this rank's block of it
        raise SystemExit("--scaling strong with N > 1 needs --traj device or pinned (a trajectory file is written from the whole device trajectory)")
    if args.traj == "pinned":                   # PCIe-inclusive variant: every batch is DMA'd from host memory
        dev_traj = traj
        traj = V.PinnedHostTrajectory(F, w["atoms"])
        traj.copy_from_device(dev_traj, beg, end)
        dev_traj.close()
    elif args.traj in ("dcd", "xtc", "trr", "xtc-resident"):    # end to end from a trajectory file (page cache after the first pass)
        import tempfile
        dev_traj = traj
        host = V.PinnedHostTrajectory(F, w["atoms"])
        host.copy_from_device(dev_traj)
        dev_traj.close()
        if args.rigid_water:
            # the synthetic box scatters the two H of a water up to 1 A per coordinate around the O (an "ideal gas of triplets"); real
            # water is a rigid triangle (0.9572 A, 104.52 deg), which is what XTC's run-length coding of small neighbours is made for:
            # same O positions, H atoms placed with that geometry in a random, seeded orientation per molecule and frame.  The RDF of
            # the O atoms - the timed script - is unchanged by it.
            nb_, nat = w["blob"], w["atoms"]
            nmol = (nat - nb_) // 3
            half = np.deg2rad(104.52 / 2)
            for f in range(F):
                rng = np.random.default_rng(7919 * (w["seed"] + 1000 * rank) + f)
                fr = host.frame(f)
                o = fr[:, nb_:nb_ + 3 * nmol:3].T.astype(np.float64)
                a = rng.normal(size=(nmol, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
                b = np.cross(a, rng.normal(size=(nmol, 3))); b /= np.linalg.norm(b, axis=1, keepdims=True)
                fr[:, nb_ + 1:nb_ + 3 * nmol:3] = (o + 0.9572 * (np.cos(half) * a + np.sin(half) * b)).T
                fr[:, nb_ + 2:nb_ + 3 * nmol:3] = (o + 0.9572 * (np.cos(half) * a - np.sin(half) * b)).T
            w["desc"] += ", rigid water geometry in the file"
        ext = "xtc" if args.traj == "xtc-resident" else args.traj
        path = os.path.join(tempfile.gettempdir(), f"viamd_amd_bench_{name}_{rank}.{ext}")
        {"dcd": V.write_dcd, "xtc": V.write_xtc, "trr": V.write_trr}[ext](path, host, cell)
        host.close()
        traj = V.DcdTrajectory(path) if args.traj == "dcd" else V.XdrTrajectory(path)
        if args.traj == "xtc" and args.ck_sidecar:
            # decoder checkpoints of an earlier process (vmd_ckcache_load): the first pass of THIS one decodes in sections.  The
            # bench writes the sidecar once (a throw-away pass over the file), drops the trajectory object and opens the file again.
            ckp = path + ".vmdck"
            ev0 = V.ScriptEval(F, script.compile_script(w["script"], topo)[0])
            assert ev0.frame_range(V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell), traj, beg, end)
            traj.save_checkpoints(ckp)
            ev0.close(); traj.close()
            traj = V.XdrTrajectory(path)
            sidecar_frames = traj.load_checkpoints(ckp, torch.cuda.current_device())
            w["desc"] += f", decoder checkpoints of {sidecar_frames} frames loaded from a sidecar file ({os.path.getsize(ckp) / F:.0f} bytes per frame)"
        if args.traj == "xtc-resident":          # the file is read and uploaded ONCE, still compressed; every step decodes from HBM
            t1 = time.perf_counter()
            traj = V.CompressedDeviceTrajectory(traj)
            resident_info = {"upload_s": time.perf_counter() - t1, "hbm_bytes": traj.device_bytes(),
                             "file_bytes_per_frame": os.path.getsize(path) / F}
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0

    ir, info = script.compile_script(w["script"], topo)
    ev = V.ScriptEval(F, ir)
    sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell)
    if world > 1 or defer_views:
        # a rank's partial volume is nobody's result: its float view (8.4 MB over PCIe per range) is derived once, from the merged counts,
        # by the finalize inside vmd_eval_reduce.  At N = 1 the view is part of the step, as VIAMD reads it.
        ev.defer_volume_views(True)

    merge_s = [0.0]

    def ranged():
        # --pool-threads N --grain G: the way VIAMD drives the boundary (src/main.cpp:993-997): N pool threads pull ranges of G frames
        # and all call frame_range on the SAME eval; every thread blocks until its frames are evaluated.  Native threads
        # (vmd_eval_frame_range_pooled): a Python thread per call costs more than a small call does
        return ev.frame_range_pooled(sysm, traj, beg, end, args.pool_threads, args.grain)

    def step():
        ev.clear_data()
        assert ranged() if args.pool_threads > 0 else ev.frame_range(sysm, traj, beg, end)
        t_m = time.perf_counter()
        reduce_eval(ev)                          # vmd_eval_reduce over RCCL: ONE merge of the accumulators per step (no-op at N = 1)
        merge_s[0] += time.perf_counter() - t_m

    first_step_s = None
    for k in range(warmup):
        t1 = time.perf_counter()
        step()
        if k == 0:
            torch.cuda.synchronize()
            first_step_s = time.perf_counter() - t1       # the first pass over the trajectory: cold caches, no decoder checkpoints yet
    lib.vmd_profile_reset()
    lib.vmd_profile_enable(True)
    lib.vmd_hip_rdf_columns(1)
    merge_s[0] = 0.0
    if dist:
        dist.barrier()
    lib.vmd_hip_marker(None)                 # an empty kernel: a counter collection of this command keeps what is dispatched after it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    lib.vmd_profile_enable(False)
    per_rank_ms = [elapsed / steps * 1e3]
    if dist:
        # every rank's own time travels in one SUM (its slot, zeros elsewhere); the step time of the job is the MAX over ranks
        t = torch.zeros(world, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) / steps * 1e3 for x in t.tolist()]
        elapsed = float(t.max().item())

    rdf_columns = int(lib.vmd_hip_rdf_columns(1))          # candidate columns this rank's pair kernel walked in the timed region
    # after the merge every rank holds the counts of all ranks' frames
    hits_per_step = sum(int(ev.property_data(n).counts.sum()) for n, d in info.items() if d["kind"] == "rdf")
    voxel_hits = sum(int(ev.property_data(n).counts.sum()) for n, d in info.items() if d["kind"] == "sdf")
    frames_per_step = F if strong else F * world          # whole job
    value = steps * frames_per_step / elapsed
    local_frames = end - beg

    kernel_ms, kernel_launches = {}, {}
    for k in ("rdf_pencil", "rdf_brute", "cells_build", "sdf_align", "sdf_scatter", "distance", "xtc_decode",
              "host_raw_upload", "host_raw_read", "host_raw_map", "host_refresh", "host_queue_to_sync", "host_settle", "host_fetch_stage", "host_sync_wait"):
        n = C.c_uint64(0)
        ms = lib.vmd_profile_ms(k.encode(), C.byref(n))
        if n.value:
            kernel_ms[k], kernel_launches[k] = ms, int(n.value)
    dom = w["kernel"]
    dispatches = max(kernel_launches.get(dom, 0), 1)
    nbt = C.c_uint64(0)
    lib.vmd_profile_ms(b"batches", C.byref(nbt))             # frame batches evaluated in the timed region
    nbatch = max(1, nbt.value)
    # A "launch" of the roofline = everything the dominant kernel does for ONE frame batch.  A single-pass script (c2, c3, c4) dispatches it
    # once per batch; a co-evaluated script (c5: 3 RDFs = several class-pair passes) dispatches it several times over the same frames, and
    # the frames' 12*N bytes are the algorithmic bytes of all of those dispatches TOGETHER (VERDICT r04 weak #4b): the time they are divided
    # by is the sum of the passes, not one of them.
    nl = nbatch
    t_launch = kernel_ms.get(dom, 0.0) / nl * 1e-3
    frames_per_launch = steps * local_frames / nbatch
    alg_bytes = 12.0 * w["atoms"] * frames_per_launch      # 12*N bytes per frame, x frames in one batch
    achieved = alg_bytes / t_launch / 1e9 if t_launch > 0 else 0.0
    step_gbs = 12.0 * w["atoms"] * local_frames * steps / elapsed / 1e9       # the same bytes against the whole timed region (this rank)

    traffic, traffic_src, valu, counters_current = None, None, None, None
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[name]
        k = pt["kernels"]["k_" + dom]
        # per frame, from the sums over every dispatch of the kernel in one step of the profiled run (scripts/pmc_traffic.py); the guide's
        # gfx950 correction inside: FETCH_SIZE counts half the bytes of a streaming read, WRITE_SIZE is 1:1 (both confirmed on known byte
        # counts in this code base: k_synth writes 12*N*F bytes -> WRITE_SIZE matches; k_sdf_scatter streams 12*N*F -> 2 x FETCH_SIZE matches)
        fps = pt.get("frames_per_step") or pt["frames_per_launch"]
        per_frame = (k.get("hbm_bytes_per_step_read_x2") or k["hbm_bytes_per_launch_read_x2"]) / fps
        traffic = per_frame * frames_per_launch
        kernel_s = kernel_ms.get(dom, 0.0) * 1e-3
        if "valu_insts_per_step" in k and kernel_s > 0:
            per_frame_insts = k["valu_insts_per_step"] / fps
            rate = per_frame_insts * steps * local_frames / kernel_s
            valu = {"insts_per_frame": per_frame_insts, "achieved": rate, "peak": VALU_PEAK_2CYCLE_WINST_S, "unit": "wave64 VALU instructions/s",
                    "frac": rate / VALU_PEAK_2CYCLE_WINST_S, "busy": k.get("valu_busy"),
                    "note": "frac: against the guide's 2 cycles per wave64 instruction per SIMD (only VGPR-only fp32 add / mul / fma issue that fast; "
                            "compares, lane prefixes, address arithmetic and SGPR-fed packed ops take ~4: profiles/r02_valu_calibration.txt). "
                            "busy: SQ_ACTIVE_INST_VALU x 4 / (128 SIMDs x GRBM_GUI_ACTIVE summed over the 8 XCDs) of the profiled run - the share of "
                            "SIMD cycles with the vector ALU at work, a ratio of two counters",
                    "source": f"SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE, profiles/pmc_traffic.json ({pt['source']})"}
        # the counters are replayed from a committed collection, not measured by this run: say whether the kernels have changed since
        import hashlib
        ksha = hashlib.sha256(open(os.path.join(ROOT, "viamd_amd", "csrc", "vmd_kernels.hip"), "rb").read()).hexdigest()[:16]
        counters_current = (pt.get("kernels_sha256_16") == ksha) if pt.get("kernels_sha256_16") else None
        traffic_src = (f"replayed, not measured in this run: profiles/pmc_traffic.json ({pt['source']}; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                       f"passes of this command), (2 x FETCH + WRITE) per frame x frames_per_launch; valid while traffic_counters_match_kernel_source")
    except Exception:
        pass
    # VIAMD_BENCH_SHARE_GPU=1 runs every rank on device 0 and merges through a host-staged gloo collective (the multi-process
    # path on a 1-GPU box): that is ONE GPU, and the line says so - n_gpus counts devices, `ranks` processes, the collective is named
    share = os.environ.get("VIAMD_BENCH_SHARE_GPU") == "1"
    n_devices = 1 if share else world
    if world == 1:
        parallelism = "one process, one GPU: no merge"
    elif share:
        parallelism = (f"{world} ranks SHARING one GPU ({args.scaling} scaling), frames block-sharded over the ranks, one vmd_eval_reduce per step through a "
                       f"host-staged gloo all-reduce - not a multi-GPU measurement")
    else:
        parallelism = (f"frames block-sharded x{world} ({args.scaling} scaling), one vmd_eval_reduce (RCCL all-reduce in place on the device "
                       f"accumulators) per step")
    out = {
        "metric": "trajectory frames/s for RDF+SDF eval (BASELINE.json metric; atom-pairs/s in pairs_per_s)",
        "value": value, "unit": "frames/s", "n_gpus": n_devices, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if os.environ.get("VIAMD_BENCH_DRYRUN") != "1" else "synthetic - DRY RUN on the CPU emulator build with tiny workloads: no timing claim",
        "per_rank_ms_per_step": per_rank_ms,
        "config": {"workload": w["desc"], "name": name, "script": w["script"], "atoms": w["atoms"],
                   "frames_per_step": frames_per_step, "frames_per_step_per_gpu": local_frames,
                   "parallelism": parallelism, "rdf_variant": args.variant,
                   "trajectory": {"device": "resident in HBM", "pinned": "pinned host memory, DMA per batch (PCIe-inclusive)",
                                  "dcd": "DCD file, native reader -> pinned staging -> DMA (file- and PCIe-inclusive)",
                                  "xtc": "GROMACS XTC file (compressed, 0.01 A grid): native decoder on host threads -> pinned staging -> DMA, or "
                                         "(xtc_device_decode) compressed bytes -> pinned staging -> DMA -> k_xtc_wave; file- and PCIe-inclusive",
                                  "xtc-resident": "GROMACS XTC file uploaded once, kept COMPRESSED in HBM (vmd_rawtraj); every step decompresses its "
                                                  "batches on the device (k_xtc_wave) - no host work, no PCIe per step",
                                  "trr": "GROMACS TRR file, native reader -> pinned staging -> DMA"}[args.traj]},
        "pairs_per_s": hits_per_step * steps / elapsed,
        # measured, not modelled: one column = one target atom against the 64 reference atoms of a chunk.  Per ORDERED
        # pair of the histograms (a same-set pass counts an unordered pair once and adds 2: halve the hits for lanes per unordered hit)
        "pair_kernel_columns": {"columns_per_step": rdf_columns / steps if steps else 0,
                                "candidate_lanes_per_ordered_pair": (64.0 * rdf_columns / steps) / (hits_per_step * local_frames / frames_per_step) if hits_per_step else None},
        "voxel_hits_per_s": voxel_hits * steps / elapsed,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_counters_match_kernel_source": counters_current,     # None: the collection predates the marker
                     "kernel": "k_" + dom, "avg_launch_ms": t_launch * 1e3, "launches": nl, "dispatches": dispatches,
                     "algorithmic_bytes_per_launch": alg_bytes, "frames_per_launch": frames_per_launch, "valu": valu,
                     "step_level": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                                    "note": "12*N bytes x frames of this rank / wall time of the timed region (all kernels, launches, host work)"},
                     "note": ("k_rdf_pencil is VALU-issue bound, not HBM bound (DESIGN.md 3.1/5): achieved is the brief's "
                              "12*N*frames/launch-time figure" if dom == "rdf_pencil" else "HBM stream kernel")},
        "kernel_ms": dict(kernel_ms, timed_region=elapsed * 1e3),
        "kernel_launches_per_step": {k_: v_ / steps for k_, v_ in kernel_launches.items() if not k_.startswith("host_")},
        "synth_s": gen_s,
    }
    # what ONE vmd_eval_reduce of this script moves per rank: the integer accumulators on the device (volumes as u32 where the
    # merged counts provably fit: count_bound x 8 ranks < 2^32) + everything host-side packed as fp64 (weights, temporal rows, the frame mask)
    try:
        dev_b = host_b = 0
        for v in ev.accum_views():
            narrow = v.num_counts >= 1 << 20 and v.count_bound and v.count_bound * 8 < 1 << 32
            dev_b += int(v.num_counts) * (4 if narrow else 8)
            host_b += 8 * (int(v.num_weights) + int(v.num_temporal))
        host_b += 8 * F
        out["merge_payload_bytes"] = {"device_counts": dev_b, "host_packed_f64": host_b, "total": dev_b + host_b}
    except Exception:
        pass
    if "cells_build" in kernel_ms:
        # the sorted copies behind the pair kernel: share of the step and (where a PMC pass exists) HBM-side traffic against
        # 12*N (the frame, read once) + 12*N_sel (the sorted rows, written once) per frame
        cb = {"ms_per_step": kernel_ms["cells_build"] / steps, "frac_of_step": kernel_ms["cells_build"] / (elapsed * 1e3)}
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[name]
            # every dispatch of every k_cells_* kernel in one step of the profiled run (a co-evaluated script sorts several selections per batch)
            per_frame = sum(k.get("hbm_bytes_per_step_read_x2") or k["hbm_bytes_per_launch_read_x2"] for kn, k in pt["kernels"].items()
                            if kn.startswith("k_cells")) / (pt.get("frames_per_step") or pt["frames_per_launch"])
            # sorted rows written once per frame: co-evaluated RDFs are split into DISJOINT atom classes, each sorted once, so the
            # selected atoms are the union of the sets; a lone RDF over two different sets sorts both
            rdfs = [d for d in info.values() if d["kind"] == "rdf"]
            if len(rdfs) > 1:
                nsel = len(np.unique(np.concatenate([np.asarray(a) for d in rdfs for a in (d["ref"], d["target"])])))
            else:
                sel = {}
                for d in rdfs:
                    for arr in (d["ref"], d["target"]):
                        sel[(len(arr), int(arr[0]), int(arr[-1]))] = len(arr)
                nsel = sum(sel.values())
            alg = 12.0 * w["atoms"] + 12.0 * nsel
            # the build's floor is 1.5 x (the records between its two levels are written once and read once: DESIGN 3.0); what the traffic above
            # that floor costs the step, at the rate the build moves its bytes
            floor = 1.5 * alg
            cb.update({"traffic_bytes_per_frame": per_frame, "algorithmic_bytes_per_frame": alg, "traffic_ratio": per_frame / alg,
                       "floor_ratio": 1.5, "excess_over_floor_frac_of_step": max(0.0, per_frame - floor) / per_frame * cb["frac_of_step"],
                       "traffic_source": f"profiles/pmc_traffic.json ({pt['source']}), kernels k_cells_*"})
        except Exception:
            pass
        try:
            ov, off = ev.cell_build_stats()
            cb.update({"launches_per_step": kernel_launches.get("cells_build", 0) / steps, "bucket_overflows_since_creation": ov, "selections_off_buckets": off})
        except Exception:
            pass
        out["cell_build"] = cb
    if share:
        out["ranks"], out["shared_gpu"] = world, True
    if dist:
        # what carried the merge, how many ranks IT saw (not WORLD_SIZE: the driver can check rccl_ranks == n_gpus), what it moved
        from viamd_amd.dist import collective_info, reduce_stats
        ci = collective_info(lib) or {}
        out["merge"] = dict(reduce_stats(ev), collective=ci.get("kind"), rccl_ranks=ci.get("ranks"), fallback_reason=ci.get("fallback_reason"),
                            host_ms_per_step=merge_s[0] / steps * 1e3,
                            note="one vmd_eval_reduce per step: all all-reduces of a merge inside one ncclGroupStart / End; device_ms = hipEvents "
                                 "around the staged collectives of the LAST step, host_ms_per_step = wall time of the call (pack, merge, "
                                 "unpack, finalize) averaged over the timed steps, this rank")
    if args.pool_threads > 0:
        out["config"]["call_pattern"] = {"pool_threads": args.pool_threads, "grain": args.grain,
                                         "note": "frame_range called from pool threads with ranges of `grain` frames on one eval (VIAMD: src/main.cpp:993-997); "
                                                 "the metric's configuration is one call per step"}
    if args.traj in ("xtc", "xtc-resident"):
        # > 0 only with --opt xtc_device_decode=N or a compressed-resident trajectory: decompressed by the k_xtc_* kernels
        out["config"]["frames_decompressed_on_device_per_step"] = ev.frames_device_decoded()
        if first_step_s:
            # the timed steps re-evaluate a trajectory whose frames have been decoded before (VIAMD: every script edit): the decoder
            # checkpoints of the first pass exist.  The first pass itself walks every bit stream from its start:
            out["config"]["first_pass"] = {"frames_per_s": local_frames / first_step_s, "ms": first_step_s * 1e3,
                                           "note": "warm-up step 0 of this run: no decoder checkpoints yet (and, for a file, a cold page cache)"}
        if args.traj == "xtc-resident":
            out["config"]["compressed_resident"] = resident_info
    ev.close()
    traj.close() if hasattr(traj, "close") else None
    if with_cpu and rank == 0:
        out["cpu_baseline"] = cpu_baseline(name, w, topo, info)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        en = (out["cpu_baseline"].get("extrapolated_to_node") or {}).get("value")
        out["gpu_over_cpu_node"] = value / en if en else None        # against every physical core of the node (extrapolated)
    return out


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher around it: spawn the N ranks through torch.distributed.run on a free local
    port (rendezvous on 127.0.0.1), same arguments, and return its exit code.  Rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dryrun_setup():
    """VIAMD_BENCH_DRYRUN=1 (tests/test_bench_dryrun.py): the same file, the same command line, on a machine without a GPU - the
    SIMT-emulator build of the library (VIAMD_AMD_LIB), gloo instead of RCCL, and workloads of a few frames so that the emulator
    finishes in seconds.  The line it prints says so in `data`; it carries no timing claim."""
    import torch
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    tiny = dict(atoms=1500, blob=0, box=40.0, frames=6, seed=2, steps=1, sec_steps=1, kernel="rdf_pencil",
                script="g = rdf(element('O'), element('O'), 12.0);", desc="DRY RUN: tiny rdf")
    tiny4 = dict(atoms=1501, blob=100, box=40.0, frames=6, seed=4, steps=1, sec_steps=1, kernel="sdf_scatter",
                 script="s = residue(5:8); v = sdf(s, element('O') and water, 10.0); d = distance(residue(1), residue(3));",
                 desc="DRY RUN: tiny sdf")
    tiny5 = dict(tiny4, kernel="rdf_pencil", script=tiny["script"] + tiny4["script"], desc="DRY RUN: tiny rdf + sdf + distance")
    WORKLOADS.update(c3=dict(tiny), c2=dict(tiny), c3d=dict(tiny), c4=tiny4, c5=tiny5)
    os.environ.setdefault("VIAMD_BENCH_BACKEND", "gloo")
    globals()["cpu_baseline"] = lambda *a, **k: {"value": 1.0, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "stub (dry run)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (weak) / in total (strong); default: per workload")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns `frames` frames of its own; strong: the workload's frames are block-sharded over the ranks")
    ap.add_argument("--variant", type=int, default=0, help="pair-kernel hit compaction: 0 = per column (default), 1 = bin in place, 2 = pair entries, 3 = 0 behind a bounding-box test of the j windows (A/B)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short c2 / c4 / c5 runs of the default N = 1 line")
    ap.add_argument("--opt", action="append", default=[], help="library tuning knob key=value (vmd_set_option)")
    ap.add_argument("--pool-threads", type=int, default=0, help="evaluate a step the way VIAMD calls the boundary: this many threads pull ranges of --grain frames and call frame_range on ONE eval (0 = one call per step, the metric's configuration)")
    ap.add_argument("--grain", type=int, default=1, help="frames per frame_range call with --pool-threads (VIAMD's create_pool_task default: 1)")
    ap.add_argument("--ck-sidecar", action="store_true", help="--traj xtc: the decoder checkpoints come from a sidecar file written by an earlier pass (vmd_ckcache_save / _load); 'first_pass' is then a first pass WITH checkpoints")
    ap.add_argument("--rigid-water", action="store_true", help="file trajectories: give the waters their real geometry before writing the file")
    ap.add_argument("--tilt", default=None, help="xy,xz,yz in Angstrom: evaluate in a sheared (triclinic) cell of the same volume")
    ap.add_argument("--traj", default="device", choices=["device", "pinned", "dcd", "xtc", "trr", "xtc-resident"],
                    help="device: frames resident in HBM (the metric); pinned: frames in pinned host memory, PCIe-inclusive; "
                         "dcd / xtc / trr: frames decoded from a trajectory file on disk by the native readers "
                         "(file -> decode on host threads -> pinned staging -> PCIe); xtc is lossy (0.01 A grid)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the bare command `python bench.py --gpus N ...`: this process becomes the launcher of N ranks (one per GPU) and relays
        # their single JSON line - the way VIAMD's one call fans out over all workers and completes once (src/main.cpp:993-1008)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    dry = os.environ.get("VIAMD_BENCH_DRYRUN") == "1"
    if dry:
        dryrun_setup()
    share = os.environ.get("VIAMD_BENCH_SHARE_GPU") == "1"        # all ranks on device 0, merge staged through the host (gloo): the
    if share:                                                      # multi-process path on a 1-GPU box (RCCL refuses two ranks per GPU)
        local_rank = 0
        os.environ["VIAMD_BENCH_BACKEND"] = "gloo"
        os.environ["VIAMD_AMD_STAGED_COLLECTIVE"] = "1"

    import torch
    import viamd_amd as V
    from viamd_amd import script, synth
    from viamd_amd.dist import close_comms, reduce_eval
    lib = V.default_lib()                       # hipcc-built library or ImportError: there is no fallback
    if lib.vmd_device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    lib.vmd_set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # VIAMD_BENCH_BACKEND=gloo: the CPU dry run of this file's N > 1 logic on the emulator build (tests/test_bench_dryrun.py)
        backend = os.environ.get("VIAMD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = dict(V=V, lib=lib, script=script, synth=synth, reduce_eval=reduce_eval, dist=dist, world=world, rank=rank)

    w = WORKLOADS[args.workload]
    steps = args.steps if args.steps is not None else w["steps"]
    warmup = args.warmup if args.warmup is not None else 2
    out = run_workload(args.workload, args, ctx, steps, warmup, frames=args.frames, with_cpu=(world == 1 and not args.no_cpu_baseline),
                       opts=args.opt)
    # the other single-GPU configurations of BASELINE.json, short runs in the same line (N = 1, default invocation only)
    if world == 1 and not args.no_secondary and args.workload == "c3" and args.traj == "device" and not args.frames:
        sec = {}
        for nm in ("c2", "c4", "c5", "c3d"):       # c3d: SURVEY 8d's C3-dense variant (every atom counted as heavy), 200 frames
            r = run_workload(nm, args, ctx, WORKLOADS[nm].get("sec_steps", 2), 1, opts=args.opt)
            rf = r["roofline"]
            sec[nm] = {"workload": r["config"]["workload"], "value": r["value"], "unit": "frames/s", "steps": r["steps"], "warmup": r["warmup"],
                       "ms_per_step": r["ms_per_step"], "frames_per_step": r["config"]["frames_per_step"],
                       "pairs_per_s": r["pairs_per_s"], "voxel_hits_per_s": r["voxel_hits_per_s"], "kernel_ms": r["kernel_ms"],
                       "kernel_launches_per_step": r.get("kernel_launches_per_step"), "cell_build": r.get("cell_build"),
                       "roofline": {"bound": "hbm", "kernel": rf["kernel"], "avg_launch_ms": rf["avg_launch_ms"], "launches": rf["launches"], "dispatches": rf["dispatches"],
                                    "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                                    "frames_per_launch": rf["frames_per_launch"], "traffic": rf["traffic"],
                                    "step_level": rf["step_level"], "valu": rf["valu"]}}
        # configs[3] is quoted on 8 GPUs, strong scaling: 10 000 frames / 8 = 1 250 per rank.  One GPU can measure what bounds that curve - the
        # step over a rank's share against the step over the whole trajectory (kernels shrink 8x, the per-step fixed part does not; the merge,
        # which only N > 1 has, comes on top)
        n1250, w1250 = (3, 1) if dry else (100, 10)
        r = run_workload("c4", args, ctx, n1250, w1250, frames=1250, opts=args.opt)   # 100 steps of ~0.6 ms: a 10-step sample moved by 20 % from call to call
        # ... and what a RANK does before the merge: the same steps with the volume's float view deferred (vmd_eval_defer_volume_views: no zeroing
        # of the view, no 8.4 MB over PCIe per step) - the merge and the one view of the merged counts come on top at N > 1
        rank_part = run_workload("c4", args, ctx, n1250, w1250, frames=1250, opts=args.opt, defer_views=True)
        r_c4_payload = r.get("merge_payload_bytes")
        sec["c4_1250"] = {"workload": r["config"]["workload"] + " - ONE rank's share at 8 GPUs (1 250 frames)", "value": r["value"], "unit": "frames/s",
                          "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"], "frames_per_step": r["config"]["frames_per_step"],
                          "kernel_ms": r["kernel_ms"],
                          "strong_scaling_bound_8_gpus": sec["c4"]["ms_per_step"] / r["ms_per_step"],
                          "rank_part_ms": rank_part["ms_per_step"],
                          "note": "ms_per_step(10 000 frames) / ms_per_step(1 250 frames) on one GPU: the upper bound of configs[3]'s 8-GPU strong scaling "
                                  "(the float view of the volume, 8.4 MB over PCIe, is inside both steps; at N > 1 a rank defers it to the merge: rank_part_ms is that "
                                  "step without the view - clear_data + frame_range only, NOT a complete evaluation)"}
        # ... and the same one-GPU bound for the other two configurations BASELINE.json quotes on 8 GPUs (configs[2] = the metric's workload,
        # configs[4]): a rank's share of their 1 000 frames is 125
        for nm in ("c3", "c5"):
            whole = out if nm == "c3" else sec[nm]
            n125, w125 = (2, 1) if dry else ((20, 3) if nm == "c3" else (8, 2))
            r = run_workload(nm, args, ctx, n125, w125, frames=125, opts=args.opt)
            rp = run_workload(nm, args, ctx, n125, w125, frames=125, opts=args.opt, defer_views=True) if nm == "c5" else r
            sec[nm + "_125"] = {"workload": r["config"]["workload"] + " - ONE rank's share at 8 GPUs (125 frames)", "value": r["value"], "unit": "frames/s",
                                "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"], "frames_per_step": r["config"]["frames_per_step"],
                                "kernel_ms": r["kernel_ms"], "strong_scaling_bound_8_gpus": whole["ms_per_step"] / r["ms_per_step"],
                                "rank_part_ms": rp["ms_per_step"], "merge_payload_bytes": r.get("merge_payload_bytes"),
                                "note": "ms_per_step(1 000 frames) / ms_per_step(125 frames) on one GPU: the upper bound of this configuration's 8-GPU STRONG "
                                        "scaling (the weak-scaling curve of the metric keeps 1 000 frames per rank and does not see this bound)"}
        sec["c4_1250"]["merge_payload_bytes"] = r_c4_payload
        # SURVEY 8d's C1 stand-in - the size class of VIAMD's default dataset (datasets/1ALA-500.pdb: 112 atoms, 500 frames) with the literal
        # default script's hot-path statements (src/main.cpp:528) - timed the way VIAMD drives the boundary (16 pool threads, grain 1) AND on the
        # same host cores through the oracle: what include/vmd_md_script_shim.h's work threshold (vmd_shim_set_min_work) is set from
        sec["c1"] = run_c1(ctx, dry)
        out["secondary"] = sec
    if world > 1 and not args.no_secondary and args.workload == "c3" and args.traj == "device" and not args.frames and args.scaling == "weak":
        # the two configurations BASELINE.json quotes on 8 GPUs, STRONG scaling (the named trajectory block-sharded over the ranks,
        # one merge per step), short runs next to the weak-scaling c3 curve.  Every rank takes the same decisions here (a failure is
        # caught on all ranks alike or the job dies as one), so the primary line is printed in any case.
        import copy
        sec = {}
        for nm in ("c4", "c5"):
            a2 = copy.copy(args)
            a2.scaling = "strong"
            try:
                r = run_workload(nm, a2, ctx, WORKLOADS[nm]["sec_steps"], 1, opts=args.opt)
                sec[nm + "_strong"] = {"workload": r["config"]["workload"], "scaling": "strong", "value": r["value"], "unit": "frames/s",
                                       "n_gpus": world, "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"],
                                       "per_rank_ms_per_step": r["per_rank_ms_per_step"], "frames_per_step": r["config"]["frames_per_step"],
                                       "frames_per_step_per_gpu": r["config"]["frames_per_step_per_gpu"], "pairs_per_s": r["pairs_per_s"],
                                       "voxel_hits_per_s": r["voxel_hits_per_s"], "merge": r.get("merge")}
            except Exception as e:          # noqa: BLE001
                sec[nm + "_strong"] = {"error": repr(e)}
        out["secondary"] = sec
    if rank == 0:
        # no fraction of a peak may exceed 1 anywhere in the line: checked here, reported in the line itself
        bad = []

        def walk(node, path):
            if isinstance(node, dict):
                for k_, v_ in node.items():
                    if isinstance(v_, (int, float)) and not isinstance(v_, bool) and (k_ == "busy" or k_.startswith("frac")) and not (0.0 <= v_ <= 1.0):
                        bad.append(f"{path}{k_}={v_:.4g}")
                    walk(v_, f"{path}{k_}.")
            elif isinstance(node, list):
                for i_, v_ in enumerate(node):
                    walk(v_, f"{path}{i_}.")
        walk(out, "")
        out["fractions_within_0_1"] = not bad
        if bad:
            print("bench.py: fraction outside [0, 1]: " + ", ".join(bad), file=sys.stderr)
        print(json.dumps(out))
    if dist:
        close_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
