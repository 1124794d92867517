#!/usr/bin/env python
"""bench.py — throughput of the RDF / SDF / distance hot path on MI355X, one JSON line (contract: the task brief).

A *step* is one pass of the hot path over one batch of synthetic input: a full evaluation (clear_data ->
frame_range over the whole resident trajectory -> merge) of the workload's script.  Inputs are resident in HBM
before the timed region (trajectories are pre-staged).  N > 1: one process per GPU
(torch.distributed / RCCL), every rank owns its own block of frames (weak scaling), one all-reduce per step
merges the integer accumulators.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--frames F]

Default (no flags): N = 1, workload c2 = BASELINE.json configs[1] (100k-atom box, 1k frames, O-O RDF r_cut 12 A).
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
VALU_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4   # 1024 SIMDs, one wave64 VALU instruction per 4 cycles (measured)

WORKLOADS = {
    # SURVEY.md 8d.  frames = frames resident per GPU and evaluated per step.
    "c2": dict(atoms=100002, blob=0, box=100.0, frames=1000, seed=2, steps=20, kernel="rdf_pencil",
               script="g = rdf(element('O'), element('O'), 12.0);",
               desc="BASELINE configs[1]: synthetic 100002-atom periodic water box, 1000 frames, O-O RDF r_cut=12 A, 1024 bins"),
    "c3": dict(atoms=1000002, blob=0, box=215.443, frames=1000, seed=3, steps=3, kernel="rdf_pencil",
               script="g = rdf(not element('H'), not element('H'), 12.0);",
               desc="BASELINE configs[2]: synthetic 1000002-atom box, 1000 frames, all-heavy-atom RDF r_cut=12 A, 1024 bins"),
    "c4": dict(atoms=100001, blob=2000, box=100.0, frames=2000, seed=4, steps=5, kernel="sdf_scatter",
               script="s = residue(5:11); v = sdf(s, element('O') and water, 10.0);",
               desc="BASELINE configs[3]: 100001-atom solvated protein-like blob, SDF 128^3 around 7 residues + reference-frame tracking"),
    "c5": dict(atoms=1001999, blob=2000, box=215.443, frames=500, seed=5, steps=3, kernel="rdf_pencil",
               script=("goo = rdf(element('O') and water, element('O') and water, 12.0);"
                       "goh = rdf(element('O') and water, element('H') and water, 12.0);"
                       "ghv = rdf(not element('H'), not element('H'), 12.0);"
                       "s = residue(5:11); v = sdf(s, element('O') and water, 10.0);"
                       "d1 = distance(1, 1990); d2 = distance(residue(1), residue(200));"
                       "d3 = distance_min(residue(3), residue(150)); d4 = distance_max(residue(10), residue(20));"),
               desc="BASELINE configs[4]: 1M-atom box + blob, 3x RDF + 1x SDF + 4x distance co-evaluated per frame"),
}


def cpu_baseline(name, w, topo, info):
    """The oracle (a port — ext/mdlib is empty, so mdlib itself cannot be timed) driven like VIAMD drives mdlib:
    all host cores, frames handed out dynamically with grain 1, on a bounded sample of the same workload."""
    from oracle import oracle as O
    from viamd_amd import synth
    cores = os.cpu_count() or 1
    cell = O.make_cell(w["box"])
    budget = 15.0                       # seconds of wall time aimed at
    f0 = synth.host_frames(O, w["seed"], w["atoms"], w["box"], 1, w["blob"])
    mass = topo.mass

    def run(traj, nthreads):
        hits = 0
        for nm, d in info.items():
            if d["kind"] == "rdf":
                hits += O.rdf_run(traj, [cell] * len(traj), d["ref"], d["target"], d["rmin"], d["rmax"], nthreads=nthreads)[2]
            elif d["kind"] == "sdf":
                O.sdf_run(traj, [cell] * len(traj), d["structures"], mass[d["structures"]], d["target"], d["cutoff"], nthreads=nthreads)
            # the distance family is O(|a||b|) on a handful of atoms: below timer resolution next to rdf/sdf
        return hits

    t = time.perf_counter()
    run(f0, 1)
    t1 = time.perf_counter() - t
    nfr = int(max(cores, min(1000, budget * cores / max(t1, 1e-3))))
    nfr = max(1, min(nfr, int(6e9 // (12 * w["atoms"]))))          # keep the host copy below ~6 GB
    traj = synth.host_frames(O, w["seed"], w["atoms"], w["box"], nfr, w["blob"])
    t = time.perf_counter()
    hits = run(traj, cores)
    dt = time.perf_counter() - t
    return {"value": nfr / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{nfr} frames of {name} ({w['atoms']} atoms), oracle (cell-list RDF / SDF align+scatter), "
                      f"OpenMP dynamic grain 1 over frames, {dt:.1f} s",
            "pairs_per_s": hits / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames resident per GPU (default: per workload)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch
    import viamd_amd as V
    from viamd_amd import script, synth
    from viamd_amd.dist import reduce_eval
    lib = V.default_lib()                       # hipcc-built library or ImportError: there is no fallback
    if lib.vmd_device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    lib.vmd_set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = dict(WORKLOADS[args.workload])
    frames = args.frames or w["frames"]
    steps = args.steps if args.steps is not None else w["steps"]
    warmup = args.warmup if args.warmup is not None else 2
    lib.vmd_set_option(b"rdf_variant", args.variant)
    if args.batch:
        lib.vmd_set_option(b"batch_frames", args.batch)

    # synthetic trajectory of this rank, generated in HBM; every rank gets its own seed -> its own block of frames
    t0 = time.perf_counter()
    traj = synth.make_device_trajectory(V, w["seed"] + 1000 * rank, w["atoms"], w["box"], frames, w["blob"])
    topo = synth.water_box_topology(w["atoms"], w["blob"])
    cell = V.make_unitcell(w["box"])
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0

    ir, info = script.compile_script(w["script"], topo)
    ev = V.ScriptEval(frames, ir)
    sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell)

    def step():
        ev.clear_data()
        assert ev.frame_range(sysm, traj, 0, frames)
        reduce_eval(ev)                          # RCCL all-reduce of the integer accumulators (no-op at N = 1)

    for _ in range(warmup):
        step()
    lib.vmd_profile_reset()
    lib.vmd_profile_enable(True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    lib.vmd_profile_enable(False)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hits_per_step = sum(int(ev.property_data(n).counts.sum()) for n, d in info.items() if d["kind"] == "rdf")
    voxel_hits = sum(int(ev.property_data(n).counts.sum()) for n, d in info.items() if d["kind"] == "sdf")
    total_frames = steps * frames * world
    value = total_frames / elapsed

    kernel_ms, kernel_launches = {}, {}
    for k in ("rdf_pencil", "rdf_brute", "cells_build", "sdf_align", "sdf_scatter", "distance"):
        n = C.c_uint64(0)
        ms = lib.vmd_profile_ms(k.encode(), C.byref(n))
        if n.value:
            kernel_ms[k], kernel_launches[k] = ms, int(n.value)
    dom = w["kernel"]
    nl = max(kernel_launches.get(dom, 0), 1)
    t_launch = kernel_ms.get(dom, 0.0) / nl * 1e-3
    n_rdf = max(1, sum(1 for d in info.values() if d["kind"] == "rdf")) if dom == "rdf_pencil" else 1
    frames_per_launch = steps * frames * n_rdf / nl      # one pencil launch handles one RDF property of one frame batch
    alg_bytes = 12.0 * w["atoms"] * frames_per_launch      # 12*N bytes per frame, x frames in one launch
    achieved = alg_bytes / t_launch / 1e9 if t_launch > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "trajectory frames/s for RDF+SDF eval (BASELINE.json metric; atom-pairs/s in pairs_per_s)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"], "script": w["script"], "atoms": w["atoms"], "frames_per_step_per_gpu": frames,
                       "parallelism": f"frames sharded x{world}, one RCCL all-reduce per step", "rdf_variant": args.variant},
            "pairs_per_s": hits_per_step * steps / elapsed,
            "voxel_hits_per_s": voxel_hits * steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_" + dom, "avg_launch_ms": t_launch * 1e3, "launches": nl,
                         "algorithmic_bytes_per_launch": alg_bytes, "frames_per_launch": frames_per_launch,
                         "note": ("k_rdf_pencil is VALU-issue bound, not HBM bound (DESIGN.md 3.1/5): achieved is the brief's "
                                  "12*N*frames/launch-time figure" if dom == "rdf_pencil" else "HBM stream kernel")},
            "kernel_ms": dict(kernel_ms, timed_region=elapsed * 1e3),
            "synth_s": gen_s,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, w, topo, info)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
