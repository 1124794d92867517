#!/usr/bin/env python
"""bench.py — throughput of the RDF/SDF hot path on MI355X, one JSON line (contract: see the task brief).

A *step* is one pass of the hot path over one batch of synthetic input: a full evaluation (clear_data ->
frame_range over the whole resident trajectory -> merge) of the workload's script.  Inputs are resident in HBM
before the timed region (trajectories are pre-staged).  N > 1: one process per GPU
(torch.distributed / RCCL), every rank owns its own block of frames (weak scaling), one all-reduce per step
merges the integer accumulators.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--frames F]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz: fp32 lane-instructions per second

WORKLOADS = {
    # name: (atoms, box, frames, seed, description)   — SURVEY.md 8d
    "c2": (100002, 100.0, 1000, 2, "BASELINE configs[1]: synthetic 100002-atom periodic water box, 1000 frames, O-O RDF r_cut=12 A, 1024 bins"),
    "c3": (1000002, 215.443, 1000, 3, "BASELINE configs[2]: synthetic 1000002-atom box, 1000 frames, heavy-atom (O) RDF r_cut=12 A, 1024 bins"),
}


def build_script(V, name, n_atoms):
    ir = V.ScriptIR()
    o = np.arange(0, n_atoms, 3, dtype=np.int32)
    if name in ("c2", "c3"):
        ir.add_rdf("g", o, o, 12.0)      # `g = rdf(element('O'), element('O'), 12.0);`  (heavy == O in the O,H,H box)
        return ir, {"g": (o, o)}
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(name, n_atoms, box, seed, pairs):
    """The oracle (a port, not mdlib: ext/mdlib is empty) driven like VIAMD drives mdlib: all host cores, frames
    handed out dynamically, on a bounded sample of the same workload."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    ref, tgt = pairs["g"]
    cell = O.make_cell(box)
    f0 = O.synth_frame(seed, n_atoms, box, 0.05, 0)
    t = time.perf_counter()
    O.rdf_run(f0[None], [cell], ref, tgt, 0.0, 12.0, nthreads=1)
    t1 = time.perf_counter() - t
    nfr = int(max(cores, min(1000, 15.0 * cores / max(t1, 1e-3))))
    nfr = min(nfr, int(6e9 // (12 * n_atoms)))          # keep the host copy below ~6 GB
    traj = np.stack([O.synth_frame(seed, n_atoms, box, 0.05, f) for f in range(nfr)])
    t = time.perf_counter()
    _, _, hits = O.rdf_run(traj, [cell] * nfr, ref, tgt, 0.0, 12.0, nthreads=cores)
    dt = time.perf_counter() - t
    return {"value": nfr / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{nfr} frames of {name} ({n_atoms} atoms), oracle cell-list RDF, OpenMP dynamic grain 1, {dt:.1f} s",
            "pairs_per_s": hits / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--frames", type=int, default=None, help="frames resident per GPU (default: the config's 1000)")
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
    from viamd_amd.dist import reduce_eval
    lib = V.default_lib()                       # hipcc-built library or ImportError: no fallback
    if lib.vmd_device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    lib.vmd_set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_atoms, box, frames, seed, desc = WORKLOADS[args.workload]
    frames = args.frames or frames
    steps = args.steps if args.steps is not None else (20 if args.workload == "c2" else 5)
    warmup = args.warmup if args.warmup is not None else 2
    lib.vmd_set_option(b"rdf_variant", args.variant)
    if args.batch:
        lib.vmd_set_option(b"batch_frames", args.batch)

    # synthetic trajectory of this rank, generated on the device by the counter-based generator (oracle S9 twin);
    # every rank gets its own seed -> its own block of frames (weak scaling)
    import ctypes as C
    t0 = time.perf_counter()
    traj = V.DeviceTrajectory(frames, n_atoms)
    traj.synth(seed + 1000 * rank, box, 0.05)
    cell = V.make_unitcell(box)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0

    ir, pairs = build_script(V, args.workload, n_atoms)
    ev = V.ScriptEval(frames, ir)
    sysm = V.MolSystem(n_atoms, unitcell=cell)

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

    pd = ev.property_data("g")
    hits_per_step = int(pd.counts.sum())          # after the merge: ordered pairs of all ranks' frames
    total_frames = steps * frames * world
    value = total_frames / elapsed

    launches = C.c_uint64(0)
    ms = lib.vmd_profile_ms(b"rdf_pencil", C.byref(launches))
    nl = max(int(launches.value), 1)
    frames_per_launch = steps * frames / nl
    t_launch = ms / nl * 1e-3
    n_sel = len(pairs["g"][0])
    alg_bytes = 12.0 * n_atoms * frames_per_launch          # SURVEY 8d per-frame figure x frames in one launch
    achieved = alg_bytes / t_launch / 1e9 if t_launch > 0 else 0.0
    cand_ms = lib.vmd_profile_ms(b"cells_build", None)

    if rank == 0:
        out = {
            "metric": "trajectory frames/s, RDF eval (BASELINE.json: trajectory frames/s and atom-pairs/s for RDF+SDF eval)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "atoms": n_atoms, "frames_per_step_per_gpu": frames, "selection_atoms": n_sel,
                       "parallelism": f"frames sharded x{world}, one RCCL all-reduce per step", "rdf_variant": args.variant},
            "pairs_per_s": hits_per_step * steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_rdf_pencil", "avg_launch_ms": t_launch * 1e3, "launches": nl,
                         "algorithmic_bytes_per_launch": alg_bytes, "frames_per_launch": frames_per_launch,
                         "note": "pair kernel is VALU-bound, not HBM-bound (DESIGN.md): see valu_lane_ops_frac"},
            "kernel_ms": {"rdf_pencil": ms, "cells_build": cand_ms, "timed_region": elapsed * 1e3},
            "synth_s": gen_s,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, n_atoms, box, seed, pairs)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
