"""bench.py's own logic (workload table, weak / strong sharding, one vmd_eval_reduce per step, max-over-ranks timing, the JSON
line with `roofline`, `cpu_baseline` and `secondary`) dry-run on the CPU: the SIMT-emulator build of the library, tiny workloads,
world_size 1 and 2 (gloo).  No GPU, no timing claims - it keeps the N > 1 path of the file the driver launches from rotting."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import torch
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
import bench
tiny = dict(atoms=1500, blob=0, box=40.0, frames=6, seed=2, steps=1, sec_steps=1, kernel="rdf_pencil",
            script="g = rdf(element('O'), element('O'), 12.0);", desc="tiny rdf")
tiny4 = dict(atoms=1501, blob=100, box=40.0, frames=6, seed=4, steps=1, sec_steps=1, kernel="sdf_scatter",
             script="s = residue(5:8); v = sdf(s, element('O') and water, 10.0); d = distance(residue(1), residue(3));", desc="tiny sdf")
bench.WORKLOADS.update(c3=dict(tiny), c2=dict(tiny), c4=tiny4, c5=dict(tiny))
bench.cpu_baseline = lambda *a, **k: {{"value": 1.0, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "stub"}}
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
'''


def _run(tmp_path, emu, nproc, extra):
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER.format(root=ROOT))
    env = dict(os.environ, VIAMD_AMD_LIB=emu, VIAMD_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if nproc == 1:
        cmd = [sys.executable, str(drv)] + extra
    else:
        port = 29700 + (os.getpid() % 200)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(drv), "--gpus", str(nproc)] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]           # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields(tmp_path, emu_lib):
    d = _run(tmp_path, emu_lib.path, 1, ["--steps", "1", "--warmup", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "secondary"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["config"]["name"] == "c3" and d["vs_baseline"] is None and d["dtype"] == "f32"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert set(d["secondary"]) == {"c2", "c4", "c5"} and d["secondary"]["c4"]["roofline"]["kernel"] == "k_sdf_scatter"
    assert d["secondary"]["c4"]["voxel_hits_per_s"] > 0 and d["pairs_per_s"] > 0
    assert 0.0 <= d["cell_build"]["frac_of_step"] < 1.0             # the sorted copies behind the pair kernel: share of the step (event times are 0 on the emulator)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks(tmp_path, emu_lib, scaling):
    one = _run(tmp_path, emu_lib.path, 1, ["--steps", "1", "--warmup", "0", "--workload", "c4", "--scaling", scaling, "--no-cpu-baseline"])
    two = _run(tmp_path, emu_lib.path, 2, ["--steps", "1", "--warmup", "0", "--workload", "c4", "--scaling", scaling, "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["scaling"] == scaling
    # the line says what carried the merge and how many ranks THAT saw (the driver checks rccl_ranks == n_gpus on the GPU node)
    m = two["merge"]
    assert m["rccl_ranks"] == 2 and m["collective"].startswith("torch.distributed") and m["allreduce_calls"] >= 2 and m["bytes"] > 0
    assert m["host_ms_per_step"] >= 0.0 and "merge" not in one
    if scaling == "strong":
        assert m["volumes_as_u32"] == 1           # 6 frames x 4 structures x 467 targets fits 32 bits: the volume travelled as u32
    if scaling == "strong":       # the same 6 frames, block-sharded: the merged volume holds exactly the hits of the 1-rank run
        assert two["config"]["frames_per_step"] == 6 and two["config"]["frames_per_step_per_gpu"] == 3
        assert round(two["voxel_hits_per_s"] * two["ms_per_step"]) == round(one["voxel_hits_per_s"] * one["ms_per_step"])
    else:                          # every rank its own 6 frames
        assert two["config"]["frames_per_step"] == 12


def test_pool_threads_call_pattern(tmp_path, emu_lib):
    """--pool-threads / --grain: a step evaluated the way VIAMD calls the boundary gives the same line (plus `call_pattern`)."""
    d = _run(tmp_path, emu_lib.path, 1, ["--steps", "1", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--pool-threads", "3", "--grain", "2"])
    assert d["config"]["call_pattern"] == {"pool_threads": 3, "grain": 2, "note": d["config"]["call_pattern"]["note"]}
    assert d["value"] > 0 and d["pairs_per_s"] > 0
