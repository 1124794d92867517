"""What a FRESH eval costs (VIAMD creates a new md_script_eval_t for every script edit, src/main.cpp:966-972): first frame_range of
new evals on a trajectory that has been evaluated before, against the steady state of a reused eval.
usage: python scripts/exp_fresh_eval.py [workload] [device|xtc]"""
import os, sys, time, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import viamd_amd as V
from viamd_amd import script, synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind = sys.argv[2] if len(sys.argv) > 2 else "device"
for kv in sys.argv[3:]:
    k, v = kv.split("="); V.default_lib().vmd_set_option(k.encode(), int(v))
w = bench.WORKLOADS[name]
F = w["frames"] if name != "c3" else 200
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
cell = V.make_unitcell(w["box"])
if kind == "xtc":
    host = V.PinnedHostTrajectory(F, w["atoms"]); host.copy_from_device(traj); traj.close()
    path = os.path.join(tempfile.gettempdir(), f"fresh_{name}.xtc")
    V.write_xtc(path, host, cell); host.close()
    traj = V.XdrTrajectory(path)
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell)

def run(ev):
    torch.cuda.synchronize(); t = time.perf_counter()
    assert ev.frame_range(sysm, traj, 0, F)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t)

out = {"workload": name, "trajectory": kind, "frames": F}
ev = V.ScriptEval(F, ir)
out["first_eval_first_pass_ms"] = run(ev)
steady = []
for _ in range(3):
    ev.clear_data(); steady.append(run(ev))
out["reused_eval_ms"] = min(steady)
fresh = []
for _ in range(4):
    t = time.perf_counter(); e2 = V.ScriptEval(F, ir); c = 1e3 * (time.perf_counter() - t)
    r = run(e2)
    t = time.perf_counter(); e2.close(); d = 1e3 * (time.perf_counter() - t)
    fresh.append({"create_ms": round(c, 2), "first_range_ms": round(r, 2), "free_ms": round(d, 2)})
out["fresh_evals"] = fresh
print(json.dumps(out))
