"""Where the fixed cost of one small frame_range call goes (the round of a combining queue fed with grain-1 ranges): N calls of one frame each.
usage: python scripts/exp_round_cost.py [workload] [frames per call]"""
import os, sys, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import viamd_amd as V
from viamd_amd import script, synth
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = bench.WORKLOADS[name]
F = 400
lib = V.default_lib()
for kv in sys.argv[3:]:
    k, v = kv.split("="); lib.vmd_set_option(k.encode(), int(v))
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
ev = V.ScriptEval(F, ir)
assert ev.frame_range(sysm, traj, 0, F)
ev.clear_data()
lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
torch.cuda.synchronize(); t = time.perf_counter()
for f in range(0, F, g):
    assert ev.frame_range(sysm, traj, f, min(F, f + g))
wall = 1e3 * (time.perf_counter() - t)
lib.vmd_profile_enable(False)
out = {"workload": name, "calls": (F + g - 1) // g, "frames_per_call": g, "wall_ms": wall, "per_call_us": 1e3 * wall / ((F + g - 1) // g)}
for k in ("host_settle", "host_queue_to_sync", "host_fetch_stage", "host_sync_wait", "host_refresh", "cells_build", "rdf_pencil", "sdf_align", "sdf_scatter", "distance", "batches"):
    n = C.c_uint64(0); ms = lib.vmd_profile_ms(k.encode(), C.byref(n))
    if n.value: out[k] = {"ms": round(ms, 3), "n": int(n.value), "us_per": round(1e3 * ms / n.value, 1)}
print(json.dumps(out))
