"""frame_range as VIAMD calls it: N pool threads pull small disjoint ranges (enkiTS, grain 1: src/main.cpp:993-997, src/task_system.cpp:73-81)
and call md_script_eval_frame_range on the SAME eval; the evaluator combines the calls into large batches.  Cost of that against one call.
usage: python scripts/exp_threads.py [workload]"""
import os, sys, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import viamd_amd as V
from viamd_amd import script, synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = bench.WORKLOADS[name]
F = w["frames"] if name != "c3" else 200
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
ev = V.ScriptEval(F, ir)

def one_call():
    ev.clear_data(); torch.cuda.synchronize(); t = time.perf_counter()
    assert ev.frame_range(sysm, traj, 0, F)
    return 1e3 * (time.perf_counter() - t)

def pooled(nthreads, grain):
    ev.clear_data(); torch.cuda.synchronize()
    nxt = [0]; lock = threading.Lock(); ok = [True]
    def work():
        while True:
            with lock:
                b = nxt[0]; nxt[0] += grain
            if b >= F: return
            if not ev.frame_range(sysm, traj, b, min(F, b + grain)): ok[0] = False; return
    ths = [threading.Thread(target=work) for _ in range(nthreads)]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    ms = 1e3 * (time.perf_counter() - t)
    assert ok[0] and ev.frames_done() == F
    return ms

one_call()
ref = ev.property_data(next(n for n in info if info[n]["kind"] in ("rdf", "sdf"))).counts.copy()
out = {"workload": name, "frames": F, "one_call_ms": min(one_call() for _ in range(3))}
import numpy as np
for nt, g in ((16, 1), (16, 4), (16, 16), (64, 1), (128, 1), (4, 1)):
    out[f"threads{nt}_grain{g}_ms"] = min(pooled(nt, g) for _ in range(3))
    assert np.array_equal(ev.property_data(next(n for n in info if info[n]["kind"] in ("rdf", "sdf"))).counts, ref)
print(json.dumps(out))
