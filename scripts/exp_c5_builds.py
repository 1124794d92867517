"""c5 (configs[4]): cell builds, pair dispatches and bucket overflows PER STEP over a few steps (round 6: the counter collection showed 3.5 builds
per batch where 2 are expected)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import viamd_amd as V
from viamd_amd import script, synth
import bench
lib = V.default_lib(); lib.vmd_set_device(0)
w = bench.WORKLOADS["c5"]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
ir, info = script.compile_script(w["script"], topo)
ev = V.ScriptEval(F, ir)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))
for step in range(6):
    lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
    ev.clear_data()
    t = time.perf_counter()
    assert ev.frame_range(sysm, traj, 0, F)
    dt = time.perf_counter() - t
    out = {}
    for k in ("cells_build", "rdf_pencil", "batches"):
        n = C.c_uint64(0); ms = lib.vmd_profile_ms(k.encode(), C.byref(n)); out[k] = (int(n.value), round(ms, 2))
    print(f"step {step}: {dt * 1e3:.1f} ms", out, "overflows / selections off buckets:", ev.cell_build_stats(), flush=True)
