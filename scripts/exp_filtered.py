"""Filtered evaluation at the bench shape: cost of keeping block partials during the full evaluation, and the
time a timeline sub-range takes with and without them.  usage: python scripts/exp_filtered.py [workload] [block_frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viamd_amd as V
from viamd_amd import script, synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
opts = {}
for kv in sys.argv[3:]:            # library options, key=value (e.g. block_superbatch=0)
    k, v = kv.split("=")
    V.default_lib().vmd_set_option(k.encode(), int(v)); opts[k] = int(v)
w = bench.WORKLOADS[name]
frames = w["frames"]
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], frames, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=V.make_unitcell(w["box"]))

def timed(ev, beg, end, reps=3):
    best = 1e9
    for _ in range(reps):
        ev.clear_data()
        t = time.perf_counter()
        assert ev.frame_range(sysm, traj, beg, end)
        best = min(best, time.perf_counter() - t)
    return best

plain = V.ScriptEval(frames, ir)
full = V.ScriptEval(frames, ir); full.set_block_frames(S)
timed(plain, 0, frames, 1)
out = {"workload": name, "frames": frames, "block_frames": S, "options": opts}
out["full_plain_ms"] = 1e3 * timed(plain, 0, frames)
out["full_with_blocks_ms"] = 1e3 * timed(full, 0, frames)
filt = V.ScriptEval(frames, ir); filt.set_source(full)
for beg, end in [(frames // 10 + 3, 9 * frames // 10 - 3), (frames // 4, frames // 2), (0, frames)]:
    a = timed(plain, beg, end); b = timed(filt, beg, end)
    for n in info:
        if info[n]["kind"] in ("rdf", "sdf"):
            assert np.array_equal(plain.property_data(n).counts, filt.property_data(n).counts), n
    out[f"range_{beg}_{end}"] = {"plain_ms": 1e3 * a, "filtered_ms": 1e3 * b, "frames_computed_reused": filt.frame_stats()}
print(json.dumps(out))
