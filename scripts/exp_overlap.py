"""Experiment (round 2): does running the HBM-bound cell build of one half of the frames next to the VALU-bound pair kernel of the
other half pay?  Two evals over [0, F/2) and [F/2, F) of the c3 trajectory, each on its own stream, driven from two host threads,
against one eval over [0, F); with the persistent pair grid at 7 / 6 / 5 blocks per CU (free wave slots for the other stream)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from viamd_amd import default_lib, script, synth
import viamd_amd as V

lib = default_lib()
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
F = w["frames"]
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
cell = V.make_unitcell(w["box"])
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell)
torch.cuda.synchronize()

def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

one = V.ScriptEval(F, ir)
def single():
    one.clear_data()
    assert one.frame_range(sysm, traj, 0, F)

def make_split(k):
    evs = [V.ScriptEval(F, ir) for _ in range(k)]
    cuts = [F * i // k for i in range(k + 1)]
    def run():
        ths = []
        for i, ev in enumerate(evs):
            ev.clear_data()
            th = threading.Thread(target=lambda ev=ev, i=i: ev.frame_range(sysm, traj, cuts[i], cuts[i + 1]))
            th.start(); ths.append(th)
        for th in ths:
            th.join()
    return run, evs

ref = None
for blocks in (2048, 1792, 1536, 1280):
    lib.vmd_set_option(b"rdf_blocks", blocks)
    ms1 = timed(single)
    if ref is None:
        ref = {n: one.property_data(n).counts.copy() for n, d in info.items() if d["kind"] == "rdf"}
    line = "rdf_blocks %4d: one eval %8.2f ms" % (blocks, ms1)
    for k in (2, 4):
        run, evs = make_split(k)
        ms = timed(run)
        for n in ref:
            tot = sum(ev.property_data(n).counts.astype("uint64") for ev in evs)
            assert (tot == ref[n]).all(), n
        line += "   %d evals in threads %8.2f ms (%+.1f %%)" % (k, ms, (ms1 / ms - 1) * 100)
        del run, evs
    print(line, flush=True)
