"""Experiment (round 2), second form: as scripts/exp_overlap.py, but the two evals run STAGGERED (the second thread starts half a
pass late and both loop back to back), so that the HBM-bound cell build of one lands in the middle of the VALU-bound pair kernel of
the other instead of next to the other's cell build."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from viamd_amd import default_lib, script, synth
import viamd_amd as V

lib = default_lib()
w = bench.WORKLOADS["c3"]
F = w["frames"]
traj = synth.make_device_trajectory(V, w["seed"], w["atoms"], w["box"], F, w["blob"])
topo = synth.water_box_topology(w["atoms"], w["blob"])
cell = V.make_unitcell(w["box"])
ir, info = script.compile_script(w["script"], topo)
sysm = V.MolSystem(w["atoms"], mass=topo.mass, unitcell=cell)
torch.cuda.synchronize()
NP = 6

one = V.ScriptEval(F, ir)
def single():
    for _ in range(NP):
        one.clear_data()
        assert one.frame_range(sysm, traj, 0, F)

def wall(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / NP

for blocks in (2048, 1536):
    lib.vmd_set_option(b"rdf_blocks", blocks)
    ms1 = wall(single)
    line = "rdf_blocks %4d: one eval %8.2f ms per pass" % (blocks, ms1)
    for stagger_ms in (0.0, 0.25 * ms1, 0.5 * ms1):
        evs = [V.ScriptEval(F, ir) for _ in range(2)]
        def run():
            def loop(ev, beg, end, delay):
                time.sleep(delay * 1e-3)
                for _ in range(NP):
                    ev.clear_data()
                    assert ev.frame_range(sysm, traj, beg, end)
            ths = [threading.Thread(target=loop, args=(evs[0], 0, F // 2, 0.0)), threading.Thread(target=loop, args=(evs[1], F // 2, F, stagger_ms))]
            for th in ths: th.start()
            for th in ths: th.join()
        ms = wall(run)
        line += "   2 evals, second %5.1f ms late: %8.2f ms (%+.1f %%)" % (stagger_ms, ms, (ms1 / ms - 1) * 100)
        del evs
    print(line, flush=True)
