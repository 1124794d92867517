"""Experiment: does running the cell build of one batch concurrently with the pair kernel of another pay?
Two host threads, each with its own evaluator (own HIP stream), evaluate the c2 workload at the same time; the aggregate
rate is compared with a single evaluator.  usage: python scripts/exp_overlap.py [rdf_blocks ...]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viamd_amd as V
from viamd_amd import script, synth

lib = V.default_lib()
atoms, box, frames, steps = 100002, 100.0, 1000, 10
traj = synth.make_device_trajectory(V, 2, atoms, box, frames, 0)
topo = synth.water_box_topology(atoms, 0)
cell = V.make_unitcell(box)
ir, info = script.compile_script("g = rdf(element('O'), element('O'), 12.0);", topo)
sysm = V.MolSystem(atoms, mass=topo.mass, unitcell=cell)

def worker(ev, n):
    for _ in range(n):
        ev.clear_data()
        assert ev.frame_range(sysm, traj, 0, frames)

def run(nthreads, blocks):
    lib.vmd_set_option(b"rdf_blocks", blocks)
    evs = [V.ScriptEval(frames, ir) for _ in range(nthreads)]
    for ev in evs: worker(ev, 2)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(ev, steps)) for ev in evs]
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    ref = int(evs[0].property_data("g").counts.sum())
    assert all(int(ev.property_data("g").counts.sum()) == ref for ev in evs)
    print(f"threads={nthreads} rdf_blocks={blocks}: {nthreads * steps * frames / dt:9.0f} frames/s")

for blocks in [int(a) for a in sys.argv[1:]] or [2048]:
    for nt in (1, 2, 3):
        run(nt, blocks)
