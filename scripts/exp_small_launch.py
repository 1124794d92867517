"""What a one-frame pair launch costs as a function of the kernel instantiation it runs (code size): same-set / two-set passes, variants.
usage: python scripts/exp_small_launch.py [key=value ...]   (prints one line per case: kernel time per call from the library's own events)"""
import os, sys, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import viamd_amd as V
from viamd_amd import script, synth
lib = V.default_lib()
busy = False
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "busy":
        busy = bool(int(v))        # keep the device busy from another stream while the small launches are timed: is it the clock?
    else:
        lib.vmd_set_option(k.encode(), int(v))
if busy:
    import threading
    stop = [False]
    def hog():
        # ONE spinning block on a side stream (torch.cuda._sleep): the device is never idle, 255 CUs stay free for the launches under test
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                torch.cuda._sleep(int(2.0e8))
                st.synchronize()
    th = threading.Thread(target=hog, daemon=True); th.start(); time.sleep(0.5)
N, box, F = 100002, 100.0, 200
traj = synth.make_device_trajectory(V, 2, N, box, F, 0)
topo = synth.water_box_topology(N, 0)
sysm = V.MolSystem(N, mass=topo.mass, unitcell=V.make_unitcell(box))
for name, text, variant in (("O-O same set (variant 0)", "g = rdf(element('O'), element('O'), 12.0);", 0),
                            ("O-H two sets (variant 0)", "g = rdf(element('O'), element('H'), 12.0);", 0),
                            ("O-O same set (variant 1: bin in place)", "g = rdf(element('O'), element('O'), 12.0);", 1),
                            ("O-O same set (variant 2: pair entries)", "g = rdf(element('O'), element('O'), 12.0);", 2)):
    lib.vmd_set_option(b"rdf_variant", variant)
    ir, info = script.compile_script(text, topo)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(sysm, traj, 0, F)
    for g in (1, 8):
        ev.clear_data()
        lib.vmd_profile_reset(); lib.vmd_profile_enable(True)
        for f in range(0, F, g):
            assert ev.frame_range(sysm, traj, f, min(F, f + g))
        lib.vmd_profile_enable(False)
        n = C.c_uint64(0); ms = lib.vmd_profile_ms(b"rdf_pencil", C.byref(n))
        print(f"{name}: {g} frame(s) per call: pair launch {1e3 * ms / n.value:.0f} us", flush=True)
    ev.close()
lib.vmd_set_option(b"rdf_variant", 0)
if busy:
    stop[0] = True; th.join()
