"""The seeded random RDF scenarios of tests/test_fuzz_emu.py (random cells incl. triclinic / open / per-frame boxes, random selections, cutoffs, options)
evaluated the way VIAMD calls the boundary - pool threads, ranges of 1 - 3 frames, random read-ahead block / region sizes - against ONE call over
the same range on the same library: counts, weights and the frame mask must be identical.  usage: python scripts/fuzz_pool.py SEED0 SEED1 [gpu] [lone]
`lone`: every third scenario runs in deferred-settle mode (option readahead_lone) - half of those with ONE caller thread - and compares after
vmd_eval_wait_settled, sometimes after a pause in which the helper thread may have settled on its own."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases, conftest
from test_fuzz_emu import scenario
import viamd_amd as V
from viamd_amd import _lib as L

s0, s1 = int(sys.argv[1]), int(sys.argv[2])
gpu = "gpu" in sys.argv[3:]
lone_mode = "lone" in sys.argv[3:]
lib = V.default_lib() if gpu else V.VmdLib(conftest.build_emu())
fails = []
t0 = time.time()
for seed in range(s0, s1):
    rng = np.random.default_rng(seed)
    coords, box, flags, props, opts, kind = scenario(seed, scale=20 if gpu else 4)
    F0 = coords.shape[0]
    reps = int(rng.integers(2, 6))
    coords = np.concatenate([coords] * reps)                       # more frames: the scenarios hold 1 - 4
    boxes = (box * reps) if isinstance(box, list) else box
    F, N = coords.shape[0], coords.shape[2]
    deferred = lone_mode and seed % 3 == 0
    ropts = dict(opts, readahead_lone=1 if deferred else 0, readahead_lone_settle_us=int(rng.choice([50, 300, 2000])), readahead_block=int(rng.choice([1, 2, 3, 4])), readahead_frames=int(rng.choice([1, 2, 4, 8])), readahead_growth=int(rng.choice([1, 2, 4])), readahead_company_us=200000)
    old = {k: lib.vmd_set_option(k.encode(), v) for k, v in ropts.items()}
    try:
        from oracle import oracle as O
        vcell = [cases.cell_pair(O, b, flags)[1] for b in boxes] if isinstance(boxes, list) else cases.cell_pair(O, boxes, flags)[1]
        ir = V.ScriptIR(lib)
        for nm, a, b, r0, r1 in props:
            ir.add_rdf(nm, a, b, (r0, r1))
        traj = cases.make_traj(lib, coords, vcell, bool(seed & 1) and gpu)
        sysm = V.MolSystem(N, unitcell=vcell[0] if isinstance(vcell, list) else vcell)
        one = V.ScriptEval(F, ir)
        o = lib.vmd_set_option(b"readahead", 0)
        ok1 = one.frame_range(sysm, traj, 0, F)
        lib.vmd_set_option(b"readahead", o)
        ev = V.ScriptEval(F, ir)
        grain, nth = int(rng.integers(1, 4)), int(rng.integers(2, 7))
        if deferred and rng.random() < 0.5: nth = 1
        starts = list(range(0, F, grain))
        if rng.random() < 0.4: starts = starts[::-1]
        nxt = [0]; lock = threading.Lock(); res = []
        def work():
            while True:
                with lock:
                    k = nxt[0]; nxt[0] += 1
                if k >= len(starts): return
                res.append(ev.frame_range(sysm, traj, starts[k], min(F, starts[k] + grain)))
        ths = [threading.Thread(target=work) for _ in range(nth)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert ok1 and all(res), "a call failed: " + lib.last_error()
        if deferred:
            if rng.random() < 0.5: time.sleep(float(rng.choice([0.0001, 0.001, 0.01])))
            ev.wait_settled()
        assert ev.frames_done() == F and ev.frame_mask().all()
        for nm, *_ in props:
            np.testing.assert_array_equal(ev.property_data(nm).counts, one.property_data(nm).counts, err_msg=nm)
            np.testing.assert_allclose(ev.property_data(nm).weights64, one.property_data(nm).weights64, rtol=1e-12, err_msg=nm)
        ev.close(); one.close()
    except Exception as ex:
        fails.append(seed); print("FAIL", seed, kind, coords.shape, flags, ropts, str(ex)[:300], flush=True)
    finally:
        for k, v in old.items(): lib.vmd_set_option(k.encode(), v)
print("pool fuzz seeds", s0, s1, "fails", fails, "%.1f s" % (time.time() - t0))
