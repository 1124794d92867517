"""Long run of the seeded random scenarios (tests/test_fuzz_emu.py) on the product build: python scripts/fuzz_gpu.py SEED0 SEED1 [SCALE]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import cases
from test_fuzz_emu import scenario, sdf_scenario
from viamd_amd import default_lib
from oracle import oracle as O

lib = default_lib()
s0, s1 = int(sys.argv[1]), int(sys.argv[2])
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fails = []
t = time.time()
for seed in range(s0, s1):
    coords, box, flags, props, opts, kind = scenario(seed, scale=scale)
    old = {k: lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
    try:
        cases.check_rdf(lib, O, coords, box, props, flags=flags, oracle_method="brute", device=bool(seed & 1))
    except Exception as ex:
        fails.append(("rdf", seed)); print("FAIL rdf", seed, kind, coords.shape, flags, opts, [(p[1].size, p[2].size, p[3], p[4]) for p in props], str(ex)[:300], flush=True)
    finally:
        for k, v in old.items(): lib.vmd_set_option(k.encode(), v)
    coords, box, flags, structures, mass, tgt, cutoff, opts, dist, kind = sdf_scenario(seed)
    old = {k: lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
    try:
        cases.check_distances(lib, O, coords, box, mass, dist, flags=flags, device=bool(seed & 1))
        cases.check_sdf(lib, O, coords, box, structures, mass, tgt, cutoff, flags=flags, allow_empty=True, device=bool(seed & 1))
    except Exception as ex:
        fails.append(("sdf", seed)); print("FAIL sdf", seed, kind, coords.shape, flags, structures.shape, tgt.size, cutoff, opts, str(ex)[:300], flush=True)
    finally:
        for k, v in old.items(): lib.vmd_set_option(k.encode(), v)
print("seeds", s0, s1, "scale", scale, "fails", fails, "%.1f s" % (time.time() - t))
