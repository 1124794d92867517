"""The seeded random scenarios of tests/test_fuzz_emu.py at 20x the atoms on the MI355X (product build: the hand-scheduled push / pop
paths, real wave scheduling, real atomics) against the oracle's all-pairs method; SDF / distance scenarios as they are."""
import numpy as np
import pytest

import cases
from test_fuzz_emu import scenario, sdf_scenario

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chunk", range(4))
def test_random_rdf_scenarios_on_gpu(gpu_lib, oracle, chunk):
    for seed in range(chunk * 8, chunk * 8 + 8):
        coords, box, flags, props, opts, kind = scenario(seed, scale=20)
        old = {k: gpu_lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
        try:
            cases.check_rdf(gpu_lib, oracle, coords, box, props, flags=flags, oracle_method="brute", device=bool(seed & 1))
        except Exception as ex:
            raise AssertionError(f"scenario seed {seed} x20 ({kind}, N {coords.shape[2]}, F {coords.shape[0]}, flags {flags}, opts {opts}, "
                                 f"props {[(p[0], p[1].size, p[2].size, p[3], p[4]) for p in props]}): {ex}") from ex
        finally:
            for k, v in old.items():
                gpu_lib.vmd_set_option(k.encode(), v)


@pytest.mark.parametrize("chunk", range(2))
def test_random_sdf_and_distance_scenarios_on_gpu(gpu_lib, oracle, chunk):
    for seed in range(chunk * 10, chunk * 10 + 10):
        coords, box, flags, structures, mass, tgt, cutoff, opts, dist, kind = sdf_scenario(seed)
        old = {k: gpu_lib.vmd_set_option(k.encode(), v) for k, v in opts.items()}
        try:
            cases.check_distances(gpu_lib, oracle, coords, box, mass, dist, flags=flags, device=bool(seed & 1))
            cases.check_sdf(gpu_lib, oracle, coords, box, structures, mass, tgt, cutoff, flags=flags, allow_empty=True, device=bool(seed & 1))
        except Exception as ex:
            raise AssertionError(f"scenario seed {seed} ({kind}, N {coords.shape[2]}, F {coords.shape[0]}, flags {flags}, K x m {structures.shape}, "
                                 f"targets {tgt.size}, cutoff {cutoff:.3f}, opts {opts}): {ex}") from ex
        finally:
            for k, v in old.items():
                gpu_lib.vmd_set_option(k.encode(), v)
