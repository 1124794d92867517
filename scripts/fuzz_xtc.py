#!/usr/bin/env python
"""Randomised cross-check of the native XTC writer / reader (viamd_amd/csrc/vmd_xdr.cpp, emulator build: the file code is plain
host C++) against the byte-wise Python restatement tests/xtc_ref.py: byte-identical files, identical decoded integers.
usage: python scripts/fuzz_xtc.py [trials] [seed] [gpu]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest          # noqa: E402
import xtc_ref           # noqa: E402
import test_xdr          # noqa: E402
import viamd_amd as V    # noqa: E402


def random_system(rng):
    n = int(rng.integers(10, 400)) if rng.random() < 0.8 else int(rng.integers(400, 6000))     # the larger ones: several tiles and checkpoints
    kind = rng.integers(0, 5)
    scale = float(10 ** rng.uniform(0, 3.2))
    if kind == 0:
        x = rng.uniform(-scale, scale, (n, 3))
    elif kind == 1:                                  # random walk with bursts
        x = np.cumsum(rng.normal(0, 10 ** rng.uniform(-2, 0.5), (n, 3)) * (1 + 20 * (rng.random((n, 1)) < 0.05)), axis=0)
    elif kind == 2:                                  # clusters
        c = rng.uniform(0, scale, (n, 3))
        x = c[np.sort(rng.integers(0, max(1, n // 4), n))] + rng.normal(0, 10 ** rng.uniform(-2, 0), (n, 3))
    elif kind == 3:                                  # one huge axis (three-field branch)
        x = rng.uniform(0, 30, (n, 3))
        x[:, rng.integers(0, 3)] *= 10 ** rng.uniform(3, 4.3)
    else:                                            # many exact repeats
        x = np.round(rng.uniform(0, 20, (n, 3)))
    return x.T.astype(np.float32)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"      # the device decoders of the product library on a real GPU
    if on_gpu:
        import torch  # noqa: F401      torch first, as in bench.py and tests/conftest.py: its bundled HIP runtime must be the one that finds the GPU
    lib = V.default_lib() if on_gpu else V.VmdLib(conftest.build_emu())      # one library per process: both export the same symbols
    dev_lib = lib
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    ndev = 0
    for t in range(trials):
        xyz = random_system(rng)
        precision = float(rng.choice([10.0, 100.0, 1000.0, 1000.0, 5000.0]))
        if abs(xyz).max() * 0.1 * precision > 2.0e9:
            precision = 10.0
        ref = xtc_ref.frame_bytes(xyz, np.diag([50.0, 60.0, 70.0]), t, 0.0, precision)
        p = os.path.join(tmp, "f.xtc")
        V.write_xtc(p, xyz[None], V.make_unitcell((50.0, 60.0, 70.0)), precision=precision, lib=lib)
        nat = open(p, "rb").read()
        # write_xtc numbers steps from 0
        assert nat[:8] == ref[:8] and nat[12:] == ref[12:], f"trial {t}: files differ (seed {seed})"
        ints = xtc_ref.to_ints((xyz * np.float32(0.1)).T, precision)
        got, _ = V.XdrTrajectory(p, lib=lib).load_frame(0)
        want = ((ints.astype(np.float32) * (np.float32(1) / np.float32(precision))) * np.float32(10)).T
        assert np.array_equal(got, want), f"trial {t}: native decode differs (seed {seed})"
        assert np.array_equal(xtc_ref.parse_frames(nat)[0]["ints"], ints), f"trial {t}: python decode differs"
        try:
            dev, status = test_xdr._device_decode(dev_lib, nat, xyz.shape[1], chunk=(0, 64, 97, -1, -2, -3, -4, -5, -6)[t % 9], gpu=on_gpu)
        except AssertionError as e:
            raise AssertionError(f"trial {t} (seed {seed}, {xyz.shape[1]} atoms, precision {precision}, variant {(0, 64, 97, -1, -2, -3, -4, -5, -6)[t % 9]}): {str(e)[:300]}")    # every device variant (-1: wave per frame, -2: + checkpointed second pass, -3 / -4: the same on streams laid out as in the file), on the SIMT emulator
        assert status[0] in (0, 2), f"trial {t}: device decoder rejected a valid stream (seed {seed})"
        if status[0] == 0:
            assert np.array_equal(dev[0], want), f"trial {t}: device decode differs (seed {seed})"
            ndev += 1
    print(f"{trials} random XTC frames: native and byte-wise implementations agree, device decoder agrees on {ndev} "
          f"(the others need more than 64 bits per number: status 2) (seed {seed})")


if __name__ == "__main__":
    main()
