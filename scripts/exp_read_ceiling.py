"""What a pure streaming read of a trajectory reaches on this box: k_bbox (min / max of all atoms of every frame: 12*N bytes per
frame, a handful of ALU ops per atom, one 1024-thread block per frame) over the c4 trajectory - the ceiling k_sdf_scatter
(same bytes, stride-3 gathers + the scatter logic) is to be read against."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import viamd_amd as V
from viamd_amd import synth

lib = V.default_lib()
torch.cuda.set_device(0); lib.vmd_set_device(0)
N, F = 100001, 10000
traj = synth.make_device_trajectory(V, 4, N, 100.0, F, 0)
ptr, fs, rs = traj.device_ptr()
out = torch.empty(F * 6, dtype=torch.float32, device="cuda")
for B in (1000, 10000):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f0 in range(0, F, B):
            assert lib.vmd_hip_bbox(None, ptr + f0 * fs * 4, fs, rs, B, N, out.data_ptr() + f0 * 24) == 0
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    print(f"k_bbox, {B} frames per launch: {ms:.3f} ms for {F} frames = {12.0 * N * F / ms / 1e9:.2f} TB/s")
