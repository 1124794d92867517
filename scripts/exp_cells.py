"""Micro-experiment: where does the cell build spend its time?  Calls the thin C-ABI layer directly on torch buffers."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import viamd_amd as V
from viamd_amd import _lib as L

lib = V.default_lib()
dev = "cuda"
N, Lbox, B = 1000002, 215.443, 64
traj = V.DeviceTrajectory(B, N)
traj.synth(3, Lbox, 0.05)
base, fs, rs = traj.device_ptr()
boxes = torch.tensor([[Lbox] * 3 + [1.0 / Lbox] * 3 + [0.0] * 3] * B, dtype=torch.float32, device=dev)


def run(sel, label, fused=0, use_aos=True):
    lib.vmd_set_option(b"cells_fused", fused)
    nsel = sel.numel()
    nsel_pad = (nsel + 63) // 64 * 64
    ny = nz = int(Lbox // 12.0)
    nxf = min(int(Lbox // 1.5), 24575 // (ny * nz))
    g = L.Grid(nxf, ny, nz, nxf * ny * nz)
    cc = torch.zeros(B * (g.ncell + 1), dtype=torch.int32, device=dev)
    cs = torch.zeros_like(cc)
    rank = torch.zeros(B * nsel, dtype=torch.int32, device=dev)
    srt = torch.zeros(B * 3 * nsel_pad + 64, dtype=torch.float32, device=dev)
    seli = sel.to(torch.int32).to(dev)
    aos = torch.zeros(B * 4 * nsel_pad, dtype=torch.float32, device=dev)
    args = (None, base, fs, rs, boxes.data_ptr(), 7, B, seli.data_ptr(), nsel, nsel_pad, g, cc.data_ptr(), rank.data_ptr(), cs.data_ptr(), srt.data_ptr(), aos.data_ptr() if use_aos else None)
    for _ in range(2):
        assert lib.vmd_hip_cells_build(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.vmd_hip_cells_build(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label:34s} nsel={nsel:8d} {ms / B * 1e3:8.2f} us/frame  {nsel * B / ms / 1e6:8.1f} Matoms/ms  {12.0 * N * B / ms / 1e6:8.1f} GB/s of 12N")


import torch
run(torch.arange(0, N, 3), "every 3rd atom (O), SoA scatter", use_aos=False)
run(torch.arange(0, N, 3), "every 3rd atom (O), AoS+repack")
run(torch.arange(0, N // 3), "first third, contiguous")
run(torch.arange(0, N), "all atoms, contiguous")
run(torch.randperm(N)[: N // 3].sort().values, "random third, sorted")
run(torch.arange(0, 100002, 3), "33k atoms (c2-like sel), 3-kernel")
run(torch.arange(0, 100002, 3), "33k atoms (c2-like sel), fused", fused=1)
run(torch.arange(0, 100002, 3), "33k atoms, fused, SoA scatter", fused=1, use_aos=False)
