"""Multi-GPU evaluation: frames are block-sharded over ranks, one process per GPU; every rank
accumulates integer histograms / volumes for its frames and ONE collective at the end merges them.

The collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
The reduction runs in place on the evaluator's own device accumulators (zero copy through
__cuda_array_interface__), sums are integers, so the merged result is bit-identical for any rank count.
"""
import ctypes as C

import numpy as np


def shard_frames(num_frames, rank, world_size):
    """Contiguous block of ceil(F/G) frames per rank: rank g owns [g*ceil(F/G), min(F, (g+1)*ceil(F/G)))."""
    per = -(-num_frames // world_size)
    beg = min(num_frames, rank * per)
    return beg, min(num_frames, beg + per)


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer without copying."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _alias_counts(view, on_gpu):
    import torch
    n = view.num_counts
    if on_gpu:
        return torch.as_tensor(_DevArray(view.counts_dev, n, "<i8"), device="cuda")
    # emulator build (tests only): "device" memory is host memory
    buf = (C.c_int64 * n).from_address(view.counts_dev)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def reduce_eval(ev, group=None):
    """Merge the accumulators of all ranks into every rank's evaluator, then refresh its host views.

    u64 counts (RDF bins, SDF voxels): SUM as int64 (two's complement sum == unsigned sum), in place on the device;
    fp64 weights, temporal rows (zero for frames a rank did not evaluate) and the frame mask: one packed fp64 SUM."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return          # nothing to merge; frame_range already refreshed the host views
    on_gpu = dist.get_backend(group) == "nccl"
    dev = "cuda" if on_gpu else "cpu"
    keep, host_parts = [], []
    for v in ev.accum_views():
        if v.counts_dev and v.num_counts:
            t = _alias_counts(v, on_gpu)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)      # in place on the evaluator's device accumulators
            keep.append(t)
        for ptr, n, ctype in ((v.weights64, v.num_weights, C.c_double), (v.temporal, v.num_temporal, C.c_float)):
            if ptr and n:
                host_parts.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)))
    # everything that lives on the host (fp64 weights, temporal rows, the frame mask) travels as ONE fp64 buffer: a frame is
    # evaluated by one rank, so SUM of the masks is 0/1 (> 0 is taken, which also covers ranks that evaluated the same frames)
    mask = np.array(ev.frame_mask(), dtype=np.uint8, copy=True)
    packed = np.concatenate([h.astype(np.float64, copy=False).ravel() for h in host_parts] + [mask.astype(np.float64)])
    t = torch.from_numpy(packed)
    if on_gpu:
        g = t.to(dev)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        t = g.cpu()
        torch.cuda.synchronize()                                      # the in-place counts are final before finalize() reads them
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    merged = t.numpy()
    off = 0
    for h in host_parts:
        h[...] = merged[off:off + h.size].astype(h.dtype)
        off += h.size
    ev.set_frame_mask((merged[off:] > 0.5).astype(np.uint8))
    ev.finalize()
