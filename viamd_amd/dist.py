"""Multi-GPU evaluation: frames are block-sharded over ranks, one process per GPU; every rank
accumulates integer histograms / volumes for its frames and ONE merge at the end combines them.

The merge is C++ behind the C ABI (vmd_eval_reduce, viamd_amd/csrc/vmd_reduce.cpp): the u64 accumulators are summed in
place on the device, the host-side parts in one packed fp64 all-reduce.  This module only supplies the collective:
  * on the GPU box the library's own RCCL communicator (vmd_comm_create; one per process, created once - rank 0 makes the
    128-byte id, torch.distributed broadcasts it: that is the only thing torch does for the merge);
  * in the CPU tests (emulator build, "device" memory is host memory) a vmd_collective_i whose callbacks run
    torch.distributed all-reduces on the gloo backend.
Sums are integers, so the merged result is bit-identical for any rank count.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def shard_frames(num_frames, rank, world_size):
    """Contiguous block of ceil(F/G) frames per rank: rank g owns [g*ceil(F/G), min(F, (g+1)*ceil(F/G)))."""
    per = -(-num_frames // world_size)
    beg = min(num_frames, rank * per)
    return beg, min(num_frames, beg + per)


class RcclComm:
    """vmd_comm_t: the library's RCCL communicator of this process (rendezvous of the 128-byte id through torch.distributed)."""

    def __init__(self, lib, group=None):
        import torch
        import torch.distributed as dist
        self.lib = lib
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = np.zeros(L.COMM_ID_BYTES, np.uint8)
        if rank == 0 and not lib.vmd_comm_unique_id(ident.ctypes.data_as(L.c_uint8_p)):
            raise RuntimeError(lib.last_error())
        t = torch.from_numpy(ident)
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:
            t = t.cuda()
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = t.cpu().numpy().copy()
        self.h = lib.vmd_comm_create(world, rank, ident.ctypes.data_as(L.c_uint8_p))
        if not self.h:
            raise RuntimeError(lib.last_error())

    def collective(self):
        return self.lib.vmd_comm_collective(self.h)

    def close(self):
        if self.h:
            self.lib.vmd_comm_destroy(self.h)
            self.h = None


class TorchCollective:
    """vmd_collective_i over torch.distributed for the CPU tests: buffers the evaluator calls "device" are host memory in the
    emulator build, so the callbacks alias them as numpy arrays and all-reduce those (gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        def allreduce(ptr, n, ctype, view):
            buf = (ctype * n).from_address(ptr)
            t = torch.from_numpy(np.ctypeslib.as_array(buf).view(view))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return True

        self._cb = (L.COMM_INT_FN(lambda _: dist.get_rank(group)), L.COMM_INT_FN(lambda _: dist.get_world_size(group)),
                    L.ALLREDUCE_U64_FN(lambda _, p, n, s: allreduce(p, n, C.c_int64, np.int64)),      # two's complement sum == unsigned sum
                    L.ALLREDUCE_F64_FN(lambda _, p, n, s: allreduce(p, n, C.c_double, np.float64)))
        self.c = L.CollectiveI(None, *self._cb)

    def collective(self):
        return C.byref(self.c)


_comms = {}


def reduce_eval(ev, group=None):
    """Merge the accumulators of all ranks into every rank's evaluator (vmd_eval_reduce), then its host views are current."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return          # nothing to merge; frame_range already refreshed the host views
    key = (id(ev.lib), group)
    if key not in _comms:
        _comms[key] = RcclComm(ev.lib, group) if dist.get_backend(group) == "nccl" else TorchCollective(group)
    if not ev.lib.vmd_eval_reduce(ev.h, _comms[key].collective(), None):
        raise RuntimeError(ev.lib.last_error())


def close_comms():
    """Destroy the cached RCCL communicators (before torch.distributed.destroy_process_group)."""
    for c in _comms.values():
        if hasattr(c, "close"):
            c.close()
    _comms.clear()
