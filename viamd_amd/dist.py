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
        # 1 flag byte + the 128-byte id.  Rank 0 ALWAYS takes part in the broadcast, also when it could not make an id: a rank that
        # skipped it would leave the others stuck in a collective of torch's process group, and everything queued behind it with them
        msg = np.zeros(1 + L.COMM_ID_BYTES, np.uint8)
        err0 = ""
        if rank == 0:
            if lib.vmd_comm_unique_id(msg[1:].ctypes.data_as(L.c_uint8_p)):
                msg[0] = 1
            else:
                err0 = lib.last_error()
        t = torch.from_numpy(msg)
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:
            t = t.cuda()
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        msg = t.cpu().numpy().copy()
        if msg[0] != 1:
            raise RuntimeError("rank 0 could not create an RCCL id" + (": " + err0 if err0 else ""))
        ident = np.ascontiguousarray(msg[1:])
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
    """vmd_collective_i over torch.distributed.  CPU tests (gloo): buffers the evaluator calls "device" are host memory in the
    emulator build, so the callbacks alias them as numpy arrays and all-reduce those.  GPU (nccl backend): the FALLBACK when the
    library's own communicator cannot be created (bench.py, N > 1: the first contact of that path with real hardware is the
    driver's run) - the callbacks alias the evaluator's device memory as torch tensors (__cuda_array_interface__, zero copy) and
    torch's RCCL communicator carries the same SUMs; torch runs them on its own stream, so every call is bracketed by device
    synchronisation (a merge per step: microseconds against a step)."""

    def __init__(self, group=None, device=False):
        import torch
        import torch.distributed as dist
        staged = device and dist.get_backend(group) != "nccl"
        self.kind = "torch.distributed (" + dist.get_backend(group) + (", device buffers staged through host memory)" if staged else ")")

        class _Dev:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

        def allreduce(ptr, n, ctype, view, typestr):
            if n == 0:
                return True
            if device and staged:
                # device memory, host collective (gloo): D2H, all-reduce, H2D - several processes on ONE GPU, where RCCL refuses
                torch.cuda.synchronize()
                t = torch.as_tensor(_Dev(ptr, n, typestr), device="cuda")
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
                torch.cuda.synchronize()
            elif device:
                torch.cuda.synchronize()
                t = torch.as_tensor(_Dev(ptr, n, typestr), device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                torch.cuda.synchronize()
            else:
                buf = (ctype * n).from_address(ptr)
                t = torch.from_numpy(np.ctypeslib.as_array(buf).view(view))
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return True

        # two's complement sums == unsigned sums
        self._cb = (L.COMM_INT_FN(lambda _: dist.get_rank(group)), L.COMM_INT_FN(lambda _: dist.get_world_size(group)),
                    L.ALLREDUCE_U64_FN(lambda _, p, n, s: allreduce(p, n, C.c_int64, np.int64, "<i8")),
                    L.ALLREDUCE_F64_FN(lambda _, p, n, s: allreduce(p, n, C.c_double, np.float64, "<f8")),
                    L.GROUP_FN(), L.GROUP_FN(),
                    L.ALLREDUCE_U32_FN(lambda _, p, n, s: allreduce(p, n, C.c_int32, np.int32, "<i4")))
        self.c = L.CollectiveI(None, *self._cb)
        self.ranks = dist.get_world_size(group)

    def collective(self):
        return C.byref(self.c)


_comms = {}


def _make_comm(lib, group):
    """The library's own RCCL communicator where it can be made on EVERY rank, torch.distributed's otherwise (all ranks decide alike:
    a mixed job would dead-lock).  VIAMD_AMD_COLLECTIVE=torch|rccl forces one of them; VIAMD_AMD_COMM_TIMEOUT (seconds, default 120)
    bounds ncclCommInitRank, which blocks until every rank has arrived."""
    import os
    import threading

    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == "nccl"
    if not on_gpu:
        # gloo: the emulator build (its "device" memory is host memory) - or, VIAMD_AMD_STAGED_COLLECTIVE=1, real device memory
        # staged through the host (several ranks sharing one GPU: tests/test_zz_late_gpu.py, bench.py VIAMD_BENCH_SHARE_GPU=1)
        return TorchCollective(group, device=os.environ.get("VIAMD_AMD_STAGED_COLLECTIVE") == "1")
    want = os.environ.get("VIAMD_AMD_COLLECTIVE", "").lower()
    comm, why = None, ""
    if want != "torch":
        box = {}
        dev = torch.cuda.current_device()

        def make():
            try:
                # the current device is a property of the host THREAD: without this every rank's helper thread would sit on device 0
                # and ncclCommInitRank would see eight ranks on one GPU
                torch.cuda.set_device(dev)
                lib.vmd_set_device(dev)
                box["comm"] = RcclComm(lib, group)
            except Exception as e:                     # noqa: BLE001 - whatever went wrong, the job falls back as one
                box["err"] = repr(e)

        t = threading.Thread(target=make, daemon=True)
        t.start()
        t.join(float(os.environ.get("VIAMD_AMD_COMM_TIMEOUT", "120")))
        comm = box.get("comm")
        why = box.get("err", "ncclCommInitRank did not return in time" if t.is_alive() else "")
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 1:
        comm.kind = "rccl (vmd_comm_create)"
        comm.ranks = int(lib.vmd_comm_size(comm.h))
        return comm
    if want == "rccl":
        raise RuntimeError("VIAMD_AMD_COLLECTIVE=rccl but the communicator could not be created on every rank: " + why)
    if comm is not None:
        comm.close()
    fb = TorchCollective(group, device=True)
    fb.fallback_reason = why or "another rank could not create its communicator"
    return fb


def reduce_eval(ev, group=None):
    """Merge the accumulators of all ranks into every rank's evaluator (vmd_eval_reduce), then its host views are current."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return          # nothing to merge; frame_range already refreshed the host views
    key = (id(ev.lib), group)
    if key not in _comms:
        _comms[key] = _make_comm(ev.lib, group)
    if not ev.lib.vmd_eval_reduce(ev.h, _comms[key].collective(), None):
        raise RuntimeError(ev.lib.last_error())


def collective_info(lib, group=None):
    """What carried the merges of this process so far: {"kind", "ranks", "fallback_reason"} or None before the first merge."""
    c = _comms.get((id(lib), group))
    if c is None:
        return None
    return {"kind": getattr(c, "kind", type(c).__name__), "ranks": getattr(c, "ranks", None), "fallback_reason": getattr(c, "fallback_reason", None)}


def reduce_stats(ev):
    st = L.ReduceStats()
    ev.lib.vmd_eval_reduce_stats(ev.h, C.byref(st))
    return {"bytes": int(st.bytes), "allreduce_calls": int(st.calls), "grouped_into_one_launch": bool(st.grouped),
            "volumes_as_u32": int(st.volumes_as_u32), "device_ms": float(st.ms)}


def close_comms():
    """Destroy the cached RCCL communicators (before torch.distributed.destroy_process_group)."""
    for c in _comms.values():
        if hasattr(c, "close"):
            c.close()
    _comms.clear()
