"""Data parallelism: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm;
'gloo' for the CPU tests).  The batch shards over ranks; the only exchanges are SUM all-reduces:
the flat per-network gradient buffers (the reference differentiates [B,1] targets, i.e. sums over
the batch: SURVEY fact 5), the 12 fp64 loss statistics, and the per-channel BN statistics (SyncBN)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .nn import Reducer


class DistReducer(Reducer):
    shard_inputs = True

    def __init__(self, group=None):
        assert dist.is_initialized(), "call init_process_group first"
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    calls = 0          # collectives issued (bench.py reports them per step)

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            self.calls += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_sum_async(self, t: torch.Tensor):
        """RCCL runs the reduction on its own stream (ordered after the kernels already queued on the
        current stream), so D/R/S gradient exchange overlaps the generator's backward sweep."""
        if self.world_size == 1:
            return None
        self.calls += 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, handle) -> None:
        if handle is not None:
            handle.wait()        # makes the current stream wait for the collective; no host block on nccl

    def broadcast_object(self, obj, src: int = 0):
        """Rank `src`'s Python object on every rank (host-side; used for the per-step host draws when a caller did not
        seed the ranks' `random` streams identically)."""
        box = [obj if self.rank == src else None]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def shard(self, t: torch.Tensor) -> torch.Tensor:
        """This rank's contiguous slice [r*B/P, (r+1)*B/P) of a per-step batch tensor."""
        B = t.shape[0]
        if B % self.world_size:
            raise ValueError("global batch %d is not divisible by world size %d" % (B, self.world_size))
        b = B // self.world_size
        return t[self.rank * b:(self.rank + 1) * b].contiguous()


class AbiReducer(DistReducer):
    """The same exchanges with the gradient / statistics all-reduces issued through the C-ABI entry
    (`sg_allreduce_sum` -> ncclAllReduce of librccl) instead of torch.distributed's wrapper.  torch.distributed is still
    used once, to hand rank 0's RCCL unique id to the other ranks.  Selected with SG_COLLECTIVE=abi (one GPU per rank).
    Collectives run on a side stream ordered after the kernels already queued on the launch stream (event), so an
    asynchronous exchange overlaps the kernels queued after it, exactly like the torch.distributed path."""

    def __init__(self, group=None):
        super().__init__(group)
        import ctypes
        from ._lib import call
        self._call = call
        ident = (ctypes.c_char * 128)()
        if self.rank == 0:
            call("sg_rccl_unique_id", ctypes.addressof(ident))
        raw = self.broadcast_object(bytes(ident.raw)) if self.world_size > 1 else bytes(ident.raw)
        comm = ctypes.c_void_p()
        call("sg_rccl_comm_init_rank", ctypes.addressof(comm), self.world_size, raw, self.rank)
        self._comm = comm.value
        self._side = torch.cuda.Stream()
        import atexit
        atexit.register(self.close)             # the communicator is destroyed at interpreter shutdown at the latest

    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 3}

    def _launch(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous()
        self.calls += 1
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        self._side.wait_event(ready)
        # the collective reads and writes `t` on the side stream: tell the caching allocator, so that a temporary freed
        # before wait() is not handed out again while RCCL still works on it
        t.record_stream(self._side)
        self._call("sg_allreduce_sum", t.data_ptr(), t.numel(), self._DT[t.dtype], self._comm, self._side.cuda_stream)
        done = torch.cuda.Event()
        done.record(self._side)
        return done

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        torch.cuda.current_stream().wait_event(self._launch(t))
        return t

    def all_reduce_sum_async(self, t: torch.Tensor):
        return self._launch(t)

    def wait(self, handle) -> None:
        if handle is not None:
            torch.cuda.current_stream().wait_event(handle)

    def close(self):
        if getattr(self, "_comm", None):
            try:
                self._side.synchronize()
                self._call("sg_rccl_comm_destroy", self._comm)
            finally:
                self._comm = None


def init_from_env(backend: str = "nccl") -> "Reducer":
    """Build the reducer from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return Reducer()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    if os.environ.get("SG_COLLECTIVE", "torch") == "abi" and backend == "nccl":
        return AbiReducer()
    return DistReducer()
