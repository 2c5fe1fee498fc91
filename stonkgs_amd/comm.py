"""The gradient exchange on the C ABI's own communicator (`stonk_comm_*`, csrc/comm.hip): RCCL collectives on a stream
the library owns, ordered by events - an alternative to `torch.distributed` for the data-parallel step
(`TrainingArguments.comm_backend = "stonk"`), and what a maintainer who binds the C ABI directly would use.

What the reference does here: nothing of its own - HF `Trainer` wraps the model in DistributedDataParallel when the job
is launched distributed (ref:src/stonkgs/models/stonkgs_pretraining.py:215-223), and DeepSpeed ZeRO-2 takes over with
`deepspeed=True` (:174-175).

The 128-byte RCCL id travels from rank 0 to the other ranks through `torch.distributed`'s default group when there is
one (any backend: `gloo` will do - it carries 128 bytes once), or through a file (`id_file`) for a program without torch's
process group."""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import torch

from . import _hip as hip

DTYPES = {torch.float32: 0, torch.bfloat16: 1}


def _exchange_id(rank: int, world: int, id_file: Optional[str]) -> bytes:
    buf = (C.c_char * 128)()
    if rank == 0:
        hip.call("stonk_comm_unique_id", C.addressof(buf))
    if world == 1:
        return bytes(buf)
    if id_file is None:
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("stonk comm: pass id_file=... or initialise torch.distributed (any backend) to carry the RCCL id")
        box = [bytes(buf) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    if rank == 0:
        tmp = id_file + ".tmp"
        with open(tmp, "wb") as f:
            f.write(bytes(buf))
        os.replace(tmp, id_file)
        return bytes(buf)
    t_end = time.time() + 120
    while not os.path.exists(id_file):
        if time.time() > t_end:
            raise TimeoutError(f"stonk comm: rank 0 did not publish {id_file}")
        time.sleep(0.05)
    with open(id_file, "rb") as f:
        return f.read()


class StonkComm:
    """One RCCL communicator per process on the library's own stream. Every `*_async` orders the collective behind what the
    CURRENT torch stream has enqueued and returns at once; `wait()` makes the current stream wait for everything issued."""

    def __init__(self, rank: int, world: int, device: int, id_file: Optional[str] = None):
        self.rank, self.world = rank, world
        uid = _exchange_id(rank, world, id_file)
        handle = C.c_void_p()
        hip.call("stonk_comm_init", C.byref(handle), world, rank, uid, device)
        self._h = handle

    @property
    def stream_ptr(self) -> int:
        return int(hip.lib().stonk_comm_stream(self._h) or 0)

    def all_reduce(self, t: torch.Tensor) -> None:
        hip.call("stonk_comm_allreduce_async", self._h, t.data_ptr(), t.numel(), DTYPES[t.dtype], hip.stream_ptr())

    def reduce_scatter(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype
        hip.call("stonk_comm_reduce_scatter_async", self._h, send.data_ptr(), recv.data_ptr(), recv.numel(), DTYPES[recv.dtype],
                 hip.stream_ptr())

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
        hip.call("stonk_comm_allgather_async", self._h, send.data_ptr(), recv.data_ptr(), send.numel(), DTYPES[send.dtype],
                 hip.stream_ptr())

    def wait(self) -> None:
        hip.call("stonk_comm_wait", self._h, hip.stream_ptr())

    def close(self) -> None:
        if self._h is not None:
            hip.call("stonk_comm_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Work:
    """What `GradSynchronizer` expects of a collective's handle: `wait()` orders the current stream behind it. A stonk
    communicator has ONE stream, so waiting for it waits for this collective and every earlier one."""

    def __init__(self, comm: StonkComm):
        self.comm = comm

    def wait(self) -> None:
        self.comm.wait()
