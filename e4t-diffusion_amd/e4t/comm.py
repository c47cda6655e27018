"""The library's own RCCL communicator (include/e4t_hip.h `e4t_comm_*`, csrc/comm.hip) as the trainer's collective back end.

The reference gets its gradient all-reduce from the DDP wrapper `accelerator.prepare` puts around the models (pretrain_e4t.py:410-412,
backward at :648).  The trainer's default is torch.distributed on the launcher's process group; `E4TTrainer(collectives="library")` routes
the gradient regions' all-reduces and the head's factor all-gather through this class instead — the same calls a host without
torch.distributed would make.  The 128-byte id is the only thing that has to cross between the ranks before `e4t_comm_init`;
`from_process_group` sends it over whatever group the launcher already made (any back end, it is a CPU object broadcast).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _C
from .ops import _stream

_DTYPES = {torch.float32: _C.COMM_F32, torch.bfloat16: _C.COMM_BF16}
_OPS = {"sum": _C.COMM_SUM, "avg": _C.COMM_AVG, "min": _C.COMM_MIN, "max": _C.COMM_MAX}


class _Handle:
    """what torch.distributed's async work object is to the trainer: wait() orders the CURRENT stream after the collective (no host wait)"""

    def __init__(self, comm):
        self.comm = comm

    def wait(self):
        self.comm.wait()


class LibraryComm:
    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError(f"LibraryComm: the id is {len(unique_id)} bytes, e4t_comm_unique_id makes 128")
        self._lib = _C.load()
        h = _C.vp()
        _C.check(self._lib.e4t_comm_init(C.byref(h), C.c_char_p(unique_id), int(rank), int(world)), "e4t_comm_init")
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _C.check(_C.load().e4t_comm_unique_id(buf), "e4t_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """rank 0 draws the id; it travels as an object broadcast over the launcher's group (torch.cuda.set_device done by the caller)"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world)

    def _check(self, t: torch.Tensor, what: str):
        if self._h is None:
            raise _C.E4TError(f"{what}: the communicator is closed")
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPES:
            raise ValueError(f"{what}: needs a contiguous fp32 / bf16 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")

    def all_reduce(self, t: torch.Tensor, op: str = "sum") -> _Handle:
        """in place; queued behind everything already on the current stream; returns at once"""
        self._check(t, "all_reduce")
        _C.check(self._lib.e4t_comm_allreduce(self._h, t.data_ptr(), t.numel(), _DTYPES[t.dtype], _OPS[op], _stream()), "e4t_comm_allreduce")
        return _Handle(self)

    def all_gather_into_tensor(self, out: torch.Tensor, inp: torch.Tensor) -> _Handle:
        self._check(out, "all_gather")
        self._check(inp, "all_gather")
        if out.dtype != inp.dtype or out.numel() != self.world * inp.numel():
            raise ValueError(f"all_gather: {tuple(out.shape)} {out.dtype} cannot hold {self.world} x {tuple(inp.shape)} {inp.dtype}")
        _C.check(self._lib.e4t_comm_allgather(self._h, inp.data_ptr(), out.data_ptr(), inp.numel(), _DTYPES[inp.dtype], _stream()), "e4t_comm_allgather")
        return _Handle(self)

    def wait(self):
        _C.check(self._lib.e4t_comm_wait(self._h, _stream()), "e4t_comm_wait")

    def stream_handle(self) -> int:
        s = _C.vp()
        _C.check(self._lib.e4t_comm_info(self._h, None, None, C.byref(s)), "e4t_comm_info")
        return int(s.value or 0)

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            _C.check(self._lib.e4t_comm_destroy(h), "e4t_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the driver may already be gone
            pass
