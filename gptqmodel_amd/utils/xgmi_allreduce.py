"""One-shot all-reduce over peer-mapped buffers for the tensor-parallel decode step (C ABI gptqhip_allreduce_oneshot,
csrc/gptqhip_comm.hip).  One process per GPU; the IPC handles travel once through torch.distributed (any backend:
all_gather_object), afterwards a call is ONE kernel launch on the current stream -- capture-safe, no library collective.

    comm = OneShotAllReduce(n_max=hidden, device=dev, group=None)
    out = comm(partial_fp32, out_dtype=torch.float16, bias=None, residual=h)     # every rank, same sequence of calls

Used by RowParallelQuantLinear(..., comm=comm) for small messages (batch-1 decode: 32 KB at hidden 8192); larger ones keep
dist.all_reduce (RCCL spreads bandwidth-bound messages over the 7 xGMI links).  NOT yet run across physical GPUs."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib, ops


class OneShotAllReduce:
    def __init__(self, n_max: int, device: torch.device, group: Optional[dist.ProcessGroup] = None):
        lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device
        self.n_max = int(n_max)
        nbytes = lib.gptqhip_comm_bytes(self.world, self.n_max)
        if nbytes == 0:
            raise ValueError(f"OneShotAllReduce: unsupported world={self.world} / n_max={n_max} (world <= 8, n_max <= 65536)")
        own = ctypes.c_void_p(0)
        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(device):
            _lib.check(lib.gptqhip_comm_alloc(nbytes, ctypes.byref(own), handle), "gptqhip_comm_alloc")
        self._own = own.value
        handles = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(handles, handle.raw, group=group)
        else:
            handles[0] = handle.raw
        self._peers = (ctypes.c_void_p * self.world)()
        self._opened = []
        for r, h in enumerate(handles):
            if r == self.rank:
                self._peers[r] = self._own
                continue
            p = ctypes.c_void_p(0)
            with torch.cuda.device(device):
                _lib.check(lib.gptqhip_comm_open(ctypes.create_string_buffer(h, 64), ctypes.byref(p)), "gptqhip_comm_open")
            self._peers[r] = p.value
            self._opened.append(p.value)
        if self.world > 1:
            dist.barrier(group=group)   # nobody pushes before everybody has mapped everybody

    def __call__(self, partial: torch.Tensor, out_dtype: torch.dtype = torch.float16, bias: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if partial.dtype != torch.float32 or not partial.is_cuda or not partial.is_contiguous():
            raise RuntimeError("OneShotAllReduce: partial must be a contiguous float32 device tensor")
        n = partial.numel()
        if n > self.n_max or n % 4 != 0:
            raise RuntimeError(f"OneShotAllReduce: {n} elements (max {self.n_max}, multiple of 4)")
        if out is None:
            out = torch.empty(partial.shape, dtype=out_dtype, device=partial.device)
        for t, what in ((bias, "bias"), (residual, "residual")):
            if t is not None and (t.dtype != out.dtype or t.numel() < n or not t.is_contiguous()):
                raise RuntimeError(f"OneShotAllReduce: {what} must be a contiguous {out.dtype} tensor with >= {n} elements")
        if stats_out is not None and (stats_out.dtype != torch.float32 or stats_out.numel() < -(-n // 16) or n % 16 != 0):
            raise RuntimeError("OneShotAllReduce: stats_out must be float32 with >= n/16 elements (n a multiple of 16)")
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(partial.device):
            rc = _lib.load().gptqhip_allreduce_oneshot(partial.data_ptr(), self._peers, self.rank, self.world, n, self.n_max, p(bias),
                                                       p(residual), out.data_ptr(), p(stats_out), ops._DT[out.dtype],
                                                       ops._stream(partial.device))
        _lib.check(rc, "gptqhip_allreduce_oneshot")
        return out

    def check_status(self) -> None:
        st = ctypes.c_uint32(0)
        _lib.check(_lib.load().gptqhip_comm_status(self._own, ctypes.byref(st)), "gptqhip_comm_status")
        if st.value != 0:
            raise RuntimeError("OneShotAllReduce: a bounded wait for a peer timed out; results are invalid")

    def close(self) -> None:
        lib = _lib.load()
        torch.cuda.synchronize(self.device)
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        for p in self._opened:
            lib.gptqhip_comm_close(p)
        self._opened = []
        if self._own:
            lib.gptqhip_comm_free(self._own)
            self._own = 0


__all__ = ["OneShotAllReduce"]
