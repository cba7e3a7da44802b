"""One-shot all-reduce over peer-mapped buffers for the tensor-parallel decode step (C ABI gptqhip_allreduce_oneshot,
csrc/gptqhip_comm.hip).  One process per GPU; the IPC handles travel once through torch.distributed (any backend:
all_gather_object), afterwards a call is ONE kernel launch on the current stream -- capture-safe, no library collective.

    comm = OneShotAllReduce(n_max=hidden, device=dev, group=None)
    out = comm(partial_fp32, out_dtype=torch.float16, bias=None, residual=h)     # every rank, same sequence of calls

Used by RowParallelQuantLinear(..., comm=comm) for small messages (batch-1 decode: 32 KB at hidden 8192); larger ones keep
dist.all_reduce (RCCL spreads bandwidth-bound messages over the 7 xGMI links).

Trust model: the build / test boxes have ONE GPU (two processes share it through real IPC mappings in tests/test_gpu_comm.py),
so a communicator proves itself where it runs: `self_test()` pushes distinct payloads through the kernel -- single calls and a
back-to-back burst with no host synchronisation in between -- and compares every result BIT FOR BIT with the rank-ordered sum
obtained through the process group's own all_gather; callers (bench.py, bench_tp.py) fall back to dist.all_reduce when it
returns False.  A peer that never arrives poisons the output with NaN and sets the sticky status word (`check_status()`)."""
from __future__ import annotations

import ctypes
import os
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
        self._timeout_ms = int(os.environ.get("GPTQHIP_COMM_TIMEOUT_MS", "0") or 0) or 10000
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

    def gather_select(self, x_local: torch.Tensor, index: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One-shot all-gather of every rank's 16-bit vector x_local [n_local] (same length on all ranks) + select:
        out[j] = concat_r(x_r)[index[j]] (index int32 device tensor; None: the whole vector).  The input exchange of an act-order
        row-parallel shard (utils.tp.shard_gptq_row(act_order="global_sort")); one kernel, capture-safe."""
        if x_local.dtype not in (torch.float16, torch.bfloat16) or not x_local.is_cuda or not x_local.is_contiguous():
            raise RuntimeError("gather_select: x_local must be a contiguous fp16 / bf16 device tensor")
        n_local = x_local.numel()
        if n_local % 8 != 0 or n_local > self.n_max:
            raise RuntimeError(f"gather_select: {n_local} elements per rank (multiple of 8, max {self.n_max})")
        if index is not None and (index.dtype != torch.int32 or not index.is_cuda or not index.is_contiguous()):
            raise RuntimeError("gather_select: index must be a contiguous int32 device tensor")
        n_out = n_local * self.world if index is None else index.numel()
        if out is None:
            out = torch.empty(n_out, dtype=x_local.dtype, device=x_local.device)
        elif out.dtype != x_local.dtype or out.numel() < n_out or not out.is_contiguous():
            raise RuntimeError("gather_select: out must be a contiguous tensor of x_local's dtype with >= n_out elements")
        with torch.cuda.device(x_local.device):
            rc = _lib.load().gptqhip_allgather_select(x_local.data_ptr(), self._peers, self.rank, self.world, n_local, self.n_max,
                                                      0 if index is None else index.data_ptr(), n_out, out.data_ptr(),
                                                      ops._DT[x_local.dtype], ops._stream(x_local.device))
        _lib.check(rc, "gptqhip_allgather_select")
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def _gather(self, t: torch.Tensor):
        """All ranks' copies of device tensor t (through the process group, whatever its backend)."""
        if self.world == 1:
            return [t.clone()]
        if dist.get_backend(self.group) == "gloo":
            parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
            dist.all_gather(parts, t.cpu(), group=self.group)
            return [p.to(t.device) for p in parts]
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return parts

    def self_test(self, calls: int = 8, burst: int = 512, n: Optional[int] = None, timeout_ms: int = 1500) -> bool:
        """Validate this communicator on the hardware it runs on (collective: every rank calls it).  `calls` single
        all-reduces with fresh random payloads (+ residual) and one burst of `burst` back-to-back launches whose payloads change
        every epoch (no host synchronisation inside the burst: the device-side epoch / parity protocol is what is being tested),
        each compared bit for bit with the rank-ordered fp32 sum of the all_gather'ed inputs.  Returns True only if every rank
        saw every result right and no wait timed out.  `timeout_ms`: the peer-wait bound while testing (raise it when the ranks
        time-share ONE GPU, as the test-suite does: there every step costs a rotation of the GPU scheduler's time slices)."""
        dev = self.device
        n = int(n or self.n_max)
        n -= n % 16
        ok = True

        def healthy() -> bool:
            torch.cuda.synchronize(dev)
            st = ctypes.c_uint32(0)
            _lib.check(_lib.load().gptqhip_comm_status(self._own, ctypes.byref(st)), "gptqhip_comm_status")
            return st.value == 0

        def agree(local_ok: bool) -> bool:
            """MIN over the ranks: every rank leaves the test loop in the same iteration (a rank that saw a failure must not skip
            process-group collectives its healthy peers are still going to enter -- mismatched collectives hang)."""
            if self.world == 1:
                return local_ok
            flag = torch.tensor([1 if local_ok else 0], dtype=torch.int32)
            if dist.get_backend(self.group) != "gloo":
                flag = flag.to(dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(int(flag.item()) == 1)

        # The process-group collectives (all_gather / all_reduce) run in the SAME order on every rank whatever any rank observes:
        # failures -- a wrong result, a timed-out peer wait, an exception -- are only recorded, and the ranks agree on them once per
        # iteration.  (A one-directional link fault or a time-out on one rank only is the normal failure shape.)
        try:
            # short peer-wait bound while testing: a link that does not deliver must cost seconds, not calls x 10 s
            self.set_timeout_ms(int(timeout_ms))
        except Exception:  # noqa: BLE001
            ok = False
        # (a local set_timeout failure must be AGREED before anything below is skipped on its account: with calls == 0 the loop that
        # agrees once per iteration never runs, and one rank skipping the burst's all_gather would hang its peers)
        ok = agree(ok)
        with torch.cuda.device(dev):
            for it in range(calls if ok else 0):
                g = torch.Generator(device=dev)
                g.manual_seed(7919 * it + self.rank)
                part = torch.randn(n, device=dev, generator=g) * 3.0
                gr = torch.Generator(device=dev)
                gr.manual_seed(104729 + it)
                res = torch.randn(n, device=dev, generator=gr).to(torch.float16)
                xl = torch.randn(512, device=dev, generator=g).to(torch.float16)
                idx = torch.randperm(512 * self.world, device=dev, generator=gr)[:640].to(torch.int32)
                parts = self._gather(part)                  # process-group collectives: unconditional
                full = torch.cat(self._gather(xl))
                try:
                    want = parts[0].clone()
                    for p in parts[1:]:
                        want = want + p
                    want = (res.float() + want.to(torch.float16).float()).to(torch.float16)
                    got = self(part, out_dtype=torch.float16, residual=res)
                    it_ok = healthy() and bool(torch.equal(got, want))
                    sel = self.gather_select(xl, idx)
                    torch.cuda.synchronize(dev)
                    it_ok = it_ok and healthy() and bool(torch.equal(sel, full[idx.long()]))
                except Exception:  # noqa: BLE001 -- a failing self test must not take the caller down: it answers False
                    it_ok = False
                ok = agree(ok and it_ok)
                if not ok:
                    break
            # burst: payload of epoch t = base_r + t (exact in fp32 for these magnitudes); mismatches counted on the device
            g = torch.Generator(device=dev)
            g.manual_seed(31337 + self.rank)
            base = torch.randn(n, device=dev, generator=g).to(torch.float16).float()   # 11 significant bits: base + t is exact
            # `ok` was AGREED across the ranks by the loop above, so skipping the gather + the burst after a failure is collective-safe
            # (every rank skips): a lost peer must cost one bounded wait, not burst x timeout
            bases = self._gather(base) if ok else []
            try:
                bad = torch.zeros((), dtype=torch.int64, device=dev)
                part = torch.empty(n, device=dev)
                out = torch.empty(n, device=dev, dtype=torch.float16)
                for t in range(burst if ok else 0):
                    # (a timed-out peer wait: stop launching, every further call would wait again -- checked after the first launch,
                    # then every 16th)
                    if (t == 1 or t % 16 == 15) and not healthy():
                        ok = False
                        break
                    torch.add(base, float(t % 64), out=part)
                    self(part, out_dtype=torch.float16, out=out)
                    want = bases[0] + float(t % 64)
                    for b_ in bases[1:]:
                        want = want + (b_ + float(t % 64))
                    bad += (out != want.to(torch.float16)).sum()
                torch.cuda.synchronize(dev)
                ok = ok and int(bad.item()) == 0
                self.check_status()
            except Exception:  # noqa: BLE001
                ok = False
        try:
            self.set_timeout_ms(self._timeout_ms)
        except Exception:  # noqa: BLE001
            ok = False
        ok = agree(ok)
        return ok

    def set_timeout_ms(self, ms: int) -> None:
        """Bound of this rank's peer waits from now on (host write into the buffer header: call between launches)."""
        torch.cuda.synchronize(self.device)
        _lib.check(_lib.load().gptqhip_comm_set_timeout(self._own, int(ms)), "gptqhip_comm_set_timeout")

    def check_status(self) -> None:
        st = ctypes.c_uint32(0)
        _lib.check(_lib.load().gptqhip_comm_status(self._own, ctypes.byref(st)), "gptqhip_comm_status")
        if st.value & 2:
            raise RuntimeError("OneShotAllReduce.gather_select: an index outside [0, n_local * world) was poisoned with NaN")
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
