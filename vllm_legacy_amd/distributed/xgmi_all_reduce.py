"""All-reduce through xGMI peer memory for the tensor-parallel group of one node.

The counterpart of vllm/distributed/device_communicators/custom_all_reduce.py:78-303
(`CustomAllreduce`), which the reference disables on ROCm (:28-31).  Same life cycle and the
same question/answer interface (`should_custom_ar`, `custom_all_reduce` returning None when the
message is not taken), but there is nothing to register: the kernel stages its input itself, so
eager calls and graph capture share one code path and `capture()` only exists for call-site
compatibility.

Set-up exchanges one 64-byte IPC handle per rank over a CPU (Gloo) group; a failure on any rank
disables the communicator on all of them (`disabled`, `reason`).  communication_op uses it when
one has been installed in parallel_state (after the caller checked and timed it against RCCL) or
when VLA_XGMI_ALLREDUCE=1 asks for it; RCCL is the default.
"""
import ctypes as C
import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _lib

_SUPPORTED_WORLD_SIZES = (2, 3, 4, 5, 6, 7, 8)


class XgmiAllReduce:

    def __init__(self, group, device: torch.device, max_size: int = 8192 * 1024,
                 num_blocks: Optional[int] = None) -> None:
        """group: a non-NCCL process group spanning the ranks (one GPU each, one node);
        max_size: largest message in bytes (the reference's default, custom_all_reduce.py:86)."""
        self.disabled = True
        self.reason = "single rank or unsupported world size"
        self._ctx = C.c_void_p()
        self._region = C.c_void_p()
        self._peers: List[Optional[int]] = []
        self._err_host = None
        self.group = group
        self.device = torch.device(device)
        self.max_size = int(max_size)
        assert dist.get_backend(group) != dist.Backend.NCCL, \
            "XgmiAllReduce exchanges handles over a CPU group"
        self.rank = dist.get_rank(group=group)
        self.world_size = dist.get_world_size(group=group)
        if self.world_size == 1 or self.world_size not in _SUPPORTED_WORLD_SIZES:
            return
        if num_blocks is None:
            num_blocks = int(os.environ.get("VLA_XGMI_BLOCKS", "32"))
        L = _lib.lib()
        self.reason = ""
        # A step that fails on one rank must not leave the others waiting in a collective: every
        # rank goes through the same exchanges and all agree on the outcome at the end.
        handle = (C.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            try:
                _lib.check(L.vla_ar_alloc_region(self.max_size, C.byref(self._region)), "ar_alloc_region")
                _lib.check(L.vla_ar_export(self._region, handle), "ar_export")
            except Exception as e:  # noqa: BLE001
                self.reason = f"rank {self.rank}: {e}"
            # the device ordinal travels with the handle; it only means something to a peer that
            # sees the same set of devices
            index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            tag = bytes([index & 0xff, torch.cuda.device_count() & 0xff])
            handles = self._exchange(bytes(handle) + tag)
            if not self.reason:
                try:
                    regions = (C.c_void_p * self.world_size)()
                    for r, h in enumerate(handles):
                        if r == self.rank:
                            regions[r] = self._region.value
                            self._peers.append(None)
                        else:
                            peer = C.c_void_p()
                            buf = (C.c_ubyte * 64).from_buffer_copy(h[:64])
                            peer_dev = h[64] if h[65] == torch.cuda.device_count() else -1
                            _lib.check(L.vla_ar_import(buf, peer_dev, C.byref(peer)), "ar_import")
                            regions[r] = peer.value
                            self._peers.append(peer.value)
                    _lib.check(L.vla_ar_create(C.byref(self._ctx), self.rank, self.world_size,
                                               regions, self.max_size, num_blocks), "ar_create")
                except Exception as e:  # noqa: BLE001
                    self.reason = f"rank {self.rank}: {e}"
        # every region is zeroed and mapped before anybody launches; and everybody made it
        ok = torch.tensor([0 if self.reason else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            self.reason = self.reason or "set-up failed on another rank"
            self._release()
            return
        self.disabled = False

    def _exchange(self, mine: bytes) -> List[bytes]:
        """All-gather of the IPC handles as byte tensors (object collectives misbehave under
        inference mode with Gloo, custom_all_reduce.py:232-236)."""
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8).clone()
        out = [torch.empty(len(mine), dtype=torch.uint8) for _ in range(self.world_size)]
        dist.all_gather(out, t, group=self.group)
        return [bytes(o.numpy().tobytes()) for o in out]

    @contextmanager
    def capture(self):
        yield

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """custom_all_reduce.cuh:442-449 / custom_all_reduce.py:253-255: whole 16-byte units, a
        dense tensor, at most max_size bytes."""
        nbytes = inp.numel() * inp.element_size()
        return (not self.disabled and inp.is_cuda and inp.is_contiguous() and nbytes % 16 == 0
                and nbytes <= self.max_size
                and inp.dtype in (torch.float16, torch.bfloat16, torch.float32))

    def all_reduce(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(inp)
        code = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}[inp.dtype]
        with torch.cuda.device(inp.device):
            _lib.check(_lib.lib().vla_ar_all_reduce(
                self._ctx, out.data_ptr(), inp.data_ptr(), inp.numel(), code,
                torch.cuda.current_stream(inp.device).cuda_stream), "ar_all_reduce")
        return out

    def takes_slabs(self, m: int, n: int, dtype: torch.dtype) -> bool:
        """Whether all_reduce_slabs can take the unreduced [m, n] product of an int4 GEMM."""
        nbytes = m * n * 2
        return (not self.disabled and dtype in (torch.float16, torch.bfloat16)
                and nbytes % 16 == 0 and nbytes <= self.max_size)

    def all_reduce_slabs(self, slabs, dtype: torch.dtype = torch.float16) -> torch.Tensor:
        """slabs: ops.SplitKSlabs of this rank's partial product.  Same bits as all_reduce of
        slabs.reduce(dtype), without the reduction launch."""
        assert self.takes_slabs(slabs.m, slabs.n, dtype)
        ws = slabs.workspace
        out = torch.empty((slabs.m, slabs.n), dtype=dtype, device=ws.device)
        numel = slabs.m * slabs.n
        code = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dtype]
        with torch.cuda.device(ws.device):
            _lib.check(_lib.lib().vla_ar_all_reduce_slabs(
                self._ctx, out.data_ptr(), ws.data_ptr(), slabs.num_slabs, numel, numel, code,
                torch.cuda.current_stream(ws.device).cuda_stream), "ar_all_reduce_slabs")
        return out

    def takes_add_rms_norm(self, m: int, n: int, dtype: torch.dtype) -> bool:
        """Whether all_reduce_add_rms_norm can take an [m, n] message."""
        return (self.takes_slabs(m, n, dtype) and n % 8 == 0 and (n // 8) % self.world_size == 0
                and n // 8 <= 1024)

    def all_reduce_add_rms_norm(self, x, residual: torch.Tensor, weight: torch.Tensor,
                                epsilon: float) -> torch.Tensor:
        """x: this rank's [m, n] partial output or its ops.SplitKSlabs.  residual is updated in
        place with sum + residual; returns RMSNorm(residual) * weight."""
        slabs = None if isinstance(x, torch.Tensor) else x
        m, n = (x.shape if slabs is None else (slabs.m, slabs.n))
        dtype = residual.dtype
        assert self.takes_add_rms_norm(m, n, dtype) and residual.is_contiguous()
        assert tuple(residual.shape) == (m, n) and weight.dtype == dtype and weight.numel() == n
        if slabs is None:
            assert x.dtype == dtype and x.is_contiguous()
            src, ns = x, 0
        else:
            src, ns = slabs.workspace, slabs.num_slabs
        out = torch.empty_like(residual)
        code = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dtype]
        with torch.cuda.device(residual.device):
            _lib.check(_lib.lib().vla_ar_all_reduce_add_rms_norm(
                self._ctx, out.data_ptr(), src.data_ptr(), ns, m * n, residual.data_ptr(),
                weight.data_ptr(), float(epsilon), m, n, code,
                torch.cuda.current_stream(residual.device).cuda_stream), "ar_all_reduce_add_rms_norm")
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        if self.disabled or not self.should_custom_ar(input):
            return None
        return self.all_reduce(input)

    def poll_error_async(self) -> None:
        """Enqueue a 4-byte copy of the timeout counter to pinned host memory on the current
        stream.  Meant to be called once per step, right before a point where the caller
        synchronises anyway (reading the sampled tokens); `polled_error()` then costs nothing."""
        if self._err_host is None:
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vla_ar_error_count_async(
                self._ctx, self._err_host.data_ptr(),
                torch.cuda.current_stream(self.device).cuda_stream), "ar_error_count_async")

    def polled_error(self) -> int:
        """The counter as of the last poll_error_async() the device has completed (no
        synchronisation).  Non-zero: a barrier wait timed out on this rank; every output since is
        NaN and the communicator must be closed on all ranks."""
        return 0 if self._err_host is None else int(self._err_host.item())

    def self_check(self, calls: int = 6) -> bool:
        """Collective.  A few all-reduces of seeded random messages against the fp32 rank-order
        sum of the inputs gathered over the CPU group, bit for bit, and no barrier timeout; the
        answer is agreed on by all ranks.  Run before trusting a communicator that nobody has
        tried on this node."""
        ok = not self.disabled
        if ok:
            g = torch.Generator(device=self.device).manual_seed(4321 + self.rank)
            for it in range(calls):
                n = 8 * (1 + it) if it % 2 else min(self.max_size // 2, 256 * 4096)
                x = torch.empty(n, dtype=torch.float16, device=self.device).uniform_(-1, 1, generator=g)
                parts = [torch.empty(n, dtype=torch.float16) for _ in range(self.world_size)]
                dist.all_gather(parts, x.cpu(), group=self.group)
                want = torch.zeros(n, dtype=torch.float32)
                for part in parts:
                    want += part.float()
                got = self.all_reduce(x)
                ok = ok and torch.equal(got.cpu(), want.half())
            ok = ok and self.error_count() == 0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def error_count(self) -> int:
        """Barrier waits that timed out on this rank (synchronises the device)."""
        n = C.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vla_ar_error_count(self._ctx, C.byref(n)), "ar_error_count")
        return n.value

    def _release(self) -> None:
        L = _lib.lib()
        with torch.cuda.device(self.device):
            if self._ctx:
                L.vla_ar_destroy(self._ctx)
            for peer in self._peers:
                if peer is not None:
                    L.vla_ar_unimport(C.c_void_p(peer))
            # nobody frees a region a peer still has mapped
            dist.barrier(group=self.group)
            if self._region:
                L.vla_ar_free_region(self._region)
        self._ctx = C.c_void_p()
        self._region = C.c_void_p()
        self._peers = []
        self.disabled = True

    def close(self) -> None:
        """Collective: every rank of the group calls it."""
        if self._ctx:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
            # nobody unmaps a region a slower peer may still be reading
            dist.barrier(group=self.group)
            self._release()

    def __del__(self):
        # explicit close() is the orderly path; never run collectives from a finaliser
        pass
