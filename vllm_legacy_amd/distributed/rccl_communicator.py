"""RCCL through ctypes: an all-reduce that is one library call on a stream the caller names.

The role of the reference's PyNcclCommunicator and its wrapper
(vllm/distributed/device_communicators/pynccl.py:18-185, pynccl_wrapper.py:128-175): during
graph capture `torch.distributed.all_reduce` makes HIP calls of its own that a capture does not
allow, so the hot all-reduce of the decode step goes straight to `ncclAllReduce` (RCCL exports
the NCCL names) on the capturing stream instead (communication_op.py:85-100 of the reference
switches the same way).  The communicator is separate from torch's process group: rank 0 makes a
unique id, a CPU (Gloo) group broadcasts its 128 bytes, every rank calls ncclCommInitRank.

Library: VLLM_NCCL_SO_PATH if set (the reference's variable, vllm/envs.py:9), else the
`librccl.so` that ships inside torch (the one torch's own process group uses), else
`librccl.so.1` from the loader path (vllm/utils.py:596-617).

What a one-GPU machine can show of this: loading, the failure paths, and a communicator of one
rank (all-reduce = copy) issued eagerly and from a captured graph.  More ranks need more GPUs:
RCCL refuses two ranks on one device.
"""
import ctypes as C
import os
from contextlib import contextmanager
from typing import Optional

import torch
import torch.distributed as dist

ncclSuccess = 0
# ncclDataType_t / ncclRedOp_t (nccl.h)
_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6,
           torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
_OPS = {dist.ReduceOp.SUM: 0, dist.ReduceOp.PRODUCT: 1, dist.ReduceOp.MAX: 2, dist.ReduceOp.MIN: 3,
        dist.ReduceOp.AVG: 4}


class ncclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def find_rccl_library() -> str:
    path = os.environ.get("VLLM_NCCL_SO_PATH")
    if path:
        return path
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return bundled if os.path.exists(bundled) else "librccl.so.1"


class RcclLibrary:
    """The handful of entry points the decode path needs, with argument types set."""

    def __init__(self, so_file: Optional[str] = None):
        self.path = so_file or find_rccl_library()
        L = C.CDLL(self.path)
        p, i, sz = C.c_void_p, C.c_int, C.c_size_t
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [i]
        L.ncclGetVersion.argtypes = [C.POINTER(i)]
        L.ncclGetUniqueId.argtypes = [C.POINTER(ncclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(p), i, ncclUniqueId, i]
        L.ncclAllReduce.argtypes = [p, p, sz, i, i, p, p]
        L.ncclCommDestroy.argtypes = [p]
        self.L = L

    def check(self, rc: int, what: str) -> None:
        if rc != ncclSuccess:
            raise RuntimeError(f"RCCL error in {what}: {self.L.ncclGetErrorString(rc).decode()}")

    def version(self) -> str:
        v = C.c_int()
        self.check(self.L.ncclGetVersion(C.byref(v)), "ncclGetVersion")
        major, rest = divmod(v.value, 10000)
        return f"{major}.{rest // 100}.{rest % 100}"


class RcclCommunicator:

    def __init__(self, group=None, device=None, library_path: Optional[str] = None,
                 allow_single_rank: bool = False):
        """group: a non-NCCL process group over the ranks (it carries the unique id); device: the
        GPU of this rank.  `available` is False (and nothing raises) when the library cannot be
        loaded or the group has one rank -- unless allow_single_rank, which builds a one-rank
        communicator (tests on a one-GPU machine)."""
        self.available = False
        self.disabled = True
        self.stream = None
        self.comm = C.c_void_p()
        self.reason = ""
        if group is None and dist.is_initialized():
            group = dist.group.WORLD
        if group is not None:
            assert dist.get_backend(group) != dist.Backend.NCCL, \
                "RcclCommunicator should be attached to a non-NCCL group."
            self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self.group = group
        if self.world_size == 1 and not allow_single_rank:
            self.reason = "single rank"
            return
        uid = ncclUniqueId()
        try:
            self.rccl = RcclLibrary(library_path)
            self.version = self.rccl.version()
            if self.rank == 0:
                self.rccl.check(self.rccl.L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        except Exception as e:  # noqa: BLE001  (missing file, missing symbol: stay unavailable)
            self.reason = f"{type(e).__name__}: {e}"
        # The constructor is collective from here on (broadcast, ncclCommInitRank): whether it goes
        # on must be one decision for the whole group, or the ranks that loaded the library would
        # wait in the broadcast for the one that did not.
        if not self._all_ranks(not self.reason):
            self.reason = self.reason or "the RCCL library could not be loaded on another rank"
            return
        if self.world_size > 1:
            t = torch.tensor(list(uid.internal), dtype=torch.uint8)
            dist.broadcast(t, src=dist.get_process_group_ranks(group)[0], group=group)
            for i, b in enumerate(t.tolist()):
                uid.internal[i] = b if b < 128 else b - 256
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            rc = self.rccl.L.ncclCommInitRank(C.byref(self.comm), self.world_size, uid, self.rank)
            if not self._all_ranks(rc == ncclSuccess):
                self.reason = ("ncclCommInitRank failed" + (
                    f": {self.rccl.L.ncclGetErrorString(rc).decode()}" if rc != ncclSuccess
                    else " on another rank"))
                self.close()
                return
            self.stream = torch.cuda.Stream()
            self.available = True
            self.disabled = False
            # warm-up, as the reference does: the first collective sets up the channels
            data = torch.zeros(1, device=self.device)
            self.all_reduce(data, stream=self.stream)
            self.stream.synchronize()
        # like the reference's, the communicator is switched on only where it is wanted
        # (change_state around a graph capture)
        self.disabled = True

    def _all_ranks(self, flag: bool) -> bool:
        """True when `flag` holds on every rank of the group (one Gloo all-reduce)."""
        if self.group is None or self.world_size == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    @staticmethod
    def supports(dtype: torch.dtype) -> bool:
        """Whether all_reduce takes tensors of this dtype (callers fall back to torch.distributed
        for the others instead of failing in the middle of a capture)."""
        return dtype in _DTYPES

    def all_reduce(self, tensor: torch.Tensor, op=dist.ReduceOp.SUM, stream=None) -> None:
        """In place, on `stream` (default: the communicator's own stream)."""
        if self.disabled:
            return
        assert tensor.device == self.device, (
            f"this communicator works on {self.device}, but the tensor is on {tensor.device}")
        assert tensor.is_contiguous()
        if stream is None:
            stream = self.stream
        self.rccl.check(self.rccl.L.ncclAllReduce(
            tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), _DTYPES[tensor.dtype], _OPS[op],
            self.comm, stream.cuda_stream), "ncclAllReduce")

    @contextmanager
    def change_state(self, enable: Optional[bool] = None, stream: Optional[torch.cuda.Stream] = None):
        """Temporarily enable (or disable) the communicator and point it at `stream`
        (pynccl.py:163-185)."""
        if enable is None:
            enable = self.available
        old = (self.disabled, self.stream)
        self.disabled, self.stream = not (enable and self.available), stream or self.stream
        try:
            yield
        finally:
            self.disabled, self.stream = old

    def close(self) -> None:
        if self.comm and getattr(self, "rccl", None) is not None:
            self.rccl.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
        self.available, self.disabled = False, True
