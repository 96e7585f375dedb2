"""Tensor-parallel process groups for the decode path: one process per GPU, RCCL over xGMI.

Mirrors the part of vllm/distributed/parallel_state.py the hot path uses
(init_distributed_environment :80-125, initialize_model_parallel :128-203, the accessors) with
one device group (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests) and one Gloo group
for host-side metadata.  Pipeline parallelism is rejected by the reference's own config
(vllm/config.py:606-608) and has no counterpart here.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP: Optional[dist.ProcessGroup] = None
_TP_CPU_GROUP: Optional[dist.ProcessGroup] = None
_TP_RANKS = None
_TP_RCCL = None          # RcclCommunicator (ctypes) of the TP group: the all-reduce under graph capture
_TP_XGMI = None          # XgmiAllReduce of the TP group, made on first use when enabled
_TP_XGMI_REFUSED = False  # made on request, failed its self-check: stay on RCCL


def init_distributed_environment(world_size: int = -1, rank: int = -1,
                                 distributed_init_method: str = "env://", local_rank: int = -1,
                                 backend: str = "nccl") -> None:
    """World group + a warm-up all-reduce (the reference does one to surface broken links
    early, parallel_state.py:107-125)."""
    if not dist.is_initialized():
        kw = {}
        if world_size >= 0:
            kw.update(world_size=world_size, rank=rank)
        dist.init_process_group(backend=backend, init_method=distributed_init_method, **kw)
    if backend == "nccl":
        if local_rank < 0:
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        t = torch.ones(1, device=f"cuda:{local_rank}")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert int(t.item()) == dist.get_world_size()


def initialize_model_parallel(tensor_model_parallel_size: int = 1, backend: Optional[str] = None) -> None:
    """Consecutive ranks form a TP group: [0..tp-1], [tp..2tp-1], ... (parallel_state.py:150-166)."""
    global _TP_GROUP, _TP_CPU_GROUP, _TP_RANKS, _TP_RCCL
    assert dist.is_initialized()
    world = dist.get_world_size()
    assert world % tensor_model_parallel_size == 0
    backend = backend or dist.get_backend()
    me = dist.get_rank()
    for i in range(world // tensor_model_parallel_size):
        ranks = list(range(i * tensor_model_parallel_size, (i + 1) * tensor_model_parallel_size))
        grp = dist.new_group(ranks, backend=backend)
        cpu_grp = dist.new_group(ranks, backend="gloo")
        if me in ranks:
            _TP_GROUP, _TP_CPU_GROUP, _TP_RANKS = grp, cpu_grp, ranks
    # A second, ctypes-level RCCL communicator over the same ranks, as the reference makes one
    # (parallel_state.py:191-194): torch.distributed's all_reduce cannot be trusted inside a graph
    # capture, a bare ncclAllReduce on the capturing stream can.  VLA_RCCL_CTYPES=0 leaves it out.
    if backend == "nccl" and tensor_model_parallel_size > 1 and torch.cuda.is_available() and \
            os.environ.get("VLA_RCCL_CTYPES", "1") != "0":
        from .rccl_communicator import RcclCommunicator
        _TP_RCCL = RcclCommunicator(_TP_CPU_GROUP, torch.device("cuda", torch.cuda.current_device()))


def model_parallel_is_initialized() -> bool:
    return _TP_GROUP is not None


def ensure_model_parallel_initialized(tensor_model_parallel_size: int, backend: Optional[str] = None) -> None:
    if not model_parallel_is_initialized():
        initialize_model_parallel(tensor_model_parallel_size, backend)
        return
    assert get_tensor_model_parallel_world_size() == tensor_model_parallel_size


def get_tensor_model_parallel_group():
    assert _TP_GROUP is not None, "tensor model parallel group is not initialized"
    return _TP_GROUP


def get_tensor_model_parallel_cpu_group():
    assert _TP_CPU_GROUP is not None
    return _TP_CPU_GROUP


def get_tensor_model_parallel_world_size() -> int:
    if _TP_GROUP is None:
        return 1
    return dist.get_world_size(group=_TP_GROUP)


def get_tensor_model_parallel_rank() -> int:
    if _TP_GROUP is None:
        return 0
    return dist.get_rank(group=_TP_GROUP)


def get_tensor_model_parallel_src_rank() -> int:
    return _TP_RANKS[0] if _TP_RANKS else 0


def get_tp_rccl_communicator():
    """The ctypes RCCL communicator of the TP group, or None (one rank, a CPU run, library not
    loadable: `reason` on the object says which)."""
    return _TP_RCCL if _TP_RCCL is not None and _TP_RCCL.available else None


def get_tp_xgmi_all_reduce(device: torch.device):
    """The peer-memory all-reduce of the TP group, or None when RCCL is to be used.  Either
    installed by the caller (set_tp_xgmi_all_reduce, after it has checked and timed it) or, with
    VLA_XGMI_ALLREDUCE=1, made on first use -- which every rank reaches together (the first
    all-reduce of a step); the reference builds its custom all-reduce the same lazy way
    (vllm/distributed/parallel_state.py:170-180 with custom_all_reduce.py:78-130)."""
    global _TP_XGMI, _TP_XGMI_REFUSED
    if _TP_XGMI is None and not _TP_XGMI_REFUSED and _TP_CPU_GROUP is not None and \
            os.environ.get("VLA_XGMI_ALLREDUCE", "0") == "1":
        # Set-up allocates, zeroes and synchronises: none of that may happen inside a stream
        # capture.  A first call made during capture keeps RCCL for the captured graph; the
        # communicator is built by the first call outside one (warm up before capturing).
        if torch.cuda.is_current_stream_capturing():
            return None
        from .xgmi_all_reduce import XgmiAllReduce
        comm = XgmiAllReduce(_TP_CPU_GROUP, device)
        if comm.disabled:
            raise RuntimeError(f"VLA_XGMI_ALLREDUCE=1 but set-up failed: {comm.reason}")
        # nobody has tried this communicator on this node: a short bit-exact check against the
        # sum of the gathered inputs, agreed on by all ranks, before any model output depends on it
        if comm.self_check():
            _TP_XGMI = comm
        else:
            comm.close()
            _TP_XGMI_REFUSED = True
            import warnings
            warnings.warn("VLA_XGMI_ALLREDUCE=1: the peer-memory all-reduce failed its self-check "
                          "on this node; staying on RCCL")
    return _TP_XGMI


def set_tp_xgmi_all_reduce(comm) -> None:
    """Install (or with None remove, without closing) the peer-memory all-reduce of the group."""
    global _TP_XGMI
    _TP_XGMI = comm


def destroy_model_parallel() -> None:
    global _TP_GROUP, _TP_CPU_GROUP, _TP_RANKS, _TP_XGMI, _TP_XGMI_REFUSED, _TP_RCCL
    _TP_XGMI_REFUSED = False
    if _TP_RCCL is not None:
        _TP_RCCL.close()
        _TP_RCCL = None
    if _TP_XGMI is not None:
        _TP_XGMI.close()
        _TP_XGMI = None
    for g in (_TP_GROUP, _TP_CPU_GROUP):
        if g is not None:
            dist.destroy_process_group(g)
    _TP_GROUP = _TP_CPU_GROUP = _TP_RANKS = None
