"""Collectives of the tensor-parallel decode path.

Mirrors vllm/distributed/communication_op.py:73-100 (all-reduce), :103-127 (all-gather),
:130-160 (gather), :227-317 (broadcast_tensor_dict).  On ROCm the reference ends up in
torch.distributed / RCCL for every one of these (its custom all-reduce is CUDA-only,
device_communicators/custom_all_reduce.py:28-31); so does this module by default, with the
peer-memory all-reduce of xgmi_all_reduce.py as an opt-in for the hot one.  The message of the hot
all-reduce is the [tokens, hidden] partial output of a row-parallel linear (2 MiB for
Llama-3-8B at 256 tokens), issued in place on the compute stream.
"""
import os
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from .parallel_state import (get_tp_rccl_communicator, get_tp_xgmi_all_reduce,
                             get_tensor_model_parallel_cpu_group,
                             get_tensor_model_parallel_group,
                             get_tensor_model_parallel_rank,
                             get_tensor_model_parallel_src_rank,
                             get_tensor_model_parallel_world_size)


def tensor_model_parallel_takes_slabs(m: int, n: int, dtype: torch.dtype, device) -> bool:
    """Whether tensor_model_parallel_all_reduce can be handed the unreduced split-K slabs
    (ops.SplitKSlabs) of an [m, n] product: a single rank passes them through to a fused consumer;
    several ranks need the peer-memory all-reduce, which sums them while it publishes them."""
    if get_tensor_model_parallel_world_size() == 1:
        return True
    xgmi = get_tp_xgmi_all_reduce(device)
    return xgmi is not None and xgmi.takes_slabs(m, n, dtype)


# VLA_XGMI_FUSE_NORM=0 (or setting this to False) keeps the all-reduce and the add + RMSNorm after
# it as separate launches even when the peer-memory all-reduce could do both.
FUSE_NORM = os.environ.get("VLA_XGMI_FUSE_NORM", "1") != "0"


class PendingAllReduce:
    """The output of a row-parallel linear (a tensor, or ops.SplitKSlabs) whose all-reduce has
    been left to the add + RMSNorm that consumes it (layers.RMSNorm accepts one)."""

    def __init__(self, value):
        self.value = value


def tensor_model_parallel_fuses_norm(m: int, n: int, dtype: torch.dtype, device) -> bool:
    """Whether an [m, n] row-parallel output may be wrapped in PendingAllReduce: several ranks
    and a peer-memory all-reduce that takes the message together with the normalisation."""
    if not FUSE_NORM or get_tensor_model_parallel_world_size() == 1:
        return False
    xgmi = get_tp_xgmi_all_reduce(device)
    return xgmi is not None and xgmi.takes_add_rms_norm(m, n, dtype)


def tensor_model_parallel_all_reduce_add_rms_norm(pending: PendingAllReduce, residual: torch.Tensor,
                                                  weight: torch.Tensor, epsilon: float):
    """all-reduce, residual += sum, RMSNorm(residual) * weight; returns (normed, residual).  One
    launch on the peer-memory all-reduce, the separate steps otherwise."""
    from .. import _custom_ops as ops
    x = pending.value
    m, n = residual.shape
    if tensor_model_parallel_fuses_norm(m, n, residual.dtype, residual.device):
        xgmi = get_tp_xgmi_all_reduce(residual.device)
        return xgmi.all_reduce_add_rms_norm(x, residual, weight, epsilon), residual
    hidden = tensor_model_parallel_all_reduce(x, residual.dtype)
    ops.fused_add_rms_norm(hidden, residual, weight, epsilon)
    return hidden, residual


def tensor_model_parallel_poll_all_reduce_error(device) -> None:
    """Once per step, right before the step's own synchronisation point (reading the sampled
    tokens): enqueue the peer-memory all-reduce's 4-byte error read.  No-op on RCCL."""
    if get_tensor_model_parallel_world_size() > 1:
        xgmi = get_tp_xgmi_all_reduce(device)
        if xgmi is not None:
            xgmi.poll_error_async()


def tensor_model_parallel_check_all_reduce(device) -> bool:
    """After that synchronisation point, collectively: False when some rank's peer-memory
    all-reduce lost a barrier (its outputs of this step are NaN on that rank).  The communicator
    is then closed on every rank and later steps run on RCCL; the caller repeats the step."""
    if get_tensor_model_parallel_world_size() == 1:
        return True
    xgmi = get_tp_xgmi_all_reduce(device)
    if xgmi is None:
        return True
    bad = torch.tensor([xgmi.polled_error()], dtype=torch.int32)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=get_tensor_model_parallel_cpu_group())
    if int(bad.item()) == 0:
        return True
    from .parallel_state import set_tp_xgmi_all_reduce
    set_tp_xgmi_all_reduce(None)
    xgmi.close()
    return False


def tensor_model_parallel_all_reduce(input_, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """SUM over the TP group, in place; returns its argument.  Unreduced split-K slabs (see
    tensor_model_parallel_takes_slabs) come back as the finished [m, n] tensor of `dtype` when
    there is more than one rank."""
    if get_tensor_model_parallel_world_size() == 1:
        return input_                     # single rank (possibly unreduced split-K slabs)
    if not isinstance(input_, torch.Tensor):
        xgmi = get_tp_xgmi_all_reduce(input_.workspace.device)
        if xgmi is not None and xgmi.takes_slabs(input_.m, input_.n, dtype):
            return xgmi.all_reduce_slabs(input_, dtype)
        input_ = input_.reduce(dtype)     # a caller that did not ask first: sum here, then as usual
    group = get_tensor_model_parallel_group()
    if input_.is_cuda:
        # opt-in: direct loads/stores over xGMI peer memory instead of an RCCL call
        # (communication_op.py:87-95 tries the custom all-reduce first in the same way)
        xgmi = get_tp_xgmi_all_reduce(input_.device)
        if xgmi is not None and xgmi.should_custom_ar(input_):
            return xgmi.all_reduce(input_, out=input_)
    if input_.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the GPU code path without RCCL (several ranks on one GPU): stage
        # through the host, where Gloo is at home
        host = input_.cpu()
        dist.all_reduce(host, group=group)
        input_.copy_(host)
        return input_
    if input_.is_cuda and input_.is_contiguous():
        # under graph capture (or always, with VLA_RCCL_CTYPES=always): one ncclAllReduce on the
        # current stream through the ctypes communicator, as the reference does while capturing
        # (communication_op.py:85-100 with pynccl.py:109-127)
        rccl = get_tp_rccl_communicator()
        if rccl is not None and rccl.supports(input_.dtype) and (
                _RCCL_CTYPES_ALWAYS or torch.cuda.is_current_stream_capturing()):
            with rccl.change_state(True, torch.cuda.current_stream(input_.device)):
                rccl.all_reduce(input_)
            return input_
    dist.all_reduce(input_, group=group)
    return input_


_RCCL_CTYPES_ALWAYS = os.environ.get("VLA_RCCL_CTYPES", "1") == "always"


def all_reduce_implementation(device, capturing: bool) -> str:
    """Which implementation tensor_model_parallel_all_reduce takes for a GPU tensor on `device`:
    "xgmi" (peer memory), "rccl-ctypes" (ncclAllReduce through ctypes; under capture unless
    VLA_RCCL_CTYPES=always), "rccl-torch" (torch.distributed), "gloo-staged" (rehearsal) or
    "none" (one rank).  For result lines and logs."""
    if get_tensor_model_parallel_world_size() == 1:
        return "none"
    if get_tp_xgmi_all_reduce(device) is not None:
        return "xgmi"
    if dist.get_backend(get_tensor_model_parallel_group()) == "gloo":
        return "gloo-staged"
    if get_tp_rccl_communicator() is not None and (capturing or _RCCL_CTYPES_ALWAYS):
        return "rccl-ctypes"
    return "rccl-torch"


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    world = get_tensor_model_parallel_world_size()
    if world == 1:
        return input_
    if dim < 0:
        dim += input_.dim()
    size = input_.size()
    # flat [world * size0, ...] output: the layout every backend's all_gather_into_tensor accepts
    group = get_tensor_model_parallel_group()
    if input_.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal without RCCL, see all-reduce above
        host = input_.contiguous().cpu()
        out = torch.empty((world * size[0], ) + tuple(size[1:]), dtype=input_.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        out = out.to(input_.device)
    else:
        out = torch.empty((world * size[0], ) + tuple(size[1:]), dtype=input_.dtype,
                          device=input_.device)
        dist.all_gather_into_tensor(out, input_.contiguous(), group=group)
    out = out.view((world, ) + tuple(size)).movedim(0, dim)
    return out.reshape(size[:dim] + (world * size[dim], ) + size[dim + 1:])


def tensor_model_parallel_gather(input_: torch.Tensor, dst: int = 0, dim: int = -1) -> Optional[torch.Tensor]:
    """Concatenate along `dim` on rank `dst` of the group; None elsewhere (logits gather,
    vllm/model_executor/layers/logits_processor.py:68)."""
    world = get_tensor_model_parallel_world_size()
    if world == 1:
        return input_
    if dim < 0:
        dim += input_.dim()
    me = get_tensor_model_parallel_rank()
    group = get_tensor_model_parallel_group()
    src = input_.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        src = src.cpu()                   # rehearsal without RCCL, see all-reduce above
    bufs = [torch.empty_like(src) for _ in range(world)] if me == dst else None
    dist.gather(src, bufs, dst=get_tensor_model_parallel_src_rank() + dst, group=group)
    return torch.cat(bufs, dim=dim).to(input_.device) if me == dst else None


def broadcast_tensor_dict(tensor_dict: Optional[Dict[str, Any]] = None, src: int = 0) -> Dict[str, Any]:
    """Driver -> workers broadcast of the per-step metadata (block tables, slot mapping,
    sequence lengths): python values and tensor shapes travel over Gloo, tensor payloads over
    the device group."""
    world = get_tensor_model_parallel_world_size()
    if world == 1:
        return tensor_dict or {}
    group, cpu_group = get_tensor_model_parallel_group(), get_tensor_model_parallel_cpu_group()
    src_global = get_tensor_model_parallel_src_rank() + src
    me = get_tensor_model_parallel_rank()
    if me == src:
        assert isinstance(tensor_dict, dict)
        meta = [(k, ("t", v.dtype, tuple(v.shape), v.device.type)) if isinstance(v, torch.Tensor)
                else (k, ("v", v)) for k, v in tensor_dict.items()]
        dist.broadcast_object_list([meta], src=src_global, group=cpu_group)
        for k, v in tensor_dict.items():
            if isinstance(v, torch.Tensor) and v.numel():
                dist.broadcast(v, src=src_global, group=group if v.is_cuda else cpu_group)
        return tensor_dict
    box = [None]
    dist.broadcast_object_list(box, src=src_global, group=cpu_group)
    out: Dict[str, Any] = {}
    for k, spec in box[0]:
        if spec[0] == "v":
            out[k] = spec[1]
            continue
        _, dtype, shape, devtype = spec
        dev = torch.device("cuda", torch.cuda.current_device()) if devtype == "cuda" else "cpu"
        t = torch.empty(shape, dtype=dtype, device=dev)
        if t.numel():
            dist.broadcast(t, src=src_global, group=group if t.is_cuda else cpu_group)
        out[k] = t
    return out
