"""The elementwise layers on either side of attention, on the gfx950 library.

Host-side mirrors of the reference's RMSNorm (vllm/model_executor/layers/layernorm.py:11-71),
RotaryEmbedding (vllm/model_executor/layers/rotary_embedding.py:47-170: cos/sin cache in the
model dtype, [max_position, rot_dim], cos half then sin half) and SiluAndMul
(vllm/model_executor/layers/activation.py:15-38).  forward() always runs the HIP kernels; there
is no eager fallback.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from vllm_legacy_amd import _custom_ops as ops
from vllm_legacy_amd.distributed.communication_op import (
    PendingAllReduce, tensor_model_parallel_all_reduce_add_rms_norm)


class RMSNorm(nn.Module):

    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype=torch.float16, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype, device=device),
                                   requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, x, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if isinstance(x, PendingAllReduce):
            # a row-parallel linear's output under tensor parallelism: the all-reduce happens here
            assert residual is not None
            return tensor_model_parallel_all_reduce_add_rms_norm(x, residual, self.weight.data,
                                                                 self.variance_epsilon)
        if isinstance(x, ops.SplitKSlabs):
            # the producing GEMM left its split-K partial sums for this kernel to add up
            assert residual is not None
            out = torch.empty_like(residual)
            ops.fused_add_rms_norm_splitk(x, out, residual, self.weight.data,
                                          self.variance_epsilon)
            return out, residual
        if residual is not None:
            ops.fused_add_rms_norm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        out = torch.empty_like(x)
        ops.rms_norm(out, x, self.weight.data, self.variance_epsilon)
        return out


class RotaryEmbedding(nn.Module):

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype=torch.float16, device=None):
        super().__init__()
        self.head_size, self.rotary_dim, self.is_neox_style = head_size, rotary_dim, is_neox_style
        inv_freq = 1.0 / (base**(torch.arange(0, rotary_dim, 2, dtype=torch.float32, device=device)
                                 / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float32, device=device)
        freqs = torch.outer(t, inv_freq)
        self.register_buffer("cos_sin_cache",
                             torch.cat([freqs.cos(), freqs.sin()], dim=-1).to(dtype),
                             persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor,
                key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        ops.rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache,
                             self.is_neox_style)
        return query, key


class SiluAndMul(nn.Module):

    def forward(self, x, dtype: torch.dtype = torch.float16) -> torch.Tensor:
        if isinstance(x, ops.SplitKSlabs):
            # the merged gate|up GEMM left its split-K partial sums for this kernel to add up
            out = torch.empty((x.m, x.n // 2), dtype=dtype, device=x.workspace.device)
            ops.silu_and_mul_splitk(x, out)
            return out
        d = x.shape[-1] // 2
        out = torch.empty(x.shape[:-1] + (d, ), dtype=x.dtype, device=x.device)
        ops.silu_and_mul(out, x)
        return out
