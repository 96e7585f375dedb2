"""GPTQ int4 linear method on the gfx950 ops (the role of
vllm/model_executor/layers/quantization/gptq.py:20-224).

Checkpoint layout: qweight [in/8, out] int32, qzeros [groups, out/8] int32, scales [groups, out]
fp16, g_idx [in] int32.  As in the reference the exllama re-layout (gptq_shuffle) happens on the
first forward pass (:209-217); row-parallel act-order layers cannot use it (:124-132).
"""
import enum
from dataclasses import dataclass
from typing import List, Optional

import torch

from vllm_legacy_amd import _custom_ops as ops


class ExllamaState(enum.Enum):
    UNUSED = enum.auto()
    UNINITIALIZED = enum.auto()
    READY = enum.auto()


@dataclass
class GPTQConfig:
    weight_bits: int = 4
    group_size: int = 128
    desc_act: bool = False

    def __post_init__(self):
        if self.weight_bits != 4:
            raise ValueError("the gfx950 backend implements 4-bit GPTQ only, "
                             f"but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    @staticmethod
    def get_name() -> str:
        return "gptq"


class GPTQLinearMethod:

    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, device=None) -> None:
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        out = sum(output_partition_sizes)
        if out % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        state = ExllamaState.UNINITIALIZED
        groups = input_size // group_size
        if input_size != input_size_per_partition and cfg.group_size != -1:
            if cfg.desc_act:
                state = ExllamaState.UNUSED       # row-parallel + act-order: plain kernel
            else:
                groups = input_size_per_partition // group_size
        P = torch.nn.Parameter
        layer.register_parameter("qweight", P(torch.empty(input_size_per_partition // cfg.pack_factor,
                                                          out, dtype=torch.int32, device=device),
                                              requires_grad=False))
        layer.register_parameter("g_idx", P(torch.arange(input_size_per_partition, dtype=torch.int32,
                                                         device=device) // cfg.group_size,
                                            requires_grad=False))
        layer.register_parameter("qzeros", P(torch.empty(groups, out // cfg.pack_factor,
                                                         dtype=torch.int32, device=device),
                                             requires_grad=False))
        layer.register_parameter("scales", P(torch.empty(groups, out, dtype=params_dtype,
                                                         device=device), requires_grad=False))
        layer.exllama_state = state

    def _shuffle_once(self, layer: torch.nn.Module) -> None:
        """First use of an exllama layer: the one-time weight shuffle (gptq.py:209-217)."""
        if layer.exllama_state == ExllamaState.UNINITIALIZED:
            if self.quant_config.desc_act:
                layer.g_idx.data = torch.argsort(layer.g_idx).to(torch.int)
            else:
                layer.g_idx.data = torch.empty((0, ), dtype=torch.int32, device=layer.g_idx.device)
            layer.exllama_state = ExllamaState.READY
            ops.gptq_shuffle(layer.qweight, layer.g_idx, self.quant_config.weight_bits)

    def _k_packed_input(self, layer: torch.nn.Module, x: torch.Tensor):
        """x as the k-packed MFMA kernels want it (columns in the shuffled weight's row order), or
        None when this layer does not run on them."""
        if (x.dim() != 2 or x.dtype != torch.float16 or self.quant_config.weight_bits != 4 or
                layer.exllama_state == ExllamaState.UNUSED or x.shape[1] % 64 != 0 or
                (x.shape[1] // layer.qzeros.shape[0]) % 32 != 0):
            return None
        self._shuffle_once(layer)
        if layer.exllama_state != ExllamaState.READY:
            return None
        return x.index_select(1, layer.g_idx) if layer.g_idx.numel() > 0 else x

    def apply_partial(self, layer: torch.nn.Module, x: torch.Tensor):
        """apply() that may return the unreduced split-K slabs (ops.SplitKSlabs) for a fused
        consumer instead of the finished [tokens, out] matrix.  2-D input, no bias."""
        xk = self._k_packed_input(layer, x)
        if xk is None:
            return self.apply(layer, x)
        return ops.q4_gemm_packed_partial(xk, layer.qweight, layer.qzeros, layer.scales, 1)

    def apply_silu_and_mul(self, layer: torch.nn.Module, x: torch.Tensor):
        """For a [gate | up] projection: SiluAndMul(apply(x)) in one launch where the library has
        one (decode batches of at most 16 rows, wide layers), else None."""
        if x.dim() != 2 or x.shape[0] > 16:
            return None
        xk = self._k_packed_input(layer, x)
        if xk is None:
            return None
        return ops.q4_gemm_packed_silu_and_mul(xk, layer.qweight, layer.qzeros, layer.scales, 1)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        out_shape = x.shape[:-1] + (layer.qweight.shape[-1], )
        x2 = x.reshape(-1, x.shape[-1])
        self._shuffle_once(layer)
        out = ops.gptq_gemm(x2, layer.qweight, layer.qzeros, layer.scales, layer.g_idx,
                            layer.exllama_state == ExllamaState.READY,
                            self.quant_config.weight_bits)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
