"""AWQ int4 linear method on the gfx950 ops (the role of
vllm/model_executor/layers/quantization/awq.py:15-175).

Weights keep the AWQ checkpoint layout -- qweight [in, out/8] int32, qzeros [in/G, out/8]
int32, scales [in/G, out] fp16 -- so tensors loaded for the reference work unchanged.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch

from vllm_legacy_amd import _custom_ops as ops

# The reference dequantises and calls a dense matmul from 256 tokens up (awq.py:164-172) because
# its fused kernel only tiles 16 rows.  The gfx950 dequant-GEMM tiles up to 256 rows, so the
# fused path is kept at every batch size unless this is set to a token count.
DEQUANT_MATMUL_MIN_TOKENS: Optional[int] = None


@dataclass
class AWQConfig:
    weight_bits: int = 4
    group_size: int = 128
    zero_point: bool = True

    def __post_init__(self):
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for AWQ, "
                             f"but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    @staticmethod
    def get_name() -> str:
        return "awq"

    @staticmethod
    def get_supported_act_dtypes() -> List[torch.dtype]:
        return [torch.half]


class AWQLinearMethod:

    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, device=None) -> None:
        g, pf = self.quant_config.group_size, self.quant_config.pack_factor
        if input_size_per_partition % g != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        out = sum(output_partition_sizes)
        if out % pf != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        P = torch.nn.Parameter
        layer.register_parameter("qweight", P(torch.empty(input_size_per_partition, out // pf,
                                                          dtype=torch.int32, device=device),
                                              requires_grad=False))
        layer.register_parameter("qzeros", P(torch.empty(input_size_per_partition // g, out // pf,
                                                         dtype=torch.int32, device=device),
                                             requires_grad=False))
        layer.register_parameter("scales", P(torch.empty(input_size_per_partition // g, out,
                                                         dtype=params_dtype, device=device),
                                             requires_grad=False))

    def apply_silu_and_mul(self, layer: torch.nn.Module, x: torch.Tensor):
        """For a [gate | up] projection: SiluAndMul(apply(x)) in one launch where the library has
        one (decode batches of at most 16 rows, wide layers), else None."""
        if x.dim() != 2 or x.shape[1] % 64 != 0 or x.shape[0] > 16:
            return None
        packed = getattr(layer, "_k_packed", None)
        if packed is None:
            packed = ops.awq_repack(layer.qweight, layer.qzeros)
            layer._k_packed = packed
        return ops.q4_gemm_packed_silu_and_mul(x, packed[0], packed[1], layer.scales, 0)

    def apply_partial(self, layer: torch.nn.Module, x: torch.Tensor):
        """apply() that may return the unreduced split-K slabs (ops.SplitKSlabs) for a fused
        consumer instead of the finished [tokens, out] matrix.  2-D input, no bias."""
        if x.dim() != 2 or x.shape[1] % 64 != 0:
            return self.apply(layer, x)
        packed = getattr(layer, "_k_packed", None)
        if packed is None:
            packed = ops.awq_repack(layer.qweight, layer.qzeros)
            layer._k_packed = packed
        return ops.q4_gemm_packed_partial(x, packed[0], packed[1], layer.scales, 0)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        out_shape = x.shape[:-1] + (qweight.shape[-1] * self.quant_config.pack_factor, )
        x2 = x.reshape(-1, x.shape[-1])
        if DEQUANT_MATMUL_MIN_TOKENS is not None and x2.shape[0] >= DEQUANT_MATMUL_MIN_TOKENS:
            out = torch.matmul(x2, ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0))
        elif x2.shape[1] % 64 == 0:
            # one-time re-layout on the first forward pass, as the reference does for GPTQ's
            # exllama kernels (quantization/gptq.py:209-217)
            packed = getattr(layer, "_k_packed", None)
            if packed is None:
                packed = ops.awq_repack(qweight, qzeros)
                layer._k_packed = packed
            out = ops.q4_gemm_packed(x2, packed[0], packed[1], scales, 0)
        else:
            out = ops.awq_gemm(x2, qweight, scales, qzeros, self.quant_config.pack_factor)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
