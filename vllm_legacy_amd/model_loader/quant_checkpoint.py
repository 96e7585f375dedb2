"""AWQ / GPTQ checkpoint tensors -> the per-rank tensors of the tensor-parallel int4 linears.

Tensor layouts (vllm/model_executor/layers/quantization/awq.py:103-144, gptq.py:134-187):

    AWQ   qweight [in, out/8] i32   qzeros [in/G, out/8] i32   scales [in/G, out] f16
    GPTQ  qweight [in/8, out] i32   qzeros [groups, out/8] i32 scales [groups, out] f16
          g_idx [in] i32

Each parameter carries (input_dim, output_dim, packed_dim, pack_factor); a shard along a packed
dimension is 1/pack_factor as long in storage (linear.py:385-392, :411-414, :609-612).  Column-
parallel layers (qkv, gate_up) shard the output dimension, merged ones shard every logical
sub-matrix separately and concatenate (linear.py:341-459, :527-660); row-parallel layers (o, down)
shard the input dimension (linear.py:729-747), with GPTQ's zeros/scales replicated when the group
table cannot be split (act-order or a single group, gptq.py:124-132, :169-187).  When there are
fewer KV heads than ranks a KV head is replicated on `tp / num_kv_heads` neighbouring ranks
(linear.py:503-509, :620-624).
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch


@dataclass(frozen=True)
class QuantParamSpec:
    input_dim: Optional[int]
    output_dim: Optional[int]
    packed_dim: Optional[int] = None
    pack_factor: int = 1


def param_specs(method: str, row_parallel: bool = False, desc_act: bool = False,
                group_size: int = 128) -> Dict[str, QuantParamSpec]:
    """Sharding attributes of the tensors of one int4 linear."""
    if method == "awq":
        return {"qweight": QuantParamSpec(0, 1, 1, 8), "qzeros": QuantParamSpec(0, 1, 1, 8),
                "scales": QuantParamSpec(0, 1)}
    if method == "gptq":
        # row-parallel + (act-order or one group for the whole input): every rank keeps the whole
        # zero/scale table and indexes it through g_idx
        sz_in = None if (row_parallel and (desc_act or group_size == -1)) else 0
        return {"qweight": QuantParamSpec(0, 1, 0, 8), "g_idx": QuantParamSpec(0, None),
                "qzeros": QuantParamSpec(sz_in, 1, 1, 8), "scales": QuantParamSpec(sz_in, 1)}
    raise ValueError(f"unknown quantization method {method}")


def _narrow(t: torch.Tensor, dim: int, start: int, size: int) -> torch.Tensor:
    if start < 0 or start + size > t.shape[dim]:
        raise ValueError(f"shard [{start}, {start + size}) outside dimension {dim} of {tuple(t.shape)}")
    return t.narrow(dim, start, size)


def shard_column(spec: QuantParamSpec, full: torch.Tensor, tp_rank: int, tp_size: int) -> torch.Tensor:
    if spec.output_dim is None:
        return full
    n = full.shape[spec.output_dim]
    if n % tp_size:
        raise ValueError("The output size is not aligned with the quantized weight shape. "
                         "This can be caused by too large tensor parallel size.")
    size = n // tp_size
    return _narrow(full, spec.output_dim, tp_rank * size, size).contiguous()


def shard_row(spec: QuantParamSpec, full: torch.Tensor, tp_rank: int, tp_size: int) -> torch.Tensor:
    if spec.input_dim is None:
        return full
    n = full.shape[spec.input_dim]
    if n % tp_size:
        raise ValueError("The input size is not aligned with the quantized weight shape. "
                         "This can be caused by too large tensor parallel size.")
    size = n // tp_size
    return _narrow(full, spec.input_dim, tp_rank * size, size).contiguous()


def shard_merged_column(spec: QuantParamSpec, parts: Sequence[torch.Tensor], tp_rank: int,
                        tp_size: int) -> torch.Tensor:
    """gate_up_proj: every logical matrix is sharded on its own, shards are concatenated."""
    if spec.output_dim is None:
        return parts[0]
    return torch.cat([shard_column(spec, p, tp_rank, tp_size) for p in parts],
                     dim=spec.output_dim).contiguous()


def shard_qkv(spec: QuantParamSpec, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
              tp_rank: int, tp_size: int, num_heads: int, num_kv_heads: int,
              head_size: int) -> torch.Tensor:
    if spec.output_dim is None:
        return q
    d = spec.output_dim
    pack = spec.pack_factor if spec.packed_dim == d else 1
    if num_heads % tp_size:
        raise ValueError("number of attention heads must be divisible by the TP size")
    if tp_size >= num_kv_heads:
        kv_local, replicas = 1, tp_size // num_kv_heads
    else:
        kv_local, replicas = num_kv_heads // tp_size, 1
    q_size = (num_heads // tp_size) * head_size // pack
    kv_size = kv_local * head_size // pack
    kv_id = tp_rank // replicas
    return torch.cat([_narrow(q, d, tp_rank * q_size, q_size), _narrow(k, d, kv_id * kv_size, kv_size),
                      _narrow(v, d, kv_id * kv_size, kv_size)], dim=d).contiguous()


_HF_LINEARS = {
    "qkv_proj": ("self_attn", ["q_proj", "k_proj", "v_proj"]),
    "o_proj": ("self_attn", ["o_proj"]),
    "gate_up_proj": ("mlp", ["gate_proj", "up_proj"]),
    "down_proj": ("mlp", ["down_proj"]),
}


def load_llama_layer_weights(tensors, layer_idx: int, method: str, tp_rank: int, tp_size: int,
                             num_heads: int, num_kv_heads: int, head_size: int,
                             desc_act: bool = False, group_size: int = 128
                             ) -> Dict[str, Dict[str, torch.Tensor]]:
    """Per-rank tensors of one decoder layer from a mapping `name -> tensor` with Hugging Face
    names (`model.layers.N.self_attn.q_proj.qweight`, ...; e.g. a safetensors `safe_open` handle
    wrapped as `{k: f.get_tensor(k) for k in f.keys()}`), stacked the way the reference's
    LlamaForCausalLM.load_weights stacks them (models/llama.py:404-416)."""
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for fused, (block, srcs) in _HF_LINEARS.items():
        row = fused in ("o_proj", "down_proj")
        specs = param_specs(method, row_parallel=row, desc_act=desc_act, group_size=group_size)
        got: Dict[str, torch.Tensor] = {}
        for pname, spec in specs.items():
            parts: List[torch.Tensor] = []
            for src in srcs:
                key = f"model.layers.{layer_idx}.{block}.{src}.{pname}"
                if key not in tensors:
                    if pname == "g_idx":
                        break
                    raise KeyError(key)
                parts.append(tensors[key])
            if not parts:
                continue
            if row:
                got[pname] = shard_row(spec, parts[0], tp_rank, tp_size)
            elif fused == "qkv_proj":
                got[pname] = shard_qkv(spec, parts[0], parts[1], parts[2], tp_rank, tp_size,
                                       num_heads, num_kv_heads, head_size)
            else:
                got[pname] = shard_merged_column(spec, parts, tp_rank, tp_size)
        if row and "g_idx" in got and specs["scales"].input_dim == 0:
            # group table was split with the rows: local group numbers start at zero
            got["g_idx"] = got["g_idx"] - got["g_idx"].min()
        out[fused] = got
    return out
