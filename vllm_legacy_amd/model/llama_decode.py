"""Synthetic-weight Llama decode stack: the caller side of the hot path.

Shapes and layer order follow the reference's vllm/model_executor/models/llama.py:86-264
(qkv_proj -> rotary -> attention -> o_proj, gate_up_proj -> SiluAndMul -> down_proj, Megatron
column/row sharding with an all-reduce after each row-parallel linear,
vllm/model_executor/layers/linear.py:755-779).  The int4 linears and the attention (cache write +
paged attention), RMSNorm, rotary embedding and SwiGLU run on the gfx950 library; the embedding
lookup and the fp16 lm_head are plain PyTorch-ROCm ops.

Weights are random (no checkpoints are available offline); this module exists to drive the
decode-step benchmark and the tensor-parallel tests, not to load models.
"""
import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from vllm_legacy_amd import _custom_ops as ops
from vllm_legacy_amd.attention.layer import Attention
from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
from vllm_legacy_amd.distributed.communication_op import (PendingAllReduce,
                                                          tensor_model_parallel_all_reduce,
                                                          tensor_model_parallel_fuses_norm,
                                                          tensor_model_parallel_gather,
                                                          tensor_model_parallel_takes_slabs)
from vllm_legacy_amd.distributed.parallel_state import (get_tensor_model_parallel_rank,
                                                        get_tensor_model_parallel_world_size)
from vllm_legacy_amd.layers import RMSNorm, RotaryEmbedding, SiluAndMul
from vllm_legacy_amd.quantization.awq import AWQConfig, AWQLinearMethod
from vllm_legacy_amd.quantization.gptq import GPTQConfig, GPTQLinearMethod


# (sequences x KV heads) below which rotary embedding and the cache write of a decode step ride in
# its attention launch
_ROPE_ATTN_BELOW = int(os.environ.get("VLA_ROPE_ATTN_BELOW", "320"))


@dataclass
class LlamaConfig:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    # > 0: the MLP of every layer is a quantized sparse-MoE block of this many experts
    # (vllm/model_executor/models/mixtral_quant.py:93-156, :283-296)
    num_local_experts: int = 0
    num_experts_per_tok: int = 2

    @staticmethod
    def mixtral_8x7b() -> "LlamaConfig":
        return LlamaConfig(vocab_size=32000, rope_theta=1e6, max_position_embeddings=32768,
                           num_local_experts=8, num_experts_per_tok=2)

    @staticmethod
    def llama3_8b() -> "LlamaConfig":
        return LlamaConfig()

    @staticmethod
    def llama3_70b() -> "LlamaConfig":
        return LlamaConfig(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80,
                           num_attention_heads=64, num_key_value_heads=8)

    @staticmethod
    def tiny() -> "LlamaConfig":
        return LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                           num_attention_heads=8, num_key_value_heads=2, head_dim=32 * 2,
                           vocab_size=1024, max_position_embeddings=512)


# VLA_FUSE_SILU=0 keeps the gate_up product and SiluAndMul as two launches at every batch size
_FUSE_SILU = os.environ.get("VLA_FUSE_SILU", "1") != "0"


class QuantLinear(nn.Module):
    """Column- or row-parallel int4 linear with random weights in the checkpoint layout."""

    def __init__(self, in_features: int, out_features: int, quant: str, group_size: int, device,
                 generator: torch.Generator, dtype=torch.float16):
        """quant: "awq", "gptq", "gptq_act" (GPTQ with activation reordering: desc_act, a random
        row -> group map as an act-ordered checkpoint carries in g_idx) or "none"."""
        super().__init__()
        self.quant = "gptq" if quant == "gptq_act" else quant
        if quant == "awq":
            self.method = AWQLinearMethod(AWQConfig(4, group_size))
        elif quant in ("gptq", "gptq_act"):
            self.method = GPTQLinearMethod(GPTQConfig(4, group_size, quant == "gptq_act"))
        elif quant == "none":
            self.method = None
        else:
            raise ValueError(f"unknown quantization {quant}")
        if self.method is None:
            w = torch.empty(in_features, out_features, dtype=dtype, device=device)
            w.uniform_(-0.02, 0.02, generator=generator)
            self.weight = nn.Parameter(w, requires_grad=False)
            return
        self.method.create_weights(self, in_features, [out_features], in_features, out_features,
                                   dtype, device=device)
        for name in ("qweight", "qzeros"):
            p = getattr(self, name)
            p.data = torch.randint(-2**31, 2**31 - 1, p.shape, dtype=torch.int64, device=device,
                                   generator=generator).to(torch.int32)
        # scales ~ U(0.5, 1.5) * 1e-2 keeps activations O(1) through the stack
        s = torch.empty(self.scales.shape, dtype=torch.float32, device=device)
        s.uniform_(0.5, 1.5, generator=generator)
        self.scales.data = (s * (1e-2 * (4096.0 / in_features)**0.5)).to(dtype)
        if quant == "gptq_act":
            perm = torch.randperm(in_features, device=device, generator=generator)
            self.g_idx.data = (perm // group_size).to(torch.int32)

    def load(self, tensors) -> None:
        """Install one rank's checkpoint tensors (already stacked and sharded: qweight, qzeros,
        scales and, for GPTQ, g_idx) and forget every first-use re-layout made from the old ones
        (the AWQ repack, the exllama shuffle: awq.py / gptq.py do them again on the next forward)."""
        from vllm_legacy_amd.quantization.gptq import ExllamaState
        for name, t in tensors.items():
            p = getattr(self, name)
            if name != "g_idx" and tuple(t.shape) != tuple(p.shape):
                raise ValueError(f"{name}: checkpoint shard {tuple(t.shape)} does not fit the "
                                 f"layer's {tuple(p.shape)}")
            p.data = t.to(device=p.device, dtype=p.dtype).contiguous()
        if hasattr(self, "_k_packed"):
            del self._k_packed
        if self.quant == "gptq" and getattr(self, "exllama_state", None) == ExllamaState.READY:
            self.exllama_state = ExllamaState.UNINITIALIZED

    def forward(self, x: torch.Tensor, defer_reduce: bool = False):
        """defer_reduce: the caller feeds the result to a kernel that can sum split-K slabs
        itself, so an ops.SplitKSlabs may come back instead of a tensor."""
        if self.method is None:
            return torch.matmul(x, self.weight)
        if defer_reduce and hasattr(self.method, "apply_partial"):
            return self.method.apply_partial(self, x)
        return self.method.apply(self, x)

    def forward_silu_and_mul(self, x: torch.Tensor):
        """SiluAndMul(forward(x)) of a [gate | up] projection as one launch, or None when the
        quantization method has no such form for this batch (the caller runs the two steps)."""
        if (_FUSE_SILU and self.method is not None and
                hasattr(self.method, "apply_silu_and_mul")):
            return self.method.apply_silu_and_mul(self, x)
        return None


class LlamaDecodeLayer(nn.Module):

    def __init__(self, cfg: LlamaConfig, quant: str, kv_cache_dtype: str, tp: int, device,
                 generator, dtype, layer_index: int = 0):
        super().__init__()
        self.cfg, self.tp = cfg, tp
        self.hq = cfg.num_attention_heads // tp
        self.hkv = max(1, cfg.num_key_value_heads // tp)   # replicated when tp > kv heads (llama.py:112-116)
        d = cfg.head_dim
        self.q_size, self.kv_size = self.hq * d, self.hkv * d
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        self.act_fn = SiluAndMul()
        self.qkv_proj = QuantLinear(cfg.hidden_size, self.q_size + 2 * self.kv_size, quant, 128,
                                    device, generator, dtype)
        self.o_proj = QuantLinear(self.q_size, cfg.hidden_size, quant, 128, device, generator, dtype)
        self.block_sparse_moe = None
        if cfg.num_local_experts > 0:
            # experts are split over the ranks, not sharded (mixtral_quant.py:109-123); every rank
            # draws all experts from the same seed and keeps its own
            from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
            self.block_sparse_moe = QuantMixtralMoE(cfg.hidden_size, cfg.intermediate_size,
                                                    cfg.num_local_experts, cfg.num_experts_per_tok,
                                                    quant, 128, device, seed=7919 + layer_index,
                                                    dtype=dtype)
        else:
            self.gate_up_proj = QuantLinear(cfg.hidden_size, 2 * cfg.intermediate_size // tp, quant,
                                            128, device, generator, dtype)
            self.down_proj = QuantLinear(cfg.intermediate_size // tp, cfg.hidden_size, quant, 128,
                                         device, generator, dtype)
        self.attn = Attention(self.hq, d, d**-0.5, self.hkv, kv_cache_dtype=kv_cache_dtype,
                              dtype=dtype)

    def _rope_in_attention(self, rotary_emb, block_size: int, dtype, num_seqs: int) -> bool:
        """Whether a decode step of this layer takes the launch that does rotary embedding, cache
        write and paged attention at once: batches of a few sequences, where it saves one of the
        layer's launches (Llama-3-8B stack from a graph: batch 1 2.30 -> 2.20 ms per step, batch 8
        2.61 -> 2.49, batch 32 4.11 -> 4.00).  Once the attention launch saturates HBM the cache
        write's scattered stores inside its read stream cost more than the launch they save
        (batch 256: 375 against 363 us per layer), and with the write in a launch of its own the
        query's rotation alone is worth nothing (16.55-16.74 against 16.71-16.76 ms per step).
        VLA_FUSE_ROPE_ATTN=0 keeps the separate launches everywhere."""
        if os.environ.get("VLA_FUSE_ROPE_ATTN", "1") == "0":
            return False
        if num_seqs * self.hkv >= _ROPE_ATTN_BELOW:
            return False
        impl = self.attn.impl
        if getattr(impl, "blocksparse", None) is not None or getattr(impl, "alibi_slopes", None) is not None:
            return False
        return ops.rope_paged_attention_supported(self.hq, self.hkv, self.cfg.head_dim,
                                                  rotary_emb.cos_sin_cache.shape[1],
                                                  rotary_emb.is_neox_style, block_size, dtype)

    def forward(self, positions, hidden, residual, kv_cache, attn_metadata, rotary_emb):
        """Layer order of vllm/model_executor/models/llama.py:196-224.  `hidden` may arrive as the
        previous layer's unreduced down projection (ops.SplitKSlabs); fused kernels are used
        where they compute the same values from fewer launches: split-K sums folded into the
        following add + RMSNorm or rotary step, rotary embedding and the cache write in one."""
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        decode_only = kv_cache is not None and attn_metadata.num_prefill_tokens == 0
        qkv = self.qkv_proj(hidden, defer_reduce=decode_only)
        slabs = qkv if isinstance(qkv, ops.SplitKSlabs) else None
        if slabs is not None:
            qkv = torch.empty((slabs.m, slabs.n), dtype=hidden.dtype, device=hidden.device)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        if decode_only:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.hkv,
                                                                   self.cfg.head_dim)
            if self._rope_in_attention(rotary_emb, value_cache.shape[3], hidden.dtype, q.shape[0]):
                # a few sequences: rotary embedding and the cache write ride in the attention launch
                dm = attn_metadata.decode_metadata
                attn_out = ops.rope_paged_attention(
                    positions, q, k, v, self.cfg.head_dim, rotary_emb.cos_sin_cache,
                    rotary_emb.is_neox_style, key_cache, value_cache, attn_metadata.slot_mapping,
                    self.attn.kv_cache_dtype, self.attn._kv_scale, self.hkv, self.attn.impl.scale,
                    dm.block_tables, dm.seq_lens_tensor, dm.max_decode_seq_len,
                    slabs=slabs, seq_order=dm.decode_seq_order,
                    use_v1=PagedAttention.use_v1(dm.max_decode_seq_len, q.shape[0], self.hq, self.hkv))
            else:
                ops.rotary_reshape_and_cache(positions, q, k, v, self.cfg.head_dim,
                                             rotary_emb.cos_sin_cache, rotary_emb.is_neox_style,
                                             key_cache, value_cache, attn_metadata.slot_mapping,
                                             self.attn.kv_cache_dtype, self.attn._kv_scale, slabs)
                attn_out = self.attn(q, k, v, kv_cache, attn_metadata, kv_cached=True)
        else:
            rotary_emb(positions, q, k)     # in place on the strided views of qkv
            attn_out = self.attn(q, k, v, kv_cache, attn_metadata)
        # Split-K slabs of the row-parallel linears are left unreduced when the next step can sum
        # them itself: the add + RMSNorm kernel on a single rank, the peer-memory all-reduce with
        # several (RCCL needs the finished tensor).  With that all-reduce in place it also does
        # the add + RMSNorm that follows: the linear's output travels on as a PendingAllReduce,
        # to this layer's second norm, the next layer's first or the model's final one.
        tokens, dt, dev = hidden.shape[0], hidden.dtype, hidden.device
        fuse = tensor_model_parallel_takes_slabs(tokens, self.cfg.hidden_size, dt, dev)
        fuse_norm = tensor_model_parallel_fuses_norm(tokens, self.cfg.hidden_size, dt, dev)
        hidden = self.o_proj(attn_out, defer_reduce=fuse)
        hidden = PendingAllReduce(hidden) if fuse_norm else tensor_model_parallel_all_reduce(hidden, dt)
        hidden, residual = self.post_attention_layernorm(hidden, residual)
        if self.block_sparse_moe is not None:
            # router, routed experts and the block's own all-reduce (mixtral_quant.py:129-156)
            return self.block_sparse_moe(hidden), residual
        # gate_up is column-parallel: whatever K slices it made are summed by the activation
        act = self.gate_up_proj.forward_silu_and_mul(hidden)   # small decode batches: one launch
        if act is None:
            act = self.act_fn(self.gate_up_proj(hidden, defer_reduce=True), dt)
        hidden = self.down_proj(act, defer_reduce=fuse)
        hidden = PendingAllReduce(hidden) if fuse_norm else tensor_model_parallel_all_reduce(hidden, dt)
        return hidden, residual


class LlamaDecodeModel(nn.Module):

    def __init__(self, cfg: LlamaConfig, quant: str = "awq", kv_cache_dtype: str = "fp8",
                 device="cuda", seed: int = 0, dtype=torch.float16):
        super().__init__()
        self.cfg, self.dtype = cfg, dtype
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.tp, self.rank = tp, rank
        g = torch.Generator(device=device).manual_seed(seed * 1000 + rank)
        assert cfg.vocab_size % tp == 0 and cfg.num_attention_heads % tp == 0
        self.vocab_shard = cfg.vocab_size // tp
        emb = torch.empty(self.vocab_shard, cfg.hidden_size, dtype=dtype, device=device)
        emb.normal_(0, 1.0, generator=g)
        self.embed_tokens = nn.Parameter(emb, requires_grad=False)
        self.layers = nn.ModuleList([
            LlamaDecodeLayer(cfg, quant, kv_cache_dtype, tp, device, g, dtype, layer_index=i)
            for i in range(cfg.num_hidden_layers)
        ])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        # [vocab shard, hidden] as the reference's ParallelLMHead keeps it
        # (layers/vocab_parallel_embedding.py:110-140); the library GEMM is also ~10 % faster on
        # this layout than on its transpose (256 x 4096 x 128256: 349 against 386 us eager)
        head = torch.empty(self.vocab_shard, cfg.hidden_size, dtype=dtype, device=device)
        head.normal_(0, 0.02, generator=g)
        self.lm_head = nn.Parameter(head, requires_grad=False)
        self.rotary_emb = RotaryEmbedding(cfg.head_dim, cfg.head_dim, cfg.max_position_embeddings,
                                          cfg.rope_theta, True, dtype, device)

    def kv_cache_shape(self, num_blocks: int, block_size: int):
        layer = self.layers[0]
        return layer.attn.backend.get_kv_cache_shape(num_blocks, block_size, layer.hkv,
                                                     self.cfg.head_dim)

    def embed(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Vocab-parallel lookup + all-reduce (layers/vocab_parallel_embedding.py:90-106)."""
        if self.tp == 1:
            return self.embed_tokens[input_ids]
        lo = self.rank * self.vocab_shard
        local = input_ids - lo
        mask = (local < 0) | (local >= self.vocab_shard)
        out = self.embed_tokens[local.masked_fill(mask, 0)]
        out.masked_fill_(mask[:, None], 0)
        return tensor_model_parallel_all_reduce(out)

    def forward(self, input_ids, positions, kv_caches: List[Optional[torch.Tensor]], attn_metadata):
        hidden = self.embed(input_ids)
        residual = None
        for layer, cache in zip(self.layers, kv_caches):
            hidden, residual = layer(positions, hidden, residual, cache, attn_metadata,
                                     self.rotary_emb)
        hidden, _ = self.norm(hidden, residual)
        return hidden

    def load_weights(self, tensors, method: Optional[str] = None, desc_act: bool = False,
                     group_size: int = 128) -> int:
        """Fill the stack from checkpoint tensors under their Hugging Face names (a mapping
        name -> tensor, e.g. read from safetensors): q/k/v and gate/up stacked into the merged
        projections and every linear cut to this rank's shard as LlamaForCausalLM.load_weights
        does through the parameters' weight loaders (vllm/model_executor/models/llama.py:411-469,
        layers/linear.py:341-459, :527-660, :729-747); embedding and lm_head rows cut to this
        rank's vocabulary shard (layers/vocab_parallel_embedding.py:90-140); a per-layer
        `self_attn.kv_scale` lands on the attention layer (llama.py:448-460).  Returns the number
        of tensors consumed."""
        from vllm_legacy_amd.model_loader import load_llama_layer_weights
        method = method or self.layers[0].qkv_proj.quant
        n = 0
        for i, layer in enumerate(self.layers):
            per_linear = load_llama_layer_weights(
                tensors, i, method, self.rank, self.tp, self.cfg.num_attention_heads,
                self.cfg.num_key_value_heads, self.cfg.head_dim, desc_act, group_size)
            for lname, got in per_linear.items():
                getattr(layer, lname).load(got)
                n += len(got)
            for norm in ("input_layernorm", "post_attention_layernorm"):
                key = f"model.layers.{i}.{norm}.weight"
                if key in tensors:
                    w = getattr(layer, norm).weight
                    w.data = tensors[key].to(device=w.device, dtype=w.dtype).contiguous()
                    n += 1
            key = f"model.layers.{i}.self_attn.kv_scale"
            if key in tensors:
                layer.attn._kv_scale = float(tensors[key])
                n += 1
        lo, hi = self.rank * self.vocab_shard, (self.rank + 1) * self.vocab_shard
        for key, param in (("model.embed_tokens.weight", self.embed_tokens),
                           ("lm_head.weight", self.lm_head), ("model.norm.weight", self.norm.weight)):
            if key in tensors:
                t = tensors[key]
                t = t if param is self.norm.weight else t[lo:hi]
                param.data = t.to(device=param.device, dtype=param.dtype).contiguous()
                n += 1
        return n

    def load_kv_cache_scales(self, quantization_param_path: str) -> int:
        """Per-layer fp8 KV-cache scaling factors from a kv_cache_scales.json
        (models/llama.py:474-493); returns how many layers received one."""
        from vllm_legacy_amd.model_loader import load_kv_cache_scales
        return load_kv_cache_scales(self.layers, quantization_param_path, self.rank, self.tp)

    def compute_logits(self, hidden: torch.Tensor) -> Optional[torch.Tensor]:
        """Column-parallel lm_head, gathered on rank 0 (layers/logits_processor.py:61-75)."""
        logits = torch.nn.functional.linear(hidden, self.lm_head)
        return tensor_model_parallel_gather(logits, dst=0, dim=-1)
