"""Host-side glue of the paged KV cache: cache views, the V1/V2 choice, scratch allocation.

Same surface as the reference's PagedAttention helper (vllm/attention/ops/paged_attn.py:30-239)
so that a backend's forward() reads the same; the ops underneath are the gfx950 library's.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from vllm_legacy_amd import _custom_ops as ops

_PARTITION_SIZE = 512  # = VLA_PARTITION_SIZE (attention_kernels.cu:846)


@dataclass
class PagedAttentionMetadata:
    """Decode-side metadata (paged_attn.py:13-27)."""
    seq_lens_tensor: Optional[torch.Tensor]   # [batch] int32, tokens seen so far per sequence
    max_decode_seq_len: int                   # 0 for a prefill-only batch
    block_tables: Optional[torch.Tensor]      # [batch, max_blocks_per_seq] int32


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """[2, NB, BS*H*D] -> K [NB, H, D/x, BS, x] and V [NB, H, D, BS], x = 16 / element size."""
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key, value, key_cache, value_cache, slot_mapping,
                             kv_cache_dtype: str, kv_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(),
                              kv_cache_dtype, kv_scale)

    @staticmethod
    def use_v1(max_seq_len: int, num_seqs: int, num_heads: int,
               num_kv_heads: Optional[int] = None) -> bool:
        """One pass (V1) or 512-token partitions + reduce (V2).

        Without num_kv_heads: the reference's rule (paged_attn.py:120-121), which counts its
        kernel's workgroups, one per (sequence, query head), and caps V1 at 8192 tokens because
        that kernel keeps all logits in LDS.  With it: the rule for the gfx950 kernels, whose
        workgroup serves a whole (sequence, KV head) group with an online softmax and has no
        length cap -- V1 as soon as those groups alone give every other CU a workgroup.  Measured
        with 8 KV heads and an fp8 cache: 4096 tokens, batch 8 V2 31 us / V1 47 us, batch 16
        equal, batch 24 V1 41 / V2 46; 16384 tokens, batch 8 V2 60 / V1 96, batch 32 V1 185 / V2
        211, batch 64 V1 369 / V2 411."""
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        if num_kv_heads is None:
            return max_seq_len <= 8192 and (max_num_partitions == 1 or num_seqs * num_heads > 512)
        return max_num_partitions == 1 or num_seqs * num_kv_heads >= 128

    @staticmethod
    def forward_decode(query, key_cache, value_cache, block_tables, seq_lens, max_seq_len: int,
                       kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], kv_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0,
                       seq_order: Optional[torch.Tensor] = None) -> torch.Tensor:
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        if PagedAttention.use_v1(max_seq_len, num_seqs, num_heads, num_kv_heads):
            ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                                   block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                                   kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks,
                                   blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step, seq_order)
        else:
            assert _PARTITION_SIZE % block_size == 0
            parts = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
            tmp_output = torch.empty((num_seqs, num_heads, parts, head_size), dtype=output.dtype,
                                     device=output.device)
            exp_sums = torch.empty((num_seqs, num_heads, parts), dtype=torch.float32,
                                   device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache,
                                   value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                                   tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step,
                                   seq_order)
        return output

    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                       key_cache: torch.Tensor, value_cache: torch.Tensor,
                       block_tables: torch.Tensor, query_start_loc: torch.Tensor,
                       seq_lens_tensor: torch.Tensor, context_lens: torch.Tensor,
                       max_query_len: int, alibi_slopes: Optional[torch.Tensor],
                       sliding_window: Optional[int], kv_cache_dtype: str = "auto",
                       kv_scale: float = 1.0, scale: Optional[float] = None) -> torch.Tensor:
        """Prompt attention over a cached prefix (vllm/attention/ops/paged_attn.py:185-216).
        The reference's kernel fixes the softmax scale at head_size ** -0.5
        (vllm/attention/ops/prefix_prefill.py:707) and reads a byte cache without its scale;
        `scale`, `kv_cache_dtype` and `kv_scale` let the caller pass the layer's own values."""
        output = torch.empty_like(query)
        head_size = query.shape[-1]
        ops.prefix_prefill_attention(
            output, query, key, value, key_cache, value_cache, block_tables, query_start_loc,
            seq_lens_tensor, context_lens, max_query_len,
            head_size ** -0.5 if scale is None else scale, alibi_slopes, sliding_window,
            kv_cache_dtype, kv_scale)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst: torch.Tensor) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        ops.copy_blocks([c[0] for c in kv_caches], [c[1] for c in kv_caches], src_to_dists)
