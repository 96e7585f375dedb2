"""The model-facing attention layer (the role of vllm/attention/layer.py:14-147): owns the
backend implementation for one layer and the per-layer KV scale of an fp8 cache."""
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from vllm_legacy_amd.attention.backends.abstract import AttentionMetadata
from vllm_legacy_amd.attention.selector import get_attn_backend


class Attention(nn.Module):

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 block_size: int = 16, dtype: torch.dtype = torch.float16,
                 blocksparse_params: Optional[Dict[str, Any]] = None) -> None:
        super().__init__()
        num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.kv_cache_dtype = kv_cache_dtype
        # one fp32 scale per layer; loaded from kv_cache_scales.json for fp8 checkpoints,
        # 1.0 otherwise (layer.py:55-56)
        self._kv_scale = 1.0
        backend = get_attn_backend(num_heads, head_size, num_kv_heads, sliding_window, dtype,
                                   kv_cache_dtype, block_size, blocksparse_params is not None)
        self.backend = backend
        self.impl = backend.get_impl_cls()(num_heads, head_size, scale, num_kv_heads, alibi_slopes,
                                           sliding_window, kv_cache_dtype, blocksparse_params)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: AttentionMetadata,
                kv_cached: bool = False) -> torch.Tensor:
        """kv_cached: key/value of this step are already in the cache (the caller used the fused
        rotary + cache-write kernel), so the backend must not write them again."""
        if kv_cached:
            return self.impl.forward(query, key, value, kv_cache, attn_metadata, self._kv_scale,
                                     kv_cached=True)
        return self.impl.forward(query, key, value, kv_cache, attn_metadata, self._kv_scale)

    def extra_repr(self) -> str:
        i = self.impl
        return (f"head_size={i.head_size}, num_heads={i.num_heads}, num_kv_heads={i.num_kv_heads}, "
                f"scale={i.scale}, backend={self.backend.get_name()}")
