"""Backend selection (the role of vllm/attention/selector.py:24-169).

The reference picks a `_Backend` from the platform and the VLLM_ATTENTION_BACKEND override
(:79-169; on ROCm it forces ROCM_FLASH, :109-119).  Here there is one device backend; the
override is honoured for its name and anything else is an error instead of a silent switch.
"""
import enum
import os
from functools import lru_cache
from typing import Optional, Type

import torch

from vllm_legacy_amd.attention.backends.abstract import AttentionBackend

VLLM_ATTENTION_BACKEND = "VLLM_ATTENTION_BACKEND"


class _Backend(enum.Enum):
    GFX950_PAGED = enum.auto()


@lru_cache(maxsize=None)
def get_attn_backend(num_heads: int, head_size: int, num_kv_heads: int,
                     sliding_window: Optional[int], dtype: torch.dtype,
                     kv_cache_dtype: Optional[str], block_size: int,
                     is_blocksparse: bool = False) -> Type[AttentionBackend]:
    if is_blocksparse:
        raise ValueError("blocksparse attention is not available in the gfx950 backend")
    which_attn_to_use(num_heads, head_size, num_kv_heads, sliding_window, dtype, kv_cache_dtype,
                      block_size)
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionBackend
    return Gfx950PagedAttentionBackend


def which_attn_to_use(num_heads, head_size, num_kv_heads, sliding_window, dtype, kv_cache_dtype,
                      block_size) -> _Backend:
    by_env = os.environ.get(VLLM_ATTENTION_BACKEND)
    if by_env is not None and by_env not in _Backend.__members__:
        raise ValueError(f"Invalid attention backend '{by_env}'. Available backends: "
                         f"{', '.join(_Backend.__members__)} ")
    if sliding_window is not None:
        raise ValueError("sliding-window attention is not available in the gfx950 backend")
    if kv_cache_dtype not in (None, "auto", "fp8", "fp8_e4m3"):
        raise ValueError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    if dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise ValueError(f"Unsupported model dtype: {dtype}")
    return _Backend.GFX950_PAGED
