"""Backend selection (the role of vllm/attention/selector.py:24-169).

The reference picks a `_Backend` from the platform and the VLLM_ATTENTION_BACKEND override
(:79-169; on ROCm it forces ROCM_FLASH, :109-119) and hands block-sparse layers to a backend of
their own (:35-39).  Here there is one device backend for all of them: its implementation takes
`blocksparse_params` and `sliding_window` itself.  The override is honoured for its name and
anything else is an error instead of a silent switch.
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
    # is_blocksparse: the same backend (the decode kernels carry the block-sparse mask,
    # csrc/attention/attention_kernels.cu:209-254); the reference switches to
    # BlocksparseFlashAttentionBackend here (selector.py:35-39)
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
    # sliding_window is accepted, as by the reference's selector; the implementation applies it
    # where the reference's ROCm backend does (prefix-enabled prompt attention)
    if kv_cache_dtype not in (None, "auto", "fp8", "fp8_e4m3"):
        raise ValueError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    if dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise ValueError(f"Unsupported model dtype: {dtype}")
    return _Backend.GFX950_PAGED
