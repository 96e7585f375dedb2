"""The backend protocol the reference's Attention layer drives
(vllm/attention/backends/abstract.py:9-133), restated as the three interfaces a backend
provides: static cache helpers, step metadata, and the per-layer implementation."""
from abc import ABC, abstractmethod
from dataclasses import dataclass, fields
from typing import Any, Dict, List, Optional, Set, Tuple, Type

import torch


class AttentionBackend(ABC):

    @staticmethod
    @abstractmethod
    def get_name() -> str: ...

    @staticmethod
    @abstractmethod
    def get_impl_cls() -> Type["AttentionImpl"]: ...

    @staticmethod
    @abstractmethod
    def make_metadata(*args, **kwargs) -> "AttentionMetadata": ...

    @staticmethod
    @abstractmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]: ...

    @staticmethod
    @abstractmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None: ...

    @staticmethod
    @abstractmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None: ...


@dataclass
class AttentionMetadata:
    num_prefills: int          # prefill requests in the batch
    num_prefill_tokens: int
    num_decode_tokens: int     # = decode requests
    slot_mapping: torch.Tensor  # [num_tokens] int64 cache slot per token, negative = padding

    @property
    def prefill_metadata(self) -> Optional["AttentionMetadata"]:
        raise NotImplementedError

    @property
    def decode_metadata(self) -> Optional["AttentionMetadata"]:
        raise NotImplementedError

    def asdict_zerocopy(self, skip_fields: Optional[Set[str]] = None) -> Dict[str, Any]:
        skip = skip_fields or set()
        return {f.name: getattr(self, f.name) for f in fields(self) if f.name not in skip}


class AttentionImpl(ABC):

    @abstractmethod
    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 blocksparse_params: Optional[Dict[str, Any]] = None) -> None: ...

    @abstractmethod
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: torch.Tensor, attn_metadata: AttentionMetadata,
                kv_scale: float = 1.0) -> torch.Tensor: ...
