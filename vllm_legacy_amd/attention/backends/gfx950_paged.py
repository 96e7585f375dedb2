"""Attention backend for MI355X: paged KV cache written and read by the gfx950 library.

Plays the role of the reference's ROCm backend (vllm/attention/backends/rocm_flash_attn.py):
same metadata fields (:57-105), same forward contract (:261-402) -- new K/V are scattered into
the paged cache, decode tokens run single-query paged attention.  Prompt tokens are outside the
decode hot path; they run causal attention through torch's scaled_dot_product_attention (the
reference uses Triton / CK flash attention or a naive loop there, :320-363).
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type

import torch

from vllm_legacy_amd import _custom_ops as ops

from vllm_legacy_amd.attention.backends.abstract import (AttentionBackend, AttentionImpl,
                                                         AttentionMetadata)
from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention, PagedAttentionMetadata


@dataclass
class BlocksparseParams:
    """Local + strided block-sparse attention (the role of
    vllm/attention/backends/blocksparse_attn.py:15-84): a query in sparse block q attends key
    block k <= q when q - k < local_blocks, or when (k + head * head_sliding_step + 1) %
    vert_stride == 0."""
    max_seqlen: int
    num_heads: int        # query heads of this tensor-parallel rank
    num_kv_heads: int
    block_size: int       # tokens per sparse block (not the KV-cache block size)
    local_blocks: int
    vert_stride: int
    homo_head: bool = False         # every head attends the same strided blocks
    homo_head_group: bool = False   # every query head of a KV group does
    head_sliding_step: int = 0
    tp_rank: int = 0

    def __post_init__(self):
        from vllm_legacy_amd.distributed.parallel_state import (
            get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size)
        assert self.block_size > 0 and self.local_blocks >= 0 and self.vert_stride >= 1
        assert self.num_heads % self.num_kv_heads == 0
        tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        if self.homo_head:
            self.head_sliding_step = 0
        elif self.homo_head_group:   # negative: the offset slides along KV heads
            self.head_sliding_step = -max(1, int(self.vert_stride / (tp * self.num_kv_heads)))
        else:
            self.head_sliding_step = max(1, int(self.vert_stride / (tp * self.num_heads)))


class Gfx950PagedAttentionBackend(AttentionBackend):

    @staticmethod
    def get_name() -> str:
        return "gfx950-paged-attn"

    @staticmethod
    def get_impl_cls() -> Type["Gfx950PagedAttentionImpl"]:
        return Gfx950PagedAttentionImpl

    @staticmethod
    def make_metadata(*args, **kwargs) -> "Gfx950PagedAttentionMetadata":
        return Gfx950PagedAttentionMetadata(*args, **kwargs)

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        PagedAttention.copy_blocks(kv_caches, src_to_dists)


@dataclass
class Gfx950PagedAttentionMetadata(AttentionMetadata, PagedAttentionMetadata):
    """Prefill tokens first, decode tokens after them (rocm_flash_attn.py:171-193)."""
    seq_lens: Optional[List[int]]
    seq_lens_tensor: Optional[torch.Tensor]
    max_query_len: Optional[int]
    max_prefill_seq_len: int
    max_decode_seq_len: int
    query_start_loc: Optional[torch.Tensor]
    seq_start_loc: Optional[torch.Tensor]
    use_cuda_graph: bool
    context_lens_tensor: Optional[torch.Tensor]
    _cached_prefill_metadata: Optional["Gfx950PagedAttentionMetadata"] = None
    _cached_decode_metadata: Optional["Gfx950PagedAttentionMetadata"] = None
    # not in the reference: the order in which the decode kernels take the decode sequences
    # (ops.make_seq_order of their lengths: longest first), computed once per step by whoever
    # builds the metadata and shared by all layers; None = batch order
    decode_seq_order: Optional[torch.Tensor] = None

    @property
    def prefill_metadata(self) -> Optional["Gfx950PagedAttentionMetadata"]:
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is None:
            n = self.num_prefills
            self._cached_prefill_metadata = Gfx950PagedAttentionMetadata(
                num_prefills=n, num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=0,
                slot_mapping=self.slot_mapping[:self.num_prefill_tokens],
                seq_lens=self.seq_lens[:n], seq_lens_tensor=self.seq_lens_tensor[:n],
                max_query_len=self.max_query_len, max_prefill_seq_len=self.max_prefill_seq_len,
                max_decode_seq_len=0, query_start_loc=self.query_start_loc[:n + 1],
                seq_start_loc=self.seq_start_loc[:n + 1],
                context_lens_tensor=self.context_lens_tensor[:n],
                block_tables=self.block_tables[:n], use_cuda_graph=False)
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["Gfx950PagedAttentionMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is None:
            self._cached_decode_metadata = Gfx950PagedAttentionMetadata(
                num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
                slot_mapping=self.slot_mapping[self.num_prefill_tokens:], seq_lens=None,
                seq_lens_tensor=self.seq_lens_tensor[self.num_prefills:], max_query_len=None,
                max_prefill_seq_len=0, max_decode_seq_len=self.max_decode_seq_len,
                query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
                block_tables=self.block_tables[self.num_prefills:],
                use_cuda_graph=self.use_cuda_graph, decode_seq_order=self.decode_seq_order)
        return self._cached_decode_metadata


class Gfx950PagedAttentionImpl(AttentionImpl):

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: Optional[int] = None, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 blocksparse_params: Optional[Dict[str, Any]] = None) -> None:
        self.blocksparse = None
        if blocksparse_params is not None:
            # as the reference's block-sparse backend (blocksparse_attn.py:274-287)
            if alibi_slopes is not None:
                raise ValueError("Alibi not support for blocksparse flash attention.")
            if sliding_window is not None:
                raise ValueError("sliding_window is invalid for blocksparse attention.")
            bp = dict(blocksparse_params)
            bp.setdefault("num_heads", num_heads)
            bp.setdefault("num_kv_heads", num_kv_heads or num_heads)
            self.blocksparse = BlocksparseParams(**bp)
        # As in the reference's ROCm backend the window only reaches prefix-enabled prompt
        # attention (rocm_flash_attn.py:215-216, :382); a prompt without a cached prefix attends
        # causally to all of itself (:327-363) and decode relies on the caller handing in block
        # tables and lengths already cut to the window (:385-399).
        self.sliding_window = sliding_window
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.alibi_slopes = (None if alibi_slopes is None else
                             torch.tensor(alibi_slopes, dtype=torch.float32))
        self.kv_cache_dtype = kv_cache_dtype
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        supported = PagedAttention.get_supported_head_sizes()
        if head_size not in supported:
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {supported}.")

    def _alibi(self, device):
        if self.alibi_slopes is not None and self.alibi_slopes.device != device:
            self.alibi_slopes = self.alibi_slopes.to(device)
        return self.alibi_slopes

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: Gfx950PagedAttentionMetadata,
                kv_scale: float = 1.0, kv_cached: bool = False) -> torch.Tensor:
        """query [T, Hq*D], key/value [T, Hkv*D], kv_cache [2, NB, BS*Hkv*D] or None -> [T, Hq*D].
        kv_cached: this step's key/value were already written to the cache by the caller."""
        num_tokens, hidden = query.shape
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size)

        key_cache = value_cache = None
        if kv_cache is not None:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
            if not kv_cached:
                PagedAttention.write_to_paged_cache(key, value, key_cache, value_cache,
                                                    attn_metadata.slot_mapping,
                                                    self.kv_cache_dtype, kv_scale)

        n_pre, n_dec = attn_metadata.num_prefill_tokens, attn_metadata.num_decode_tokens
        assert key.shape[0] == n_pre + n_dec and value.shape[0] == n_pre + n_dec
        output = torch.empty_like(query)

        prefill_meta = attn_metadata.prefill_metadata
        if prefill_meta is not None:
            bt = prefill_meta.block_tables
            if self.blocksparse is not None:
                # outside the decode hot path: masked attention per prompt through torch (the
                # reference runs a Triton block-sparse kernel here and, like it, refuses a cached
                # prefix, blocksparse_attn.py:374-387)
                assert kv_cache is None or bt is None or bt.numel() == 0, \
                    "Does not support prefix-enabled attention."
                output[:n_pre] = _blocksparse_prompt_attention(
                    query[:n_pre], key[:n_pre], value[:n_pre], prefill_meta.seq_lens, self.scale,
                    self.num_queries_per_kv, self.blocksparse)
            elif kv_cache is not None and bt is not None and bt.numel() > 0:
                # prefix-enabled attention (rocm_flash_attn.py:368-383): the prompt continues
                # tokens that are already in the cache
                block_size = value_cache.shape[3]
                if not ops.prefix_prefill_supported(query.dtype, self.head_size, block_size,
                                                    self.kv_cache_dtype):
                    raise NotImplementedError(
                        f"prefix prefill: dtype {query.dtype}, head size {self.head_size}, "
                        f"block size {block_size} has no gfx950 kernel")
                output[:n_pre] = PagedAttention.forward_prefix(
                    query[:n_pre], key[:n_pre], value[:n_pre], key_cache, value_cache,
                    bt.to(torch.int32), prefill_meta.query_start_loc.to(torch.int32),
                    prefill_meta.seq_lens_tensor.to(torch.int32),
                    prefill_meta.context_lens_tensor.to(torch.int32),
                    int(prefill_meta.max_query_len), self._alibi(query.device),
                    self.sliding_window, self.kv_cache_dtype, kv_scale, self.scale)
            elif ops.prefill_attention_supported(query.dtype, self.head_size):
                # HIP flash-style kernel over the packed prompt tokens; seq_start_loc is the
                # cumulative prompt length table the reference's metadata carries
                # (rocm_flash_attn.py:84-88)
                cu = prefill_meta.seq_start_loc
                if cu is None:
                    lens = torch.as_tensor(prefill_meta.seq_lens, dtype=torch.int32)
                    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
                    cu[1:] = torch.cumsum(lens, 0)
                cu = cu.to(device=query.device, dtype=torch.int32)
                ops.prefill_attention(output[:n_pre], query[:n_pre], key[:n_pre], value[:n_pre],
                                      cu, int(prefill_meta.max_prefill_seq_len), self.scale,
                                      self._alibi(query.device))
            else:
                output[:n_pre] = _causal_prompt_attention(query[:n_pre], key[:n_pre],
                                                          value[:n_pre], prefill_meta.seq_lens,
                                                          self.scale, self.num_queries_per_kv)
        decode_meta = attn_metadata.decode_metadata
        if decode_meta is not None:
            assert kv_cache is not None, "decode needs a KV cache"
            bsp = self.blocksparse
            sparse = {} if bsp is None else dict(
                tp_rank=bsp.tp_rank, blocksparse_local_blocks=bsp.local_blocks,
                blocksparse_vert_stride=bsp.vert_stride, blocksparse_block_size=bsp.block_size,
                blocksparse_head_sliding_step=bsp.head_sliding_step)
            output[n_pre:] = PagedAttention.forward_decode(
                query[n_pre:], key_cache, value_cache, decode_meta.block_tables,
                decode_meta.seq_lens_tensor, decode_meta.max_decode_seq_len, self.kv_cache_dtype,
                self.num_kv_heads, self.scale, self._alibi(query.device), kv_scale,
                seq_order=decode_meta.decode_seq_order, **sparse)
        return output.view(num_tokens, hidden)


def _causal_prompt_attention(q, k, v, seq_lens: List[int], scale: float, qpk: int) -> torch.Tensor:
    """Per-sequence causal attention over the prompt tokens (no cached prefix)."""
    out = torch.empty_like(q)
    start = 0
    for n in seq_lens:
        qs = q[start:start + n].transpose(0, 1)                       # [Hq, n, D]
        ks = k[start:start + n].transpose(0, 1).repeat_interleave(qpk, dim=0)
        vs = v[start:start + n].transpose(0, 1).repeat_interleave(qpk, dim=0)
        o = torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=True, scale=scale)
        out[start:start + n] = o.transpose(0, 1)
        start += n
    return out


def _blocksparse_prompt_attention(q, k, v, seq_lens: List[int], scale: float, qpk: int,
                                  bsp: BlocksparseParams) -> torch.Tensor:
    """Causal local + strided block-sparse attention over each prompt: the dense mask of
    vllm/attention/ops/blocksparse_attention/utils.py:180-192 applied through torch."""
    out = torch.empty_like(q)
    hq = q.shape[1]
    heads = torch.arange(hq, device=q.device)
    if bsp.head_sliding_step >= 0:
        offs = (bsp.tp_rank * hq + heads) * bsp.head_sliding_step + 1
    else:
        offs = (bsp.tp_rank * (hq // qpk) + heads // qpk) * (-bsp.head_sliding_step) + 1
    start = 0
    for n in seq_lens:
        pos = torch.arange(n, device=q.device)
        blk = pos // bsp.block_size
        qb, kb = blk[:, None], blk[None, :]
        strided = ((kb[None] + offs[:, None, None]) % bsp.vert_stride) == 0      # [Hq, 1, n]
        mask = (pos[:, None] >= pos[None, :])[None] & (((qb - kb) < bsp.local_blocks)[None] | strided)
        qs = q[start:start + n].transpose(0, 1)
        ks = k[start:start + n].transpose(0, 1).repeat_interleave(qpk, dim=0)
        vs = v[start:start + n].transpose(0, 1).repeat_interleave(qpk, dim=0)
        o = torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask, scale=scale)
        out[start:start + n] = o.transpose(0, 1)
        start += n
    return out
