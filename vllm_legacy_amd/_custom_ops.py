"""Drop-in for the hot-path part of the reference's `vllm._custom_ops`.

Same function names, argument order and in-place/return conventions as
vllm/_custom_ops.py:37-99 (paged attention), :138-161 (AWQ / GPTQ) and :304-354 (cache ops),
bound to the gfx950 C-ABI library instead of `vllm._C`.  Tensors stay torch tensors; only raw
pointers, sizes and the current HIP stream cross into the library.

The KV caches may be torch tensors or any object with `data_ptr()`, `stride(i)` and `size(i)`
(the fork's managed-memory cache objects, vllm/worker/cache_engine/map_cache_engine.py:35-66),
which is why the fork passes them to `vllm._C` as pointer strings; `vllm_legacy_amd._C` offers
that pointer-string form too.
"""
import ctypes as C
from typing import List, Optional

import torch

from . import _lib
from ._lib import check

_DTYPE = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPE[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported input type of kv cache: {t.dtype}") from None


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and isinstance(t, torch.Tensor) and not t.is_cuda:
            raise RuntimeError("expected a tensor on the GPU")


# ---------------------------------------------------------------- layer glue
# (vllm/_custom_ops.py:16-17 silu_and_mul, :103-112 rotary_embedding, :126-133 layer norms)


def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _require_cuda(out, x)
    d = x.shape[-1] // 2
    assert x.is_contiguous() and out.is_contiguous() and out.shape[-1] == d
    with torch.cuda.device(x.device):
        check(_lib.lib().vla_silu_and_mul(out.data_ptr(), x.data_ptr(), x.numel() // (2 * d), d,
                                          _dt(x), _stream(x.device)), "silu_and_mul")


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                     head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """In place on query [T, Hq * D] and key [T, Hkv * D] (or [T, H, D]); rows may be strided."""
    _require_cuda(positions, query, key, cos_sin_cache)
    assert positions.dtype == torch.int64 and positions.is_contiguous()
    assert cos_sin_cache.dtype == query.dtype and cos_sin_cache.is_contiguous()
    num_tokens = positions.numel()
    q2, k2 = query.view(num_tokens, -1), key.view(num_tokens, -1)
    assert q2.stride(1) == 1 and k2.stride(1) == 1
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_rotary_embedding(
            positions.data_ptr(), query.data_ptr(), key.data_ptr(), cos_sin_cache.data_ptr(),
            num_tokens, q2.shape[1] // head_size, k2.shape[1] // head_size, head_size,
            cos_sin_cache.shape[1], q2.stride(0), k2.stride(0), 1 if is_neox else 0, _dt(query),
            _stream(query.device)), "rotary_embedding")


def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
             epsilon: float) -> None:
    _require_cuda(out, input, weight)
    assert input.is_contiguous() and out.is_contiguous() and weight.is_contiguous()
    h = input.shape[-1]
    with torch.cuda.device(input.device):
        check(_lib.lib().vla_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), epsilon,
                                      input.numel() // h, h, _dt(input), _stream(input.device)),
              "rms_norm")


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       epsilon: float) -> None:
    _require_cuda(input, residual, weight)
    assert input.is_contiguous() and residual.is_contiguous() and weight.is_contiguous()
    h = input.shape[-1]
    with torch.cuda.device(input.device):
        check(_lib.lib().vla_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(),
                                                weight.data_ptr(), epsilon, input.numel() // h, h,
                                                _dt(input), _stream(input.device)),
              "fused_add_rms_norm")


# ---------------------------------------------------------------- paged attention


def make_seq_order(seq_lens: torch.Tensor) -> torch.Tensor:
    """The launch order that takes the longest sequences first (int32 permutation; two small
    device kernels, no synchronisation).  Worth computing once per step for a ragged decode
    batch and sharing between the layers."""
    return torch.argsort(seq_lens, descending=True, stable=True).to(torch.int32)


def _check_seq_order(seq_order, num_seqs: int) -> None:
    if seq_order is not None:
        assert seq_order.dtype == torch.int32 and seq_order.is_contiguous()
        assert seq_order.numel() == num_seqs


def paged_attention_v1(
    out: torch.Tensor,
    query: torch.Tensor,
    key_cache,
    value_cache,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
    seq_order: Optional[torch.Tensor] = None,
) -> None:
    """seq_order (not in the reference): int32 [num_seqs] permutation, the order in which the
    launch takes the sequences -- longest first evens out a ragged batch (make_seq_order)."""
    paged_attention_v1_raw(out, query, key_cache.data_ptr(), key_cache.stride(0),
                           key_cache.stride(1), value_cache.data_ptr(), num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                           kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks,
                           blocksparse_vert_stride, blocksparse_block_size,
                           blocksparse_head_sliding_step, seq_order)


def paged_attention_v1_raw(out, query, key_cache_ptr: int, kv_block_stride: int,
                           kv_head_stride: int, value_cache_ptr: int, num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                           kv_cache_dtype, kv_scale, tp_rank=0, blocksparse_local_blocks=0,
                           blocksparse_vert_stride=0, blocksparse_block_size=64,
                           blocksparse_head_sliding_step=0, seq_order=None) -> None:
    _require_cuda(out, query, block_tables, seq_lens, alibi_slopes, seq_order)
    num_seqs, num_heads, head_size = query.shape
    _check_seq_order(seq_order, num_seqs)
    assert out.is_contiguous() and query.stride(1) == head_size and query.stride(2) == 1
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert block_tables.stride(1) == 1
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_paged_attention_v1_ordered(
            out.data_ptr(), query.data_ptr(), key_cache_ptr, value_cache_ptr, num_seqs, num_heads,
            num_kv_heads, head_size, block_size, scale, block_tables.data_ptr(),
            seq_lens.data_ptr(), block_tables.stride(0), max_seq_len, _ptr(alibi_slopes),
            query.stride(0), kv_block_stride, kv_head_stride, _dt(query),
            _lib.kv_dtype_code(kv_cache_dtype), kv_scale, tp_rank, blocksparse_local_blocks,
            blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step,
            _ptr(seq_order), _stream(query.device)), "paged_attention_v1")


def paged_attention_v2(
    out: torch.Tensor,
    exp_sum: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache,
    value_cache,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
    seq_order: Optional[torch.Tensor] = None,
) -> None:
    paged_attention_v2_raw(out, exp_sum, max_logits, tmp_out, query, key_cache.data_ptr(),
                           key_cache.stride(0), key_cache.stride(1), value_cache.data_ptr(),
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                           blocksparse_local_blocks, blocksparse_vert_stride,
                           blocksparse_block_size, blocksparse_head_sliding_step, seq_order)


def paged_attention_v2_raw(out, exp_sum, max_logits, tmp_out, query, key_cache_ptr: int,
                           kv_block_stride: int, kv_head_stride: int, value_cache_ptr: int,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0,
                           blocksparse_local_blocks=0, blocksparse_vert_stride=0,
                           blocksparse_block_size=64, blocksparse_head_sliding_step=0,
                           seq_order=None) -> None:
    _require_cuda(out, exp_sum, max_logits, tmp_out, query, block_tables, seq_lens, alibi_slopes,
                  seq_order)
    num_seqs, num_heads, head_size = query.shape
    _check_seq_order(seq_order, num_seqs)
    num_partitions = (max_seq_len + _lib.PARTITION_SIZE - 1) // _lib.PARTITION_SIZE
    assert out.is_contiguous() and query.stride(1) == head_size and query.stride(2) == 1
    assert exp_sum.is_contiguous() and max_logits.is_contiguous() and tmp_out.is_contiguous()
    assert exp_sum.dtype == torch.float32 and max_logits.dtype == torch.float32
    assert tuple(exp_sum.shape) == (num_seqs, num_heads, num_partitions), \
        "exp_sums must be [num_seqs, num_heads, ceil(max_seq_len / 512)]"
    assert tuple(tmp_out.shape) == (num_seqs, num_heads, num_partitions, head_size)
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert block_tables.stride(1) == 1
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_paged_attention_v2_ordered(
            out.data_ptr(), exp_sum.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(),
            query.data_ptr(), key_cache_ptr, value_cache_ptr, num_seqs, num_heads, num_kv_heads,
            head_size, block_size, scale, block_tables.data_ptr(), seq_lens.data_ptr(),
            block_tables.stride(0), max_seq_len, _ptr(alibi_slopes), query.stride(0),
            kv_block_stride, kv_head_stride, _dt(query), _lib.kv_dtype_code(kv_cache_dtype),
            kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
            blocksparse_block_size, blocksparse_head_sliding_step, _ptr(seq_order),
            _stream(query.device)), "paged_attention_v2")


def paged_attention_v2_noalloc(out: torch.Tensor, query: torch.Tensor, key_cache, value_cache,
                               num_kv_heads: int, scale: float, block_tables: torch.Tensor,
                               seq_lens: torch.Tensor, block_size: int, max_seq_len: int,
                               alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str,
                               kv_scale: float, tp_rank: int = 0,
                               blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                               blocksparse_block_size: int = 64,
                               blocksparse_head_sliding_step: int = 0) -> None:
    """paged_attention_v2 without the exp_sum / max_logits / tmp_out arguments (not in the
    reference): they live in a static device array inside the library, so neither the caller nor
    the call allocates anything.  Same bits as paged_attention_v2.  Calls that overlap on one GPU
    share that array and race; the library rejects problems beyond _lib.PA_V2_SCRATCH_BYTES."""
    _require_cuda(out, query, block_tables, seq_lens, alibi_slopes)
    num_seqs, num_heads, head_size = query.shape
    assert out.is_contiguous() and query.stride(1) == head_size and query.stride(2) == 1
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert block_tables.stride(1) == 1
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_paged_attention_v2_noalloc(
            out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            num_seqs, num_heads, num_kv_heads, head_size, block_size, scale,
            block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0), max_seq_len,
            _ptr(alibi_slopes), query.stride(0), key_cache.stride(0), key_cache.stride(1),
            _dt(query), _lib.kv_dtype_code(kv_cache_dtype), kv_scale, tp_rank,
            blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
            blocksparse_head_sliding_step, _stream(query.device)), "paged_attention_v2_noalloc")


def paged_attention_v2_noalloc_prepare(device=None) -> None:
    """Look up the static scratch of paged_attention_v2_noalloc on `device` (default: the current
    one).  Needed once per device before the op's first call may come from inside a graph capture:
    the lookup is a HIP API call, which a capture may not contain."""
    with torch.cuda.device(device):
        check(_lib.lib().vla_paged_attention_v2_noalloc_prepare(), "paged_attention_v2_noalloc_prepare")


def paged_attention_kernel_name(num_seqs: int, num_kv_heads: int, head_size: int, block_size: int,
                                max_seq_len: int, dtype: torch.dtype, kv_cache_dtype: str,
                                partitioned: bool = False) -> str:
    """Name of the kernel instantiation a paged-attention call of this shape launches, as a
    profiler's kernel trace spells it (measurement tooling; not in the reference)."""
    buf = C.create_string_buffer(160)
    check(_lib.lib().vla_paged_attention_kernel_name(
        buf, len(buf), num_seqs, num_kv_heads, head_size, block_size, max_seq_len, _DTYPE[dtype],
        _lib.kv_dtype_code(kv_cache_dtype), 1 if partitioned else 0))
    return buf.value.decode()


def prefill_attention(out: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                      value: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int,
                      scale: float, alibi_slopes: Optional[torch.Tensor] = None) -> None:
    """Causal attention over packed prompt tokens: query/out [T, Hq, D], key/value [T, Hkv, D]
    (row-strided views are fine), cu_seqlens [num_seqs + 1] int32 on the device."""
    _require_cuda(out, query, key, value, cu_seqlens, alibi_slopes)
    assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous()
    t, hq, d = query.shape
    hkv = key.shape[1]
    for x in (out, query, key, value):
        assert x.stride(2) == 1 and x.stride(1) == d
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_prefill_attention(
            out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
            cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, max_seqlen, hq, hkv, d, query.stride(0),
            key.stride(0), value.stride(0), out.stride(0), scale, _ptr(alibi_slopes), _dt(query),
            _stream(query.device)), "prefill_attention")


def prefill_attention_supported(dtype: torch.dtype, head_size: int) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and head_size in (24, 64, 80, 96, 112, 128)


def prefix_prefill_attention(out: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                             value: torch.Tensor, key_cache: torch.Tensor,
                             value_cache: torch.Tensor, block_tables: torch.Tensor,
                             query_start_loc: torch.Tensor, seq_lens: torch.Tensor,
                             context_lens: torch.Tensor, max_query_len: int, scale: float,
                             alibi_slopes: Optional[torch.Tensor] = None,
                             sliding_window: Optional[int] = None,
                             kv_cache_dtype: str = "auto", kv_scale: float = 1.0) -> None:
    """Prompt attention over a cached prefix (the reference's context_attention_fwd,
    vllm/attention/ops/prefix_prefill.py:684-744): query/out [T, Hq, D] and key/value
    [T, Hkv, D] hold the new tokens of every sequence, packed at query_start_loc [B + 1];
    key_cache [NB, Hkv, D/x, BS, x] / value_cache [NB, Hkv, D, BS] hold the context_lens [B]
    earlier tokens at block_tables [B, max_blocks]; seq_lens [B] = context + new.  The index
    tensors are int32 on the device."""
    _require_cuda(out, query, key, value, key_cache, value_cache, block_tables, query_start_loc,
                  seq_lens, context_lens, alibi_slopes)
    for x in (block_tables, query_start_loc, seq_lens, context_lens):
        assert x.dtype == torch.int32 and x.is_contiguous()
    t, hq, d = query.shape
    hkv = key.shape[1]
    for x in (out, query, key, value):
        assert x.stride(2) == 1 and x.stride(1) == d
    assert key_cache.is_contiguous() and value_cache.is_contiguous()
    nb, hkv_c, d_c, bs = value_cache.shape
    assert hkv_c == hkv and d_c == d and key_cache.shape[3] == bs
    num_seqs = seq_lens.numel()
    assert query_start_loc.numel() >= num_seqs + 1 and context_lens.numel() == num_seqs
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_prefix_prefill_attention(
            out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
            key_cache.data_ptr(), value_cache.data_ptr(), block_tables.data_ptr(),
            block_tables.stride(0), query_start_loc.data_ptr(), seq_lens.data_ptr(),
            context_lens.data_ptr(), num_seqs, max_query_len, hq, hkv, d, bs, query.stride(0),
            key.stride(0), value.stride(0), out.stride(0), value_cache.stride(0),
            value_cache.stride(1), scale, _ptr(alibi_slopes),
            int(sliding_window) if sliding_window else 0, _dt(query),
            _lib.kv_dtype_code(kv_cache_dtype), kv_scale, _stream(query.device)),
            "prefix_prefill_attention")


def prefix_prefill_supported(dtype: torch.dtype, head_size: int, block_size: int,
                             kv_cache_dtype: str = "auto") -> bool:
    sizes = (16, 32, 64) if kv_cache_dtype != "auto" else (8, 16, 32, 64)
    if kv_cache_dtype != "auto" and head_size % 16:
        return False        # a byte cache packs 16 dims per K piece
    return prefill_attention_supported(dtype, head_size) and block_size in sizes


# ---------------------------------------------------------------- cache ops


def reshape_and_cache(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache,
    value_cache,
    slot_mapping: torch.Tensor,
    kv_cache_dtype: str,
    kv_scale: float,
) -> None:
    reshape_and_cache_raw(key, value, key_cache.data_ptr(), key_cache.size(3), key_cache.size(4),
                          value_cache.data_ptr(), slot_mapping, kv_cache_dtype, kv_scale)


def reshape_and_cache_raw(key, value, key_cache_ptr: int, block_size: int, x: int,
                          value_cache_ptr: int, slot_mapping, kv_cache_dtype, kv_scale) -> None:
    _require_cuda(key, value, slot_mapping)
    num_tokens, num_heads, head_size = key.shape
    assert key.stride(1) == head_size and key.stride(2) == 1
    assert value.stride(1) == head_size and value.stride(2) == 1
    assert slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous()
    assert slot_mapping.numel() >= num_tokens
    with torch.cuda.device(key.device):
        check(_lib.lib().vla_reshape_and_cache(
            key.data_ptr(), value.data_ptr(), key_cache_ptr, value_cache_ptr,
            slot_mapping.data_ptr(), num_tokens, num_heads, head_size, block_size, x,
            key.stride(0), value.stride(0), _dt(key), _lib.kv_dtype_code(kv_cache_dtype),
            kv_scale, _stream(key.device)), "reshape_and_cache")


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    num_layers = len(key_caches)
    if num_layers != len(value_caches):
        raise RuntimeError("key_caches and value_caches must have the same length")
    if num_layers == 0:
        return
    dev = key_caches[0].device
    if not key_caches[0].is_cuda:
        raise RuntimeError("copy_blocks: caches must be on the GPU")
    assert block_mapping.is_cuda and block_mapping.dtype == torch.int64
    bm = block_mapping.contiguous().view(-1, 2)
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size()
    kp = (C.c_void_p * num_layers)(*[t.data_ptr() for t in key_caches])
    vp = (C.c_void_p * num_layers)(*[t.data_ptr() for t in value_caches])
    with torch.cuda.device(dev):
        check(_lib.lib().vla_copy_blocks(kp, vp, num_layers, bm.data_ptr(), bm.shape[0],
                                         block_bytes, _stream(dev)), "copy_blocks")


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")
        direction, dev = _lib.SWAP_D2D, src.device
    elif src.is_cuda and not dst.is_cuda:
        direction, dev = _lib.SWAP_D2H, src.device
    elif not src.is_cuda and dst.is_cuda:
        direction, dev = _lib.SWAP_H2D, dst.device
    else:
        raise RuntimeError("Invalid device combination")
    if block_mapping.is_cuda:
        raise RuntimeError("block_mapping must be on CPU")
    bm = block_mapping.to(torch.int64).contiguous().view(-1, 2)
    assert src.is_contiguous() and dst.is_contiguous()
    block_bytes = src[0].numel() * src.element_size()
    with torch.cuda.device(dev):
        check(_lib.lib().vla_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr(),
                                         bm.shape[0], block_bytes, direction, _stream(dev)),
              "swap_blocks")


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0,
                kv_dtype: str = "fp8") -> None:
    _require_cuda(output, input)
    if output.device != input.device:
        raise RuntimeError("src and dst must be on the same GPU")
    assert output.is_contiguous() and input.is_contiguous() and output.numel() == input.numel()
    if input.dtype == torch.uint8 and output.dtype != torch.uint8:
        to_fp8, dt = 0, _dt(output)
    elif output.dtype == torch.uint8 and input.dtype != torch.uint8:
        to_fp8, dt = 1, _dt(input)
    else:
        raise RuntimeError("convert_fp8 needs exactly one uint8 side")
    with torch.cuda.device(input.device):
        check(_lib.lib().vla_convert_fp8(output.data_ptr(), input.data_ptr(), input.numel(), dt,
                                         _lib.kv_dtype_code(kv_dtype), scale, to_fp8,
                                         _stream(input.device)), "convert_fp8")


# ---------------------------------------------------------------- int4 weights


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor,
                   split_k_iters: int = 0, thx: int = 0, thy: int = 0) -> torch.Tensor:
    """[in_c, out_c] fp16. split_k_iters / thx / thy only shape the reference's launch
    (csrc/quantization/awq/gemm_kernels.cu:445-461) and are accepted and ignored."""
    _require_cuda(qweight, scales, zeros)
    assert qweight.dtype == torch.int32 and zeros.dtype == torch.int32
    assert scales.dtype == torch.float16 and qweight.is_contiguous() and scales.is_contiguous()
    in_c, qoc = qweight.shape
    out = torch.empty((in_c, qoc * 8), dtype=scales.dtype, device=scales.device)
    with torch.cuda.device(scales.device):
        check(_lib.lib().vla_awq_dequantize(qweight.data_ptr(), scales.data_ptr(),
                                            zeros.data_ptr(), out.data_ptr(), in_c, qoc * 8,
                                            in_c // scales.shape[0], _stream(scales.device)),
              "awq_dequantize")
    return out


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor,
             qzeros: torch.Tensor, split_k_iters: int = 8) -> torch.Tensor:
    """Positional order is (x, qweight, scales, qzeros, split_k_iters) as in csrc/ops.h:69-71
    (the reference wrapper's parameter names are swapped, vllm/_custom_ops.py:145-147, its
    caller passes scales third, quantization/awq.py:171).  Split-K is not needed here: the
    contraction is accumulated in fp32 in one pass, so `split_k_iters` is ignored."""
    _require_cuda(input, qweight, scales, qzeros)
    assert input.dtype == torch.float16 and scales.dtype == torch.float16
    assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32
    x = input.contiguous()
    m, in_c = x.shape
    out_c = qweight.shape[1] * 8
    out = torch.empty((m, out_c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().vla_awq_gemm(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                      qzeros.data_ptr(), out.data_ptr(), m, in_c, out_c,
                                      in_c // scales.shape[0], _stream(x.device)), "awq_gemm")
    return out


def awq_repack(qweight: torch.Tensor, qzeros: torch.Tensor):
    """One-time re-layout of an AWQ weight for q4_gemm_packed: returns (qweight [in/8, out],
    qzeros [in/G, out/8]) in the k-packed pair order (the order gptq_shuffle writes)."""
    _require_cuda(qweight, qzeros)
    in_c, qoc = qweight.shape
    out_c = qoc * 8
    pw = torch.empty((in_c // 8, out_c), dtype=torch.int32, device=qweight.device)
    pz = torch.empty_like(qzeros)
    with torch.cuda.device(qweight.device):
        check(_lib.lib().vla_awq_repack(qweight.data_ptr(), qzeros.data_ptr(), pw.data_ptr(),
                                        pz.data_ptr(), in_c, out_c, in_c // qzeros.shape[0],
                                        _stream(qweight.device)), "awq_repack")
    return pw, pz


def q4_gemm_packed(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                   scales: torch.Tensor, zero_offset: int) -> torch.Tensor:
    """x [m, k] fp16 times k-packed int4 weights [k/8, n] -> [m, n] fp16."""
    _require_cuda(x, qweight, qzeros, scales)
    assert x.dtype == torch.float16 and scales.dtype == torch.float16
    x = x.contiguous()
    m, k = x.shape
    n = qweight.shape[1]
    group = k // qzeros.shape[0]
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    L = _lib.lib()
    ws_bytes = L.vla_q4_gemm_workspace_bytes(m, n, k, group)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        check(L.vla_q4_gemm_packed(x.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                   scales.data_ptr(), out.data_ptr(), m, n, k, group, zero_offset,
                                   _ptr(ws), ws_bytes, _stream(x.device)), "q4_gemm_packed")
    return out


def q4_gemm_grouped(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                    scales: torch.Tensor, row_offsets: torch.Tensor, zero_offset: int,
                    max_rows_per_set: Optional[int] = None,
                    x_rows: Optional[torch.Tensor] = None,
                    zero_outside: bool = True) -> torch.Tensor:
    """x [rows, k] fp16, rows sorted by weight set; qweight [sets, k/8, n], qzeros
    [sets, k/group, n/8], scales [sets, k/group, n] (k-packed); row_offsets [>= sets + 1] int32 on
    the device.  Returns [rows, n]; rows outside every set come back zero.
    With x_rows (int32 [rows] on the device) product row r reads x[x_rows[r]] -- the gather of the
    routed tokens without its copy.  With x_rows or zero_outside=False rows outside every set are
    left unwritten (no fill of the output, no clearing of K-split slabs)."""
    _require_cuda(x, qweight, qzeros, scales, row_offsets)
    assert x.dtype == torch.float16 and scales.dtype == torch.float16
    assert row_offsets.dtype == torch.int32 and row_offsets.is_contiguous()
    assert qweight.is_contiguous() and qzeros.is_contiguous() and scales.is_contiguous()
    x = x.contiguous()
    k = x.shape[1]
    sets, _, n = qweight.shape
    assert row_offsets.numel() >= sets + 1
    group = k // qzeros.shape[1]
    L = _lib.lib()
    gather = x_rows is not None or not zero_outside
    if x_rows is not None:
        assert x_rows.dtype == torch.int32 and x_rows.is_contiguous() and x_rows.is_cuda
        rows = x_rows.numel()
    else:
        rows = x.shape[0]
    out = (torch.empty if gather else torch.zeros)((rows, n), dtype=x.dtype, device=x.device)
    ws_bytes = L.vla_q4_gemm_grouped_workspace_bytes(rows, n, k, group, sets)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    bound = rows if max_rows_per_set is None else max_rows_per_set
    with torch.cuda.device(x.device):
        if gather:
            check(L.vla_q4_gemm_grouped_gather(
                x.data_ptr(), _ptr(x_rows), qweight.data_ptr(), qzeros.data_ptr(),
                scales.data_ptr(), out.data_ptr(), row_offsets.data_ptr(), sets, rows, bound, n, k,
                group, zero_offset, _ptr(ws), ws_bytes, _stream(x.device)), "q4_gemm_grouped_gather")
        else:
            check(L.vla_q4_gemm_grouped(
                x.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                out.data_ptr(), row_offsets.data_ptr(), sets, rows, bound, n, k, group, zero_offset,
                _ptr(ws), ws_bytes, _stream(x.device)), "q4_gemm_grouped")
    return out


def moe_route_topk(router_logits: torch.Tensor, top_k: int):
    """softmax(router_logits, fp32) -> top_k -> renormalise, one launch.  Returns (routing_weights
    fp32 [tokens, top_k], selected_experts int64 [tokens, top_k]); at most 64 experts."""
    _require_cuda(router_logits)
    router_logits = router_logits.contiguous()
    tokens, experts = router_logits.shape
    w = torch.empty((tokens, top_k), dtype=torch.float32, device=router_logits.device)
    sel = torch.empty((tokens, top_k), dtype=torch.int64, device=router_logits.device)
    with torch.cuda.device(router_logits.device):
        check(_lib.lib().vla_moe_route_topk(
            router_logits.data_ptr(), tokens, experts, top_k, w.data_ptr(), sel.data_ptr(),
            _dt(router_logits), _stream(router_logits.device)), "moe_route_topk")
    return w, sel


def moe_gate_topk(hidden_states: torch.Tensor, gate: torch.Tensor, top_k: int,
                  want_logits: bool = False):
    """The router in one launch: logits = hidden_states @ gate (fp32 sums rounded to the model
    dtype), softmax (fp32) -> top_k -> renormalise.  gate [hidden, experts], 8 or 16 experts.
    Returns (routing_weights fp32 [tokens, top_k], selected_experts int64 [tokens, top_k]) and,
    with want_logits, the logits."""
    _require_cuda(hidden_states, gate)
    assert hidden_states.dtype == gate.dtype and hidden_states.dim() == 2
    x, g = hidden_states.contiguous(), gate.contiguous()
    tokens, hidden = x.shape
    experts = g.shape[1]
    assert g.shape[0] == hidden
    w = torch.empty((tokens, top_k), dtype=torch.float32, device=x.device)
    sel = torch.empty((tokens, top_k), dtype=torch.int64, device=x.device)
    logits = torch.empty((tokens, experts), dtype=x.dtype, device=x.device) if want_logits else None
    with torch.cuda.device(x.device):
        check(_lib.lib().vla_moe_gate_topk(
            x.data_ptr(), g.data_ptr(), tokens, hidden, experts, top_k, w.data_ptr(),
            sel.data_ptr(), _ptr(logits), _dt(x), _stream(x.device)), "moe_gate_topk")
    return (w, sel, logits) if want_logits else (w, sel)


def moe_sort_pairs(selected_experts: torch.Tensor, bucket_of: torch.Tensor, num_buckets: int,
                   want_row_bucket: bool = False):
    """Stable bucketing of the (token, choice) pairs of a top-k routing by local expert, on the
    device.  selected_experts int64 [tokens, top_k]; bucket_of int32 [experts] (the trailing bucket
    num_buckets - 1 takes the experts of other ranks).  Returns (row_offsets int32
    [num_buckets + 1], sorted_token int32 [pairs], pair_row int32 [pairs], row_bucket int32 [pairs]
    or None).  Raises when the batch is too large for the one-workgroup sort."""
    _require_cuda(selected_experts, bucket_of)
    assert selected_experts.dtype == torch.int64 and selected_experts.is_contiguous()
    assert bucket_of.dtype == torch.int32 and bucket_of.is_contiguous()
    tokens, top_k = selected_experts.shape
    dev = selected_experts.device
    pairs = tokens * top_k
    row_offsets = torch.empty(num_buckets + 1, dtype=torch.int32, device=dev)
    sorted_token = torch.empty(pairs, dtype=torch.int32, device=dev)
    pair_row = torch.empty(pairs, dtype=torch.int32, device=dev)
    row_bucket = torch.empty(pairs, dtype=torch.int32, device=dev) if want_row_bucket else None
    with torch.cuda.device(dev):
        check(_lib.lib().vla_moe_sort_pairs(
            selected_experts.data_ptr(), bucket_of.data_ptr(), bucket_of.numel(), tokens, top_k,
            num_buckets, row_offsets.data_ptr(), sorted_token.data_ptr(), pair_row.data_ptr(),
            _ptr(row_bucket), _stream(dev)), "moe_sort_pairs")
    return row_offsets, sorted_token, pair_row, row_bucket


MOE_SORT_MAX_COUNTERS = 15 * 1024   # segments of 64 pairs x buckets the one-workgroup sort holds


def moe_combine(y: torch.Tensor, routing_weights: torch.Tensor, selected_experts: torch.Tensor,
                pair_row: torch.Tensor, bucket_of: torch.Tensor, num_local: int) -> torch.Tensor:
    """out[t] = sum over token t's local experts (ascending expert index) of
    model_dtype(y[pair_row[t, e]] * weight[t, e]), partial sums rounded to the model dtype."""
    _require_cuda(y, routing_weights, selected_experts, pair_row, bucket_of)
    assert routing_weights.dtype == torch.float32 and routing_weights.is_contiguous()
    assert selected_experts.dtype == torch.int64 and selected_experts.is_contiguous()
    assert y.is_contiguous() and pair_row.dtype == torch.int32 and pair_row.is_contiguous()
    tokens, top_k = selected_experts.shape
    hidden = y.shape[1]
    out = torch.empty((tokens, hidden), dtype=y.dtype, device=y.device)
    with torch.cuda.device(y.device):
        check(_lib.lib().vla_moe_combine(
            y.data_ptr(), routing_weights.data_ptr(), selected_experts.data_ptr(),
            pair_row.data_ptr(), bucket_of.data_ptr(), bucket_of.numel(), num_local, tokens, top_k,
            hidden, out.data_ptr(), _dt(y), _stream(y.device)), "moe_combine")
    return out


def moe_gather_perm(x: torch.Tensor, src_row: Optional[torch.Tensor], row_set: torch.Tensor,
                    perm: torch.Tensor) -> torch.Tensor:
    """out[r, c] = x[src_row[r], perm[row_set[r], c]] (src_row None: row r): the rows of an MoE
    block's sorted (token, expert) pairs with each row's columns in its own expert's act-order
    permutation, in one launch.  x 16-bit [*, cols] (row-strided is fine), perm int32 [sets, cols]."""
    _require_cuda(x, src_row, row_set, perm)
    assert x.element_size() == 2 and x.stride(1) == 1
    assert perm.dtype == torch.int32 and perm.is_contiguous() and perm.shape[1] == x.shape[1]
    assert row_set.dtype == torch.int32 and row_set.is_contiguous()
    if src_row is not None:
        assert src_row.dtype == torch.int32 and src_row.is_contiguous() and src_row.numel() == row_set.numel()
    rows, cols = row_set.numel(), x.shape[1]
    out = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().vla_moe_gather_perm(x.data_ptr(), _ptr(src_row), row_set.data_ptr(),
                                             perm.data_ptr(), out.data_ptr(), rows, cols, x.stride(0),
                                             _stream(x.device)), "moe_gather_perm")
    return out


class SplitKSlabs:
    """fp32 partial products [num_slabs, m, n] of an int4 GEMM whose K range was split, waiting
    for a consumer (fused_add_rms_norm_splitk / rotary_reshape_and_cache) to sum them."""

    def __init__(self, workspace: torch.Tensor, num_slabs: int, m: int, n: int):
        self.workspace, self.num_slabs, self.m, self.n = workspace, num_slabs, m, n

    def reduce(self, dtype=torch.float16) -> torch.Tensor:
        """What the GEMM's own reduction launch computes (slab order, one rounding)."""
        s = self.workspace[:self.num_slabs * self.m * self.n * 4].view(torch.float32)
        s = s.view(self.num_slabs, self.m, self.n)
        acc = s[0].clone()
        for p in range(1, self.num_slabs):
            acc += s[p]
        return acc.to(dtype)


def q4_gemm_packed_partial(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                           scales: torch.Tensor, zero_offset: int):
    """q4_gemm_packed without the reduction launch.  Returns an [m, n] tensor when the product
    was not split, else a SplitKSlabs."""
    _require_cuda(x, qweight, qzeros, scales)
    assert x.dtype == torch.float16 and scales.dtype == torch.float16
    x = x.contiguous()
    m, k = x.shape
    n = qweight.shape[1]
    group = k // qzeros.shape[0]
    L = _lib.lib()
    ws_bytes = L.vla_q4_gemm_workspace_bytes(m, n, k, group)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    # `out` always exists: the library may decide not to split after all (it then writes the
    # finished product there and reports zero slabs), whatever the workspace query said
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    ns = C.c_int(0)
    with torch.cuda.device(x.device):
        check(L.vla_q4_gemm_packed_partial(x.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                           scales.data_ptr(), out.data_ptr(), m, n, k, group,
                                           zero_offset, _ptr(ws), ws_bytes, C.byref(ns),
                                           _stream(x.device)), "q4_gemm_packed_partial")
    if ns.value > 1:
        return SplitKSlabs(ws, ns.value, m, n)
    return out


def q4_gemm_packed_silu_and_mul(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                                scales: torch.Tensor, zero_offset: int) -> Optional[torch.Tensor]:
    """silu(x @ W[:, :n/2]) * (x @ W[:, n/2:]) in one launch for a k-packed [gate | up] weight
    and a decode batch of at most 16 rows; None when the library does not take the shape (the
    caller then runs q4_gemm_packed_partial and silu_and_mul, which give the same bits)."""
    _require_cuda(x, qweight, qzeros, scales)
    if x.dtype != torch.float16 or x.dim() != 2:
        return None
    m, k = x.shape
    n = qweight.shape[1]
    group = k // qzeros.shape[0]
    L = _lib.lib()
    if m == 0 or not L.vla_q4_gemm_silu_and_mul_supported(m, n, k, group):
        return None
    x = x.contiguous()
    out = torch.empty((m, n // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(L.vla_q4_gemm_packed_silu_and_mul(
            x.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(), out.data_ptr(),
            m, n, k, group, zero_offset, _stream(x.device)), "q4_gemm_packed_silu_and_mul")
    return out


def rotary_reshape_and_cache(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                             value: torch.Tensor, head_size: int, cos_sin_cache: torch.Tensor,
                             is_neox: bool, key_cache, value_cache, slot_mapping: torch.Tensor,
                             kv_cache_dtype: str, kv_scale: float,
                             slabs: Optional[SplitKSlabs] = None, skip_query: bool = False) -> None:
    """rotary_embedding(positions, query, key, ...) followed by reshape_and_cache(key, value, ...)
    in one launch.  skip_query: the query is left alone (rope_paged_attention(write_kv=False)
    rotates it inside the attention launch).  query [T, Hq*D], key/value [T, Hkv*D] may be strided row views of one qkv
    tensor.  With `slabs` (the unreduced qkv projection, columns laid out q | k | v) the three
    tensors are outputs only."""
    _require_cuda(positions, query, key, value, cos_sin_cache, slot_mapping)
    assert positions.dtype == torch.int64 and positions.is_contiguous()
    assert slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous()
    assert cos_sin_cache.dtype == query.dtype and cos_sin_cache.is_contiguous()
    t = positions.numel()
    q2, k2, v2 = query.view(t, -1), key.view(t, -1), value.view(t, -1)
    assert q2.stride(1) == 1 and k2.stride(1) == 1 and v2.stride(1) == 1
    hq, hkv = q2.shape[1] // head_size, k2.shape[1] // head_size
    sp, ns, sstride, srow = None, 0, 0, 0
    if slabs is not None:
        assert slabs.m == t and slabs.n == (hq + 2 * hkv) * head_size
        sp, ns, sstride, srow = slabs.workspace.data_ptr(), slabs.num_slabs, slabs.m * slabs.n, slabs.n
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_rotary_reshape_and_cache(
            positions.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
            cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            slot_mapping.data_ptr(), t, 0 if skip_query else hq, hkv, head_size,
            cos_sin_cache.shape[1], q2.stride(0),
            k2.stride(0), v2.stride(0), key_cache.size(3), key_cache.size(4),
            1 if is_neox else 0, _dt(query), _lib.kv_dtype_code(kv_cache_dtype), kv_scale, sp, ns,
            sstride, srow, 0, hq * head_size, (hq + hkv) * head_size, _stream(query.device)),
            "rotary_reshape_and_cache")


def rope_paged_attention_supported(num_heads: int, num_kv_heads: int, head_size: int, rot_dim: int,
                                   is_neox: bool, block_size: int, dtype: torch.dtype) -> bool:
    """Whether rope_paged_attention takes a decode step of this shape."""
    if dtype not in _DTYPE:
        return False
    return bool(_lib.lib().vla_rope_paged_attention_supported(
        num_heads, num_kv_heads, head_size, rot_dim, 1 if is_neox else 0, block_size, _DTYPE[dtype]))


def rope_paged_attention(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                         value: torch.Tensor, head_size: int, cos_sin_cache: torch.Tensor,
                         is_neox: bool, key_cache, value_cache, slot_mapping: torch.Tensor,
                         kv_cache_dtype: str, kv_scale: float, num_kv_heads: int, scale: float,
                         block_tables: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int,
                         alibi_slopes: Optional[torch.Tensor] = None,
                         slabs: Optional[SplitKSlabs] = None,
                         seq_order: Optional[torch.Tensor] = None,
                         use_v1: bool = True, write_kv: bool = True) -> torch.Tensor:
    """rotary_reshape_and_cache followed by paged_attention_v1 / _v2 of a decode step (token t is
    sequence t) in one launch: same cache bytes, same output.  query [T, Hq*D], key / value
    [T, Hkv*D] hold the UNROTATED rows (strided row views of one qkv tensor are fine) and are left
    as they are; with `slabs` (the unreduced qkv projection, columns q | k | v) they are not read.
    write_kv=False: key and value of the step are in the cache already (rotary_reshape_and_cache
    with an empty query puts them there); only the query's rotation rides in the launch.
    Returns [T, Hq*D]."""
    _require_cuda(positions, query, key, value, cos_sin_cache, slot_mapping, block_tables, seq_lens,
                  alibi_slopes, seq_order)
    assert positions.dtype == torch.int64 and positions.is_contiguous()
    assert slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous()
    assert cos_sin_cache.dtype == query.dtype and cos_sin_cache.is_contiguous()
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert block_tables.stride(1) == 1
    t = positions.numel()
    q2, k2, v2 = query.view(t, -1), key.view(t, -1), value.view(t, -1)
    assert q2.stride(1) == 1 and k2.stride(1) == 1 and v2.stride(1) == 1
    hq, hkv = q2.shape[1] // head_size, k2.shape[1] // head_size
    assert hkv == num_kv_heads
    _check_seq_order(seq_order, t)
    sp, ns, sstride, srow = None, 0, 0, 0
    if slabs is not None:
        assert slabs.m == t and slabs.n == (hq + 2 * hkv) * head_size
        sp, ns, sstride, srow = slabs.workspace.data_ptr(), slabs.num_slabs, slabs.m * slabs.n, slabs.n
    out = torch.empty((t, hq * head_size), dtype=query.dtype, device=query.device)
    block_size = value_cache.shape[3]
    es = ml = tmp = None
    if not use_v1:
        parts = (max_seq_len + 511) // 512
        tmp = torch.empty((t, hq, parts, head_size), dtype=query.dtype, device=query.device)
        es = torch.empty((t, hq, parts), dtype=torch.float32, device=query.device)
        ml = torch.empty_like(es)
    with torch.cuda.device(query.device):
        check(_lib.lib().vla_rope_paged_attention(
            out.data_ptr(), _ptr(es), _ptr(ml), _ptr(tmp), q2.data_ptr(), k2.data_ptr(),
            v2.data_ptr(), q2.stride(0), k2.stride(0), v2.stride(0), positions.data_ptr(),
            cos_sin_cache.data_ptr(), cos_sin_cache.shape[1], 1 if is_neox else 0,
            key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr(), sp, ns, sstride,
            srow, 0, hq * head_size, (hq + hkv) * head_size, t, hq, hkv, head_size, block_size,
            scale, block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0),
            max_seq_len, _ptr(alibi_slopes), key_cache.stride(0), key_cache.stride(1), _dt(query),
            _lib.kv_dtype_code(kv_cache_dtype), kv_scale, _ptr(seq_order), 1 if write_kv else 0,
            _stream(query.device)), "rope_paged_attention")
    return out


def fused_add_rms_norm_splitk(slabs: SplitKSlabs, out: torch.Tensor, residual: torch.Tensor,
                              weight: torch.Tensor, epsilon: float) -> None:
    """fused_add_rms_norm whose input is the unreduced product `slabs`: residual += sum(slabs)
    (rounded to the model dtype first), out = rms_norm(residual) * weight."""
    _require_cuda(out, residual, weight)
    assert out.is_contiguous() and residual.is_contiguous() and weight.is_contiguous()
    h = residual.shape[-1]
    t = residual.numel() // h
    assert slabs.m == t and slabs.n == h and out.shape == residual.shape
    with torch.cuda.device(out.device):
        check(_lib.lib().vla_fused_add_rms_norm_splitk(
            slabs.workspace.data_ptr(), slabs.num_slabs, slabs.m * slabs.n, out.data_ptr(),
            residual.data_ptr(), weight.data_ptr(), epsilon, t, h, _dt(out), _stream(out.device)),
            "fused_add_rms_norm_splitk")


def silu_and_mul_splitk(slabs: SplitKSlabs, out: torch.Tensor) -> None:
    """silu_and_mul whose input [tokens, 2 d] is the unreduced product `slabs` (summed and rounded
    to out's dtype first): out [tokens, d]."""
    _require_cuda(out)
    assert out.is_contiguous() and out.dim() == 2
    t, d = out.shape
    assert slabs.m == t and slabs.n == 2 * d
    with torch.cuda.device(out.device):
        check(_lib.lib().vla_silu_and_mul_splitk(
            slabs.workspace.data_ptr(), slabs.num_slabs, slabs.m * slabs.n, out.data_ptr(), t, d,
            _dt(out), _stream(out.device)), "silu_and_mul_splitk")


def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor,
              b_gptq_scales: torch.Tensor, b_g_idx: torch.Tensor, use_exllama: bool,
              bit: int) -> torch.Tensor:
    _require_cuda(a, b_q_weight, b_gptq_qzeros, b_gptq_scales)
    assert a.dtype == torch.float16 and b_gptq_scales.dtype == torch.float16
    x = a.contiguous()
    m, k = x.shape
    n = b_q_weight.shape[1]
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    gidx = None
    if b_g_idx is not None and b_g_idx.device.type != "meta" and b_g_idx.numel() > 0:
        assert b_g_idx.dtype == torch.int32 and b_g_idx.is_cuda
        gidx = b_g_idx
    group = k // b_gptq_qzeros.shape[0]
    if use_exllama and bit == 4 and k % 64 == 0 and group % 32 == 0:
        # exllama layout = the k-packed layout of the MFMA kernel.  An act-ordered checkpoint was
        # shuffled with its sort permutation (rows grouped), so the activations' columns go
        # through the same permutation first (q_gemm.cu:224-227 reads a[perm[k]]): one gather.
        if gidx is not None:
            x = x.index_select(1, gidx)
        return q4_gemm_packed(x, b_q_weight, b_gptq_qzeros, b_gptq_scales, 1)
    with torch.cuda.device(x.device):
        check(_lib.lib().vla_gptq_gemm(x.data_ptr(), b_q_weight.data_ptr(),
                                       b_gptq_qzeros.data_ptr(), b_gptq_scales.data_ptr(),
                                       _ptr(gidx), out.data_ptr(), m, n, k,
                                       b_gptq_qzeros.shape[0], 1 if use_exllama else 0, bit,
                                       _stream(x.device)), "gptq_gemm")
    return out


def gptq_shuffle(q_weight: torch.Tensor, q_perm: torch.Tensor, bit: int) -> None:
    _require_cuda(q_weight)
    perm = None
    scratch = None
    if q_perm is not None and q_perm.device.type != "meta" and q_perm.numel() > 0:
        assert q_perm.dtype == torch.int32 and q_perm.is_cuda
        perm = q_perm
        scratch = torch.empty_like(q_weight)
    with torch.cuda.device(q_weight.device):
        check(_lib.lib().vla_gptq_shuffle(q_weight.data_ptr(), _ptr(perm), _ptr(scratch),
                                          q_weight.shape[0] * 32 // bit, q_weight.shape[1], bit,
                                          _stream(q_weight.device)), "gptq_shuffle")


def gptq_dequantize(b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx=None,
                    shuffled: bool = False) -> torch.Tensor:
    """The weight reconstruction half of the reference's gptq_gemm at M > 50
    (csrc/quantization/gptq/q_gemm.cu:1496-1513), exposed for tests and for callers that
    want the fp16 matrix."""
    k, n = b_q_weight.shape[0] * 8, b_q_weight.shape[1]
    out = torch.empty((k, n), dtype=torch.float16, device=b_q_weight.device)
    gidx = b_g_idx if (b_g_idx is not None and b_g_idx.numel() > 0) else None
    with torch.cuda.device(out.device):
        check(_lib.lib().vla_gptq_dequantize(b_q_weight.data_ptr(), b_gptq_qzeros.data_ptr(),
                                             b_gptq_scales.data_ptr(), _ptr(gidx),
                                             out.data_ptr(), k, n, b_gptq_qzeros.shape[0],
                                             1 if shuffled else 0, _stream(out.device)),
              "gptq_dequantize")
    return out


def get_max_shared_memory_per_block_device_attribute(device: int = 0) -> int:
    return _lib.lib().vla_get_max_shared_memory_per_block(device)


def set_attention_path(mode: int) -> None:
    """Test hook: 0 = pick the kernel per shape, 1 = force the generic kernel."""
    check(_lib.lib().vla_set_attention_path(mode))


def set_q4_row_block_kernel(mode: int) -> None:
    """Where the int4 GEMM kernel for row blocks of 17..80 rows runs: 0 grouped launches only
    (default), 1 dense products of 17..80 rows too (tests, kernel studies), 2 nowhere."""
    check(_lib.lib().vla_set_q4_row_block_kernel(mode))


def set_prefill_form(mode: int) -> None:
    """Which form of the prompt attention kernel runs: 0 by the longest prompt of the call
    (default), 1 the 256-thread form, 2 the 512-thread form (tests, kernel studies)."""
    check(_lib.lib().vla_set_prefill_form(mode))


def set_kv_load_policy(mode: int) -> None:
    """Cache policy of the attention kernel's K / V loads: 0 by size (default), 1 always
    nontemporal, 2 never.  A cache that stays resident between launches (a benchmark over a small
    block pool) is read faster without the hint; results are the same either way."""
    check(_lib.lib().vla_set_kv_load_policy(mode))
