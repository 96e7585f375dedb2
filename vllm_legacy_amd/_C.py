"""Stand-in for the hot-path part of the reference's compiled module `vllm._C`.

`vllm._C` (csrc/pybind.cpp:6-111) exposes three sub-modules, `ops`, `cache_ops` and `cuda_utils`;
`vllm/_custom_ops.py:6-12` imports them as `vllm_ops` / `vllm_cache_ops`.  The objects below have
the same names and the fork's positional signatures -- including its pointer-string form of the
KV cache (csrc/ops.h:14-30, csrc/cache.h:15-25: `str(tensor.data_ptr())` plus explicit strides) --
so that `from vllm_legacy_amd._C import ops as vllm_ops, cache_ops as vllm_cache_ops` is a
drop-in for the functions listed here.  Anything of `vllm._C` that is not on the decode hot path
(gelu variants, marlin, aqlm, squeezellm, MoE alignment, fp8/int8 activation quantisation, the
CUDA custom all-reduce) is deliberately absent and raises AttributeError.
"""
from types import SimpleNamespace

from . import _custom_ops as _o


def _paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache_data_ptr_str,
                        key_cache_stride_0, key_cache_stride_1, value_cache_data_ptr_str,
                        num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                        alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0,
                        blocksparse_local_blocks=0, blocksparse_vert_stride=0,
                        blocksparse_block_size=64, blocksparse_head_sliding_step=0):
    _o.paged_attention_v2_raw(out, exp_sums, max_logits, tmp_out, query,
                              int(key_cache_data_ptr_str), key_cache_stride_0, key_cache_stride_1,
                              int(value_cache_data_ptr_str), num_kv_heads, scale, block_tables,
                              seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                              kv_scale, tp_rank, blocksparse_local_blocks,
                              blocksparse_vert_stride, blocksparse_block_size,
                              blocksparse_head_sliding_step)


def _reshape_and_cache(key, value, key_cache_data_ptr_str, key_cache_block_size, key_cache_x,
                       value_cache_data_ptr_str, slot_mapping, kv_cache_dtype, kv_scale):
    _o.reshape_and_cache_raw(key, value, int(key_cache_data_ptr_str), key_cache_block_size,
                             key_cache_x, int(value_cache_data_ptr_str), slot_mapping,
                             kv_cache_dtype, kv_scale)


def _awq_gemm(input, qweight, scales, qzeros, split_k_iters):
    # csrc/ops.h:72-74 order: (in_feats, kernel, scaling_factors, zeros, split_k_iters)
    return _o.awq_gemm(input, qweight, scales, qzeros, split_k_iters)


ops = SimpleNamespace(
    paged_attention_v1=_o.paged_attention_v1,      # tensor form (csrc/ops.h:5-12)
    paged_attention_v2=_paged_attention_v2,        # pointer-string form (csrc/ops.h:14-30)
    silu_and_mul=_o.silu_and_mul,
    rms_norm=_o.rms_norm,
    fused_add_rms_norm=_o.fused_add_rms_norm,
    rotary_embedding=_o.rotary_embedding,
    awq_gemm=_awq_gemm,
    awq_dequantize=_o.awq_dequantize,
    gptq_gemm=_o.gptq_gemm,
    gptq_shuffle=_o.gptq_shuffle,
)

cache_ops = SimpleNamespace(
    swap_blocks=_o.swap_blocks,
    copy_blocks=_o.copy_blocks,
    reshape_and_cache=_reshape_and_cache,
    convert_fp8=_o.convert_fp8,
)

cuda_utils = SimpleNamespace(
    get_max_shared_memory_per_block_device_attribute=(
        _o.get_max_shared_memory_per_block_device_attribute),
)
