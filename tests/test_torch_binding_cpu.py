"""CPU tests of the compiled torch-extension binding (vllm_legacy_amd/csrc/torch_binding.cpp,
built by torch.utils.cpp_extension into vllm_legacy_amd/vla_torch_C.so): the module a maintainer
would build in place of `vllm._C` (csrc/pybind.cpp:6-111).  Without a GPU it must load, expose the
reference's sub-modules and names, and report the library's argument validation as RuntimeError
with the reference's messages; nothing is launched on tensors that are not on a GPU."""
import pytest
import torch


@pytest.fixture(scope="module")
def C():
    from vllm_legacy_amd import _lib
    return _lib.torch_binding()


def test_module_layout(C):
    for name in ("paged_attention_v1", "paged_attention_v2", "awq_gemm", "awq_dequantize",
                 "gptq_gemm", "gptq_shuffle", "rms_norm", "fused_add_rms_norm", "rotary_embedding",
                 "silu_and_mul"):
        assert callable(getattr(C.ops, name))
    for name in ("swap_blocks", "copy_blocks", "reshape_and_cache", "convert_fp8"):
        assert callable(getattr(C.cache_ops, name))
    assert callable(C.cuda_utils.get_max_shared_memory_per_block_device_attribute)


def _attn_args(head_size=64, heads=8, kv_heads=8, block_size=16, kv="auto", bs=(0, 0, 0, 64, 0)):
    q = torch.zeros((1, heads, head_size), dtype=torch.float16)
    kc = torch.zeros((4, kv_heads, head_size // 8, 16, 8), dtype=torch.float16)
    vc = torch.zeros((4, kv_heads, head_size, 16), dtype=torch.float16)
    bt = torch.zeros((1, 4), dtype=torch.int32)
    sl = torch.ones(1, dtype=torch.int32)
    return (torch.empty_like(q), q, kc, vc, kv_heads, 1.0, bt, sl, block_size, 64, None, kv, 1.0) + bs


def test_attention_validation_statuses(C):
    # the cases of tests/test_cabi_cpu.py::test_argument_validation_without_gpu, through the binding
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        C.ops.paged_attention_v1(*_attn_args(head_size=72))
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        C.ops.paged_attention_v1(*_attn_args(block_size=4))
    with pytest.raises(RuntimeError, match="Invalid argument"):
        C.ops.paged_attention_v1(*_attn_args(kv_heads=3))
    with pytest.raises(RuntimeError, match="[Bb]locksparse"):
        C.ops.paged_attention_v1(*_attn_args(bs=(0, 0, 8, 0, 0)))
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache: fp8_e5m2"):
        C.ops.paged_attention_v1(*_attn_args(kv="fp8_e5m2"))
    # valid arguments, but not on a GPU: refused after validation, nothing launched
    with pytest.raises(RuntimeError, match="expected a tensor on the GPU"):
        C.ops.paged_attention_v1(*_attn_args())


def test_pointer_string_form_validation(C):
    out, q, kc, vc, hkv, scale, bt, sl, bsz, msl, al, kv, kvs, *bs = _attn_args(head_size=72)
    es = torch.zeros((1, 8, 1), dtype=torch.float32)
    tmp = torch.zeros((1, 8, 1, 72), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        C.ops.paged_attention_v2(out, es, es.clone(), tmp, q, str(kc.data_ptr()), kc.stride(0),
                                 kc.stride(1), str(vc.data_ptr()), hkv, scale, bt, sl, bsz, msl, al,
                                 kv, kvs, *bs)
    with pytest.raises(Exception):      # std::stoull on a non-number, as in the reference
        C.ops.paged_attention_v2(out, es, es.clone(), tmp, q, "not-a-pointer", 1, 1, "0", hkv, scale,
                                 bt, sl, bsz, msl, al, kv, kvs, *bs)
    k = torch.zeros((2, 8, 64), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="expected a tensor on the GPU"):
        C.cache_ops.reshape_and_cache(k, k, str(kc.data_ptr()), 16, 8, str(vc.data_ptr()),
                                      torch.zeros(2, dtype=torch.int64), "auto", 1.0)


def test_quant_validation_statuses(C):
    z = torch.zeros
    with pytest.raises(RuntimeError, match="Unsupported quantized weight shape"):
        C.ops.awq_gemm(z((4, 128), dtype=torch.float16), z((128, 4), dtype=torch.int32),
                       z((1, 32), dtype=torch.float16), z((1, 4), dtype=torch.int32), 8)
    with pytest.raises(RuntimeError, match="Unsupported quantized weight shape"):
        C.ops.gptq_gemm(z((4, 128), dtype=torch.float16), z((16, 64), dtype=torch.int32),
                        z((1, 8), dtype=torch.int32), z((1, 64), dtype=torch.float16),
                        torch.empty(0, dtype=torch.int32), True, 8)
    with pytest.raises(RuntimeError, match="expected a tensor on the GPU"):
        C.ops.gptq_gemm(z((4, 128), dtype=torch.float16), z((16, 64), dtype=torch.int32),
                        z((1, 8), dtype=torch.int32), z((1, 64), dtype=torch.float16),
                        torch.empty(0, dtype=torch.int32), True, 4)


def test_cache_op_device_checks(C):
    a = torch.zeros((4, 64), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="Invalid device combination"):     # cache_kernels.cu:31-45
        C.cache_ops.swap_blocks(a, a.clone(), torch.zeros((1, 2), dtype=torch.int64))
    with pytest.raises(RuntimeError, match="on a GPU"):
        C.cache_ops.convert_fp8(torch.zeros(8, dtype=torch.uint8), torch.zeros(8, dtype=torch.float16),
                                1.0, "fp8")
    C.cache_ops.copy_blocks([], [], torch.zeros((0, 2), dtype=torch.int64))   # no layers: no-op
