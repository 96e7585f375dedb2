"""The boundary in the form the fork exports it: `vllm_legacy_amd._C.ops` / `.cache_ops` called
exactly as the reference's wrappers call `vllm._C` -- the KV cache as a decimal pointer string
plus strides / sizes (vllm/_custom_ops.py:87-99, :313-325; csrc/ops.h:14-30, csrc/cache.h:15-25).

The cache is NOT a torch tensor here: it is a raw device allocation (torch's allocator hands out
the address, as the fork's managed-memory cache hands out `managed_ptr.ptr`,
vllm/worker/cache_engine/map_cache_engine.py:35-66), described by a duck-typed object with
data_ptr() / stride(i) / size(i).  Expectations come from the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu


class RawDeviceBuffer:
    """A device allocation that is not a tensor, with the three accessors the reference's
    wrappers use on a cache object."""

    def __init__(self, shape, itemsize, typestr):
        self.shape = tuple(shape)
        self.nbytes = int(np.prod(shape)) * itemsize
        self.ptr = torch.cuda.caching_allocator_alloc(self.nbytes)
        st, acc = [], 1
        for n in reversed(self.shape):
            st.append(acc)
            acc *= n
        self._stride = tuple(reversed(st))
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": typestr,
                                         "data": (self.ptr, False), "version": 2}

    def data_ptr(self):
        return self.ptr

    def stride(self, i):
        return self._stride[i]

    def size(self, i):
        return self.shape[i]

    def view(self):
        """Test-side access to the bytes (the ops under test never see this)."""
        return torch.as_tensor(self, device="cuda")

    def free(self):
        torch.cuda.caching_allocator_delete(self.ptr)


@pytest.fixture(scope="module", params=["ctypes", "torch_extension"])
def C(request):
    """Both faces of the C-ABI with the layout of `vllm._C`: the ctypes module and the compiled
    torch extension (csrc/torch_binding.cpp)."""
    if request.param == "ctypes":
        from vllm_legacy_amd import _C
        return _C
    from vllm_legacy_amd import _lib
    return _lib.torch_binding()


def test_module_layout_matches_vllm_C(C):
    # csrc/pybind.cpp:6-111: sub-modules ops, cache_ops, cuda_utils
    for name in ("paged_attention_v1", "paged_attention_v2", "awq_gemm", "awq_dequantize",
                 "gptq_gemm", "gptq_shuffle", "rms_norm", "fused_add_rms_norm", "rotary_embedding",
                 "silu_and_mul"):
        assert callable(getattr(C.ops, name))
    for name in ("swap_blocks", "copy_blocks", "reshape_and_cache", "convert_fp8"):
        assert callable(getattr(C.cache_ops, name))
    assert C.cuda_utils.get_max_shared_memory_per_block_device_attribute(0) >= 64 * 1024
    with pytest.raises(AttributeError):
        C.ops.marlin_gemm


def test_compiled_binding_quant_and_cache_ops():
    """The remaining ops of the compiled binding against the ctypes wrappers (bit for bit: both
    call the same library)."""
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd import _lib
    B = _lib.torch_binding()
    g = torch.Generator(device="cuda").manual_seed(3)
    k, n, group, m = 512, 256, 128, 37
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), generator=g, device="cuda", dtype=torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), generator=g, device="cuda", dtype=torch.int32)
    sc = (torch.rand((k // group, n), generator=g, device="cuda") * 0.01 + 0.005).half()
    x = (torch.rand((m, k), generator=g, device="cuda") * 2 - 1).half()
    assert torch.equal(B.ops.awq_dequantize(qw, sc, qz, 0, 0, 0), ops.awq_dequantize(qw, sc, qz))
    assert torch.equal(B.ops.awq_gemm(x, qw, sc, qz, 8), ops.awq_gemm(x, qw, sc, qz, 8))
    gw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), generator=g, device="cuda", dtype=torch.int32)
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    a, b = gw.clone(), gw.clone()
    B.ops.gptq_shuffle(a, empty, 4)
    ops.gptq_shuffle(b, empty, 4)
    assert torch.equal(a, b)
    # the Python wrapper hands the library a split-K workspace, the binding (like the reference's
    # gptq_gemm) does not: the same fp32 sums in another order
    torch.testing.assert_close(B.ops.gptq_gemm(x, a, qz, sc, empty, True, 4),
                               ops.gptq_gemm(x, b, qz, sc, empty, True, 4), rtol=2e-3, atol=2e-3)
    perm = torch.randperm(k, generator=g, device="cuda").to(torch.int32)
    a, b = gw.clone(), gw.clone()
    B.ops.gptq_shuffle(a, perm, 4)
    ops.gptq_shuffle(b, perm, 4)
    assert torch.equal(a, b)
    # glue
    h = (torch.rand((5, 1024), generator=g, device="cuda") - 0.5).half()
    w = torch.rand(1024, generator=g, device="cuda").half()
    o1, o2 = torch.empty_like(h), torch.empty_like(h)
    B.ops.rms_norm(o1, h, w, 1e-5)
    ops.rms_norm(o2, h, w, 1e-5)
    assert torch.equal(o1, o2)
    s1, s2 = torch.empty((5, 512), dtype=h.dtype, device="cuda"), torch.empty((5, 512), dtype=h.dtype, device="cuda")
    B.ops.silu_and_mul(s1, h)
    ops.silu_and_mul(s2, h)
    assert torch.equal(s1, s2)
    # cache ops: convert, copy, swap
    src = (torch.rand((6, 2048), generator=g, device="cuda") - 0.5).half()
    f1, f2 = torch.empty(src.shape, dtype=torch.uint8, device="cuda"), torch.empty(src.shape, dtype=torch.uint8, device="cuda")
    B.cache_ops.convert_fp8(f1, src, 1.0, "fp8")
    ops.convert_fp8(f2, src, 1.0, "fp8")
    assert torch.equal(f1, f2)
    kc, vc = src.clone(), (src * 2).clone()
    B.cache_ops.copy_blocks([kc], [vc], torch.tensor([[0, 3], [1, 5]], dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(kc[3], src[0]) and torch.equal(vc[5], src[1] * 2)
    host = torch.zeros((6, 2048), dtype=torch.float16).pin_memory()
    B.cache_ops.swap_blocks(src, host, torch.tensor([[2, 4], [3, 0]], dtype=torch.int64))
    torch.cuda.synchronize()
    assert torch.equal(host[4], src[2].cpu()) and torch.equal(host[0], src[3].cpu())


@pytest.mark.parametrize("kv_fp8", [False, True])
def test_pointer_string_decode_step_on_a_raw_buffer(C, kv_fp8):
    """One decode step the way the fork's wrappers issue it: reshape_and_cache of the newest
    token, then paged_attention_v2, both on `str(cache.data_ptr())`."""
    c = T.make_attention_case(51, 5, (32, 8), 128, 16, O.DT_F16, kv_fp8, False, max_seq_len=900,
                              num_blocks=320, unique_blocks=True)
    esz, typestr = (1, "|u1") if kv_fp8 else (2, "<f2")
    kbuf = RawDeviceBuffer(c.kc.shape, esz, typestr)
    vbuf = RawDeviceBuffer(c.vc.shape, esz, typestr)
    try:
        kbuf.view().copy_(c.t_kc)
        vbuf.view().copy_(c.t_vc)
        # ---- cache_ops.reshape_and_cache(key, value, str, block_size, x, str, slots, dtype, scale)
        rng = np.random.default_rng(1)
        k = O.to_bits(rng.uniform(-c.scale, c.scale, (5, 8, 128)).astype(np.float32), O.DT_F16)
        v = O.to_bits(rng.uniform(-c.scale, c.scale, (5, 8, 128)).astype(np.float32), O.DT_F16)
        slots = np.array([int(c.block_tables[s, (n - 1) // 16]) * 16 + (n - 1) % 16
                          for s, n in enumerate(c.seq_lens)], dtype=np.int64)
        kvd = "fp8" if kv_fp8 else "auto"
        C.cache_ops.reshape_and_cache(T.from_bits(k, O.DT_F16, "cuda"),
                                      T.from_bits(v, O.DT_F16, "cuda"), str(kbuf.data_ptr()),
                                      kbuf.size(3), kbuf.size(4), str(vbuf.data_ptr()),
                                      torch.from_numpy(slots).cuda(), kvd, 1.0)
        O.reshape_and_cache(k, v, c.kc, c.vc, slots, O.DT_F16, kv=c.kv)
        torch.cuda.synchronize()
        assert np.array_equal(T.bits(kbuf.view()), c.kc)      # bit-exact indexing and codec
        assert np.array_equal(T.bits(vbuf.view()), c.vc)
        # ---- ops.paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, str, s0, s1, str, ...)
        S, H, D = c.t_q.shape
        P = (c.max_seq_len + 511) // 512
        out = torch.empty_like(c.t_q)
        tmp = torch.empty((S, H, P, D), dtype=out.dtype, device="cuda")
        es = torch.empty((S, H, P), dtype=torch.float32, device="cuda")
        ml = torch.empty_like(es)
        C.ops.paged_attention_v2(out, es, ml, tmp, c.t_q, str(kbuf.data_ptr()), kbuf.stride(0),
                                 kbuf.stride(1), str(vbuf.data_ptr()), c.hkv, c.scale, c.t_bt,
                                 c.t_sl, c.bs, c.max_seq_len, None, kvd, 1.0, 0, 0, 0, 64, 0)
        torch.cuda.synchronize()
        want = T.to_f32(T.oracle_attention(c, "v2"), O.DT_F16)
        np.testing.assert_allclose(out.float().cpu().numpy(), want, atol=1e-2 if kv_fp8 else 1e-3,
                                   rtol=1e-5)
    finally:
        kbuf.free()
        vbuf.free()


def test_custom_ops_accept_a_duck_typed_cache(C):
    """vllm_legacy_amd._custom_ops (the wrapper layer) takes the same non-tensor cache objects
    the fork's wrappers take: only data_ptr(), stride(i) and size(i) are used."""
    from vllm_legacy_amd import _custom_ops as ops
    c = T.make_attention_case(52, 4, (8, 1), 128, 16, O.DT_BF16, False, False, max_seq_len=400,
                              num_blocks=120, unique_blocks=True)
    kbuf = RawDeviceBuffer(c.kc.shape, 2, "<u2")
    vbuf = RawDeviceBuffer(c.vc.shape, 2, "<u2")
    try:
        kbuf.view().copy_(c.t_kc.view(torch.uint16))
        vbuf.view().copy_(c.t_vc.view(torch.uint16))
        out = torch.empty_like(c.t_q)
        ops.paged_attention_v1(out, c.t_q, kbuf, vbuf, c.hkv, c.scale, c.t_bt, c.t_sl, c.bs,
                               c.max_seq_len, None, "auto", 1.0)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.float().cpu().numpy(),
                                   T.to_f32(T.oracle_attention(c, "v1"), O.DT_BF16), atol=1e-3,
                                   rtol=1e-5)
    finally:
        kbuf.free()
        vbuf.free()


def test_pointer_string_errors(C):
    q = torch.zeros((2, 8, 72), dtype=torch.float16, device="cuda")
    z = torch.zeros(64, dtype=torch.float16, device="cuda")
    bt = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    sl = torch.ones(2, dtype=torch.int32, device="cuda")
    es = torch.zeros((2, 8, 1), dtype=torch.float32, device="cuda")
    tmp = torch.zeros((2, 8, 1, 72), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        C.ops.paged_attention_v2(torch.empty_like(q), es, es.clone(), tmp, q, str(z.data_ptr()),
                                 64, 8, str(z.data_ptr()), 8, 0.1, bt, sl, 16, 40, None, "auto",
                                 1.0, 0, 0, 0, 64, 0)
    with pytest.raises((ValueError, RuntimeError)):     # std::stoull on a non-number raises in the reference too
        C.cache_ops.reshape_and_cache(q, q, "not-a-pointer", 16, 8, "0", sl.long(), "auto", 1.0)
