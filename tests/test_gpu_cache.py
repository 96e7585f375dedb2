"""GPU parity tests for reshape_and_cache / copy_blocks / swap_blocks / convert_fp8.

Modelled on the reference's tests/kernels/test_cache.py (copy :31-102, reshape :105-193,
swap :274-344, fp8 round trip :347-380); expectations come from the oracle and are bit-exact
(including the fp8 bytes, which the reference only checks after dequantisation).
"""
import os
import random

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu

DTYPES = [O.DT_F16, O.DT_BF16, O.DT_F32]
HEAD_SIZES = [64, 80, 96, 112, 128, 192, 256]
BLOCK_SIZES = [8, 16, 32]


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def _make_caches(rng, nb, h, d, bs, dt, kv_fp8):
    esz = 1 if kv_fp8 else (4 if dt == O.DT_F32 else 2)
    x = 16 // esz
    kf = rng.uniform(-1, 1, (nb, h, d // x, bs, x)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nb, h, d, bs)).astype(np.float32)
    if kv_fp8:
        return O.convert_fp8(kf, O.DT_F32, True), O.convert_fp8(vf, O.DT_F32, True), x
    return O.to_bits(kf, dt), O.to_bits(vf, dt), x


def _up(a, dt, kv_fp8):
    return torch.from_numpy(a).cuda() if kv_fp8 else T.from_bits(a, dt, "cuda")


@pytest.mark.parametrize("head_size", HEAD_SIZES)
@pytest.mark.parametrize("block_size", BLOCK_SIZES)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_reshape_and_cache(ops, head_size, block_size, dt, kv_fp8):
    rng = np.random.default_rng(head_size * 7 + block_size)
    random.seed(0)
    T_, H, NB = 42, 8, 128
    kc, vc, x = _make_caches(rng, NB, H, head_size, block_size, dt, kv_fp8)
    slots = np.array(random.sample(range(NB * block_size), T_), dtype=np.int64)
    slots[5] = -1  # padding token
    qkv = rng.standard_normal((T_, 3, H, head_size)).astype(np.float32)
    t_qkv = T.from_bits(O.to_bits(qkv, dt), dt, "cuda")
    key, value = t_qkv[:, 1], t_qkv[:, 2]  # strided views (test_cache.py:137-138)
    t_kc, t_vc = _up(kc, dt, kv_fp8), _up(vc, dt, kv_fp8)
    kv_scale = 0.5 if kv_fp8 else 1.0
    ops.reshape_and_cache(key, value, t_kc, t_vc, torch.from_numpy(slots).cuda(),
                          "fp8" if kv_fp8 else "auto", kv_scale)
    torch.cuda.synchronize()
    O.reshape_and_cache(T.bits(key), T.bits(value), kc, vc, slots, dt,
                        kv=O.KV_FP8 if kv_fp8 else O.KV_AUTO, kv_scale=kv_scale)
    assert np.array_equal(T.bits(t_kc), kc)
    assert np.array_equal(T.bits(t_vc), vc)


def test_reshape_and_cache_decode_shape(ops):
    # the decode step of the metric config: 256 tokens, 8 KV heads x 128, fp16 -> fp8
    rng = np.random.default_rng(1)
    B, H, D, BS, NB = 256, 8, 128, 16, 4096
    kc, vc, x = _make_caches(rng, NB, H, D, BS, O.DT_F16, True)
    slots = rng.permutation(NB * BS)[:B].astype(np.int64)
    k = O.to_bits(rng.standard_normal((B, H, D)).astype(np.float32), O.DT_F16)
    v = O.to_bits(rng.standard_normal((B, H, D)).astype(np.float32), O.DT_F16)
    t_kc, t_vc = _up(kc, O.DT_F16, True), _up(vc, O.DT_F16, True)
    ops.reshape_and_cache(T.from_bits(k, O.DT_F16, "cuda"), T.from_bits(v, O.DT_F16, "cuda"), t_kc,
                          t_vc, torch.from_numpy(slots).cuda(), "fp8", 1.0)
    torch.cuda.synchronize()
    O.reshape_and_cache(k, v, kc, vc, slots, O.DT_F16, kv=O.KV_FP8)
    assert np.array_equal(T.bits(t_kc), kc) and np.array_equal(T.bits(t_vc), vc)


@pytest.mark.parametrize("head_size", [64, 80, 128, 256])
@pytest.mark.parametrize("block_size", BLOCK_SIZES)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("num_layers", [1, 3])
def test_copy_blocks(ops, head_size, block_size, dt, kv_fp8, num_layers):
    rng = np.random.default_rng(3)
    random.seed(0)
    NB, H, M = 1024, 8, 256
    src = random.sample(range(NB), M)
    rest = list(set(range(NB)) - set(src))
    dst = random.sample(rest, 2 * M)
    mapping = []
    for i in range(M):  # each source to two destinations (test_cache.py:60-72)
        mapping += [(src[i], dst[2 * i]), (src[i], dst[2 * i + 1])]
    caches = [_make_caches(rng, NB, H, head_size, block_size, dt, kv_fp8)[:2]
              for _ in range(num_layers)]
    t_k = [_up(k, dt, kv_fp8) for k, _ in caches]
    t_v = [_up(v, dt, kv_fp8) for _, v in caches]
    bm = torch.tensor(mapping, dtype=torch.int64, device="cuda").view(-1, 2)
    ops.copy_blocks(t_k, t_v, bm)
    torch.cuda.synchronize()
    for (k, v), tk, tv in zip(caches, t_k, t_v):
        O.copy_blocks(k, v, np.array(mapping))
        assert np.array_equal(T.bits(tk), k) and np.array_equal(T.bits(tv), v)


def test_copy_blocks_many_layers(ops):
    # more layers than one launch carries in its argument block (80 = Llama-3-70B)
    rng = np.random.default_rng(4)
    L, NB = 80, 16
    ks = [torch.from_numpy(rng.integers(0, 255, (NB, 256), dtype=np.uint8)).cuda() for _ in range(L)]
    vs = [torch.from_numpy(rng.integers(0, 255, (NB, 256), dtype=np.uint8)).cuda() for _ in range(L)]
    want_k = [k.clone() for k in ks]
    want_v = [v.clone() for v in vs]
    for wk, wv in zip(want_k, want_v):
        wk[9], wv[9] = wk[2].clone(), wv[2].clone()
        wk[11], wv[11] = wk[3].clone(), wv[3].clone()
    ops.copy_blocks(ks, vs, torch.tensor([[2, 9], [3, 11]], dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ks, want_k))
    assert all(torch.equal(a, b) for a, b in zip(vs, want_v))
    ops.copy_blocks([], [], torch.zeros((0, 2), dtype=torch.int64, device="cuda"))  # no-op


@pytest.mark.parametrize("direction", [("cuda", "cpu"), ("cuda", "cuda"), ("cpu", "cuda")])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("contiguous_runs", [False, True])
def test_swap_blocks(ops, direction, dt, kv_fp8, contiguous_runs):
    rng = np.random.default_rng(5)
    random.seed(0)
    NB, H, D, BS, M = 1024, 8, 128, 16, 256
    if contiguous_runs:  # runs of consecutive pairs exercise the merged-copy path
        src_blocks = list(range(100, 100 + M))
        dst_blocks = list(range(600, 600 + M // 2)) + list(range(10, 10 + M // 2))
    else:
        src_blocks = random.sample(range(NB), M)
        rest = list(set(range(NB)) - set(src_blocks))
        dst_blocks = random.sample(rest if direction[0] == direction[1] else range(NB), M)
    mapping = np.array(list(zip(src_blocks, dst_blocks)), dtype=np.int64)
    sk, sv, _ = _make_caches(rng, NB, H, D, BS, dt, kv_fp8)
    dk, dv, _ = _make_caches(rng, NB, H, D, BS, dt, kv_fp8)

    def put(a, dev):
        t = torch.from_numpy(a) if kv_fp8 else T.from_bits(a, dt, "cpu")
        return t.cuda() if dev == "cuda" else t.pin_memory()

    for s, d in ((sk, dk), (sv, dv)):
        ts, td = put(s, direction[0]), put(d, direction[1])
        ops.swap_blocks(ts, td, torch.from_numpy(mapping))
        torch.cuda.synchronize()
        want = d.copy()
        O.swap_blocks(s, want, mapping)
        assert np.array_equal(T.bits(td), want)
        assert np.array_equal(T.bits(ts), s)  # source untouched


def test_swap_blocks_errors(ops):
    a = torch.zeros((4, 64), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="Invalid device combination"):
        ops.swap_blocks(a, a.clone(), torch.zeros((1, 2), dtype=torch.int64))
    with pytest.raises(RuntimeError, match="block_mapping must be on CPU"):
        ops.swap_blocks(a.cuda(), a.cuda(), torch.zeros((1, 2), dtype=torch.int64).cuda())


@pytest.mark.parametrize("dt", DTYPES)
def test_convert_fp8_bit_exact_all_values(ops, dt):
    # every fp8 byte decodes to the oracle's value; every finite fp16 value (and a spread of
    # other floats) encodes to the oracle's byte, with and without a scale
    codes = np.arange(256, dtype=np.uint8)
    out = torch.empty(256, dtype=T.TORCH_DT[dt], device="cuda")
    ops.convert_fp8(out, torch.from_numpy(codes).cuda())
    torch.cuda.synchronize()
    want = O.from_bits(O.convert_fp8(codes, dt, False), dt)
    got = out.float().cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])

    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    rng = np.random.default_rng(6)
    x = np.concatenate([h[np.isfinite(h)], rng.uniform(-500, 500, 20000).astype(np.float32),
                        (rng.standard_normal(20000) * 0.02).astype(np.float32),
                        np.array([np.inf, -np.inf, 448, 464, 465, -1e30], dtype=np.float32)])
    xb = O.to_bits(x, dt)
    for scale in (1.0, 0.37):
        got8 = torch.empty(x.size, dtype=torch.uint8, device="cuda")
        ops.convert_fp8(got8, T.from_bits(xb, dt, "cuda"), scale, "fp8_e4m3")
        torch.cuda.synchronize()
        assert np.array_equal(got8.cpu().numpy(), O.convert_fp8(xb, dt, True, scale))
        back = torch.empty(x.size, dtype=T.TORCH_DT[dt], device="cuda")
        ops.convert_fp8(back, got8, scale)
        torch.cuda.synchronize()
        gb, wb = T.bits(back), O.convert_fp8(got8.cpu().numpy(), dt, False, scale)
        bad = np.nonzero(gb != wb)[0]
        assert bad.size == 0, (scale, bad[:8], got8.cpu().numpy()[bad[:8]], gb[bad[:8]], wb[bad[:8]])


def test_convert_fp8_nan_and_roundtrip_tolerance(ops):
    x = torch.tensor([float("nan"), 1.0, -3.5], dtype=torch.float16, device="cuda")
    y = torch.empty(3, dtype=torch.uint8, device="cuda")
    ops.convert_fp8(y, x)
    assert y.cpu().tolist()[0] & 0x7F == 0x7F
    # the reference's own round-trip check (test_cache.py:368-380): U(-224, 224), atol 1e-3, rtol 0.1
    cache = torch.empty((64, 8, 128, 16), dtype=torch.float16, device="cuda").uniform_(-224, 224)
    c8 = torch.empty_like(cache, dtype=torch.uint8)
    ops.convert_fp8(c8, cache)
    back = torch.empty_like(cache)
    ops.convert_fp8(back, c8)
    assert torch.allclose(cache, back, atol=0.001, rtol=0.1)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.convert_fp8(c8, cache, 1.0, "fp8_e5m2")


@pytest.mark.parametrize("name", ["cache_f32", "cache_bf16"])
def test_against_reference_cpu_kernel_fixtures(ops, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    dt = {"float32": O.DT_F32, "bfloat16": O.DT_BF16}[str(z["dtype"])]
    kc, vc = T.from_bits(z["key_cache_in"], dt, "cuda"), T.from_bits(z["value_cache_in"], dt, "cuda")
    ops.reshape_and_cache(T.from_bits(z["key"], dt, "cuda"), T.from_bits(z["value"], dt, "cuda"),
                          kc, vc, torch.from_numpy(z["slot_mapping"]).cuda(), "auto", 1.0)
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(kc), z["key_cache_out"])
    assert np.array_equal(T.bits(vc), z["value_cache_out"])
    ops.copy_blocks([kc], [vc], torch.from_numpy(z["block_mapping"]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(kc), z["key_cache_copied"])
    assert np.array_equal(T.bits(vc), z["value_cache_copied"])


@pytest.mark.parametrize("direction", [("cuda", "cpu"), ("cpu", "cuda")])
def test_swap_blocks_scattered_pageable_host(ops, direction):
    """Host memory that is not pinned cannot be addressed by the copy kernel: the per-run copy
    path must give the same bytes."""
    rng = np.random.default_rng(9)
    random.seed(1)
    NB, M = 256, 64
    src = rng.integers(0, 255, (NB, 4096)).astype(np.uint8)
    dst = rng.integers(0, 255, (NB, 4096)).astype(np.uint8)
    mapping = np.array(list(zip(random.sample(range(NB), M), random.sample(range(NB), M))), dtype=np.int64)
    ts = torch.from_numpy(src.copy())
    td = torch.from_numpy(dst.copy())
    ts = ts.cuda() if direction[0] == "cuda" else ts
    td = td.cuda() if direction[1] == "cuda" else td
    ops.swap_blocks(ts, td, torch.from_numpy(mapping))
    torch.cuda.synchronize()
    want = dst.copy()
    O.swap_blocks(src, want, mapping)
    assert np.array_equal(td.cpu().numpy(), want)
