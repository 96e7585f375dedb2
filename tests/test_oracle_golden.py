"""CPU tests: the oracle (oracle/pa_oracle.c) against fixtures produced by the reference's
own CPU kernels (tests/golden/make_golden.py) and against independent codecs."""
import os

import numpy as np
import pytest

from oracle import oracle as O

DT = {"float32": O.DT_F32, "bfloat16": O.DT_BF16, "float16": O.DT_F16}


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return z, DT[str(z["dtype"])]


@pytest.mark.parametrize("name,atol", [("attn_f32_gqa", 2e-6), ("attn_f32_d80", 2e-6),
                                       ("attn_bf16_alibi", 1e-3), ("attn_bf16_opt125m", 1e-3),
                                       ("attn_bf16_llama70b_tp8", 1e-3)])
def test_attention_matches_reference_cpu_kernels(golden_dir, name, atol):
    # The reference CPU kernel divides by the plain sum (csrc/cpu/attention.cpp:62-64) and keeps
    # probabilities in fp32, the GPU kernel (and the oracle) uses 1/(sum+1e-6) and rounds them
    # to the model dtype: fp32 agrees to ~1e-6, bf16 to the reference test's own atol 1e-3
    # (tests/kernels/test_attention.py:271).
    z, dt = _load(golden_dir, name)
    al = z["alibi"] if z["alibi"].size else None
    scale = float(z["scale"])
    out1 = O.paged_attention_v1(z["query"], z["key_cache"], z["value_cache"], z["block_tables"],
                                z["seq_lens"], scale, dt, alibi_slopes=al)
    out2, es, ml, tmp = O.paged_attention_v2(z["query"], z["key_cache"], z["value_cache"],
                                             z["block_tables"], z["seq_lens"], scale, dt,
                                             alibi_slopes=al)
    for got, want in ((out1, z["out_v1"]), (out2, z["out_v2"])):
        g, w = O.from_bits(got, dt), O.from_bits(want, dt)
        assert np.isfinite(g).all()
        np.testing.assert_allclose(g, w, atol=atol, rtol=1e-5)


@pytest.mark.parametrize("name", ["cache_f32", "cache_bf16"])
def test_cache_ops_bit_exact(golden_dir, name):
    z, dt = _load(golden_dir, name)
    kc, vc = z["key_cache_in"].copy(), z["value_cache_in"].copy()
    O.reshape_and_cache(z["key"], z["value"], kc, vc, z["slot_mapping"], dt)
    assert np.array_equal(kc, z["key_cache_out"])
    assert np.array_equal(vc, z["value_cache_out"])
    O.copy_blocks(kc, vc, z["block_mapping"])
    assert np.array_equal(kc, z["key_cache_copied"])
    assert np.array_equal(vc, z["value_cache_copied"])


def test_empty_sequence_rows_are_zero():
    # seq_len 0 (a padding row): zero logits summed, 1 / (0 + 1e-6) times zero accumulators
    rng = np.random.default_rng(5)
    q = O.to_bits(rng.standard_normal((3, 4, 64)).astype(np.float32), O.DT_F16)
    kc = O.to_bits(rng.standard_normal((6, 2, 8, 16, 8)).astype(np.float32), O.DT_F16)
    vc = O.to_bits(rng.standard_normal((6, 2, 64, 16)).astype(np.float32), O.DT_F16)
    bt = rng.integers(0, 6, (3, 3)).astype(np.int32)
    sl = np.array([0, 40, 0], np.int32)
    o1 = O.paged_attention_v1(q, kc, vc, bt, sl, 0.125, O.DT_F16)
    o2 = O.paged_attention_v2(q, kc, vc, bt, sl, 0.125, O.DT_F16)[0]
    for o in (o1, o2):
        assert not o[[0, 2]].any() and o[1].any()


def test_swap_blocks_exact():
    rng = np.random.default_rng(0)
    src = rng.integers(0, 255, (10, 64), dtype=np.uint8)
    dst = rng.integers(0, 255, (12, 64), dtype=np.uint8)
    want = dst.copy()
    pairs = [(1, 3), (4, 0), (9, 11)]
    for s, d in pairs:
        want[d] = src[s]
    O.swap_blocks(src, dst, np.array(pairs))
    assert np.array_equal(dst, want)


def test_fnuz_codec_matches_reference_codec(golden_dir):
    z = np.load(os.path.join(golden_dir, "fp8_fnuz_codec.npz"))
    lib = O.lib()
    dec = np.array([lib.oracle_e4m3fnuz_decode(int(c)) for c in range(256)], dtype=np.float32)
    want = z["decode_table"]
    assert np.array_equal(np.isnan(dec), np.isnan(want))
    assert np.array_equal(dec[~np.isnan(dec)], want[~np.isnan(want)])
    x = z["encode_in"]
    got = O.convert_fp8(x, O.DT_F32, to_fp8=True, kv=O.KV_FP8_FNUZ)
    assert np.array_equal(got, z["encode_out"])


def test_e4m3fn_codec_matches_torch():
    # OCP e4m3fn, saturating (the reference's CUDA path: __NV_SATFINITE,
    # csrc/quantization/fp8/nvidia/quant_utils.cuh:456-487). torch.float8_e4m3fn is an
    # independent implementation of the same format; its cast does not saturate, so clamp.
    import torch
    lib = O.lib()
    codes = torch.arange(256, dtype=torch.uint8)
    want = codes.view(torch.float8_e4m3fn).float().numpy()
    dec = np.array([lib.oracle_e4m3fn_decode(int(c)) for c in range(256)], dtype=np.float32)
    assert np.array_equal(np.isnan(dec), np.isnan(want))
    assert np.array_equal(dec[~np.isnan(dec)], want[~np.isnan(want)])
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    h = h[np.isfinite(h)]
    rng = np.random.default_rng(1)
    x = np.concatenate([h, rng.uniform(-500, 500, 8192).astype(np.float32),
                        (rng.standard_normal(8192) * 0.02).astype(np.float32),
                        np.array([447.9, 448, 463.9, 464, 480, 1e9, -1e9, 2.0**-9, 2.0**-10,
                                  1.5 * 2.0**-10, 3 * 2.0**-10], dtype=np.float32)])
    got = O.convert_fp8(x, O.DT_F32, to_fp8=True)
    want = torch.from_numpy(x).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(got, want)
    assert O.convert_fp8(np.array([np.nan, np.inf, -np.inf], np.float32), O.DT_F32,
                         True).tolist() == [0x7F, 0x7E, 0xFE]


def test_half_and_bf16_conversions_match_numpy_and_torch():
    import torch
    lib = O.lib()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.standard_normal(4096).astype(np.float32) * 100,
                        rng.standard_normal(4096).astype(np.float32) * 1e-6,
                        np.array([65504, 65519.9, 65520, 1e9, 2.0**-24, 2.0**-25, 1.5 * 2.0**-24,
                                  0, -0.0], dtype=np.float32)])
    got = np.array([lib.oracle_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    with np.errstate(over="ignore"):
        assert np.array_equal(got, x.astype(np.float16).view(np.uint16))
    gotb = np.array([lib.oracle_f32_to_bf16(float(v)) for v in x], dtype=np.uint16)
    wantb = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(gotb, wantb)
    assert np.array_equal(O.to_bits(x, O.DT_BF16), wantb)
    allh = np.arange(65536, dtype=np.uint16)
    dec = np.array([lib.oracle_f16_to_f32(int(v)) for v in allh], dtype=np.float32)
    want = allh.view(np.float16).astype(np.float32)
    ok = ~np.isnan(want)
    assert np.array_equal(dec[ok], want[ok]) and np.isnan(dec[~ok]).all()


def test_fp8_attention_consistent_with_dequantised_cache():
    # fp8 KV: attention over the fp8 cache == attention over the cache dequantised to the
    # model dtype first (how the reference's test builds its expectation,
    # tests/kernels/test_attention.py:232-248).
    rng = np.random.default_rng(3)
    S, Hq, Hkv, D, BS, NB = 3, 8, 2, 64, 16, 20
    scale = D**-0.5
    dt = O.DT_F16
    q = O.to_bits(rng.uniform(-scale, scale, (S, Hq, D)).astype(np.float32), dt)
    kc8 = O.convert_fp8(O.to_bits(rng.uniform(-scale, scale, (NB, Hkv, D // 16, BS, 16))
                                  .astype(np.float32), dt), dt, True)
    vc8 = O.convert_fp8(O.to_bits(rng.uniform(-scale, scale, (NB, Hkv, D, BS))
                                  .astype(np.float32), dt), dt, True)
    seq_lens = np.array([5, 160, 77], np.int32)
    bt = rng.integers(0, NB, (S, 10)).astype(np.int32)
    out8 = O.paged_attention_v1(q, kc8, vc8, bt, seq_lens, scale, dt, kv=O.KV_FP8)
    # dequantised 16-bit cache has x = 8, so re-tile K from [.., D/16, BS, 16] to [.., D/8, BS, 8]
    kd = O.convert_fp8(kc8, dt, False)
    kd = kd.reshape(NB, Hkv, D // 16, BS, 2, 8).transpose(0, 1, 2, 4, 3, 5).reshape(
        NB, Hkv, D // 8, BS, 8).copy()
    vd = O.convert_fp8(vc8, dt, False)
    out16 = O.paged_attention_v1(q, kd, vd, bt, seq_lens, scale, dt)
    assert np.array_equal(out8, out16)


@pytest.mark.parametrize("name", ["glue_f32", "glue_bf16"])
def test_glue_matches_reference_cpu_twins(golden_dir, name):
    """RMSNorm, fused add+RMSNorm, rotary embedding and SiLU-and-mul against the outputs of the
    reference's CPU twins.  The oracle restates the CUDA kernels, which round at slightly
    different points than the CPU twins (RMSNorm: `(scalar_t)(x * rstd) * w` in
    csrc/layernorm_kernels.cu:43-45 against an all-fp32 product in csrc/cpu/layernorm.cpp:30-36):
    fp32 agrees to a few ulp, bf16 to two bf16 ulp of the value."""
    z, dt = _load(golden_dir, name)
    f = (lambda a: O.from_bits(a, dt)) if dt != O.DT_F32 else (lambda a: np.asarray(a, np.float32))
    rtol = 4e-6 if dt == O.DT_F32 else 2**-6    # bf16: two roundings against one
    atol = 1e-6 if dt == O.DT_F32 else 2e-3
    eps = float(z["eps"])
    np.testing.assert_allclose(f(O.rms_norm(z["x"], z["weight"], eps, dt)), f(z["rms_out"]), rtol=rtol, atol=atol)
    fx, fr = O.fused_add_rms_norm(z["x"], z["residual"], z["weight"], eps, dt)
    np.testing.assert_allclose(f(fr), f(z["fused_residual"]), rtol=rtol, atol=atol)
    np.testing.assert_allclose(f(fx), f(z["fused_out"]), rtol=rtol, atol=2 * atol)
    # the CUDA rotary kernel multiplies and subtracts in the model dtype
    # (csrc/pos_encoding_kernels.cu:10-40), the CPU twin in fp32: where x cos - y sin cancels, a
    # bf16 result carries the rounding of its O(1) products
    rot_atol = atol if dt == O.DT_F32 else 3e-2
    for tag, neox in (("neox", True), ("gptj", False)):
        q, k = O.rotary_embedding(z["positions"], z[f"rot_{tag}_q"], z[f"rot_{tag}_k"], int(z["head_size"]),
                                  z[f"rot_{tag}_cache"], neox, dt)
        np.testing.assert_allclose(f(q), f(z[f"rot_{tag}_q_out"]), rtol=rtol, atol=rot_atol)
        np.testing.assert_allclose(f(k), f(z[f"rot_{tag}_k_out"]), rtol=rtol, atol=rot_atol)
    np.testing.assert_allclose(f(O.silu_and_mul(z["act_in"], dt)), f(z["act_out"]), rtol=rtol, atol=atol)
