"""Per-op comparison of the tiny decode stack against the oracle with identical inputs per op."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from tests import pa_testlib as T
from vllm_legacy_amd import _custom_ops as ops
from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
DT = O.DT_F16
kv = sys.argv[1] if len(sys.argv) > 1 else "fp8"
dev = torch.device("cuda", 0)
cfg = LlamaConfig.tiny()
model = LlamaDecodeModel(cfg, quant="awq", kv_cache_dtype=kv, device=dev, seed=3)
hq, hkv, d, bs = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, 16
seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64], dtype=np.int32)
B = len(seq_lens); mb = (int(seq_lens.max()) + bs - 1) // bs; nb = B * mb + 3
rng = np.random.default_rng(11)
bt = rng.permutation(nb)[:B * mb].reshape(B, mb).astype(np.int32)
pos = (seq_lens - 1).astype(np.int64)
slots = np.array([bt[b, p // bs] * bs + p % bs for b, p in enumerate(pos)], dtype=np.int64)
ids = rng.integers(0, cfg.vocab_size, B).astype(np.int64)
shape = model.kv_cache_shape(nb, bs)
g = torch.Generator(device=dev).manual_seed(5)
def f(a): return O.from_bits(a, DT).astype(np.float64)
def G(a): return T.from_bits(a, DT, dev)
def cmp(name, got_t, want_bits):
    gf, wf = got_t.float().cpu().numpy().astype(np.float64), f(want_bits)
    ulp = np.abs(gf - wf) / np.maximum(np.abs(wf) * 2.0**-10, 1e-8)
    print(f"{name:24s} max|x|={np.abs(wf).max():9.3f} max abs err={np.abs(gf-wf).max():.3e} max ulp={ulp.max():8.1f} frac>1ulp={(ulp>1.01).mean():.4f}")
x = 16 if kv == "fp8" else 8
kvc = O.KV_FP8 if kv == "fp8" else O.KV_AUTO
eps = cfg.rms_norm_eps
cs = T.bits(model.rotary_emb.cos_sin_cache)
h = T.bits(model.embed_tokens)[ids]; res = None
W = lambda lin: O.awq_dequantize(T.bits(lin.qweight), T.bits(lin.scales), T.bits(lin.qzeros))
with torch.inference_mode():
  for li, layer in enumerate(model.layers):
    src = torch.empty(shape, dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    if kv == "fp8":
        cache_t = torch.empty(shape, dtype=torch.uint8, device=dev); ops.convert_fp8(cache_t, src)
    else:
        cache_t = src
    ln1, ln2 = T.bits(layer.input_layernorm.weight), T.bits(layer.post_attention_layernorm.weight)
    if res is None:
        res = h.copy(); hn = O.rms_norm(h, ln1, eps, DT)
        cmp(f"L{li} rms_norm", layer.input_layernorm(G(h)), hn)
    else:
        gh, gr = layer.input_layernorm(G(h), G(res))
        hn, res = O.fused_add_rms_norm(h, res, ln1, eps, DT)
        cmp(f"L{li} add_rms_norm", gh, hn); cmp(f"L{li}   residual", gr, res)
    h = hn
    qkv = O.gemm_f16(h, W(layer.qkv_proj)); cmp(f"L{li} qkv_proj", layer.qkv_proj(G(h)), qkv)
    gq = G(qkv)
    q = np.ascontiguousarray(qkv[:, :hq*d]); k = np.ascontiguousarray(qkv[:, hq*d:(hq+hkv)*d]); v = np.ascontiguousarray(qkv[:, (hq+hkv)*d:])
    q, k = O.rotary_embedding(pos, q, k, d, cs, True, DT)
    tq, tk, tv = gq.split([hq*d, hkv*d, hkv*d], dim=-1)
    model.rotary_emb(torch.from_numpy(pos).to(dev), tq, tk)
    cmp(f"L{li} rotary q", tq, q); cmp(f"L{li} rotary k", tk, k)
    cache = T.bits(cache_t)
    kc = np.ascontiguousarray(cache[0]).reshape(nb, hkv, d // x, bs, x); vc = np.ascontiguousarray(cache[1]).reshape(nb, hkv, d, bs)
    O.reshape_and_cache(k.reshape(B, hkv, d), v.reshape(B, hkv, d), kc, vc, slots, DT, kvc, 1.0)
    attn = O.paged_attention_v1(q.reshape(B, hq, d), kc, vc, bt, seq_lens, d**-0.5, DT, kvc, 1.0)
    meta = Gfx950PagedAttentionMetadata(num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B,
        slot_mapping=torch.from_numpy(slots).to(dev), seq_lens=None, seq_lens_tensor=torch.from_numpy(seq_lens).to(dev),
        max_query_len=None, max_prefill_seq_len=0, max_decode_seq_len=int(seq_lens.max()), query_start_loc=None,
        seq_start_loc=None, use_cuda_graph=False, context_lens_tensor=None, block_tables=torch.from_numpy(bt).to(dev))
    ga = layer.attn(G(q), G(k), G(v), cache_t, meta)
    cmp(f"L{li} attention", ga, attn.reshape(B, hq*d))
    a2 = attn.reshape(B, hq*d)
    o = O.gemm_f16(a2, W(layer.o_proj)); cmp(f"L{li} o_proj", layer.o_proj(G(a2)), o)
    gh, gr = layer.post_attention_layernorm(G(o), G(res))
    h, res = O.fused_add_rms_norm(o, res, ln2, eps, DT); cmp(f"L{li} post_ln", gh, h)
    gu = O.gemm_f16(h, W(layer.gate_up_proj)); cmp(f"L{li} gate_up", layer.gate_up_proj(G(h)), gu)
    act = O.silu_and_mul(gu, DT); cmp(f"L{li} silu_mul", layer.act_fn(G(gu)), act)
    hd = O.gemm_f16(act, W(layer.down_proj)); cmp(f"L{li} down", layer.down_proj(G(act)), hd)
    h = hd
