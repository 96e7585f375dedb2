"""The model-facing `Attention` layer (the role of vllm/attention/layer.py:14-147) reaches what the
library implements: a sliding window (a Mistral-style layer) in prefix-enabled prompt attention,
and local + strided block-sparse attention (the reference's block-sparse backend,
vllm/attention/backends/blocksparse_attn.py) in prompt and decode.  Expectations: the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T
from tests.test_gpu_prefix_prefill import _case

pytestmark = pytest.mark.gpu


def _meta(**kw):
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    base = dict(num_prefills=0, num_prefill_tokens=0, num_decode_tokens=0, slot_mapping=None,
                seq_lens=None, seq_lens_tensor=None, max_query_len=None, max_prefill_seq_len=0,
                max_decode_seq_len=0, query_start_loc=None, seq_start_loc=None, use_cuda_graph=False,
                context_lens_tensor=None, block_tables=None)
    base.update(kw)
    return Gfx950PagedAttentionMetadata(**base)


def test_selector_accepts_window_and_blocksparse():
    from vllm_legacy_amd.attention.selector import get_attn_backend
    a = get_attn_backend(32, 128, 8, 4096, torch.float16, "fp8", 16)
    b = get_attn_backend(32, 128, 8, None, torch.float16, "auto", 16, True)
    assert a is b and a.get_name() == "gfx950-paged-attn"


@pytest.mark.parametrize("window", [48, 200])
def test_attention_layer_with_sliding_window_runs_prefix_prefill(window):
    from vllm_legacy_amd.attention.layer import Attention
    dev = torch.device("cuda", 0)
    hq, hkv, d, bs, dt = 8, 2, 128, 16, O.DT_F16
    qlens, ctx = [40, 7, 130], [100, 64, 33]
    q, k, v, kc, vc, table, _ = _case(23, qlens, ctx, hq, hkv, d, bs, dt, False)
    scale = d**-0.5
    want = O.prefix_prefill_attention(q, k, v, kc.copy(), vc.copy(), table, qlens, ctx, scale, dt,
                                      sliding_window=window)
    tq, tk, tv, tkc, tvc = [T.from_bits(a, dt, "cuda") for a in (q, k, v, kc, vc)]
    nb = tkc.shape[0]
    cache = torch.stack([tkc.reshape(nb, -1), tvc.reshape(nb, -1)])
    total = np.add(ctx, qlens)
    meta = _meta(num_prefills=len(qlens), num_prefill_tokens=sum(qlens),
                 slot_mapping=torch.full((sum(qlens), ), -1, dtype=torch.int64, device=dev),
                 seq_lens=[int(a) for a in total],
                 seq_lens_tensor=torch.tensor(total, dtype=torch.int32, device=dev),
                 max_query_len=max(qlens), max_prefill_seq_len=int(total.max()),
                 query_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(qlens)]),
                                              dtype=torch.int32, device=dev),
                 seq_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(total)]),
                                            dtype=torch.int32, device=dev),
                 context_lens_tensor=torch.tensor(ctx, dtype=torch.int32, device=dev),
                 block_tables=torch.from_numpy(table).cuda())
    layer = Attention(hq, d, scale, hkv, sliding_window=window, kv_cache_dtype="auto", block_size=bs)
    out = layer(tq.view(-1, hq * d), tk.view(-1, hkv * d), tv.view(-1, hkv * d), cache, meta)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.view(-1, hq, d).float().cpu().numpy(), O.from_bits(want, dt),
                               atol=1e-3, rtol=1e-3)


def _bs_mask(n_q_pos, n, head, kv_head, hq, hkv, bsp):
    """Keys a query at position n_q_pos (0-based, attending keys 0..n_q_pos) may see."""
    step = bsp.head_sliding_step
    off = ((bsp.tp_rank * hq + head) * step + 1) if step >= 0 else \
        ((bsp.tp_rank * hkv + kv_head) * (-step) + 1)
    t = np.arange(n)
    kb, qb = t // bsp.block_size, n_q_pos // bsp.block_size
    return (t <= n_q_pos) & (((kb + off) % bsp.vert_stride == 0) | (kb > qb - bsp.local_blocks))


@pytest.mark.parametrize("homo_head", [False, True])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_attention_layer_blocksparse_prompt_then_decode(homo_head, kv_fp8):
    """Prompt through the layer (torch masked attention), cache write, then one decode step on the
    block-sparse kernels; both against float64 over the attended keys, decode also against the
    oracle's restatement of the kernel mask."""
    from vllm_legacy_amd.attention.layer import Attention
    dev = torch.device("cuda", 0)
    hq, hkv, d, bs, dt = 8, 2, 128, 16, O.DT_F16
    lens = [150, 333]
    params = dict(max_seqlen=512, block_size=32, local_blocks=2, vert_stride=3, homo_head=homo_head)
    scale = d**-0.5
    layer = Attention(hq, d, scale, hkv, kv_cache_dtype="fp8" if kv_fp8 else "auto", block_size=bs,
                      blocksparse_params=params)
    bsp = layer.impl.blocksparse
    assert bsp.head_sliding_step == (0 if homo_head else 1)
    rng = np.random.default_rng(3)
    t = sum(lens)
    q, k, v = [O.to_bits(rng.uniform(-1, 1, (t, h, d)).astype(np.float32), dt) for h in (hq, hkv, hkv)]
    tq, tk, tv = [T.from_bits(a, dt, "cuda") for a in (q, k, v)]
    mb = (max(lens) + 1 + bs - 1) // bs
    nb = 2 * mb
    table = rng.permutation(nb).reshape(2, mb).astype(np.int32)
    x = 16 if kv_fp8 else 8
    cache = torch.zeros((2, nb, bs * hkv * d), dtype=torch.uint8 if kv_fp8 else torch.float16, device=dev)
    slots = np.concatenate([table[s, np.arange(n) // bs].astype(np.int64) * bs + np.arange(n) % bs
                            for s, n in enumerate(lens)])
    meta = _meta(num_prefills=2, num_prefill_tokens=t, slot_mapping=torch.from_numpy(slots).cuda(),
                 seq_lens=lens, seq_lens_tensor=torch.tensor(lens, dtype=torch.int32, device=dev),
                 max_query_len=max(lens), max_prefill_seq_len=max(lens),
                 query_start_loc=torch.tensor([0, lens[0], t], dtype=torch.int32, device=dev),
                 seq_start_loc=torch.tensor([0, lens[0], t], dtype=torch.int32, device=dev),
                 context_lens_tensor=torch.zeros(2, dtype=torch.int32, device=dev),
                 block_tables=torch.empty((2, 0), dtype=torch.int32, device=dev))
    out = layer(tq.view(t, -1), tk.view(t, -1), tv.view(t, -1), cache, meta).view(t, hq, d)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    qf, kf, vf = [O.from_bits(a, dt).astype(np.float64) for a in (q, k, v)]
    start = 0
    for n in lens:
        for i in (0, 31, 32, 95, n - 1):
            for h in range(hq):
                m = _bs_mask(i, n, h, h // 4, hq, hkv, bsp)
                lg = scale * (kf[start:start + n][m, h // 4] @ qf[start + i, h])
                p = np.exp(lg - lg.max())
                ref = (p / p.sum()) @ vf[start:start + n][m, h // 4]
                np.testing.assert_allclose(got[start + i, h], ref, atol=2e-3, rtol=1e-3)
        start += n

    # ---- one decode step per sequence on top of the cached prompts
    kc_np = np.zeros((nb, hkv, d // x, bs, x), dtype=np.uint8 if kv_fp8 else np.uint16)
    vc_np = np.zeros((nb, hkv, d, bs), dtype=kc_np.dtype)
    kv = O.KV_FP8 if kv_fp8 else O.KV_AUTO
    O.reshape_and_cache(k, v, kc_np, vc_np, slots, dt, kv=kv)
    assert np.array_equal(T.bits(cache[0]).reshape(kc_np.shape), kc_np)    # the prompt's cache write
    q1, k1, v1 = [O.to_bits(rng.uniform(-1, 1, (2, h, d)).astype(np.float32), dt) for h in (hq, hkv, hkv)]
    new_lens = np.array([n + 1 for n in lens], dtype=np.int32)
    slots1 = np.array([int(table[s, n // bs]) * bs + n % bs for s, n in enumerate(lens)], dtype=np.int64)
    O.reshape_and_cache(k1, v1, kc_np, vc_np, slots1, dt, kv=kv)
    dmeta = _meta(num_decode_tokens=2, slot_mapping=torch.from_numpy(slots1).cuda(),
                  seq_lens_tensor=torch.from_numpy(new_lens).cuda(), max_decode_seq_len=int(new_lens.max()),
                  block_tables=torch.from_numpy(table).cuda())
    tq1, tk1, tv1 = [T.from_bits(a, dt, "cuda") for a in (q1, k1, v1)]
    dout = layer(tq1.view(2, -1), tk1.view(2, -1), tv1.view(2, -1), cache, dmeta).view(2, hq, d)
    torch.cuda.synchronize()
    want = O.paged_attention_v1(q1, kc_np, vc_np, table, new_lens, scale, dt, kv=kv,
                                blocksparse=(0, bsp.local_blocks, bsp.vert_stride, bsp.block_size,
                                             bsp.head_sliding_step))
    np.testing.assert_allclose(dout.float().cpu().numpy(), O.from_bits(want, dt),
                               atol=1e-2 if kv_fp8 else 1e-3, rtol=1e-5)


def test_blocksparse_rejects_alibi_and_window():
    from vllm_legacy_amd.attention.layer import Attention
    params = dict(max_seqlen=512, block_size=32, local_blocks=2, vert_stride=3)
    with pytest.raises(ValueError):
        Attention(8, 128, 0.1, 2, alibi_slopes=[0.5] * 8, blocksparse_params=params)
    with pytest.raises(ValueError):
        Attention(8, 128, 0.1, 2, sliding_window=64, blocksparse_params=params)
