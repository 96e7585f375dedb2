"""GPU parity tests of the HIP prompt-attention kernel against the oracle's restatement of the
reference expectation (ref_masked_attention with a causal mask per sequence,
tests/kernels/test_attention.py:43-55, :333-366; tolerances of that test: atol = rtol = 1e-3 for
fp16, looser for bf16), on packed variable-length prompts with grouped KV heads."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


@pytest.fixture(autouse=True, params=[1, 2], ids=["wg256", "wg512"])
def form(request, ops):
    """Every test runs with each form of the kernel forced: the 256-thread one (a 128-query tile
    of one head per workgroup) and the 512-thread one (two heads of a KV group, or two query
    tiles, per workgroup); by default the longest prompt of a call picks."""
    ops.set_prefill_form(request.param)
    yield request.param
    ops.set_prefill_form(0)


def _run(ops, seq_lens, hq, hkv, d, dt, alibi, seed, strided=False):
    rng = np.random.default_rng(seed)
    t = int(sum(seq_lens))
    scale = d**-0.5
    n = (hq + 2 * hkv) * d
    qkv = O.to_bits(rng.uniform(-1, 1, (t, n)).astype(np.float32), dt)
    q = np.ascontiguousarray(qkv[:, :hq * d]).reshape(t, hq, d)
    k = np.ascontiguousarray(qkv[:, hq * d:(hq + hkv) * d]).reshape(t, hkv, d)
    v = np.ascontiguousarray(qkv[:, (hq + hkv) * d:]).reshape(t, hkv, d)
    slopes = (0.5**np.arange(1, hq + 1)).astype(np.float32) if alibi else None
    want = O.prefill_attention(q, k, v, seq_lens, scale, dt, slopes)
    if strided:  # q/k/v as column slices of one qkv tensor, the way the model hands them over
        t_qkv = T.from_bits(qkv, dt, "cuda")
        tq, tk, tv = [x.view(t, -1, d) for x in t_qkv.split([hq * d, hkv * d, hkv * d], dim=-1)]
    else:
        tq, tk, tv = T.from_bits(q, dt, "cuda"), T.from_bits(k, dt, "cuda"), T.from_bits(v, dt, "cuda")
    out = torch.full((t, hq, d), float("nan"), dtype=tq.dtype, device="cuda")
    cu = torch.tensor(np.concatenate([[0], np.cumsum(seq_lens)]), dtype=torch.int32, device="cuda")
    ops.prefill_attention(out, tq, tk, tv, cu, int(max(seq_lens)), scale,
                          None if slopes is None else torch.from_numpy(slopes).cuda())
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    tol = 1e-3 if dt == O.DT_F16 else 8e-3
    np.testing.assert_allclose(got, O.from_bits(want, dt), atol=tol, rtol=tol)


@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("heads", [(8, 8), (8, 2), (4, 1)])
@pytest.mark.parametrize("alibi", [False, True])
def test_prefill_matches_oracle(ops, dt, d, heads, alibi):
    _run(ops, [1, 17, 64, 129, 300, 5], heads[0], heads[1], d, dt, alibi, seed=d + heads[1])


@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
@pytest.mark.parametrize("d", [24, 80, 96, 112])
def test_prefill_other_head_sizes(ops, dt, d):
    _run(ops, [1, 17, 64, 129, 300, 5], 4, 2, d, dt, False, seed=d)


@pytest.mark.parametrize("heads", [(6, 2), (3, 3), (2, 1)])
def test_prefill_group_sizes(ops, heads):
    """odd KV group sizes and no grouping: the 512-thread form then pairs query tiles, not heads"""
    _run(ops, [300, 1, 257, 64, 640], heads[0], heads[1], 128, O.DT_F16, False, seed=heads[0])


@pytest.mark.parametrize("case", [((8, 2), [32, 7, 1, 20]), ((8, 2), [64, 33, 50]), ((8, 1), [16, 31]),
                                  ((6, 3), [30, 64]), ((4, 4), [20, 5])])
@pytest.mark.parametrize("d", [128, 64])
def test_prefill_short_prompts_share_tiles(ops, case, d):
    """prompts of at most 64 / 32 tokens: 2 / 4 heads of a KV group per 128-row tile (256-thread form)"""
    _run(ops, case[1], case[0][0], case[0][1], d, O.DT_BF16, case[0][0] == 6, seed=d + len(case[1]))


@pytest.mark.parametrize("heads", [(4, 2), (2, 2)])
def test_prefill_long_prompts(ops, heads, form):
    """prompts long enough for the default choice of form to differ from the short tests'"""
    if form == 2:
        ops.set_prefill_form(0)
    _run(ops, [3100, 100, 2177], heads[0], heads[1], 128, O.DT_F16, False, seed=11, strided=True)


@pytest.mark.parametrize("seq_lens", [[128], [127, 129], [63, 64, 65], [1, 1, 1], [513]])
def test_prefill_tile_boundaries(ops, seq_lens):
    _run(ops, seq_lens, 4, 2, 128, O.DT_F16, False, seed=len(seq_lens), strided=True)


def test_prefill_rejects_unsupported(ops):
    q = torch.zeros((4, 2, 72), dtype=torch.float16, device="cuda")
    cu = torch.tensor([0, 4], dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="head size"):
        ops.prefill_attention(torch.empty_like(q), q, q, q, cu, 4, 1.0)
    q32 = torch.zeros((4, 2, 64), dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError):
        ops.prefill_attention(torch.empty_like(q32), q32, q32, q32, cu, 4, 1.0)


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_backend_prefill_then_decode_is_consistent(kv):
    """Through the attention backend: prefill n tokens of each prompt (cache write + prompt
    attention), then decode token n+1 from the paged cache.  The decode output must equal the
    last row of a prefill over n+1 tokens (exactly the same keys and values; with an fp8 cache
    the decode side sees quantised K/V, so the bar is the fp8 one)."""
    from vllm_legacy_amd.attention.backends.gfx950_paged import (Gfx950PagedAttentionBackend,
                                                                 Gfx950PagedAttentionMetadata)
    dev = torch.device("cuda", 0)
    hq, hkv, d, bs = 8, 2, 128, 16
    lens = [5, 33, 100, 257]
    g = torch.Generator(device=dev).manual_seed(3)
    full = [torch.empty((n + 1, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
            for n in lens]
    impl = Gfx950PagedAttentionBackend.get_impl_cls()(hq, d, d**-0.5, hkv, None, None, kv, None)
    max_blocks = (max(lens) + 1 + bs - 1) // bs
    nb = len(lens) * max_blocks
    cache = torch.zeros(Gfx950PagedAttentionBackend.get_kv_cache_shape(nb, bs, hkv, d),
                        dtype=torch.uint8 if kv == "fp8" else torch.float16, device=dev)
    bt = torch.arange(nb, dtype=torch.int32, device=dev).view(len(lens), max_blocks)

    def split(x):
        return x.split([hq * d, hkv * d, hkv * d], dim=-1)

    # ---- prefill of the first n tokens of every prompt
    pre = torch.cat([f[:n] for f, n in zip(full, lens)])
    slots = torch.cat([bt[i, torch.arange(n, device=dev) // bs].long() * bs + torch.arange(n, device=dev) % bs
                       for i, n in enumerate(lens)])
    meta = Gfx950PagedAttentionMetadata(
        num_prefills=len(lens), num_prefill_tokens=sum(lens), num_decode_tokens=0, slot_mapping=slots,
        seq_lens=lens, seq_lens_tensor=torch.tensor(lens, dtype=torch.int32, device=dev),
        max_query_len=max(lens), max_prefill_seq_len=max(lens), max_decode_seq_len=0,
        query_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev),
        seq_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev),
        use_cuda_graph=False, context_lens_tensor=torch.zeros(len(lens), dtype=torch.int32, device=dev),
        block_tables=bt)
    q, k, v = split(pre)
    out_pre = impl.forward(q, k, v, cache, meta, 1.0)
    # ---- decode of token n+1
    dec = torch.stack([f[n] for f, n in zip(full, lens)])
    dslots = torch.stack([bt[i, n // bs].long() * bs + n % bs for i, n in enumerate(lens)])
    dmeta = Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=len(lens), slot_mapping=dslots,
        seq_lens=None, seq_lens_tensor=torch.tensor([n + 1 for n in lens], dtype=torch.int32, device=dev),
        max_query_len=None, max_prefill_seq_len=0, max_decode_seq_len=max(lens) + 1,
        query_start_loc=None, seq_start_loc=None, use_cuda_graph=False, context_lens_tensor=None,
        block_tables=bt)
    q, k, v = split(dec)
    out_dec = impl.forward(q, k, v, cache, dmeta, 1.0)
    # ---- reference: prefill over n+1 tokens, last row of each prompt
    allp = torch.cat(full)
    lens1 = [n + 1 for n in lens]
    q, k, v = [x.view(-1, h, d) for x, h in zip(split(allp), (hq, hkv, hkv))]
    ref = torch.empty((allp.shape[0], hq, d), dtype=torch.float16, device=dev)
    from vllm_legacy_amd import _custom_ops as ops
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens1)]), dtype=torch.int32, device=dev)
    ops.prefill_attention(ref, q, k, v, cu, max(lens1), d**-0.5)
    torch.cuda.synchronize()
    last = torch.tensor(np.cumsum(lens1) - 1, device=dev)
    tol = 2e-2 if kv == "fp8" else 2e-3
    torch.testing.assert_close(out_dec.view(len(lens), hq, d).float(), ref[last].float(), atol=tol, rtol=tol)
    # and the prefill rows themselves agree with the longer prefill's rows
    keep = torch.cat([torch.arange(n, device=dev) + off for n, off in zip(lens, np.concatenate([[0], np.cumsum(lens1)[:-1]]))])
    torch.testing.assert_close(out_pre.view(-1, hq, d).float(), ref[keep].float(), atol=2e-3, rtol=2e-3)
