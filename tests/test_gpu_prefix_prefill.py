"""GPU parity tests of prompt attention over a cached prefix (the replacement of the reference's
Triton context_attention_fwd) against the oracle, shaped after the reference's own test
(tests/kernels/test_prefix_prefill.py:20-208: random context and query lengths per sequence,
shuffled block table, grouped KV heads, head sizes 128 and 96, block size 32, sliding windows,
ALiBi; its bar is the xformers result within atol 1e-6 on +-1e-3 inputs -- here the inputs are
+-1 and the bar is the prompt-attention one, atol = rtol = 1e-3 for fp16)."""
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
    """Every test runs with each form of the kernel forced (256 threads: one 128-query tile of one
    head per workgroup; 512 threads: two heads of a KV group, or two query tiles); by default the
    longest query run of a call picks."""
    ops.set_prefill_form(request.param)
    yield request.param
    ops.set_prefill_form(0)


def _case(seed, query_lens, ctx_lens, hq, hkv, d, bs, dt, kv_fp8, kv_scale=1.0):
    """Builds the packed new tokens and a shuffled paged cache holding each sequence's context.
    The cache is written through the oracle's reshape_and_cache, so that an fp8 cache holds what
    the product's cache writer would have put there."""
    rng = np.random.default_rng(seed)
    x = 16 if kv_fp8 else 8
    nseq = len(query_lens)
    max_blocks = max(1, max((c + bs - 1) // bs for c in ctx_lens)) + 1
    nb = nseq * max_blocks + 3
    table = rng.permutation(nb)[:nseq * max_blocks].reshape(nseq, max_blocks).astype(np.int32)
    kv = O.KV_FP8 if kv_fp8 else O.KV_AUTO
    cdt = np.uint8 if kv_fp8 else np.uint16
    # unused slots hold a NaN pattern: nothing past a sequence's context may leak into a result
    fill = 0x7f if kv_fp8 else (0x7e00 if dt == O.DT_F16 else 0x7fc0)
    kc = np.full((nb, hkv, d // x, bs, x), fill, dtype=cdt)
    vc = np.full((nb, hkv, d, bs), fill, dtype=cdt)
    tctx = int(sum(ctx_lens))
    if tctx:
        ck = O.to_bits(rng.uniform(-1, 1, (tctx, hkv, d)).astype(np.float32), dt)
        cv = O.to_bits(rng.uniform(-1, 1, (tctx, hkv, d)).astype(np.float32), dt)
        slots = np.concatenate([table[s, np.arange(c) // bs].astype(np.int64) * bs + np.arange(c) % bs
                                for s, c in enumerate(ctx_lens)])
        O.reshape_and_cache(ck, cv, kc, vc, slots, dt, kv=kv, kv_scale=kv_scale)
    t = int(sum(query_lens))
    q = O.to_bits(rng.uniform(-1, 1, (t, hq, d)).astype(np.float32), dt)
    k = O.to_bits(rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32), dt)
    v = O.to_bits(rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32), dt)
    return q, k, v, kc, vc, table, kv


def _run(ops, seed, query_lens, ctx_lens, hq, hkv, d, bs, dt, kv_fp8=False, kv_scale=1.0,
         alibi=False, window=0):
    q, k, v, kc, vc, table, kv = _case(seed, query_lens, ctx_lens, hq, hkv, d, bs, dt, kv_fp8, kv_scale)
    scale = d**-0.5
    slopes = (0.5**np.arange(1, hq + 1)).astype(np.float32) if alibi else None
    want = O.prefix_prefill_attention(q, k, v, kc, vc, table, query_lens, ctx_lens, scale, dt, kv=kv,
                                      kv_scale=kv_scale, alibi_slopes=slopes, sliding_window=window)
    tq, tk, tv = [T.from_bits(a, dt, "cuda") for a in (q, k, v)]
    tkc = torch.from_numpy(kc).cuda() if kv_fp8 else T.from_bits(kc, dt, "cuda")
    tvc = torch.from_numpy(vc).cuda() if kv_fp8 else T.from_bits(vc, dt, "cuda")
    out = torch.full(tq.shape, float("nan"), dtype=tq.dtype, device="cuda")
    qs = torch.tensor(np.concatenate([[0], np.cumsum(query_lens)]), dtype=torch.int32, device="cuda")
    ctx = torch.tensor(ctx_lens, dtype=torch.int32, device="cuda")
    sl = torch.tensor(np.add(ctx_lens, query_lens), dtype=torch.int32, device="cuda")
    ops.prefix_prefill_attention(out, tq, tk, tv, tkc, tvc, torch.from_numpy(table).cuda(), qs, sl, ctx,
                                 int(max(query_lens)), scale,
                                 None if slopes is None else torch.from_numpy(slopes).cuda(), window,
                                 "fp8" if kv_fp8 else "auto", kv_scale)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    tol = 1e-3 if dt == O.DT_F16 else 8e-3
    np.testing.assert_allclose(got, O.from_bits(want, dt), atol=tol, rtol=tol)


def _random_lens(seed, n, lo, hi):
    rng = np.random.default_rng(seed)
    return [int(a) for a in rng.integers(lo, hi, n)], [int(a) for a in rng.integers(lo, hi, n)]


@pytest.mark.parametrize("qpk", [1, 4, 8])
@pytest.mark.parametrize("d", [128, 96])
@pytest.mark.parametrize("window", [0, 16, 64, 128, 256])
def test_contexted_kv_attention(ops, qpk, d, window):
    """The reference test's grid at a size the oracle finishes quickly: block size 32, fp16."""
    ql, cl = _random_lens(qpk + d + window, 6, 16, 300)
    _run(ops, window + d, ql, cl, 8, 8 // qpk, d, 32, O.DT_F16, window=window)


@pytest.mark.parametrize("qpk", [1, 8])
@pytest.mark.parametrize("d", [128, 64])
def test_contexted_kv_attention_alibi(ops, qpk, d):
    ql, cl = _random_lens(7 * qpk + d, 5, 16, 300)
    _run(ops, d + qpk, ql, cl, 8, 8 // qpk, d, 32, O.DT_F16, alibi=True)


@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
@pytest.mark.parametrize("d", [64, 80, 96, 112, 128])
@pytest.mark.parametrize("bs", [16, 32])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_prefix_shapes(ops, dt, d, bs, kv_fp8):
    _run(ops, d + bs, [1, 40, 129, 200], [100, 0, 63, 257], 4, 2, d, bs, dt, kv_fp8=kv_fp8)


@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
@pytest.mark.parametrize("qpk", [1, 8])
@pytest.mark.parametrize("bs", [8, 32])
@pytest.mark.parametrize("window", [0, 64])
def test_prefix_head_size_24(ops, dt, qpk, bs, window):
    """The smallest head of the reference's grid (tests/kernels/test_prefix_prefill.py HEAD_SIZES
    = [128, 96, 24]): three 8-dim chunks per row, so the last 16-deep slice of the q.K product is
    half padding.  Unused cache slots hold NaN patterns in _run, so any padding that is read
    without being zeroed shows."""
    ql, cl = _random_lens(24 + qpk + bs, 5, 1, 200)
    _run(ops, 24 + bs + window, ql, cl, 8, 8 // qpk, 24, bs, dt, window=window)


def test_prefix_head_size_24_needs_a_16_bit_cache(ops):
    """A byte cache packs 16 dims per K piece ([D/16][BS][16]): no such layout exists for 24 dims,
    and the entry point says so (the reference's 'Unsupported head size') before it reads one."""
    from vllm_legacy_amd import _lib
    assert not ops.prefix_prefill_supported(torch.float16, 24, 16, "fp8")
    assert ops.prefix_prefill_supported(torch.float16, 24, 16, "auto")
    buf = torch.zeros(4096, dtype=torch.int32, device="cuda")
    p = buf.data_ptr()
    st = _lib.lib().vla_prefix_prefill_attention(
        p, p, p, p, p, p, p, 4, p, p, p, 1, 5, 4, 4, 24, 16, 96, 96, 96, 96, 1536, 384, 0.2, None, 0,
        _lib.F16, _lib.kv_dtype_code("fp8"), 1.0, None)
    assert st == 1          # VLA_ERR_HEAD_SIZE


@pytest.mark.parametrize("case", [
    (8, 2, [32, 7, 1, 20], [300, 64, 129, 0]),      # runs of <= 32 tokens, groups of 4: four heads per tile
    (8, 2, [64, 33, 50], [200, 17, 640]),           # <= 64 tokens: two heads per tile
    (8, 1, [16, 31], [257, 100]),                   # groups of 8
    (6, 3, [30, 64], [90, 70]),                     # groups of 2
    (4, 4, [20, 5], [100, 260]),                    # no grouping: one head per tile
])
@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("window", [0, 48])
def test_prefix_short_query_runs(ops, case, kv_fp8, window):
    """A chunk of a prompt (or a few tokens to verify) behind a cached prefix: the 256-thread form
    packs 2 or 4 heads of a KV group into one 128-row tile."""
    hq, hkv, ql, cl = case
    _run(ops, hq + len(ql) + window, ql, cl, hq, hkv, 128, 16, O.DT_F16, kv_fp8=kv_fp8, window=window,
         alibi=(hq == 6))


@pytest.mark.parametrize("bs", [8, 64])
def test_prefix_other_block_sizes(ops, bs):
    _run(ops, bs, [33, 5, 130], [70, 129, 8], 4, 1, 128, bs, O.DT_F16)


def test_prefix_fp8_scaled(ops):
    """A scaled fp8 cache is decoded with the scale that wrote it."""
    _run(ops, 5, [64, 17], [200, 31], 8, 2, 128, 16, O.DT_F16, kv_fp8=True, kv_scale=0.5)
    _run(ops, 6, [64, 17], [200, 31], 8, 2, 128, 16, O.DT_BF16, kv_fp8=True, kv_scale=2.0)


@pytest.mark.parametrize("lens", [([128], [64]), ([128], [63]), ([127, 129], [65, 1]),
                                  ([1, 1, 1], [1, 64, 128]), ([300], [0])])
def test_prefix_tile_boundaries(ops, lens):
    _run(ops, len(lens[0]), lens[0], lens[1], 4, 2, 128, 16, O.DT_F16)


def test_prefix_equals_plain_prefill(ops):
    """A prompt cut in two -- the first part cached, the second attending it -- gives the rows a
    single prompt-attention call over the whole prompt gives for the second part (same keys and
    values, 16-bit cache: agreement to rounding of the different summation order)."""
    hq, hkv, d, bs = 8, 2, 128, 16
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(11)
    lens, cuts = [300, 77, 1000], [100, 64, 513]
    whole = [torch.empty((n, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
             for n in lens]

    def split(x):
        q, k, v = x.split([hq * d, hkv * d, hkv * d], dim=-1)
        return q.reshape(-1, hq, d), k.reshape(-1, hkv, d), v.reshape(-1, hkv, d)

    q, k, v = split(torch.cat(whole))
    ref = torch.empty_like(q)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    ops.prefill_attention(ref, q, k, v, cu, max(lens), d**-0.5)
    # cache the first `cut` tokens of each prompt
    max_blocks = (max(cuts) + bs - 1) // bs
    nb = len(lens) * max_blocks
    kc = torch.zeros((nb, hkv, d // 8, bs, 8), dtype=torch.float16, device=dev)
    vc = torch.zeros((nb, hkv, d, bs), dtype=torch.float16, device=dev)
    bt = torch.randperm(nb, device=dev, generator=g).to(torch.int32).view(len(lens), max_blocks)
    head = torch.cat([w[:c] for w, c in zip(whole, cuts)])
    _, hk, hv = split(head)
    slots = torch.cat([bt[i, torch.arange(c, device=dev) // bs].long() * bs + torch.arange(c, device=dev) % bs
                       for i, c in enumerate(cuts)])
    ops.reshape_and_cache(hk, hv, kc, vc, slots, "auto", 1.0)
    tail = torch.cat([w[c:] for w, c in zip(whole, cuts)])
    tq, tk, tv = split(tail)
    qlens = [n - c for n, c in zip(lens, cuts)]
    out = torch.empty_like(tq)
    ops.prefix_prefill_attention(
        out, tq, tk, tv, kc, vc, bt,
        torch.tensor(np.concatenate([[0], np.cumsum(qlens)]), dtype=torch.int32, device=dev),
        torch.tensor(lens, dtype=torch.int32, device=dev), torch.tensor(cuts, dtype=torch.int32, device=dev),
        max(qlens), d**-0.5)
    torch.cuda.synchronize()
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    keep = torch.cat([torch.arange(c, n, device=dev) + int(s) for n, c, s in zip(lens, cuts, starts)])
    torch.testing.assert_close(out.float(), ref[keep].float(), atol=1e-3, rtol=1e-3)


def test_prefix_rejects_unsupported(ops):
    q = torch.zeros((4, 2, 128), dtype=torch.float16, device="cuda")
    i32 = dict(dtype=torch.int32, device="cuda")
    qs, sl, ctx = torch.tensor([0, 4], **i32), torch.tensor([8], **i32), torch.tensor([4], **i32)
    bt = torch.zeros((1, 2), **i32)
    kc = torch.zeros((2, 2, 16, 4, 8), dtype=torch.float16, device="cuda")    # block size 4
    vc = torch.zeros((2, 2, 128, 4), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="block size"):
        ops.prefix_prefill_attention(torch.empty_like(q), q, q, q, kc, vc, bt, qs, sl, ctx, 4, 1.0)
    q72 = torch.zeros((4, 2, 72), dtype=torch.float16, device="cuda")
    kc = torch.zeros((2, 2, 9, 16, 8), dtype=torch.float16, device="cuda")
    vc = torch.zeros((2, 2, 72, 16), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="head size"):
        ops.prefix_prefill_attention(torch.empty_like(q72), q72, q72, q72, kc, vc, bt, qs, sl, ctx, 4, 1.0)


def test_backend_passes_its_sliding_window_to_prefix_prefill(ops):
    """The backend's constructor takes a sliding window as the reference's ROCm backend does
    (rocm_flash_attn.py:215-216) and hands it to prefix-enabled prompt attention (:382): same
    result as the op called with that window, different from the window-less result."""
    from vllm_legacy_amd.attention.backends.gfx950_paged import (Gfx950PagedAttentionBackend,
                                                                 Gfx950PagedAttentionMetadata)
    dev = torch.device("cuda", 0)
    hq, hkv, d, bs, window = 8, 2, 128, 16, 48
    qlens, ctx = [40, 7, 130], [100, 64, 33]
    q, k, v, kc, vc, table, _ = _case(21, qlens, ctx, hq, hkv, d, bs, O.DT_F16, False)
    tq, tk, tv, tkc, tvc = [T.from_bits(a, O.DT_F16, "cuda") for a in (q, k, v, kc, vc)]
    nb = tkc.shape[0]
    cache = torch.stack([tkc.reshape(nb, -1), tvc.reshape(nb, -1)])
    qs = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]), dtype=torch.int32, device=dev)
    sl = torch.tensor(np.add(ctx, qlens), dtype=torch.int32, device=dev)
    cl = torch.tensor(ctx, dtype=torch.int32, device=dev)
    bt = torch.from_numpy(table).cuda()
    meta = Gfx950PagedAttentionMetadata(
        num_prefills=len(qlens), num_prefill_tokens=sum(qlens), num_decode_tokens=0,
        slot_mapping=torch.full((sum(qlens), ), -1, dtype=torch.int64, device=dev),
        seq_lens=[int(a) for a in np.add(ctx, qlens)], seq_lens_tensor=sl, max_query_len=max(qlens),
        max_prefill_seq_len=int(max(np.add(ctx, qlens))), max_decode_seq_len=0, query_start_loc=qs,
        seq_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(np.add(ctx, qlens))]), dtype=torch.int32,
                                   device=dev),
        use_cuda_graph=False, context_lens_tensor=cl, block_tables=bt)
    outs = {}
    for w in (None, window):
        impl = Gfx950PagedAttentionBackend.get_impl_cls()(hq, d, d**-0.5, hkv, None, w, "auto", None)
        outs[w] = impl.forward(tq.view(-1, hq * d), tk.view(-1, hkv * d), tv.view(-1, hkv * d), cache,
                               meta, 1.0).view(-1, hq, d)
    want = torch.empty_like(tq)
    ops.prefix_prefill_attention(want, tq, tk, tv, tkc, tvc, bt, qs, sl, cl, max(qlens), d**-0.5, None,
                                 window)
    torch.cuda.synchronize()
    assert torch.equal(outs[window], want)
    assert not torch.allclose(outs[None].float(), want.float(), atol=1e-2)
