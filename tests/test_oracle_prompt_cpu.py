"""CPU tests of the oracle's prompt-attention restatements against an independent float64 numpy
evaluation of the definition the reference tests use (ref_masked_attention,
tests/kernels/test_attention.py:43-55; the bottom-right causal mask with a cached prefix and the
local-attention window of tests/kernels/test_prefix_prefill.py:186-190).  The reference's Triton
kernels cannot run here and its tests hold no fixtures for them: these restatements are pinned by
definition only."""
import numpy as np
import pytest

from oracle import oracle as O


def _f64_attention(q, keys, vals, qpos0, scale, slope=None, window=0):
    """q [n, D], keys/vals [L, D] (positions 0..L-1), query i at position qpos0 + i."""
    n, L = q.shape[0], keys.shape[0]
    s = scale * (q.astype(np.float64) @ keys.astype(np.float64).T)
    qpos = qpos0 + np.arange(n)[:, None]
    kpos = np.arange(L)[None, :]
    if slope is not None:
        s = s + slope * (kpos - qpos)
    live = kpos <= qpos
    if window > 0:
        live &= (qpos - kpos) < window
    s = np.where(live, s, -np.inf)
    p = np.exp(s - s.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    return p @ vals.astype(np.float64)


@pytest.mark.parametrize("alibi", [False, True])
def test_prefill_oracle_matches_float64_definition(alibi):
    rng = np.random.default_rng(1)
    hq, hkv, d, lens = 4, 2, 32, [1, 7, 40]
    t = sum(lens)
    q = rng.uniform(-1, 1, (t, hq, d)).astype(np.float32)
    k = rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32)
    v = rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32)
    slopes = (0.5**np.arange(1, hq + 1)).astype(np.float32) if alibi else None
    got = O.prefill_attention(q, k, v, lens, d**-0.5, O.DT_F32, slopes)
    start = 0
    for n in lens:
        for h in range(hq):
            want = _f64_attention(q[start:start + n, h], k[start:start + n, h // 2], v[start:start + n, h // 2],
                                  0, d**-0.5, None if slopes is None else float(slopes[h]))
            np.testing.assert_allclose(got[start:start + n, h], want, atol=2e-6, rtol=1e-5)
        start += n


@pytest.mark.parametrize("window", [0, 5, 64])
@pytest.mark.parametrize("alibi", [False, True])
def test_prefix_prefill_oracle_matches_float64_definition(window, alibi):
    rng = np.random.default_rng(2 + window)
    hq, hkv, d, bs, x = 4, 2, 32, 16, 4
    qlens, ctx = [3, 20, 1], [37, 0, 16]
    nseq = len(qlens)
    max_blocks = 4
    nb = nseq * max_blocks
    table = rng.permutation(nb).reshape(nseq, max_blocks).astype(np.int32)
    kc = np.full((nb, hkv, d // x, bs, x), np.nan, dtype=np.float32)
    vc = np.full((nb, hkv, d, bs), np.nan, dtype=np.float32)
    ctx_k = [rng.uniform(-1, 1, (c, hkv, d)).astype(np.float32) for c in ctx]
    ctx_v = [rng.uniform(-1, 1, (c, hkv, d)).astype(np.float32) for c in ctx]
    for s in range(nseq):
        for j in range(ctx[s]):
            b, off = table[s, j // bs], j % bs
            kc[b, :, :, off, :] = ctx_k[s][j].reshape(hkv, d // x, x)
            vc[b, :, :, off] = ctx_v[s][j]
    t = sum(qlens)
    q = rng.uniform(-1, 1, (t, hq, d)).astype(np.float32)
    k = rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32)
    v = rng.uniform(-1, 1, (t, hkv, d)).astype(np.float32)
    slopes = (0.5**np.arange(1, hq + 1)).astype(np.float32) if alibi else None
    got = O.prefix_prefill_attention(q, k, v, kc, vc, table, qlens, ctx, d**-0.5, O.DT_F32,
                                     alibi_slopes=slopes, sliding_window=window)
    assert np.isfinite(got).all()          # nothing read past a context (those slots hold NaN)
    start = 0
    for s, n in enumerate(qlens):
        for h in range(hq):
            keys = np.concatenate([ctx_k[s][:, h // 2], k[start:start + n, h // 2]])
            vals = np.concatenate([ctx_v[s][:, h // 2], v[start:start + n, h // 2]])
            want = _f64_attention(q[start:start + n, h], keys, vals, ctx[s], d**-0.5,
                                  None if slopes is None else float(slopes[h]), window)
            np.testing.assert_allclose(got[start:start + n, h], want, atol=2e-6, rtol=1e-5)
        start += n


def test_prefix_prefill_without_context_is_prefill():
    rng = np.random.default_rng(3)
    hq, hkv, d, lens = 4, 4, 16, [5, 33]
    t = sum(lens)
    q, k, v = [O.to_bits(rng.uniform(-1, 1, (t, h, d)).astype(np.float32), O.DT_F16) for h in (hq, hkv, hkv)]
    kc = np.zeros((2, hkv, d // 8, 16, 8), dtype=np.uint16)
    vc = np.zeros((2, hkv, d, 16), dtype=np.uint16)
    a = O.prefill_attention(q, k, v, lens, 0.25, O.DT_F16)
    b = O.prefix_prefill_attention(q, k, v, kc, vc, np.zeros((2, 1), np.int32), lens, [0, 0], 0.25, O.DT_F16)
    assert np.array_equal(a, b)
