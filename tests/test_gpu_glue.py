"""GPU parity tests for the layer-glue ops next to attention (RMSNorm, fused add + RMSNorm,
rotary embedding, SwiGLU gating).  Shapes follow the reference's tests
(tests/kernels/test_layernorm.py, test_pos_encoding.py, test_activation.py: tokens 7/83/4096,
hidden 768..8199, head sizes 64..256); expectations come from the oracle's restatement of the CUDA
kernels' rounding order.  Rotary is compared bit for bit; the norms and SiLU within one unit of
the model dtype's last place (rsqrt / exp differ between libm and the GPU; for float32 the bound
is 4e-6 relative, which also covers the different summation order of the variance)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu
DTYPES = [O.DT_F16, O.DT_BF16, O.DT_F32]
ULP = {O.DT_F16: 2.0**-10, O.DT_BF16: 2.0**-7, O.DT_F32: 2.0**-19}


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def _close(got, want_bits, dt):
    g, w = got.float().cpu().numpy(), O.from_bits(want_bits, dt)
    np.testing.assert_allclose(g, w, rtol=2 * ULP[dt], atol=1e-6)


@pytest.mark.parametrize("num_tokens,hidden", [(7, 768), (83, 5120), (256, 4096), (5, 8192)])
@pytest.mark.parametrize("dt", DTYPES)
def test_rms_norm_and_fused(ops, num_tokens, hidden, dt):
    rng = np.random.default_rng(hidden + num_tokens)
    x = O.to_bits(rng.standard_normal((num_tokens, hidden)).astype(np.float32), dt)
    r = O.to_bits(rng.standard_normal((num_tokens, hidden)).astype(np.float32), dt)
    w = O.to_bits((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32), dt)
    tx, tw = T.from_bits(x, dt, "cuda"), T.from_bits(w, dt, "cuda")
    out = torch.empty_like(tx)
    ops.rms_norm(out, tx, tw, 1e-6)
    torch.cuda.synchronize()
    _close(out, O.rms_norm(x, w, 1e-6, dt), dt)
    tr = T.from_bits(r, dt, "cuda")
    ops.fused_add_rms_norm(tx, tr, tw, 1e-5)
    torch.cuda.synchronize()
    wx, wr = O.fused_add_rms_norm(x, r, w, 1e-5, dt)
    assert np.array_equal(T.bits(tr), wr)  # the residual sum is a single rounded add: exact
    _close(tx, wx, dt)


@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("head_size,rot_dim", [(64, 64), (128, 128), (80, 80), (256, 64)])
@pytest.mark.parametrize("dt", DTYPES)
def test_rotary_embedding_bit_exact(ops, is_neox, head_size, rot_dim, dt):
    rng = np.random.default_rng(head_size + rot_dim)
    num_tokens, hq, hkv, max_pos = 83, 8, 2, 8192
    inv_freq = 1.0 / (10000.0**(np.arange(0, rot_dim, 2, dtype=np.float32) / rot_dim))
    freqs = np.outer(np.arange(max_pos, dtype=np.float32), inv_freq)
    cache = O.to_bits(np.concatenate([np.cos(freqs), np.sin(freqs)], axis=-1).astype(np.float32), dt)
    pos = rng.integers(0, max_pos, num_tokens).astype(np.int64)
    qkv = O.to_bits(rng.standard_normal((num_tokens, (hq + 2 * hkv) * head_size)).astype(np.float32), dt)
    t_qkv = T.from_bits(qkv, dt, "cuda")
    q, k = t_qkv[:, :hq * head_size], t_qkv[:, hq * head_size:(hq + hkv) * head_size]  # strided rows
    ops.rotary_embedding(torch.from_numpy(pos).cuda(), q, k, head_size, T.from_bits(cache, dt, "cuda"),
                         is_neox)
    torch.cuda.synchronize()
    wq, wk = O.rotary_embedding(pos, qkv[:, :hq * head_size], qkv[:, hq * head_size:(hq + hkv) * head_size],
                                head_size, cache, is_neox, dt)
    assert np.array_equal(T.bits(q), wq)
    assert np.array_equal(T.bits(k), wk)
    # v is untouched
    assert np.array_equal(T.bits(t_qkv[:, (hq + hkv) * head_size:]), qkv[:, (hq + hkv) * head_size:])


@pytest.mark.parametrize("num_tokens,d", [(7, 512), (83, 13824), (256, 14336), (3, 8)])
@pytest.mark.parametrize("dt", DTYPES)
def test_silu_and_mul(ops, num_tokens, d, dt):
    rng = np.random.default_rng(d)
    x = O.to_bits(rng.standard_normal((num_tokens, 2 * d)).astype(np.float32), dt)
    tx = T.from_bits(x, dt, "cuda")
    out = torch.empty((num_tokens, d), dtype=tx.dtype, device="cuda")
    ops.silu_and_mul(out, tx)
    torch.cuda.synchronize()
    _close(out, O.silu_and_mul(x, dt), dt)
