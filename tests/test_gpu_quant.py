"""GPU parity tests for the int4 weight path (AWQ / GPTQ dequantise, exllama shuffle, dequant-GEMM).

The reference has no kernel-level test for these ops (only Marlin, tests/kernels/test_marlin_gemm.py)
and AWQ has no ROCm build at all, so kernel-level parity is pinned by the oracle's restatement of
the CUDA sources: dequantised weights and the shuffle are compared bit for bit; GEMM outputs within
fp16 accumulation-order tolerance ("parity unpinned" against any reference binary).
Weight recipes follow vllm/model_executor/layers/quantization/utils/quant_utils.py:39-146.
"""
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


def _awq_weights(rng, ic, oc, group):
    qweight = rng.integers(-2**31, 2**31, (ic, oc // 8), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2**31, 2**31, (ic // group, oc // 8), dtype=np.int64).astype(np.int32)
    scales = O.to_bits((rng.uniform(0.5, 1.5, (ic // group, oc)) * 1e-2).astype(np.float32), O.DT_F16)
    return qweight, qzeros, scales


def _gptq_weights(rng, k, n, group):
    qweight = rng.integers(-2**31, 2**31, (k // 8, n), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2**31, 2**31, (k // group, n // 8), dtype=np.int64).astype(np.int32)
    scales = O.to_bits((rng.uniform(0.5, 1.5, (k // group, n)) * 1e-2).astype(np.float32), O.DT_F16)
    return qweight, qzeros, scales


def _cuda(a, dt=None):
    if dt is not None:
        return T.from_bits(a, dt, "cuda")
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("ic,oc,group", [(256, 512, 128), (128, 64, 32), (4096, 6144, 128)])
def test_awq_dequantize_bit_exact(ops, ic, oc, group):
    rng = np.random.default_rng(ic + oc)
    qw, qz, sc = _awq_weights(rng, ic, oc, group)
    got = ops.awq_dequantize(_cuda(qw), _cuda(sc, O.DT_F16), _cuda(qz), 0, 0, 0)
    torch.cuda.synchronize()
    assert got.shape == (ic, oc) and got.dtype == torch.float16
    assert np.array_equal(T.bits(got), O.awq_dequantize(qw, sc, qz))


@pytest.mark.parametrize("k,n,group", [(256, 512, 128), (128, 64, 32), (4096, 1280, 128)])
@pytest.mark.parametrize("act_order", [False, True])
def test_gptq_dequantize_bit_exact(ops, k, n, group, act_order):
    rng = np.random.default_rng(k + n)
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    g_idx = None
    if act_order:
        g_idx = rng.permutation(np.arange(k) // group).astype(np.int32)
    got = ops.gptq_dequantize(_cuda(qw), _cuda(qz), _cuda(sc, O.DT_F16),
                              None if g_idx is None else _cuda(g_idx))
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(got), O.gptq_dequantize(qw, qz, sc, g_idx))


@pytest.mark.parametrize("with_perm", [False, True])
def test_gptq_shuffle_bit_exact_and_dequant_consistent(ops, with_perm):
    rng = np.random.default_rng(7)
    k, n, group = 512, 256, 128
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    perm = rng.permutation(k).astype(np.int32) if with_perm else None
    t_qw = _cuda(qw.copy())
    ops.gptq_shuffle(t_qw, torch.empty(0, dtype=torch.int32, device="cuda") if perm is None
                     else _cuda(perm), 4)
    torch.cuda.synchronize()
    want = O.gptq_shuffle(qw, perm)
    assert np.array_equal(t_qw.cpu().numpy(), want)
    if not with_perm:
        # dequantising the shuffled words gives the same matrix as the plain ones
        a = ops.gptq_dequantize(t_qw, _cuda(qz), _cuda(sc, O.DT_F16), None, shuffled=True)
        torch.cuda.synchronize()
        assert np.array_equal(T.bits(a), O.gptq_dequantize(qw, qz, sc))


def _gemm_close(got: torch.Tensor, want_bits: np.ndarray):
    g = got.float().cpu().numpy()
    w = O.from_bits(want_bits, O.DT_F16)
    # both sides accumulate in fp32 and round once to fp16; only the summation order differs
    np.testing.assert_allclose(g, w, rtol=2e-3, atol=4e-3)


@pytest.mark.parametrize("m", [1, 7, 16, 50, 256, 300])
@pytest.mark.parametrize("ic,oc,group", [(512, 256, 128), (4096, 4096, 128)])
def test_awq_gemm(ops, m, ic, oc, group):
    if ic == 4096 and m not in (1, 256):
        pytest.skip("large shape only at the two batch sizes that matter")
    rng = np.random.default_rng(m + ic)
    qw, qz, sc = _awq_weights(rng, ic, oc, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, ic)).astype(np.float32), O.DT_F16)
    got = ops.awq_gemm(_cuda(x, O.DT_F16), _cuda(qw), _cuda(sc, O.DT_F16), _cuda(qz), 8)
    torch.cuda.synchronize()
    _gemm_close(got, O.gemm_f16(x, O.awq_dequantize(qw, sc, qz)))


@pytest.mark.parametrize("m", [1, 8, 51, 256])
@pytest.mark.parametrize("mode", ["plain", "plain_gidx", "exllama", "exllama_perm"])
def test_gptq_gemm(ops, m, mode):
    rng = np.random.default_rng(m)
    k, n, group = 1024, 512, 128
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    if mode == "plain":
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), _cuda(qw), _cuda(qz), _cuda(sc, O.DT_F16), empty,
                            False, 4)
        want = O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc))
    elif mode == "plain_gidx":
        g_idx = rng.permutation(np.arange(k) // group).astype(np.int32)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), _cuda(qw), _cuda(qz), _cuda(sc, O.DT_F16),
                            _cuda(g_idx), False, 4)
        want = O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc, g_idx))
    elif mode == "exllama":
        t_qw = _cuda(qw.copy())
        ops.gptq_shuffle(t_qw, empty, 4)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), t_qw, _cuda(qz), _cuda(sc, O.DT_F16), empty, True, 4)
        want = O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc))
    else:
        # act-order through exllama: rows sorted by group (q_perm = argsort(g_idx),
        # quantization/gptq.py:209-217), activations gathered by the same permutation
        g_idx = rng.permutation(np.arange(k) // group).astype(np.int32)
        perm = np.argsort(g_idx, kind="stable").astype(np.int32)
        t_qw = _cuda(qw.copy())
        ops.gptq_shuffle(t_qw, _cuda(perm), 4)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), t_qw, _cuda(qz), _cuda(sc, O.DT_F16), _cuda(perm),
                            True, 4)
        want = O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc, g_idx))
    torch.cuda.synchronize()
    _gemm_close(got, want)


def test_quant_shape_errors(ops):
    z = torch.zeros
    with pytest.raises(RuntimeError, match="Unsupported quantized weight shape"):
        ops.awq_gemm(z((4, 128), dtype=torch.float16, device="cuda"),
                     z((128, 4), dtype=torch.int32, device="cuda"),       # OC = 32: not % 64
                     z((1, 32), dtype=torch.float16, device="cuda"),
                     z((1, 4), dtype=torch.int32, device="cuda"), 8)
    with pytest.raises(RuntimeError, match="Unsupported quantized weight shape"):
        ops.gptq_gemm(z((4, 128), dtype=torch.float16, device="cuda"),
                      z((16, 64), dtype=torch.int32, device="cuda"),
                      z((1, 8), dtype=torch.int32, device="cuda"),
                      z((1, 64), dtype=torch.float16, device="cuda"),
                      torch.empty(0, dtype=torch.int32, device="cuda"), True, 8)  # 8-bit


def _nib(a, pos):
    return (a.view(np.uint32) >> (4 * pos)) & 15


def test_awq_repack_layout(ops):
    rng = np.random.default_rng(11)
    ic, oc, group = 256, 384, 128
    qw, qz, sc = _awq_weights(rng, ic, oc, group)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    torch.cuda.synchronize()
    pw, pz = pw.cpu().numpy(), pz.cpu().numpy()
    order = [0, 4, 1, 5, 2, 6, 3, 7]                      # AWQ nibble of column 8j + i
    pairpos = [0, 4, 1, 5, 2, 6, 3, 7]                    # nibble position of row 8r + e
    for i in range(8):
        assert np.array_equal(_nib(pz, i), _nib(qz, order[i]))
        for e in range(8):
            assert np.array_equal(_nib(pw[:, i::8], pairpos[e]), _nib(qw[e::8, :], order[i]))


@pytest.mark.parametrize("m", [1, 17, 64, 200, 256, 300, 513])
@pytest.mark.parametrize("ic,oc,group", [(512, 256, 128), (1024, 320, 64), (4096, 6144, 128),
                                         (14336, 4096, 128), (4096, 28672, 128)])
def test_awq_packed_gemm(ops, m, ic, oc, group):
    if ic >= 4096 and m not in (1, 256, 300):
        pytest.skip("large shapes only at a few batch sizes")
    if oc == 28672 and m != 256:
        pytest.skip("gate_up of Llama-3-8B (the largest GEMM of the decode step) at its batch")
    rng = np.random.default_rng(m + ic)
    qw, qz, sc = _awq_weights(rng, ic, oc, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, ic)).astype(np.float32), O.DT_F16)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
    torch.cuda.synchronize()
    w = O.awq_dequantize(qw, sc, qz)
    if oc > 8192:  # the CPU product of the full matrix takes minutes: every 13th column
        cols = np.arange(0, oc, 13)
        w, got = np.ascontiguousarray(w[:, cols]), got[:, torch.from_numpy(cols).cuda()]
    want = O.gemm_f16(x, w)
    if ic > 4096:  # longer sums: fp16 output of magnitude ~3, allow two ulps
        np.testing.assert_allclose(got.float().cpu().numpy(), O.from_bits(want, O.DT_F16),
                                   rtol=3e-3, atol=8e-3)
    else:
        _gemm_close(got, want)


def test_awq_linear_method_uses_packed_path(ops):
    from vllm_legacy_amd.quantization.awq import AWQConfig, AWQLinearMethod
    rng = np.random.default_rng(5)
    ic, oc, group, m = 512, 256, 128, 33
    qw, qz, sc = _awq_weights(rng, ic, oc, group)
    layer = torch.nn.Module()
    meth = AWQLinearMethod(AWQConfig(4, group))
    meth.create_weights(layer, ic, [oc], ic, oc, torch.float16, device="cuda")
    layer.qweight.data, layer.qzeros.data, layer.scales.data = _cuda(qw), _cuda(qz), _cuda(sc, O.DT_F16)
    x = O.to_bits(rng.uniform(-1, 1, (3, 11, ic)).astype(np.float32), O.DT_F16)
    out = meth.apply(layer, _cuda(x, O.DT_F16))
    torch.cuda.synchronize()
    assert out.shape == (3, 11, oc) and hasattr(layer, "_k_packed")
    _gemm_close(out.reshape(m, oc), O.gemm_f16(x.reshape(m, ic), O.awq_dequantize(qw, sc, qz)))


# Per-GPU layer shapes of the other target configurations: Llama-3-70B GPTQ at TP=8
# (qkv 8192->1280, o 1024->8192, gate_up 8192->7168, down 3584->8192) through the GPTQ linear
# method's exllama path, and Mixtral-8x7B AWQ experts at TP=2 (w1/w3 4096->7168, w2 7168->4096).
@pytest.mark.parametrize("m", [48, 256])
@pytest.mark.parametrize("k,n", [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)])
def test_gptq_linear_method_llama70b_tp8_shapes(ops, k, n, m):
    from vllm_legacy_amd.quantization.gptq import GPTQConfig, GPTQLinearMethod
    rng = np.random.default_rng(k + n)
    group = 128
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    layer = torch.nn.Module()
    meth = GPTQLinearMethod(GPTQConfig(4, group, False))
    meth.create_weights(layer, k, [n], k, n, torch.float16, device="cuda")
    layer.qweight.data, layer.qzeros.data, layer.scales.data = _cuda(qw.copy()), _cuda(qz), _cuda(sc, O.DT_F16)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    out = meth.apply(layer, _cuda(x, O.DT_F16))
    torch.cuda.synchronize()
    want = O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc))
    np.testing.assert_allclose(out.float().cpu().numpy(), O.from_bits(want, O.DT_F16), rtol=3e-3,
                               atol=8e-3)


# The unsliced 128-row launches (128 x 64 wave tiles, two K parts per workgroup) at their edges:
# a last column tile that is partly outside the matrix, a last row tile of 44 rows, 64-deep
# groups, and the GPTQ zero convention (stored zero - 1).
# ... and a wide layer at 33..64 rows (64-row tiles: two K slices of two K parts).
@pytest.mark.parametrize("kind,m,k,n,group", [("awq", 300, 4096, 16456, 64), ("awq", 129, 2048, 28680, 128),
                                              ("gptq", 256, 4096, 28672, 128), ("awq", 48, 4096, 28672, 128),
                                              ("gptq", 64, 4096, 24584, 128)])
def test_unsliced_wide_layers_at_tile_edges(ops, kind, m, k, n, group):
    rng = np.random.default_rng(m + n)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    if kind == "awq":
        qw, qz, sc = _awq_weights(rng, k, n, group)
        pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
        got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        w = O.awq_dequantize(qw, sc, qz)
    else:
        qw, qz, sc = _gptq_weights(rng, k, n, group)
        gq = _cuda(qw.copy())
        ops.gptq_shuffle(gq, torch.empty(0, dtype=torch.int32, device="cuda"), 4)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16),
                            torch.empty(0, dtype=torch.int32, device="cuda"), True, 4)
        w = O.gptq_dequantize(qw, qz, sc)
    torch.cuda.synchronize()
    assert got.shape == (m, n)
    cols = np.unique(np.concatenate([np.arange(0, n, 13), np.arange(n - 40, n)]))   # incl. the last tile
    want = O.gemm_f16(x, np.ascontiguousarray(w[:, cols]))
    np.testing.assert_allclose(got[:, torch.from_numpy(cols).cuda()].float().cpu().numpy(),
                               O.from_bits(want, O.DT_F16), rtol=3e-3, atol=8e-3)


@pytest.mark.parametrize("k,n", [(4096, 7168), (7168, 4096)])
def test_awq_packed_gemm_mixtral_tp2_expert_shapes(ops, k, n):
    rng = np.random.default_rng(k)
    m, group = 40, 128
    qw, qz, sc = _awq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
    torch.cuda.synchronize()
    want = O.gemm_f16(x, O.awq_dequantize(qw, sc, qz))
    np.testing.assert_allclose(got.float().cpu().numpy(), O.from_bits(want, O.DT_F16), rtol=3e-3,
                               atol=8e-3)


@pytest.mark.parametrize("ic,oc", [(512, 256), (1024, 4096), (2048, 320)])
@pytest.mark.parametrize("counts,tail", [([5, 0, 70, 1], 0), ([64, 64, 64, 64], 0), ([0, 0, 130, 3], 9),
                                         ([1, 1, 1, 1], 4)])
def test_grouped_gemm_matches_per_set_products(ops, ic, oc, counts, tail):
    """Several weight sets in one launch (an MoE block's experts): every row block equals the
    oracle's product with its own set; empty sets are fine; rows behind the last set (tokens
    routed to another rank's experts) come back zero."""
    rng = np.random.default_rng(ic + sum(counts))
    group, sets = 128, len(counts)
    ws = [_awq_weights(rng, ic, oc, group) for _ in range(sets)]
    packed = [ops.awq_repack(_cuda(qw), _cuda(qz)) for qw, qz, _ in ws]
    pw = torch.stack([p[0] for p in packed]).contiguous()
    pz = torch.stack([p[1] for p in packed]).contiguous()
    sc = torch.stack([_cuda(w[2], O.DT_F16) for w in ws]).contiguous()
    rows = sum(counts) + tail
    x = O.to_bits(rng.uniform(-1, 1, (rows, ic)).astype(np.float32), O.DT_F16)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device="cuda")
    got = ops.q4_gemm_grouped(_cuda(x, O.DT_F16), pw, pz, sc, offsets, 0, rows)
    torch.cuda.synchronize()
    got = got.float().cpu().numpy()
    start = 0
    for (qw, qz, s), c in zip(ws, counts):
        if c:
            want = O.gemm_f16(x[start:start + c], O.awq_dequantize(qw, s, qz))
            np.testing.assert_allclose(got[start:start + c], O.from_bits(want, O.DT_F16), rtol=3e-3, atol=4e-3)
        start += c
    assert not got[start:].any()


@pytest.mark.parametrize("desc_act", [False, True])
@pytest.mark.parametrize("m,k,n", [(256, 8192, 1280), (5, 1024, 8192), (9, 512, 16400)])
def test_gptq_linear_method_partial_and_fused_forms_equal_apply(ops, desc_act, m, k, n):
    """GPTQLinearMethod.apply_partial (unreduced split-K slabs for a fused consumer) and
    apply_silu_and_mul (the activation in the product's epilogue) give the bits of apply()
    followed by the reduction / by SiluAndMul."""
    from vllm_legacy_amd.quantization.gptq import GPTQConfig, GPTQLinearMethod
    rng = np.random.default_rng(m + k + n)
    group = 128
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    x = _cuda(O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16), O.DT_F16)

    def make():
        layer = torch.nn.Module()
        meth = GPTQLinearMethod(GPTQConfig(4, group, desc_act))
        meth.create_weights(layer, k, [n], k, n, torch.float16, device="cuda")
        layer.qweight.data, layer.qzeros.data = _cuda(qw.copy()), _cuda(qz)
        layer.scales.data = _cuda(sc, O.DT_F16) * 4
        if desc_act:
            perm = np.random.default_rng(7).permutation(k)
            layer.g_idx.data = torch.from_numpy((perm // group).astype(np.int32)).cuda()
        return layer, meth
    layer, meth = make()
    want = meth.apply(layer, x)
    layer2, meth2 = make()
    part = meth2.apply_partial(layer2, x)
    got = part.reduce() if isinstance(part, ops.SplitKSlabs) else part
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    act = meth2.apply_silu_and_mul(layer2, x)
    if m <= 16 and n == 16400:
        assert act is not None
    if act is not None:
        two = torch.empty((m, n // 2), dtype=torch.float16, device="cuda")
        ops.silu_and_mul(two, want)
        torch.cuda.synchronize()
        assert torch.equal(act.view(torch.int16), two.view(torch.int16))


# The K-sliced 128-row launches (128 x 32 wave tiles, four column waves x two K parts per
# workgroup, fp32 slabs summed by the reduction launch) at their edges: a last column tile that is
# partly outside the matrix, a last row tile of a few rows, 64-deep groups, a contraction whose
# slices hold an odd number of groups per part, and the GPTQ zero convention (stored zero - 1).
@pytest.mark.parametrize("kind,m,k,n,group", [("awq", 200, 4096, 4136, 128), ("awq", 129, 4096, 6152, 64),
                                              ("awq", 256, 14336, 4096, 128), ("awq", 70, 5120, 1032, 128),
                                              ("gptq", 256, 4096, 6144, 128), ("gptq", 131, 3584, 8200, 128)])
def test_k_part_launches_at_tile_edges(ops, kind, m, k, n, group):
    rng = np.random.default_rng(m + n + k)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    if kind == "awq":
        qw, qz, sc = _awq_weights(rng, k, n, group)
        pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
        got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        part = ops.q4_gemm_packed_partial(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        w = O.awq_dequantize(qw, sc, qz)
    else:
        qw, qz, sc = _gptq_weights(rng, k, n, group)
        gq = _cuda(qw.copy())
        ops.gptq_shuffle(gq, torch.empty(0, dtype=torch.int32, device="cuda"), 4)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16),
                            torch.empty(0, dtype=torch.int32, device="cuda"), True, 4)
        part = ops.q4_gemm_packed_partial(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16), 1)
        w = O.gptq_dequantize(qw, qz, sc)
    torch.cuda.synchronize()
    # the slabs a fused consumer would sum give the same bits as the library's own reduction
    if isinstance(part, ops.SplitKSlabs):
        assert torch.equal(part.reduce(), got)
    cols = np.unique(np.concatenate([np.arange(0, n, 7), np.arange(max(0, n - 40), n)]))
    want = O.gemm_f16(x, np.ascontiguousarray(w[:, cols]))
    np.testing.assert_allclose(got[:, torch.from_numpy(cols).cuda()].float().cpu().numpy(),
                               O.from_bits(want, O.DT_F16), rtol=3e-3, atol=8e-3)


# Row blocks of 17..80 rows (16-row MFMA blocks against weights rebuilt once, x shared through LDS
# by the four waves of a workgroup; K slices for narrow layers): every block count, row counts
# that end inside a block, a last 256-column workgroup that is partly or wholly outside the
# matrix for some of its waves, a contraction of a single 128-deep chunk, groups of two chunks,
# and the GPTQ zero convention.
@pytest.mark.parametrize("kind,m,k,n,group", [
    ("awq", 17, 512, 256, 128), ("awq", 32, 128, 264, 128), ("awq", 33, 1024, 320, 128),
    ("awq", 48, 4096, 4136, 128), ("awq", 64, 4096, 6144, 128), ("awq", 65, 2048, 840, 256),
    ("awq", 80, 14336, 4096, 128), ("awq", 79, 4096, 28672, 128), ("awq", 75, 1536, 8, 128),
    ("gptq", 40, 3584, 8200, 128), ("gptq", 81, 8192, 1280, 128)])
def test_mid_row_block_launches(ops, kind, m, k, n, group):
    ops.set_q4_row_block_kernel(1)         # dense products take the tile kernels by default
    try:
        _mid_row_block_case(ops, kind, m, k, n, group)
    finally:
        ops.set_q4_row_block_kernel(0)


def _mid_row_block_case(ops, kind, m, k, n, group):
    rng = np.random.default_rng(m + n + k)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    if kind == "awq":
        qw, qz, sc = _awq_weights(rng, k, n, group)
        pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
        got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        part = ops.q4_gemm_packed_partial(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        w = O.awq_dequantize(qw, sc, qz)
    else:
        qw, qz, sc = _gptq_weights(rng, k, n, group)
        gq = _cuda(qw.copy())
        ops.gptq_shuffle(gq, torch.empty(0, dtype=torch.int32, device="cuda"), 4)
        got = ops.gptq_gemm(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16),
                            torch.empty(0, dtype=torch.int32, device="cuda"), True, 4)
        part = ops.q4_gemm_packed_partial(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16), 1)
        w = O.gptq_dequantize(qw, qz, sc)
    torch.cuda.synchronize()
    assert got.shape == (m, n)
    if isinstance(part, ops.SplitKSlabs):
        assert torch.equal(part.reduce(), got)
    cols = np.unique(np.concatenate([np.arange(0, n, 7), np.arange(max(0, n - 40), n)]))
    want = O.gemm_f16(x, np.ascontiguousarray(w[:, cols]))
    np.testing.assert_allclose(got[:, torch.from_numpy(cols).cuda()].float().cpu().numpy(),
                               O.from_bits(want, O.DT_F16), rtol=3e-3, atol=8e-3)


@pytest.mark.parametrize("counts,tail,gather", [([70, 20, 80, 0, 81, 1, 16, 33], 0, False),
                                                ([64] * 8, 5, False), ([31, 90, 2, 77, 48, 65, 80, 17], 3, True)])
def test_grouped_mid_row_blocks(ops, counts, tail, gather):
    """The grouped form of the 17..80-row kernel (an MoE block's experts at a few dozen rows each):
    sets with no rows, with exactly one 80-row tile, with one row more than a tile (a second pass
    over the set's weights), and the row gather."""
    rng = np.random.default_rng(sum(counts) + tail)
    ic, oc, group, sets = 512, 8192, 128, len(counts)
    ws = [_awq_weights(rng, ic, oc, group) for _ in range(sets)]
    packed = [ops.awq_repack(_cuda(qw), _cuda(qz)) for qw, qz, _ in ws]
    pw = torch.stack([p[0] for p in packed]).contiguous()
    pz = torch.stack([p[1] for p in packed]).contiguous()
    sc = torch.stack([_cuda(w[2], O.DT_F16) for w in ws]).contiguous()
    rows = sum(counts) + tail
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device="cuda")
    if gather:
        src = O.to_bits(rng.uniform(-1, 1, (200, ic)).astype(np.float32), O.DT_F16)
        idx = rng.integers(0, 200, rows).astype(np.int32)
        x = src[idx]
        got = ops.q4_gemm_grouped(_cuda(src, O.DT_F16), pw, pz, sc, offsets, 0, max(counts),
                                  x_rows=torch.from_numpy(idx).cuda())
    else:
        x = O.to_bits(rng.uniform(-1, 1, (rows, ic)).astype(np.float32), O.DT_F16)
        got = ops.q4_gemm_grouped(_cuda(x, O.DT_F16), pw, pz, sc, offsets, 0, max(counts))
    torch.cuda.synchronize()
    got = got.float().cpu().numpy()
    cols = np.arange(0, oc, 5)
    start = 0
    for (qw, qz, s), c in zip(ws, counts):
        if c:
            w = np.ascontiguousarray(O.awq_dequantize(qw, s, qz)[:, cols])
            want = O.gemm_f16(x[start:start + c], w)
            np.testing.assert_allclose(got[start:start + c][:, cols], O.from_bits(want, O.DT_F16),
                                       rtol=3e-3, atol=4e-3)
        start += c
    if not gather:
        assert not got[start:].any()
