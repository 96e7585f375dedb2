"""GPU parity tests for the int4 dequant-GEMM at decode batches of at most 16 rows
(q4_gemm_skinny.hip: 16x16x32 MFMAs with the weights as the A operand, x straight from global
memory, K parts of a workgroup summed through LDS, K slices across workgroups as fp32 slabs).

As for the larger batches the reference holds no fixture for the GEMM itself: outputs are compared
with the oracle's fp32-accumulate product of the bit-exact dequantised weights, and with the
library's own 32-row kernel (VLA_Q4_SKINNY=0 is read when the library first decides, so the
comparison runs on the launch-shape override instead).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T
from tests.test_gpu_quant import _awq_weights, _cuda, _gptq_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def _close(got, want_bits, long_sum):
    want = O.from_bits(want_bits, O.DT_F16)
    rtol, atol = (3e-3, 8e-3) if long_sum else (2e-3, 2e-3)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=rtol, atol=atol)


# group 128 takes chunks of four K steps, groups of 32 / 64 / 96 chunks of one; 328 and 8 columns
# leave a last strip that is mostly outside the matrix; 14336 / 128 = 112 chunks do not divide
# evenly among 8 or 16 waves
@pytest.mark.parametrize("m", [1, 2, 5, 16])
@pytest.mark.parametrize("k,n,group", [(512, 256, 128), (1024, 328, 64), (384, 8, 96), (128, 64, 32),
                                       (4096, 4096, 128), (14336, 4096, 128), (4096, 6144, 128),
                                       (2048, 1024, 2048)])
def test_awq_packed_gemm_small_batches(ops, m, k, n, group):
    if k * n > 4096 * 4096 and m not in (1, 16):
        pytest.skip("largest shapes at two batch sizes")
    rng = np.random.default_rng(m + k + n)
    qw, qz, sc = _awq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
    torch.cuda.synchronize()
    assert got.shape == (m, n)
    _close(got, O.gemm_f16(x, O.awq_dequantize(qw, sc, qz)), k > 4096)


@pytest.mark.parametrize("m", [1, 9, 16])
@pytest.mark.parametrize("k,n,group", [(1024, 8192, 128), (8192, 1280, 128), (512, 320, 64)])
def test_gptq_exllama_gemm_small_batches(ops, m, k, n, group):
    """GPTQ stores zero - 1 (q_gemm.cu:1413): the other zero offset of the same kernel."""
    rng = np.random.default_rng(m + k)
    qw, qz, sc = _gptq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    gq = _cuda(qw.copy())
    none = torch.empty(0, dtype=torch.int32, device="cuda")
    ops.gptq_shuffle(gq, none, 4)
    got = ops.gptq_gemm(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16), none, True, 4)
    torch.cuda.synchronize()
    _close(got, O.gemm_f16(x, O.gptq_dequantize(qw, qz, sc)), k > 4096)


@pytest.mark.parametrize("tune", ["4,1", "8,1", "16,1", "4,4", "8,2", "16,8", "8,7"])
def test_every_launch_shape_gives_the_same_product(ops, tune):
    """Waves per workgroup and K slices change the order of the fp32 sums only."""
    rng = np.random.default_rng(3)
    m, k, n, group = 7, 3584, 1096, 128
    qw, qz, sc = _awq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    old = os.environ.get("VLA_Q4_SKINNY_TUNE")
    os.environ["VLA_Q4_SKINNY_TUNE"] = tune
    try:
        got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["VLA_Q4_SKINNY_TUNE"]
        else:
            os.environ["VLA_Q4_SKINNY_TUNE"] = old
    _close(got, O.gemm_f16(x, O.awq_dequantize(qw, sc, qz)), False)


def test_small_batch_product_from_a_graph_and_with_padding_rows_untouched(ops):
    """Rows past the batch are never written (the kernel computes them from a repeated row and
    drops them), and the launch replays from a graph."""
    rng = np.random.default_rng(8)
    m, k, n, group = 3, 4096, 6144, 128
    qw, qz, sc = _awq_weights(rng, k, n, group)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    xd, scd = _cuda(x, O.DT_F16), _cuda(sc, O.DT_F16)
    eager = ops.q4_gemm_packed(xd, pw, pz, scd, 0)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        ops.q4_gemm_packed(xd, pw, pz, scd, 0)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            out = ops.q4_gemm_packed(xd, pw, pz, scd, 0)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, eager)
    _close(out, O.gemm_f16(x, O.awq_dequantize(qw, sc, qz)), False)


# ---------------------------------------------------------------- grouped (MoE experts at decode)
# num_sets x ceil(oc / 64) must exceed 256 for the grouped launch to take this kernel (enough
# column strips to fill the chip without K slices); 8 sets x 2112 columns = 264 strips.
@pytest.mark.parametrize("counts,tail,gather", [([5, 0, 16, 1, 0, 3, 2, 7], 0, False),
                                                ([1, 1, 1, 1, 1, 1, 1, 1], 3, False),
                                                ([0, 0, 0, 9, 0, 0, 0, 0], 0, True),
                                                ([16, 16, 2, 0, 1, 4, 8, 3], 5, True),
                                                # more than 16 rows in a set: a second and a third
                                                # 16-row tile of the same launch
                                                ([20, 0, 33, 1, 5, 0, 48, 2], 0, False),
                                                ([17, 3, 0, 32, 1, 1, 2, 40], 2, True)])
@pytest.mark.parametrize("ic,oc,group", [(512, 2112, 128), (1024, 2176, 64)])
def test_grouped_product_of_sets_of_a_few_rows(ops, counts, tail, gather, ic, oc, group):
    """Several weight sets in one launch where the sets hold a few rows each: every row block
    equals the oracle's product with its own set, empty sets are fine, rows behind the last set
    come back zero (or stay untouched when the rows are gathered through x_rows)."""
    rng = np.random.default_rng(ic + sum(counts) + tail)
    sets = len(counts)
    ws = [_awq_weights(rng, ic, oc, group) for _ in range(sets)]
    packed = [ops.awq_repack(_cuda(qw), _cuda(qz)) for qw, qz, _ in ws]
    pw = torch.stack([p[0] for p in packed]).contiguous()
    pz = torch.stack([p[1] for p in packed]).contiguous()
    sc = torch.stack([_cuda(w[2], O.DT_F16) for w in ws]).contiguous()
    rows = sum(counts) + tail
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device="cuda")
    if gather:
        tokens = 11
        src = O.to_bits(rng.uniform(-1, 1, (tokens, ic)).astype(np.float32), O.DT_F16)
        x_rows = rng.integers(0, tokens, rows).astype(np.int32)
        x = src[x_rows]
        got = ops.q4_gemm_grouped(_cuda(src, O.DT_F16), pw, pz, sc, offsets, 0, max(counts),
                                  x_rows=torch.from_numpy(x_rows).cuda())
    else:
        x = O.to_bits(rng.uniform(-1, 1, (rows, ic)).astype(np.float32), O.DT_F16)
        got = ops.q4_gemm_grouped(_cuda(x, O.DT_F16), pw, pz, sc, offsets, 0, max(counts))
    torch.cuda.synchronize()
    assert got.shape == (rows, oc)
    got = got.float().cpu().numpy()
    start = 0
    for (qw, qz, s), c in zip(ws, counts):
        if c:
            want = O.gemm_f16(x[start:start + c], O.awq_dequantize(qw, s, qz))
            np.testing.assert_allclose(got[start:start + c], O.from_bits(want, O.DT_F16), rtol=3e-3,
                                       atol=4e-3)
        start += c
    if not gather:
        assert not got[start:].any()


def test_grouped_small_sets_equal_the_tiled_grouped_launch(ops):
    """The same grouped product with a loose row bound (which keeps it on the LDS-staged tiles)
    and with the tight one (small-batch kernel): equal within the GEMM tolerance."""
    rng = np.random.default_rng(77)
    counts, ic, oc, group = [3, 0, 12, 1, 6, 0, 16, 2], 1024, 2112, 128
    ws = [_awq_weights(rng, ic, oc, group) for _ in counts]
    packed = [ops.awq_repack(_cuda(qw), _cuda(qz)) for qw, qz, _ in ws]
    pw = torch.stack([p[0] for p in packed]).contiguous()
    pz = torch.stack([p[1] for p in packed]).contiguous()
    sc = torch.stack([_cuda(w[2], O.DT_F16) for w in ws]).contiguous()
    rows = sum(counts)
    x = _cuda(O.to_bits(rng.uniform(-1, 1, (rows, ic)).astype(np.float32), O.DT_F16), O.DT_F16)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device="cuda")
    tiled = ops.q4_gemm_grouped(x, pw, pz, sc, offsets, 0, rows)
    small = ops.q4_gemm_grouped(x, pw, pz, sc, offsets, 0, 16)
    torch.cuda.synchronize()
    np.testing.assert_allclose(small.float().cpu().numpy(), tiled.float().cpu().numpy(), rtol=3e-3,
                               atol=4e-3)


def test_random_small_shapes(ops):
    """Twenty random (batch, K, N, group) draws within the kernel's contract -- N a multiple of 8,
    K a multiple of 64 and of the group, groups of 32 / 64 / 128 / K -- against the oracle, AWQ and
    GPTQ zero conventions alternating."""
    rng = np.random.default_rng(2024)
    none = torch.empty(0, dtype=torch.int32, device="cuda")
    for case in range(20):
        m = int(rng.integers(1, 17))
        k = 64 * int(rng.integers(1, 33))
        n = 8 * int(rng.integers(1, 76))
        groups = [g for g in (32, 64, 128, k) if k % g == 0]
        group = int(groups[int(rng.integers(0, len(groups)))])
        x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
        if case % 2 == 0:
            qw, qz, sc = _awq_weights(rng, k, n, group)
            pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
            got = ops.q4_gemm_packed(_cuda(x, O.DT_F16), pw, pz, _cuda(sc, O.DT_F16), 0)
            w = O.awq_dequantize(qw, sc, qz)
        else:
            qw, qz, sc = _gptq_weights(rng, k, n, group)
            gq = _cuda(qw.copy())
            ops.gptq_shuffle(gq, none, 4)
            got = ops.gptq_gemm(_cuda(x, O.DT_F16), gq, _cuda(qz), _cuda(sc, O.DT_F16), none, True, 4)
            w = O.gptq_dequantize(qw, qz, sc)
        torch.cuda.synchronize()
        assert got.shape == (m, n), (case, m, k, n, group)
        want = O.from_bits(O.gemm_f16(x, w), O.DT_F16)
        np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=2e-3, atol=2e-3,
                                   err_msg=f"case {case}: m={m} k={k} n={n} group={group}")


# ---------------------------------------------------------------- product + SiluAndMul in one launch
@pytest.mark.parametrize("m", [1, 7, 16])
@pytest.mark.parametrize("k,n,group,zo", [(4096, 28672, 128, 0), (512, 16400, 128, 0),
                                          (256, 16448, 64, 1), (1024, 16384 + 64, 32, 1)])
def test_gate_up_product_with_silu_and_mul_in_its_epilogue(ops, m, k, n, group, zo):
    """out = silu(gate) * up from the product's epilogue equals the product followed by
    silu_and_mul bit for bit (halves of 8200 / 8224 / 8224 columns end inside a workgroup's 32)."""
    rng = np.random.default_rng(m + k + n)
    qw, qz, sc = _awq_weights(rng, k, n, group)          # k-packing is the same for both kinds
    pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
    x = _cuda(O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16), O.DT_F16)
    scd = _cuda(sc, O.DT_F16) * 4                          # gate values of a few units
    fused = ops.q4_gemm_packed_silu_and_mul(x, pw, pz, scd, zo)
    assert fused is not None and fused.shape == (m, n // 2)
    prod = ops.q4_gemm_packed(x, pw, pz, scd, zo)
    two = torch.empty((m, n // 2), dtype=torch.float16, device="cuda")
    ops.silu_and_mul(two, prod)
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.int16), two.view(torch.int16))
    assert fused.float().abs().max().item() > 0.05        # not a product of zeros


def test_gate_up_with_silu_is_refused_for_shapes_the_kernel_does_not_take(ops):
    rng = np.random.default_rng(1)
    for m, k, n in ((1, 512, 256), (17, 512, 16400), (4, 4096, 6144)):
        qw, qz, sc = _awq_weights(rng, k, n, 128)
        pw, pz = ops.awq_repack(_cuda(qw), _cuda(qz))
        x = _cuda(O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16), O.DT_F16)
        assert ops.q4_gemm_packed_silu_and_mul(x, pw, pz, _cuda(sc, O.DT_F16), 0) is None
