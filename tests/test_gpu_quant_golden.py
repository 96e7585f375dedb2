"""GPU parity of the GPTQ int4 path against fixtures written by the reference's own packer
(tests/golden/make_golden_quant.py: quantize_weights / gptq_pack / sort_weights of
vllm/model_executor/layers/quantization/utils/quant_utils.py:39-146).

`w_ref` is the reference's dequantised matrix.  Dequantisation (plain layout, with g_idx, and
after gptq_shuffle) must reproduce it bit for bit; the GEMM is compared with the oracle's
fp32-accumulate product of the activations with `w_ref` (the reference's large-batch path is
reconstruct + hgemm, csrc/quantization/gptq/q_gemm.cu:1496-1518) at fp16 rounding tolerance.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(HERE, "golden", "gptq_*.npz")))
EMPTY = dict(dtype=torch.int32, device="cuda")


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def f16(a):
    return T.from_bits(a, O.DT_F16, "cuda")


def test_fixtures_present():
    assert len(FIXTURES) >= 6


@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_dequantize_equals_reference_w_ref(ops, name):
    z = load(name)
    g_idx = cuda(z["g_idx"]) if z["g_idx"].size else None
    got = ops.gptq_dequantize(cuda(z["qweight"]), cuda(z["qzeros"]), f16(z["s"]), g_idx)
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(got), z["w_ref"])


@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_shuffle_then_dequantize_equals_reference_w_ref(ops, name):
    z = load(name)
    qw = cuda(z["qweight"].copy())
    if z["g_idx"].size:
        perm = z["sort_indices"]
        ops.gptq_shuffle(qw, cuda(perm), 4)
        want = z["w_ref"][perm]
        # the reference's own sorted packing, shuffled without a permutation, is the same tensor
        qs = cuda(z["qweight_sorted"].copy())
        ops.gptq_shuffle(qs, torch.empty(0, **EMPTY), 4)
        assert torch.equal(qw, qs)
    else:
        ops.gptq_shuffle(qw, torch.empty(0, **EMPTY), 4)
        want = z["w_ref"]
    got = ops.gptq_dequantize(qw, cuda(z["qzeros"]), f16(z["s"]), None, shuffled=True)
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(got), want)


def _close(got, x_bits, w_ref_bits):
    want = O.from_bits(O.gemm_f16(x_bits, w_ref_bits), O.DT_F16)
    # both sides accumulate in fp32 and round once to fp16; only the summation order differs
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("m", [1, 8, 51, 256])
@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_gemm_exllama_against_reference_weights(ops, name, m):
    """gptq_shuffle + gptq_gemm(use_exllama=True) as GPTQLinearMethod.apply issues them
    (quantization/gptq.py:209-221), act-ordered checkpoints with q_perm = argsort(g_idx)."""
    z = load(name)
    k, n = z["w_ref"].shape
    rng = np.random.default_rng(m + k)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    qw = cuda(z["qweight"].copy())
    perm = cuda(z["sort_indices"]) if z["g_idx"].size else torch.empty(0, **EMPTY)
    ops.gptq_shuffle(qw, perm, 4)
    got = ops.gptq_gemm(f16(x), qw, cuda(z["qzeros"]), f16(z["s"]), perm, True, 4)
    torch.cuda.synchronize()
    _close(got, x, z["w_ref"])


@pytest.mark.parametrize("m", [3, 256])
@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_gemm_plain_layout_against_reference_weights(ops, name, m):
    """use_exllama=False: the unshuffled words with g_idx as the row -> group map
    (q_gemm.cu:1386-1420), the path row-parallel act-order layers keep."""
    z = load(name)
    k, n = z["w_ref"].shape
    rng = np.random.default_rng(m + n)
    x = O.to_bits(rng.uniform(-1, 1, (m, k)).astype(np.float32), O.DT_F16)
    g_idx = cuda(z["g_idx"]) if z["g_idx"].size else torch.empty(0, **EMPTY)
    got = ops.gptq_gemm(f16(x), cuda(z["qweight"]), cuda(z["qzeros"]), f16(z["s"]), g_idx, False, 4)
    torch.cuda.synchronize()
    _close(got, x, z["w_ref"])


@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_linear_method_on_reference_checkpoint(ops, name):
    from vllm_legacy_amd.quantization.gptq import GPTQConfig, GPTQLinearMethod
    z = load(name)
    k, n = z["w_ref"].shape
    act = bool(z["g_idx"].size)
    layer = torch.nn.Module()
    meth = GPTQLinearMethod(GPTQConfig(4, int(z["group_size"]), act))
    meth.create_weights(layer, k, [n], k, n, torch.float16, device="cuda")
    assert layer.qzeros.shape == z["qzeros"].shape and layer.scales.shape == z["s"].shape
    layer.qweight.data, layer.qzeros.data, layer.scales.data = (cuda(z["qweight"].copy()),
                                                                cuda(z["qzeros"]), f16(z["s"]))
    if act:
        layer.g_idx.data = cuda(z["g_idx"])
    rng = np.random.default_rng(k)
    x = O.to_bits(rng.uniform(-1, 1, (2, 64, k)).astype(np.float32), O.DT_F16)
    out = meth.apply(layer, f16(x))
    out2 = meth.apply(layer, f16(x))          # second call: already shuffled
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    _close(out.reshape(128, n), x.reshape(128, k), z["w_ref"])
