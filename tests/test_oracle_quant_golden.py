"""CPU tests: the oracle's GPTQ dequantisation and exllama shuffle against fixtures written by the
reference's own weight recipe (tests/golden/make_golden_quant.py runs quantize_weights, gptq_pack
and sort_weights of vllm/model_executor/layers/quantization/utils/quant_utils.py:39-146).

`w_ref` in a fixture is the reference's dequantised matrix, (q - 8).half() * s; the oracle has to
reproduce it bit for bit from the packed words, which pins the nibble order, the zero - 1
convention of qzeros and the group lookup through g_idx.
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(HERE, "golden", "gptq_*.npz")))


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def test_fixtures_present():
    assert len(FIXTURES) >= 6


@pytest.mark.parametrize("name", FIXTURES)
def test_packed_words_hold_the_reference_nibbles(name):
    z = load(name)
    qw = z["qweight"].view(np.uint32)
    for i in range(8):     # row 8r + i sits in nibble i (quant_utils.py:142-143)
        assert np.array_equal((qw >> (4 * i)) & 15, z["q_w"][i::8])
    assert (z["qzeros"].view(np.uint32) == 0x77777777).all()


@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_dequantize_equals_reference_w_ref(name):
    z = load(name)
    g_idx = z["g_idx"] if z["g_idx"].size else None
    got = O.gptq_dequantize(z["qweight"], z["qzeros"], z["s"], g_idx)
    assert np.array_equal(got, z["w_ref"])


@pytest.mark.parametrize("name", FIXTURES)
def test_gptq_shuffle_then_dequantize_equals_reference_w_ref(name):
    """The exllama route: gptq_shuffle (with the sort permutation when the checkpoint is
    act-ordered, quantization/gptq.py:209-217), then dequantisation of the shuffled words."""
    z = load(name)
    if z["g_idx"].size:
        perm = z["sort_indices"]
        shuffled = O.gptq_shuffle(z["qweight"], perm)
        # the reference's own sorted packing, shuffled without a permutation, is the same tensor
        assert np.array_equal(shuffled, O.gptq_shuffle(z["qweight_sorted"], None))
        got = O.gptq_dequantize(shuffled, z["qzeros"], z["s"], None, shuffled=True)
        assert np.array_equal(got, z["w_ref"][perm])
    else:
        shuffled = O.gptq_shuffle(z["qweight"], None)
        got = O.gptq_dequantize(shuffled, z["qzeros"], z["s"], None, shuffled=True)
        assert np.array_equal(got, z["w_ref"])
