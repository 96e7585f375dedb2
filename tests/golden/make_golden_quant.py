"""Generate tests/golden/gptq_*.npz with the REFERENCE's own GPTQ weight recipe.

Runs only in the build container (needs /root/reference): it loads
vllm/model_executor/layers/quantization/utils/quant_utils.py from the reference tree by file
path (the module needs nothing but numpy and torch) and calls its `quantize_weights`
(:39-106, which also yields the dequantised matrix `w_ref = (q - 8).half() * s`), `gptq_pack`
(:125-146, the [K/8, N] int32 wire format: row 8r+i in nibble i) and `sort_weights` (:109-122).
The .npz files hold data only -- inputs and the reference's outputs -- and travel to the GPU box;
the reference does not.

What the fixtures pin: the nibble order of `qweight`, the `zero - 1` storage convention of
`qzeros` (a symmetric 4-bit zero point of 8 is stored as 7, so every word is 0x77777777; the
kernels add the 1 back, csrc/quantization/gptq/q_gemm.cu:1410-1418), group size 128 and -1, and
activation reordering (`g_idx`, with the sort permutation the exllama path shuffles by,
vllm/model_executor/layers/quantization/gptq.py:209-217).

K is the full contraction length of the layer named in each case (Llama-3-8B and the Llama-3-70B
shard of an 8-way tensor-parallel rank); N is a slice of the layer's columns so that the files
stay small (every column is quantised on its own).

    python tests/golden/make_golden_quant.py
"""
import importlib.util
import os

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
REF_QUANT_UTILS = ("/root/reference/vllm/model_executor/layers/quantization/utils/"
                   "quant_utils.py")

# name, K, N, group_size, act_order, seed
CASES = [
    ("gptq_llama8b_k4096_g128", 4096, 128, 128, False, 11),       # qkv / o / gate_up of Llama-3-8B
    ("gptq_llama8b_k14336_g128", 14336, 64, 128, False, 12),      # down_proj of Llama-3-8B
    ("gptq_llama8b_k4096_gfull", 4096, 64, -1, False, 13),        # one group per column
    ("gptq_llama70b_tp8_k8192_g128_act", 8192, 64, 128, True, 14),  # qkv / gate_up shard, act-order
    ("gptq_llama70b_tp8_k1024_g128", 1024, 128, 128, False, 15),  # o_proj shard
    ("gptq_llama70b_tp8_k3584_g128_act", 3584, 64, 128, True, 16),  # down_proj shard, act-order
]


def load_quant_utils():
    spec = importlib.util.spec_from_file_location("ref_quant_utils", REF_QUANT_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def f16_bits(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.float16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    qu = load_quant_utils()
    for name, k, n, group, act_order, seed in CASES:
        torch.manual_seed(seed)
        w = (torch.randn(k, n) * 0.02).half()
        w_ref, q_w, s, g_idx, rand_perm = qu.quantize_weights(w, 4, group, act_order)
        assert w_ref.dtype == torch.float16 and s.dtype == torch.float16
        qweight = qu.gptq_pack(q_w, 4, k, n)
        groups = s.shape[0]
        qzeros = np.full((groups, n // 8), 0x77777777, dtype=np.int32)
        rec = dict(w_ref=f16_bits(w_ref), q_w=q_w.numpy().astype(np.uint8),
                   s=f16_bits(s), qweight=qweight.numpy(), qzeros=qzeros,
                   g_idx=g_idx.numpy().astype(np.int32),
                   rand_perm=rand_perm.numpy().astype(np.int32), group_size=group)
        if act_order:
            q_sorted, g_sorted, sort_idx = qu.sort_weights(q_w, g_idx)
            rec.update(sort_indices=sort_idx.numpy().astype(np.int32),
                       qweight_sorted=qu.gptq_pack(q_sorted, 4, k, n).numpy(),
                       g_idx_sorted=g_sorted.numpy().astype(np.int32))
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **rec)
        print(name, "K", k, "N", n, "groups", groups, "act_order", act_order,
              os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
