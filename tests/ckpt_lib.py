"""Test infrastructure: Hugging Face-named AWQ / GPTQ checkpoints of a small Llama stack, written as
safetensors files, together with the dense fp16 matrices they must dequantise to.

GPTQ tensors are cut out of tests/golden/gptq_*.npz -- produced by the reference's own packer
(quantize_weights / gptq_pack of vllm/model_executor/layers/quantization/utils/quant_utils.py:39-146,
script tests/golden/make_golden_quant.py) -- so the expected dense matrix is the reference's `w_ref`,
not an oracle result: GPTQ quantises every column on its own and every 128 rows as one group, so any
group-aligned row range of any set of fixture columns is again a valid GPTQ tensor with the matching
rows of `w_ref`; a column whose scales are doubled (exact in fp16) has a doubled `w_ref` column,
which makes repeated columns distinct.  AWQ has no packer in the reference: random words, expected
matrix from the oracle's restatement of dequantize.cuh (parity unpinned).
"""
import os

import numpy as np
import torch

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_GPTQ_SOURCES = ("gptq_llama70b_tp8_k1024_g128", "gptq_llama8b_k4096_g128", "gptq_llama8b_k14336_g128")
G = 128


def _f16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.float16)


class GptqPool:
    """Columns of group-size-128 GPTQ fixtures with at least k rows, as (qweight, scales, w_ref)
    pieces of k rows."""

    def __init__(self, k: int):
        self.pieces = []
        for name in _GPTQ_SOURCES:
            d = np.load(os.path.join(GOLD, name + ".npz"))
            kf = d["w_ref"].shape[0]
            for r0 in range(0, kf - k + 1, k):
                self.pieces.append((d["qweight"][r0 // 8:(r0 + k) // 8], d["s"][r0 // G:(r0 + k) // G],
                                    d["w_ref"][r0:r0 + k]))
        assert self.pieces, f"no GPTQ fixture has {k} rows"
        self.cursor = 0

    def take(self, n: int):
        """n columns -> qweight [k/8, n] i32, qzeros [k/128, n/8] i32, scales [k/128, n] f16 bits,
        g_idx [k] i32, w_ref [k, n] f16 bits."""
        qw, sc, wr = [], [], []
        have = 0
        while have < n:
            q, s, w = self.pieces[self.cursor % len(self.pieces)]
            mult = (1.0, 2.0, 0.5, 4.0)[(self.cursor // len(self.pieces)) % 4]
            self.cursor += 1
            cols = min(q.shape[1], n - have)
            qw.append(q[:, :cols])
            sc.append(O.to_bits(O.from_bits(s[:, :cols], O.DT_F16) * mult, O.DT_F16))
            wr.append(O.to_bits(O.from_bits(w[:, :cols], O.DT_F16) * mult, O.DT_F16))
            have += cols
        qweight, scales, w_ref = (np.ascontiguousarray(np.concatenate(x, axis=1)) for x in (qw, sc, wr))
        k = w_ref.shape[0]
        return dict(qweight=qweight.astype(np.int32),
                    qzeros=np.full((k // G, n // 8), 0x77777777, dtype=np.int32), scales=scales,
                    g_idx=(np.arange(k, dtype=np.int32) // G), w_ref=w_ref)


def awq_matrix(rng, k: int, n: int):
    qweight = rng.integers(-2**31, 2**31 - 1, size=(k, n // 8), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2**31, 2**31 - 1, size=(k // G, n // 8), dtype=np.int64).astype(np.int32)
    # a linear that keeps the RMS of its input: (q - z) has standard deviation sqrt(42.5)
    scales = (rng.uniform(0.5, 1.5, (k // G, n)) * (k**-0.5 / 42.5**0.5)).astype(np.float16).view(np.uint16)
    return dict(qweight=qweight, qzeros=qzeros, scales=scales,
                w_ref=O.awq_dequantize(qweight, scales, qzeros))


def build_checkpoint(method: str, cfg, seed: int = 0, kv_scales=None):
    """Returns (tensors, dense): `tensors` maps Hugging Face names to torch tensors (what a
    safetensors file of the model holds), `dense[layer][proj]` is the fp16 bit matrix [K, N] each
    projection must dequantise to."""
    rng = np.random.default_rng(seed)
    h, inter, d = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    hq, hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    shapes = {"self_attn.q_proj": (h, hq * d), "self_attn.k_proj": (h, hkv * d),
              "self_attn.v_proj": (h, hkv * d), "self_attn.o_proj": (hq * d, h),
              "mlp.gate_proj": (h, inter), "mlp.up_proj": (h, inter), "mlp.down_proj": (inter, h)}
    pools = {}
    tensors, dense = {}, []
    for li in range(cfg.num_hidden_layers):
        dl = {}
        for name, (k, n) in shapes.items():
            if method == "gptq":
                pool = pools.setdefault(k, GptqPool(k))
                m = pool.take(n)
            else:
                m = awq_matrix(rng, k, n)
            dl[name] = m.pop("w_ref")
            for pname, arr in m.items():
                t = _f16(arr) if pname == "scales" else torch.from_numpy(arr)
                tensors[f"model.layers.{li}.{name}.{pname}"] = t
        for norm in ("input_layernorm", "post_attention_layernorm"):
            tensors[f"model.layers.{li}.{norm}.weight"] = torch.from_numpy(
                rng.uniform(0.8, 1.2, h).astype(np.float16))
        if kv_scales is not None:
            tensors[f"model.layers.{li}.self_attn.kv_scale"] = torch.tensor(kv_scales[li], dtype=torch.float32)
        dense.append(dl)
    tensors["model.embed_tokens.weight"] = torch.from_numpy(rng.standard_normal((cfg.vocab_size, h)).astype(np.float16))
    tensors["lm_head.weight"] = torch.from_numpy((rng.standard_normal((cfg.vocab_size, h)) * h**-0.5).astype(np.float16))
    tensors["model.norm.weight"] = torch.from_numpy(rng.uniform(0.8, 1.2, h).astype(np.float16))
    return tensors, dense


def stacked_dense(dl):
    """The merged projections' dense matrices of one layer, unsharded (q | k | v, gate | up)."""
    cat = lambda names: np.ascontiguousarray(np.concatenate([dl[n] for n in names], axis=1))  # noqa: E731
    return dict(qkv=cat(["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"]),
                o=dl["self_attn.o_proj"], gate_up=cat(["mlp.gate_proj", "mlp.up_proj"]),
                down=dl["mlp.down_proj"])


def read_safetensors(path: str):
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}
