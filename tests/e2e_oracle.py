"""Test infrastructure: one decode step of a Llama-shaped stack restated with the oracle.

The pipeline rounds wherever the reference's kernels round (every linear's fp16 output, rotary,
the cache write, attention probabilities, norms, SiLU): layer order of
vllm/model_executor/models/llama.py:196-224 and :157-187.  Weights come in as dense fp16 bit
matrices [K, N] so that the caller decides where they come from (the oracle's AWQ dequantisation,
or the `w_ref` matrices the reference's own GPTQ packer produced).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from oracle import oracle as O

DT = O.DT_F16


@dataclass
class OracleLayer:
    ln1: np.ndarray
    ln2: np.ndarray
    qkv: np.ndarray          # [hidden, (Hq + 2 Hkv) * D] fp16 bits
    o: np.ndarray            # [Hq * D, hidden]
    gate_up: np.ndarray      # [hidden, 2 * inter]
    down: np.ndarray         # [inter, hidden]
    kv_scale: float = 1.0


@dataclass
class OracleStack:
    embed: np.ndarray        # [vocab, hidden] fp16 bits
    lm_head: np.ndarray      # [vocab, hidden] fp16 bits
    norm: np.ndarray
    layers: List[OracleLayer]
    cos_sin: np.ndarray
    hq: int
    hkv: int
    d: int
    eps: float


@dataclass
class StepResult:
    hidden: np.ndarray                       # final normed hidden state, fp16 bits [B, hidden]
    caches: List[tuple] = field(default_factory=list)   # per layer (key_cache, value_cache) after the write
    new_kv: List[tuple] = field(default_factory=list)   # per layer (k, v) rows that were written
    trace: Dict[str, np.ndarray] = field(default_factory=dict)

    def logits64(self, lm_head_bits) -> np.ndarray:
        return (O.from_bits(self.hidden, DT).astype(np.float64)
                @ O.from_bits(lm_head_bits, DT).astype(np.float64).T)


def decode_step(st: OracleStack, input_ids, positions, block_tables, seq_lens, slot_mapping,
                caches_before: List[np.ndarray], kv: str, block_size: int = 16,
                attend_to: Optional[List[np.ndarray]] = None, embed_rows=None) -> StepResult:
    """caches_before: per layer the [2, NB, BS*Hkv*D] cache as storage bits (uint8 for fp8).
    attend_to: per layer a [2, NB, BS*Hkv*D] cache the attention reads INSTEAD of the one this
    pipeline wrote (the cache of the implementation under test, after the caller has checked its
    written bytes against `caches`): quantising to fp8 is a step function, so a last-place
    difference of a new key or value that straddles a rounding boundary is stored as a whole fp8
    step, and reading each side's own bytes would compare two different problems from there on.
    embed_rows: replaces the embedding lookup (sensitivity experiments)."""
    kvc = O.KV_FP8 if kv == "fp8" else O.KV_AUTO
    x = 16 if kv == "fp8" else 8
    hq, hkv, d, bs = st.hq, st.hkv, st.d, block_size
    B = len(input_ids)
    res = StepResult(hidden=None)
    h = st.embed[input_ids] if embed_rows is None else embed_rows
    residual: Optional[np.ndarray] = None
    for li, L in enumerate(st.layers):
        if residual is None:
            residual = h.copy()
            h = O.rms_norm(h, L.ln1, st.eps, DT)
        else:
            h, residual = O.fused_add_rms_norm(h, residual, L.ln1, st.eps, DT)
        res.trace[f"{li}.ln1"] = h
        qkv = O.gemm_f16(h, L.qkv)
        res.trace[f"{li}.qkv"] = qkv
        q = np.ascontiguousarray(qkv[:, :hq * d])
        k = np.ascontiguousarray(qkv[:, hq * d:(hq + hkv) * d])
        v = np.ascontiguousarray(qkv[:, (hq + hkv) * d:])
        q, k = O.rotary_embedding(positions, q, k, d, st.cos_sin, True, DT)
        cache = caches_before[li]
        nb = cache.shape[1]
        kc = np.ascontiguousarray(cache[0]).reshape(nb, hkv, d // x, bs, x)
        vc = np.ascontiguousarray(cache[1]).reshape(nb, hkv, d, bs)
        O.reshape_and_cache(k.reshape(B, hkv, d), v.reshape(B, hkv, d), kc, vc, slot_mapping, DT, kvc,
                            L.kv_scale)
        res.caches.append((kc, vc))
        res.new_kv.append((k, v))
        res.trace[f"{li}.q_rot"] = q
        kc_r, vc_r = kc, vc
        if attend_to is not None:
            kc_r = np.ascontiguousarray(attend_to[li][0]).reshape(nb, hkv, d // x, bs, x)
            vc_r = np.ascontiguousarray(attend_to[li][1]).reshape(nb, hkv, d, bs)
        attn = O.paged_attention_v1(q.reshape(B, hq, d), kc_r, vc_r, block_tables, seq_lens, d**-0.5, DT,
                                    kvc, L.kv_scale)
        res.trace[f"{li}.attn"] = attn.reshape(B, hq * d)
        o = O.gemm_f16(attn.reshape(B, hq * d), L.o)
        res.trace[f"{li}.o"] = o
        h, residual = O.fused_add_rms_norm(o, residual, L.ln2, st.eps, DT)
        gu = O.gemm_f16(h, L.gate_up)
        act = O.silu_and_mul(gu, DT)
        res.trace[f"{li}.act"] = act
        h = O.gemm_f16(act, L.down)
        res.trace[f"{li}.down"] = h
    h, _ = O.fused_add_rms_norm(h, residual, st.norm, st.eps, DT)
    res.hidden = h
    return res
