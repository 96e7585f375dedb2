"""How far apart two CORRECT fp16 evaluations of one decode step are: the oracle pipeline
(tests/e2e_oracle.py) against itself, the second run with one fp16 ulp added to 2 % of the
embedding's elements -- the size of difference that a GPU rsqrt against libm's, or another
summation order, puts into the first norm's output.  Two layers at the widths of Llama-3-8B,
normalised random AWQ weights, 16-bit KV cache, as in tests/test_gpu_e2e_logits.py.

This is what the bars of the GPU test are measured against: the fp32 logits of the two oracle runs
differ by more than 1e-3 (so no implementation can be held to 1e-3 absolute on these logits by a
restatement that rounds to fp16 between steps), and stay within the GPU test's bars (hidden 2^-6 and
relative L2 1.5e-3, logits 2^-7).
"""
import numpy as np

from oracle import oracle as O
from tests import e2e_oracle as E

DT = O.DT_F16


def _rand_i32(rng, shape):
    return rng.integers(-2**31, 2**31 - 1, size=shape, dtype=np.int64).astype(np.int32)


def _awq_dense(rng, k, n):
    s = (rng.uniform(0.5, 1.5, (k // 128, n)) * (k**-0.5 / 42.5**0.5)).astype(np.float16).view(np.uint16)
    return O.awq_dequantize(_rand_i32(rng, (k, n // 8)), s, _rand_i32(rng, (k // 128, n // 8)))


def test_a_one_ulp_perturbation_moves_the_logits_by_more_than_1e_3():
    rng = np.random.default_rng(3)
    hidden, inter, hq, hkv, d, vocab, bs = 4096, 14336, 32, 8, 128, 2048, 16
    layers = [E.OracleLayer(np.full(hidden, 0x3C00, np.uint16), np.full(hidden, 0x3C00, np.uint16),
                            _awq_dense(rng, hidden, (hq + 2 * hkv) * d), _awq_dense(rng, hq * d, hidden),
                            _awq_dense(rng, hidden, 2 * inter), _awq_dense(rng, inter, hidden))
              for _ in range(2)]
    maxpos = 512
    inv = 1.0 / (500000.0**(np.arange(0, d, 2, dtype=np.float32) / d))
    fr = np.outer(np.arange(maxpos, dtype=np.float32), inv)
    cos_sin = O.to_bits(np.concatenate([np.cos(fr), np.sin(fr)], axis=-1), DT)
    st = E.OracleStack(O.to_bits(rng.standard_normal((vocab, hidden)).astype(np.float32), DT),
                       O.to_bits((rng.standard_normal((vocab, hidden)) * 0.02).astype(np.float32), DT),
                       np.full(hidden, 0x3C00, np.uint16), layers, cos_sin, hq, hkv, d, 1e-5)
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64], dtype=np.int32)
    B = len(seq_lens)
    max_blocks = (int(seq_lens.max()) + bs - 1) // bs
    nb = B * max_blocks + 3
    block_tables = rng.permutation(nb)[:B * max_blocks].reshape(B, max_blocks).astype(np.int32)
    positions = (seq_lens - 1).astype(np.int64)
    slots = np.array([block_tables[b, p // bs] * bs + p % bs for b, p in enumerate(positions)], dtype=np.int64)
    ids = rng.integers(0, vocab, B).astype(np.int64)
    caches = [O.to_bits(rng.uniform(-1, 1, (2, nb, bs * hkv * d)).astype(np.float32), DT) for _ in range(2)]
    a = E.decode_step(st, ids, positions, block_tables, seq_lens, slots, [c.copy() for c in caches], "auto")
    rows = st.embed[ids].copy()
    flip = rng.random(rows.shape) < 0.02
    rows[flip] += 1                                   # the next fp16 value away from zero
    b = E.decode_step(st, ids, positions, block_tables, seq_lens, slots, [c.copy() for c in caches], "auto",
                      embed_rows=rows)
    la, lb = a.logits64(st.lm_head), b.logits64(st.lm_head)
    ha, hb = (O.from_bits(r.hidden, DT).astype(np.float64) for r in (a, b))
    l_err = float(np.abs(la - lb).max())
    h_rel = float(np.linalg.norm(ha - hb) / np.linalg.norm(ha))
    print(f"oracle vs oracle with one ulp on 2 % of the embedding: hidden max|diff| "
          f"{np.abs(ha - hb).max():.3e}, rel L2 {h_rel:.3e}, fp32 logits max|diff| {l_err:.3e} "
          f"(|logit| <= {np.abs(la).max():.2f}, rms {np.sqrt(np.mean(la**2)):.2f})")
    assert l_err > 1e-3
    assert l_err <= 2.0**-7 and h_rel <= 1.5e-3 and np.abs(ha - hb).max() <= 2.0**-7
    assert (la.argmax(-1) == lb.argmax(-1)).all()
