"""Shared helpers for the parity tests: seeded synthetic inputs in the shape of the
reference's kernel tests (tests/kernels/test_attention.py:137-177, vllm/utils.py:398-446),
torch <-> oracle conversions, and an independent float64 restatement of the reference test's
expectation (ref_single_query_cached_kv_attention, tests/kernels/test_attention.py:58-111)."""
import random

import numpy as np
import torch

from oracle import oracle as O

TORCH_DT = {O.DT_F16: torch.float16, O.DT_BF16: torch.bfloat16, O.DT_F32: torch.float32}
DT_OF = {v: k for k, v in TORCH_DT.items()}


def bits(t: torch.Tensor) -> np.ndarray:
    """torch tensor (any device) -> numpy storage array as the oracle wants it."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8):
        return t.numpy().copy()
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def from_bits(a: np.ndarray, dt: int, device="cpu") -> torch.Tensor:
    if dt == O.DT_F32:
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(TORCH_DT[dt]).to(device)


def to_f32(a: np.ndarray, dt: int) -> np.ndarray:
    return O.from_bits(a, dt)


class AttnCase:
    pass


def make_attention_case(seed, num_seqs, num_heads, head_size, block_size, dt, kv_fp8, use_alibi,
                        max_seq_len, num_blocks=None, seq_lens=None, unique_blocks=False,
                        device="cuda"):
    """Inputs as in the reference test: q ~ U(-scale, scale), caches ~ U(-scale, scale)
    (fp8: quantised from fp16 draws, vllm/utils.py:324-341), random block tables with repeats,
    seq_lens random with the last one at the maximum."""
    rng = np.random.default_rng(seed)
    random.seed(seed)
    hq, hkv = num_heads
    scale = float(1.0 / (head_size**0.5))
    c = AttnCase()
    c.dt, c.kv, c.scale, c.hq, c.hkv, c.d, c.bs = dt, (O.KV_FP8 if kv_fp8 else O.KV_AUTO), scale, hq, hkv, head_size, block_size
    if seq_lens is None:
        seq_lens = [random.randint(1, max_seq_len) for _ in range(num_seqs)]
        seq_lens[-1] = max_seq_len
    c.seq_lens = np.array(seq_lens, dtype=np.int32)
    c.max_seq_len = int(c.seq_lens.max())
    mb = (c.max_seq_len + block_size - 1) // block_size
    if num_blocks is None:
        num_blocks = 1024
    if unique_blocks:
        assert num_blocks >= num_seqs * mb
        c.block_tables = rng.permutation(num_blocks)[:num_seqs * mb].reshape(num_seqs, mb).astype(np.int32)
    else:
        c.block_tables = rng.integers(0, num_blocks, (num_seqs, mb)).astype(np.int32)
    esz = 1 if kv_fp8 else (4 if dt == O.DT_F32 else 2)
    x = 16 // esz
    c.x = x
    q = rng.uniform(-scale, scale, (num_seqs, hq, head_size)).astype(np.float32)
    c.q = O.to_bits(q, dt)
    kshape = (num_blocks, hkv, head_size // x, block_size, x)
    vshape = (num_blocks, hkv, head_size, block_size)
    kf = rng.uniform(-scale, scale, kshape).astype(np.float32)
    vf = rng.uniform(-scale, scale, vshape).astype(np.float32)
    if kv_fp8:
        c.kc = O.convert_fp8(O.to_bits(kf, O.DT_F16), O.DT_F16, True)
        c.vc = O.convert_fp8(O.to_bits(vf, O.DT_F16), O.DT_F16, True)
    else:
        c.kc, c.vc = O.to_bits(kf, dt), O.to_bits(vf, dt)
    c.alibi = rng.standard_normal(hq).astype(np.float32) if use_alibi else None
    if device is not None:
        c.t_q = from_bits(c.q, dt, device)
        c.t_kc = torch.from_numpy(c.kc).to(device) if kv_fp8 else from_bits(c.kc, dt, device)
        c.t_vc = torch.from_numpy(c.vc).to(device) if kv_fp8 else from_bits(c.vc, dt, device)
        c.t_bt = torch.from_numpy(c.block_tables).to(device)
        c.t_sl = torch.from_numpy(c.seq_lens).to(device)
        c.t_alibi = torch.from_numpy(c.alibi).to(device) if use_alibi else None
    return c


def oracle_attention(c, version="v1", kv_scale=1.0, seq_subset=None):
    q, bt, sl = c.q, c.block_tables, c.seq_lens
    if seq_subset is not None:
        q, bt, sl = np.ascontiguousarray(q[seq_subset]), np.ascontiguousarray(bt[seq_subset]), sl[seq_subset]
    if version == "v1":
        return O.paged_attention_v1(q, c.kc, c.vc, bt, sl, c.scale, c.dt, kv=c.kv,
                                    kv_scale=kv_scale, alibi_slopes=c.alibi)
    return O.paged_attention_v2(q, c.kc, c.vc, bt, sl, c.scale, c.dt, kv=c.kv, kv_scale=kv_scale,
                                alibi_slopes=c.alibi, max_seq_len=int(c.seq_lens.max()))[0]


def run_hip_attention(ops, c, version="v1", kv_scale=1.0, seq_order=None):
    """Call the HIP op exactly as the reference test calls vllm._custom_ops
    (tests/kernels/test_attention.py:180-227)."""
    out = torch.empty_like(c.t_q)
    kvd = "fp8" if c.kv == O.KV_FP8 else "auto"
    if version == "v1":
        ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl, c.bs,
                               c.max_seq_len, c.t_alibi, kvd, kv_scale, seq_order=seq_order)
    else:
        S, H, D = out.shape
        P = (c.max_seq_len + 511) // 512
        tmp = torch.empty((S, H, P, D), dtype=out.dtype, device=out.device)
        es = torch.empty((S, H, P), dtype=torch.float32, device=out.device)
        ml = torch.empty_like(es)
        ops.paged_attention_v2(out, es, ml, tmp, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt,
                               c.t_sl, c.bs, c.max_seq_len, c.t_alibi, kvd, kv_scale,
                               seq_order=seq_order)
    torch.cuda.synchronize()
    return out


def float64_reference(c, kv_scale=1.0):
    """Gather-and-softmax in float64 (no rounding of probabilities): the 'true' answer both the
    kernel and the oracle approximate."""
    S = len(c.seq_lens)
    out = np.zeros((S, c.hq, c.d), dtype=np.float64)
    if c.kv == O.KV_FP8:
        kc = O.from_bits(O.convert_fp8(c.kc, c.dt, False, kv_scale), c.dt).astype(np.float64)
        vc = O.from_bits(O.convert_fp8(c.vc, c.dt, False, kv_scale), c.dt).astype(np.float64)
    else:
        kc, vc = O.from_bits(c.kc, c.dt).astype(np.float64), O.from_bits(c.vc, c.dt).astype(np.float64)
    q = O.from_bits(c.q, c.dt).astype(np.float64)
    qpk = c.hq // c.hkv
    for s in range(S):
        n = int(c.seq_lens[s])
        t = np.arange(n)
        blk, off = c.block_tables[s][t // c.bs], t % c.bs
        k = kc[blk, :, :, off, :].reshape(n, c.hkv, c.d)   # [n, hkv, d]
        v = vc[blk, :, :, off]                              # [n, hkv, d]
        for h in range(c.hq):
            logits = c.scale * (k[:, h // qpk, :] @ q[s, h])
            if c.alibi is not None:
                logits = logits + float(c.alibi[h]) * (t - n + 1)
            p = np.exp(logits - logits.max())
            p /= p.sum()
            out[s, h] = p @ v[:, h // qpk, :]
    return out
