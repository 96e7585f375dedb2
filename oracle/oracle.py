"""TEST INFRASTRUCTURE ONLY -- numpy/ctypes front end of the CPU restatement.

Loads oracle/liboracle.so (built from oracle/pa_oracle.c by oracle/Makefile) and
offers array-level helpers.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.

Arrays are numpy; 16-bit float tensors travel as uint16 bit patterns together with
a dtype code (DT_F16 / DT_BF16 / DT_F32) so that bfloat16 needs no numpy dtype.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")

DT_F16, DT_BF16, DT_F32 = 0, 1, 2
KV_AUTO, KV_FP8, KV_FP8_FNUZ = 0, 1, 2
PARTITION_SIZE = 512

_lib = None


def build() -> None:
    src = os.path.join(HERE, "pa_oracle.c")
    if (not os.path.exists(SO)) or (os.path.exists(src) and
                                    os.path.getmtime(SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.oracle_f16_to_f32.restype = C.c_float
        _lib.oracle_f16_to_f32.argtypes = [C.c_uint16]
        _lib.oracle_f32_to_f16.restype = C.c_uint16
        _lib.oracle_f32_to_f16.argtypes = [C.c_float]
        _lib.oracle_bf16_to_f32.restype = C.c_float
        _lib.oracle_bf16_to_f32.argtypes = [C.c_uint16]
        _lib.oracle_f32_to_bf16.restype = C.c_uint16
        _lib.oracle_f32_to_bf16.argtypes = [C.c_float]
        for n in ("oracle_e4m3fn_encode", "oracle_e4m3fnuz_encode"):
            getattr(_lib, n).restype = C.c_uint8
            getattr(_lib, n).argtypes = [C.c_float]
        for n in ("oracle_e4m3fn_decode", "oracle_e4m3fnuz_decode"):
            getattr(_lib, n).restype = C.c_float
            getattr(_lib, n).argtypes = [C.c_uint8]
        _lib.oracle_paged_attention_v1.restype = C.c_int
        _lib.oracle_paged_attention_v2.restype = C.c_int
        _lib.oracle_paged_attention_v1_bs.restype = C.c_int
        _lib.oracle_paged_attention_v2_bs.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return a


def scalar_np_dtype(dt):
    return np.float32 if dt == DT_F32 else np.uint16


# ------------------------------------------------------------------ conversions


def to_bits(x_f32: np.ndarray, dt: int) -> np.ndarray:
    """float32 values -> storage array of the model dtype (RNE)."""
    x = np.ascontiguousarray(x_f32, dtype=np.float32)
    if dt == DT_F32:
        return x.copy()
    if dt == DT_F16:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def from_bits(b: np.ndarray, dt: int) -> np.ndarray:
    if dt == DT_F32:
        return np.asarray(b, dtype=np.float32)
    if dt == DT_F16:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


def convert_fp8(src: np.ndarray, dt: int, to_fp8: bool, scale: float = 1.0,
                kv: int = KV_FP8) -> np.ndarray:
    """to_fp8: src holds model-dtype storage -> uint8; else src uint8 -> model dtype."""
    src = _c(np.ascontiguousarray(src))
    dst = np.empty(src.shape, dtype=np.uint8 if to_fp8 else scalar_np_dtype(dt))
    lib().oracle_convert_fp8(_p(dst), _p(src), C.c_int64(src.size), C.c_int(dt), C.c_int(kv),
                             C.c_float(scale), C.c_int(1 if to_fp8 else 0))
    return dst


# ------------------------------------------------------------------ cache ops


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, dt, kv=KV_AUTO,
                      kv_scale=1.0, key_stride=None, value_stride=None):
    """key/value: [T, H, D] storage arrays; caches updated in place.
    key_cache: [NB, H, D/x, BS, x], value_cache: [NB, H, D, BS]."""
    T, H, D = key.shape
    NB, H2, Dx, BS, x = key_cache.shape
    assert H2 == H and Dx * x == D
    ks = H * D if key_stride is None else key_stride
    vs = H * D if value_stride is None else value_stride
    slot_mapping = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    lib().oracle_reshape_and_cache(_p(key), _p(value), _p(_c(key_cache)), _p(_c(value_cache)),
                                   _p(slot_mapping), C.c_int(T), C.c_int(H), C.c_int(D),
                                   C.c_int(BS), C.c_int(x), C.c_int64(ks), C.c_int64(vs),
                                   C.c_int(dt), C.c_int(kv), C.c_float(kv_scale))


def copy_blocks(key_cache, value_cache, block_mapping):
    bm = np.ascontiguousarray(block_mapping, dtype=np.int64).reshape(-1, 2)
    block_bytes = key_cache[0].nbytes
    lib().oracle_copy_blocks(_p(_c(key_cache)), _p(_c(value_cache)), _p(bm), C.c_int(len(bm)),
                             C.c_int64(block_bytes))


def swap_blocks(src, dst, block_mapping):
    bm = np.ascontiguousarray(block_mapping, dtype=np.int64).reshape(-1, 2)
    lib().oracle_swap_blocks(_p(_c(src)), _p(_c(dst)), _p(bm), C.c_int(len(bm)),
                             C.c_int64(src[0].nbytes))


# ------------------------------------------------------------------ attention


def _attn_args(query, key_cache, value_cache, block_tables, seq_lens):
    S, Hq, D = query.shape
    NB, Hkv, Dx, BS, x = key_cache.shape
    assert value_cache.shape == (NB, Hkv, D, BS)
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
    return S, Hq, D, NB, Hkv, BS, x, bt, sl


def _bsp(blocksparse):
    """blocksparse: None or (tp_rank, local_blocks, vert_stride, block_size, head_sliding_step)."""
    if blocksparse is None:
        return None
    return np.ascontiguousarray(blocksparse, dtype=np.int32)


def paged_attention_v1(query, key_cache, value_cache, block_tables, seq_lens, scale, dt,
                       kv=KV_AUTO, kv_scale=1.0, alibi_slopes=None, blocksparse=None):
    """query: [S, Hq, D] storage array (contiguous). Returns out storage array."""
    S, Hq, D, NB, Hkv, BS, x, bt, sl = _attn_args(query, key_cache, value_cache, block_tables,
                                                  seq_lens)
    out = np.zeros((S, Hq, D), dtype=scalar_np_dtype(dt))
    al = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    bsp = _bsp(blocksparse)
    rc = lib().oracle_paged_attention_v1_bs(
        _p(out), _p(_c(query)), _p(_c(key_cache)), _p(_c(value_cache)), C.c_int(S), C.c_int(Hq),
        C.c_int(Hkv), C.c_int(D), C.c_int(BS), C.c_int(x), C.c_float(scale), _p(bt), _p(sl),
        C.c_int(bt.shape[1]), _p(al), C.c_int64(Hq * D), C.c_int64(Hkv * D * BS),
        C.c_int64(D * BS), C.c_int(dt), C.c_int(kv), C.c_float(kv_scale), _p(bsp))
    assert rc == 0
    return out


def paged_attention_v2(query, key_cache, value_cache, block_tables, seq_lens, scale, dt,
                       kv=KV_AUTO, kv_scale=1.0, alibi_slopes=None, max_seq_len=None,
                       blocksparse=None):
    """Returns (out, exp_sums, max_logits, tmp_out)."""
    S, Hq, D, NB, Hkv, BS, x, bt, sl = _attn_args(query, key_cache, value_cache, block_tables,
                                                  seq_lens)
    if max_seq_len is None:
        max_seq_len = int(sl.max())
    P = (max_seq_len + PARTITION_SIZE - 1) // PARTITION_SIZE
    out = np.zeros((S, Hq, D), dtype=scalar_np_dtype(dt))
    tmp = np.zeros((S, Hq, P, D), dtype=scalar_np_dtype(dt))
    es = np.zeros((S, Hq, P), dtype=np.float32)
    ml = np.zeros((S, Hq, P), dtype=np.float32)
    al = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    bsp = _bsp(blocksparse)
    rc = lib().oracle_paged_attention_v2_bs(
        _p(out), _p(es), _p(ml), _p(tmp), _p(_c(query)), _p(_c(key_cache)), _p(_c(value_cache)),
        C.c_int(S), C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(BS), C.c_int(x),
        C.c_float(scale), _p(bt), _p(sl), C.c_int(bt.shape[1]), C.c_int(P), _p(al),
        C.c_int64(Hq * D), C.c_int64(Hkv * D * BS), C.c_int64(D * BS), C.c_int(dt), C.c_int(kv),
        C.c_float(kv_scale), _p(bsp))
    assert rc == 0
    return out, es, ml, tmp


def prefill_attention(q, k, v, seq_lens, scale, dt, alibi_slopes=None):
    """q [T, Hq, D], k/v [T, Hkv, D] contiguous storage arrays; seq_lens: prompt lengths.
    Causal attention within each sequence; returns out [T, Hq, D]."""
    T_, Hq, D = q.shape
    Hkv = k.shape[1]
    cu = np.zeros(len(seq_lens) + 1, dtype=np.int32)
    cu[1:] = np.cumsum(seq_lens)
    assert cu[-1] == T_
    out = np.zeros_like(q)
    al = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    lib().oracle_prefill_attention(_p(out), _p(_c(q)), _p(_c(k)), _p(_c(v)), _p(cu),
                                   C.c_int(len(seq_lens)), C.c_int(Hq), C.c_int(Hkv), C.c_int(D),
                                   C.c_int64(Hq * D), C.c_int64(Hkv * D), C.c_int64(Hkv * D),
                                   C.c_int64(Hq * D), C.c_float(scale), _p(al), C.c_int(dt))
    return out


def prefix_prefill_attention(q, k, v, key_cache, value_cache, block_tables, query_lens, ctx_lens,
                             scale, dt, kv=KV_AUTO, kv_scale=1.0, alibi_slopes=None,
                             sliding_window=0):
    """q [T, Hq, D], k/v [T, Hkv, D]: the new tokens of every sequence, packed; key_cache
    [NB, Hkv, D/x, BS, x] and value_cache [NB, Hkv, D, BS] hold the ctx_lens[s] earlier tokens of
    sequence s at block_tables[s].  Returns out [T, Hq, D]."""
    T_, Hq, D = q.shape
    Hkv = k.shape[1]
    NB, _, _, BS, x = key_cache.shape
    assert value_cache.shape == (NB, Hkv, D, BS)
    qs = np.zeros(len(query_lens) + 1, dtype=np.int32)
    qs[1:] = np.cumsum(query_lens)
    assert qs[-1] == T_
    ctx = np.ascontiguousarray(ctx_lens, dtype=np.int32)
    sl = (ctx + np.asarray(query_lens, dtype=np.int32)).astype(np.int32)
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    out = np.zeros_like(q)
    al = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    lib().oracle_prefix_prefill_attention(
        _p(out), _p(_c(q)), _p(_c(k)), _p(_c(v)), _p(_c(key_cache)), _p(_c(value_cache)), _p(bt),
        C.c_int(bt.shape[1]), _p(qs), _p(sl), _p(ctx), C.c_int(len(query_lens)), C.c_int(Hq),
        C.c_int(Hkv), C.c_int(D), C.c_int(BS), C.c_int(x), C.c_int64(Hq * D), C.c_int64(Hkv * D),
        C.c_int64(Hkv * D), C.c_int64(Hq * D), C.c_int64(Hkv * D * BS), C.c_int64(D * BS),
        C.c_float(scale), _p(al), C.c_int(sliding_window), C.c_int(dt), C.c_int(kv),
        C.c_float(kv_scale))
    return out


# ------------------------------------------------------------------ int4 weights


def awq_dequantize(qweight, scales_f16bits, qzeros):
    """qweight [IC, OC/8] i32, scales [IC/G, OC] f16 bits, qzeros [IC/G, OC/8] i32 -> [IC, OC]."""
    ic, qoc = qweight.shape
    oc = qoc * 8
    group = ic // scales_f16bits.shape[0]
    out = np.empty((ic, oc), dtype=np.uint16)
    lib().oracle_awq_dequantize(_p(_c(np.ascontiguousarray(qweight, dtype=np.int32))),
                                _p(_c(scales_f16bits)), _p(_c(np.ascontiguousarray(
                                    qzeros, dtype=np.int32))), _p(out), C.c_int(ic), C.c_int(oc),
                                C.c_int(group))
    return out


def gptq_dequantize(qweight, qzeros, scales_f16bits, g_idx=None, shuffled=False):
    """qweight [K/8, N] i32, qzeros [groups, N/8] i32, scales [groups, N] -> [K, N] f16 bits."""
    k8, n = qweight.shape
    k = k8 * 8
    groups = qzeros.shape[0]
    out = np.empty((k, n), dtype=np.uint16)
    gi = None if g_idx is None else np.ascontiguousarray(g_idx, dtype=np.int32)
    lib().oracle_gptq_dequantize(_p(_c(np.ascontiguousarray(qweight, dtype=np.int32))),
                                 _p(_c(np.ascontiguousarray(qzeros, dtype=np.int32))),
                                 _p(_c(scales_f16bits)), _p(gi), _p(out), C.c_int(k), C.c_int(n),
                                 C.c_int(groups), C.c_int(1 if shuffled else 0))
    return out


def gptq_shuffle(qweight, q_perm=None):
    """Returns the exllama-ordered copy of qweight [K/8, N] i32."""
    w = np.ascontiguousarray(qweight, dtype=np.int32).copy()
    k8, n = w.shape
    qp = None if q_perm is None or len(q_perm) == 0 else np.ascontiguousarray(q_perm,
                                                                               dtype=np.int32)
    lib().oracle_gptq_shuffle(_p(w), _p(qp), C.c_int(k8 * 8), C.c_int(n))
    return w


def gemm_f16(a_bits, w_bits, a_perm=None):
    m, k = a_bits.shape
    k2, n = w_bits.shape
    assert k == k2
    out = np.empty((m, n), dtype=np.uint16)
    ap = None if a_perm is None else np.ascontiguousarray(a_perm, dtype=np.int32)
    lib().oracle_gemm_f16(_p(_c(a_bits)), _p(_c(w_bits)), _p(out), C.c_int(m), C.c_int(n),
                          C.c_int(k), _p(ap))
    return out


# ------------------------------------------------------------------ layer glue


def rms_norm(x, weight, eps, dt):
    x = _c(np.ascontiguousarray(x))
    out = np.empty_like(x)
    lib().oracle_rms_norm(_p(out), _p(x), None, _p(_c(weight)), C.c_float(eps),
                          C.c_int(x.size // x.shape[-1]), C.c_int(x.shape[-1]), C.c_int(dt),
                          C.c_int(0))
    return out


def fused_add_rms_norm(x, residual, weight, eps, dt):
    """Returns (new_input, new_residual)."""
    x, r = np.ascontiguousarray(x).copy(), np.ascontiguousarray(residual).copy()
    lib().oracle_rms_norm(None, _p(x), _p(r), _p(_c(weight)), C.c_float(eps),
                          C.c_int(x.size // x.shape[-1]), C.c_int(x.shape[-1]), C.c_int(dt),
                          C.c_int(1))
    return x, r


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, dt):
    """query [T, Hq*D], key [T, Hkv*D] contiguous storage arrays; returns rotated copies."""
    q, k = np.ascontiguousarray(query).copy(), np.ascontiguousarray(key).copy()
    pos = np.ascontiguousarray(positions, dtype=np.int64)
    T = pos.size
    q2, k2 = q.reshape(T, -1), k.reshape(T, -1)
    lib().oracle_rotary_embedding(_p(pos), _p(q), _p(k), _p(_c(cos_sin_cache)), C.c_int(T),
                                  C.c_int(q2.shape[1] // head_size),
                                  C.c_int(k2.shape[1] // head_size), C.c_int(head_size),
                                  C.c_int(cos_sin_cache.shape[1]), C.c_int64(q2.shape[1]),
                                  C.c_int64(k2.shape[1]), C.c_int(1 if is_neox else 0), C.c_int(dt))
    return q, k


def silu_and_mul(x, dt):
    x = _c(np.ascontiguousarray(x))
    d = x.shape[-1] // 2
    out = np.empty(x.shape[:-1] + (d, ), dtype=x.dtype)
    lib().oracle_silu_and_mul(_p(out), _p(x), C.c_int(x.size // (2 * d)), C.c_int(d), C.c_int(dt))
    return out


def num_threads() -> int:
    return lib().oracle_num_threads()
