"""GPU parity tests for paged_attention_v1 / v2 through the C-ABI library.

Modelled on the reference's tests/kernels/test_attention.py:114-274 (same parameter axes,
same input recipe, same tolerances: atol 1e-3 / rtol 1e-5, fp8 KV atol 1e-2) with the oracle
(oracle/pa_oracle.c) as the expectation instead of the in-test PyTorch loop.
"""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu

ATOL, RTOL, ATOL_FP8 = 1e-3, 1e-5, 1e-2  # tests/kernels/test_attention.py:271-273


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def _check(got: torch.Tensor, want_bits: np.ndarray, dt, fp8):
    g = got.float().cpu().numpy()
    w = T.to_f32(want_bits, dt)
    assert np.isfinite(g).all()
    np.testing.assert_allclose(g, w, atol=ATOL_FP8 if fp8 else ATOL, rtol=RTOL)


# (40, 40) and (64, 8) are the reference test's; (32, 8) is Llama-3-8B, (12, 12) OPT-125M and
# (8, 1) one rank of Llama-3-70B under 8-way tensor parallelism (64 / 8 query heads, 8 / 8 KV
# heads, vllm/model_executor/models/llama.py:103-119)
NUM_HEADS = [(40, 40), (64, 8), (32, 8), (12, 12), (8, 1)]
HEAD_SIZES = [64, 80, 96, 112, 128]
DTYPES = [O.DT_F16, O.DT_BF16]


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", NUM_HEADS)
@pytest.mark.parametrize("head_size", HEAD_SIZES)
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("use_alibi", [False, True])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_paged_attention_matrix(ops, version, num_heads, head_size, block_size, use_alibi, dt,
                                kv_fp8):
    seed = zlib.crc32(repr((version, num_heads, head_size, block_size, use_alibi, dt,
                            kv_fp8)).encode()) % 1000
    c = T.make_attention_case(seed, 7, num_heads, head_size, block_size, dt, kv_fp8, use_alibi,
                              max_seq_len=1500, num_blocks=256)
    out = T.run_hip_attention(ops, c, version)
    _check(out, T.oracle_attention(c, version), dt, kv_fp8)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("head_size", [192, 256])
@pytest.mark.parametrize("block_size", [8, 16])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16, O.DT_F32])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_paged_attention_large_heads_small_blocks_fp32(ops, version, head_size, block_size, dt,
                                                       kv_fp8):
    c = T.make_attention_case(head_size + block_size, 5, (16, 4), head_size, block_size, dt,
                              kv_fp8, True, max_seq_len=700, num_blocks=200)
    out = T.run_hip_attention(ops, c, version)
    _check(out, T.oracle_attention(c, version), dt, kv_fp8)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(32, 8), (12, 12), (8, 1)])
@pytest.mark.parametrize("head_size", [64, 80, 128])
@pytest.mark.parametrize("use_alibi", [False, True])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_block_size_8(ops, version, num_heads, head_size, use_alibi, dt, kv_fp8):
    """8-token blocks (the smallest the reference builds, attention_kernels.cu:789-803) on the
    throughput kernel: a 16-token tile spans two blocks, a byte cache's value rows are 8 bytes."""
    seed = zlib.crc32(repr((version, num_heads, head_size, use_alibi, dt, kv_fp8)).encode()) % 1000
    c = T.make_attention_case(seed, 6, num_heads, head_size, 8, dt, kv_fp8, use_alibi,
                              max_seq_len=1300, num_blocks=400)
    out = T.run_hip_attention(ops, c, version)
    _check(out, T.oracle_attention(c, version), dt, kv_fp8)


@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_block_size_8_edges_and_poisoned_tails(ops, kv_fp8, version):
    """Lengths around block, tile, step and partition boundaries with 8-token blocks; every cache
    slot past a sequence's end, and every unused block, holds NaN."""
    seq_lens = [1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 511, 512, 513, 1030]
    c = T.make_attention_case(21, len(seq_lens), (32, 8), 128, 8, O.DT_F16, kv_fp8, False,
                              max_seq_len=max(seq_lens), num_blocks=1900, seq_lens=seq_lens,
                              unique_blocks=True)
    want = T.oracle_attention(c, version)
    kc, vc = c.t_kc.clone(), c.t_vc.clone()
    poison = 0x7F if kv_fp8 else float("nan")
    used = set()
    for s, n in enumerate(c.seq_lens):
        nb = (int(n) + 7) // 8
        used.update(int(c.block_tables[s, b]) for b in range(nb))
        off = int(n) % 8
        if off:
            last = int(c.block_tables[s, nb - 1])
            kc[last, :, :, off:, :] = poison
            vc[last, :, :, off:] = poison
    for b in range(kc.shape[0]):
        if b not in used:
            kc[b] = poison
            vc[b] = poison
    c.t_kc, c.t_vc = kc, vc
    out = T.run_hip_attention(ops, c, version)
    _check(out, want, O.DT_F16, kv_fp8)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_F32])
def test_generic_kernel_on_fast_shapes(ops, version, dt):
    # force the coverage kernel onto a shape the MFMA kernel would take
    ops.set_attention_path(1)
    try:
        c = T.make_attention_case(11, 6, (32, 8), 128, 16, dt, dt == O.DT_F16, False,
                                  max_seq_len=1100, num_blocks=128)
        out = T.run_hip_attention(ops, c, version)
        _check(out, T.oracle_attention(c, version), dt, dt == O.DT_F16)
    finally:
        ops.set_attention_path(0)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_long_context_like_reference_max(ops, version, kv_fp8):
    # the reference test's MAX_SEQ_LEN is shared_memory / 4 - 512 (test_attention.py:17);
    # 64 KiB of LDS gives 15872. V2 additionally gets a context past V1's 8192 limit.
    max_len = 15872
    c = T.make_attention_case(5, 3, (32, 8), 128, 16, O.DT_F16, kv_fp8, False, max_seq_len=max_len,
                              num_blocks=1200)
    out = T.run_hip_attention(ops, c, version)
    _check(out, T.oracle_attention(c, version), O.DT_F16, kv_fp8)


@pytest.mark.parametrize("seq_lens", [[1], [1, 2, 15, 16, 17], [31, 32, 33, 63, 64, 65],
                                      [511, 512, 513, 1023, 1024, 1025]])
@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_edge_sequence_lengths(ops, seq_lens, kv_fp8, version):
    c = T.make_attention_case(3, len(seq_lens), (32, 8), 128, 16, O.DT_BF16, kv_fp8, False,
                              max_seq_len=max(seq_lens), num_blocks=80, seq_lens=seq_lens)
    out = T.run_hip_attention(ops, c, version)
    _check(out, T.oracle_attention(c, version), O.DT_BF16, kv_fp8)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("path", ["mfma", "generic"])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_empty_sequences_give_zero_rows(ops, version, path, kv_fp8):
    # a padding row of a captured decode batch has seq_len 0: the reference kernels store zeros
    # for it (V1: exp_sum 0 and zero accumulators, attention_kernels.cu:333-345, :483-495; V2:
    # the reduce over zero partitions, :637-668), never stale memory
    seq_lens = [0, 5, 0, 700, 0]
    c = T.make_attention_case(17, len(seq_lens), (32, 8), 128, 16, O.DT_F16, kv_fp8, False,
                              max_seq_len=700, num_blocks=96, seq_lens=seq_lens)
    want = T.oracle_attention(c, version)
    assert not T.to_f32(want, O.DT_F16)[[0, 2, 4]].any()
    ops.set_attention_path(1 if path == "generic" else 0)
    try:
        out = torch.full_like(c.t_q, float("nan"))
        kvd = "fp8" if kv_fp8 else "auto"
        if version == "v1":
            ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl,
                                   c.bs, c.max_seq_len, None, kvd, 1.0)
        else:
            S, H, D = out.shape
            tmp = torch.full((S, H, 2, D), float("nan"), dtype=out.dtype, device="cuda")
            es = torch.full((S, H, 2), float("nan"), dtype=torch.float32, device="cuda")
            ml = torch.full_like(es, float("nan"))
            ops.paged_attention_v2(out, es, ml, tmp, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale,
                                   c.t_bt, c.t_sl, c.bs, c.max_seq_len, None, kvd, 1.0)
        torch.cuda.synchronize()
    finally:
        ops.set_attention_path(0)
    assert not out[[0, 2, 4]].float().cpu().numpy().any()
    _check(out, want, O.DT_F16, kv_fp8)


@pytest.mark.parametrize("kv_fp8", [False, True])
def test_tail_garbage_is_ignored(ops, kv_fp8):
    # tokens past seq_len inside the last block (and whole unused blocks) may hold NaN/Inf
    # (attention_kernels.cu:420-430): poison them and expect the same answer
    c = T.make_attention_case(9, 4, (32, 8), 128, 16, O.DT_F16, kv_fp8, False, max_seq_len=300,
                              num_blocks=96, seq_lens=[5, 70, 129, 300], unique_blocks=True)
    want = T.oracle_attention(c, "v1")
    kc, vc = c.t_kc.clone(), c.t_vc.clone()
    poison = 0x7F if kv_fp8 else float("nan")
    used = set()
    for s, n in enumerate(c.seq_lens):
        nb = (int(n) + 15) // 16
        for b in range(nb):
            used.add(int(c.block_tables[s, b]))
        last = int(c.block_tables[s, nb - 1])
        off = int(n) % 16
        if off:
            kc[last, :, :, off:, :] = poison
            vc[last, :, :, off:] = poison
    for b in range(kc.shape[0]):
        if b not in used:
            kc[b] = poison
            vc[b] = poison
    c.t_kc, c.t_vc = kc, vc
    for version in ("v1", "v2"):
        out = T.run_hip_attention(ops, c, version)
        _check(out, want, O.DT_F16, kv_fp8)


def test_kv_scale_is_applied(ops):
    c = T.make_attention_case(21, 4, (32, 8), 128, 16, O.DT_F16, True, False, max_seq_len=600,
                              num_blocks=64)
    for kv_scale in (0.5, 2.0):
        out = T.run_hip_attention(ops, c, "v1", kv_scale=kv_scale)
        _check(out, T.oracle_attention(c, "v1", kv_scale=kv_scale), O.DT_F16, True)


def test_strided_query(ops):
    # q is a slice of a fused qkv projection in the model (attention_kernels.cu:177-179)
    c = T.make_attention_case(31, 5, (32, 8), 128, 16, O.DT_F16, False, False, max_seq_len=400,
                              num_blocks=64)
    qkv = torch.zeros((5, 48, 128), dtype=torch.float16, device="cuda")
    qkv[:, :32] = c.t_q
    c.t_q = qkv[:, :32]
    assert not c.t_q.is_contiguous()
    out = torch.empty((5, 32, 128), dtype=torch.float16, device="cuda")
    ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl, c.bs,
                           c.max_seq_len, None, "auto", 1.0)
    torch.cuda.synchronize()
    _check(out, T.oracle_attention(c, "v1"), O.DT_F16, False)


@pytest.mark.parametrize("name", ["attn_f32_gqa", "attn_bf16_alibi", "attn_f32_d80",
                                  "attn_bf16_opt125m", "attn_bf16_llama70b_tp8"])
def test_against_reference_cpu_kernel_fixtures(ops, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    dt = {"float32": O.DT_F32, "bfloat16": O.DT_BF16}[str(z["dtype"])]
    q = T.from_bits(z["query"], dt, "cuda")
    kc = T.from_bits(z["key_cache"], dt, "cuda")
    vc = T.from_bits(z["value_cache"], dt, "cuda")
    bt = torch.from_numpy(z["block_tables"]).cuda()
    sl = torch.from_numpy(z["seq_lens"]).cuda()
    al = torch.from_numpy(z["alibi"]).cuda() if z["alibi"].size else None
    hkv, bs, max_len = kc.shape[1], vc.shape[3], int(z["seq_lens"].max())
    out = torch.empty_like(q)
    ops.paged_attention_v1(out, q, kc, vc, hkv, float(z["scale"]), bt, sl, bs, max_len, al,
                           "auto", 1.0)
    torch.cuda.synchronize()
    tol = 2e-6 if dt == O.DT_F32 else 1e-3
    np.testing.assert_allclose(out.float().cpu().numpy(), T.to_f32(z["out_v1"], dt), atol=tol,
                               rtol=1e-5)
    P = (max_len + 511) // 512
    S, H, D = q.shape
    out2 = torch.empty_like(q)
    tmp = torch.empty((S, H, P, D), dtype=q.dtype, device="cuda")
    es = torch.empty((S, H, P), dtype=torch.float32, device="cuda")
    ml = torch.empty_like(es)
    ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, hkv, float(z["scale"]), bt, sl, bs,
                           max_len, al, "auto", 1.0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.float().cpu().numpy(), T.to_f32(z["out_v2"], dt), atol=tol,
                               rtol=1e-5)


def test_errors_match_reference_messages(ops):
    c = T.make_attention_case(1, 2, (8, 8), 64, 16, O.DT_F16, False, False, max_seq_len=40,
                              num_blocks=8)
    out = torch.empty_like(c.t_q)
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, 8, c.scale, c.t_bt, c.t_sl, 4, 40,
                               None, "auto", 1.0)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, 8, c.scale, c.t_bt, c.t_sl, 16, 40,
                               None, "fp8_e5m2", 1.0)
    q72 = torch.zeros((2, 8, 72), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        ops.paged_attention_v1(torch.empty_like(q72), q72, c.t_kc, c.t_vc, 8, c.scale, c.t_bt,
                               c.t_sl, 16, 40, None, "auto", 1.0)
    with pytest.raises(RuntimeError, match="[Bb]locksparse"):   # sparse block size must be positive
        ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, 8, c.scale, c.t_bt, c.t_sl, 16, 40,
                               None, "auto", 1.0, 0, 16, 8, 0, 0)


# ---------------------------------------------------------------- full-size properties


@pytest.mark.parametrize("kv_fp8", [True, False])
def test_full_size_properties(ops, kv_fp8):
    """BASELINE sizes (bs=256, seq=4k, Llama-3-8B heads) through size-independent checks:
    a sub-sample of sequences against the oracle, v1 == v2 within tolerance, invariance under
    a re-numbering of the physical blocks, and convexity (outputs inside V's range)."""
    B, S, BS = 256, 4096, 16
    nb = B * S // BS
    dt = O.DT_F16
    g = torch.Generator(device="cuda").manual_seed(0)
    scale = 128**-0.5
    q = (torch.rand((B, 32, 128), generator=g, device="cuda") * 2 - 1).mul_(scale).half()
    shape_k = (nb, 8, 128 // (16 if kv_fp8 else 8), BS, 16 if kv_fp8 else 8)
    shape_v = (nb, 8, 128, BS)
    if kv_fp8:
        kc = torch.empty(shape_k, dtype=torch.uint8, device="cuda")
        vc = torch.empty(shape_v, dtype=torch.uint8, device="cuda")
        for cache in (kc, vc):
            for i in range(0, nb, 8192):
                tmp = (torch.rand(cache[i:i + 8192].shape, generator=g, device="cuda") * 2 - 1)
                tmp = tmp.mul_(scale).half()
                ops.convert_fp8(cache[i:i + 8192], tmp)
    else:
        kc = (torch.rand(shape_k, generator=g, device="cuda") * 2 - 1).mul_(scale).half()
        vc = (torch.rand(shape_v, generator=g, device="cuda") * 2 - 1).mul_(scale).half()
    perm = torch.randperm(nb, generator=g, device="cuda").to(torch.int32)
    bt = perm.view(B, S // BS).contiguous()
    sl = torch.full((B, ), S, dtype=torch.int32, device="cuda")
    kvd = "fp8" if kv_fp8 else "auto"

    def v1(kc_, vc_, bt_):
        out = torch.empty_like(q)
        ops.paged_attention_v1(out, q, kc_, vc_, 8, scale, bt_, sl, BS, S, None, kvd, 1.0)
        return out

    out1 = v1(kc, vc, bt)
    P = S // 512
    out2 = torch.empty_like(q)
    tmp = torch.empty((B, 32, P, 128), dtype=q.dtype, device="cuda")
    es = torch.empty((B, 32, P), dtype=torch.float32, device="cuda")
    ml = torch.empty_like(es)
    ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, 8, scale, bt, sl, BS, S, None, kvd, 1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(out1).all()
    tol = ATOL_FP8 if kv_fp8 else ATOL
    assert (out1.float() - out2.float()).abs().max().item() <= tol
    assert out1.float().abs().max().item() <= scale * 1.001  # convex combination of V

    # sub-sample against the oracle
    pick = [0, 101, 255]
    c = T.AttnCase()
    c.dt, c.kv, c.scale, c.alibi = dt, (O.KV_FP8 if kv_fp8 else O.KV_AUTO), scale, None
    c.q = T.bits(q[pick])
    # gather only the blocks these sequences use, renumbered compactly
    used = bt[pick].reshape(-1).long()
    c.kc = T.bits(kc[used])
    c.vc = T.bits(vc[used])
    c.block_tables = np.arange(len(pick) * (S // BS), dtype=np.int32).reshape(len(pick), -1)
    c.seq_lens = np.full(len(pick), S, dtype=np.int32)
    want = T.oracle_attention(c, "v1")
    _check(out1[pick], want, dt, kv_fp8)

    # renumbering physical blocks must not change a single bit
    perm2 = torch.randperm(nb, generator=g, device="cuda")
    inv = torch.empty_like(perm2)
    inv[perm2] = torch.arange(nb, device="cuda")
    kc2, vc2 = kc[perm2], vc[perm2]          # block j of the new cache = old block perm2[j]
    bt2 = inv[bt.long()].to(torch.int32)     # old id -> new id
    del kc, vc
    out3 = v1(kc2, vc2, bt2)
    torch.cuda.synchronize()
    assert torch.equal(out1, out3)


# ---------------------------------------------------------------- block-sparse attention
# The reference's block-sparse variant of the same kernels (is_block_sparse = vert_stride > 1,
# csrc/attention/attention_kernels.cu:209-254, :385-393, :821; its test:
# tests/kernels/test_blocksparse_attention.py:141-306 with local_blocks 16, vert_stride 8,
# sparse block size 64, head sliding step 0 and 1).  Expectation: the oracle's restatement of the
# mask, plus an independent float64 softmax over the attended tokens only.
def _bs_mask(seq_len, block_size, head, kv_head, hq, hkv, tp_rank, local_blocks, vert_stride,
             bs_block_size, step):
    offset = ((tp_rank * hq + head) * step + 1) if step >= 0 else ((tp_rank * hkv + kv_head) * (-step) + 1)
    t = np.arange(seq_len)
    k_bs = (t // block_size) * block_size // bs_block_size
    q_bs = (seq_len - 1) // bs_block_size
    return ((k_bs + offset) % vert_stride == 0) | (k_bs > q_bs - local_blocks)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_fp8", [False, True])
@pytest.mark.parametrize("params", [(0, 16, 8, 64, 0), (0, 2, 4, 64, 1), (1, 1, 3, 32, -1), (0, 4, 2, 16, 2)])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
def test_blocksparse_attention(ops, version, kv_fp8, params, dt):
    tp_rank, local_blocks, vert_stride, bs_block_size, step = params
    c = T.make_attention_case(zlib.crc32(repr((version, kv_fp8, params, dt)).encode()), 5, (8, 2), 128,
                              16, dt, kv_fp8, False, 1500)
    out = torch.empty_like(c.t_q)
    kvd = "fp8" if kv_fp8 else "auto"
    bs_args = (tp_rank, local_blocks, vert_stride, bs_block_size, step)
    if version == "v1":
        ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl, c.bs,
                               c.max_seq_len, None, kvd, 1.0, *bs_args)
        want = O.paged_attention_v1(c.q, c.kc, c.vc, c.block_tables, c.seq_lens, c.scale, dt, kv=c.kv,
                                    blocksparse=bs_args)
    else:
        S, H, D = out.shape
        P = (c.max_seq_len + 511) // 512
        tmp = torch.empty((S, H, P, D), dtype=out.dtype, device="cuda")
        es = torch.empty((S, H, P), dtype=torch.float32, device="cuda")
        ml = torch.empty_like(es)
        ops.paged_attention_v2(out, es, ml, tmp, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt,
                               c.t_sl, c.bs, c.max_seq_len, None, kvd, 1.0, *bs_args)
        want = O.paged_attention_v2(c.q, c.kc, c.vc, c.block_tables, c.seq_lens, c.scale, dt, kv=c.kv,
                                    max_seq_len=c.max_seq_len, blocksparse=bs_args)[0]
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    atol = 1e-2 if kv_fp8 else (1e-3 if dt == O.DT_F16 else 2e-3)
    np.testing.assert_allclose(got, O.from_bits(want, dt), atol=atol, rtol=1e-5)
    # independent float64 expectation: softmax over the attended tokens only
    if kv_fp8:
        kc = O.from_bits(O.convert_fp8(c.kc, dt, False, 1.0), dt).astype(np.float64)
        vc = O.from_bits(O.convert_fp8(c.vc, dt, False, 1.0), dt).astype(np.float64)
    else:
        kc, vc = O.from_bits(c.kc, dt).astype(np.float64), O.from_bits(c.vc, dt).astype(np.float64)
    q = O.from_bits(c.q, dt).astype(np.float64)
    for s in range(len(c.seq_lens)):
        n = int(c.seq_lens[s])
        t = np.arange(n)
        blk, off = c.block_tables[s][t // c.bs], t % c.bs
        k = kc[blk, :, :, off, :].reshape(n, c.hkv, c.d)
        v = vc[blk, :, :, off]
        for h in range(c.hq):
            m = _bs_mask(n, c.bs, h, h // (c.hq // c.hkv), c.hq, c.hkv, *bs_args)
            logits = c.scale * (k[m, h // (c.hq // c.hkv), :] @ q[s, h])
            p = np.exp(logits - logits.max())
            ref = (p / p.sum()) @ v[m, h // (c.hq // c.hkv), :]
            np.testing.assert_allclose(got[s, h], ref, atol=3 * atol, rtol=1e-3)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("head_size,block_size", [(128, 16), (112, 8)])   # MFMA and generic kernel
def test_sequence_order_does_not_change_results(ops, version, head_size, block_size):
    """seq_order only says which grid row takes which sequence: with the longest-first order
    of make_seq_order, and with an arbitrary permutation, every output bit stays the same."""
    c = T.make_attention_case(21, 33, (8, 2), head_size, block_size, O.DT_F16, head_size == 128,
                              False, 1500)
    base = T.run_hip_attention(ops, c, version)
    order = ops.make_seq_order(c.t_sl)
    lens = c.t_sl.cpu().numpy()
    assert sorted(order.cpu().tolist()) == list(range(33))
    assert (np.diff(lens[order.cpu().numpy()]) <= 0).all()
    for o in (order, torch.randperm(33, device="cuda").to(torch.int32)):
        got = T.run_hip_attention(ops, c, version, seq_order=o)
        assert torch.equal(got.view(torch.int16), base.view(torch.int16))


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_fp8", [False, True])
def test_kv_load_policy_does_not_change_results(ops, version, kv_fp8):
    # nontemporal or default-policy K/V loads (vla_set_kv_load_policy; no counterpart in the
    # reference) is a cache hint: the output must be the same bits, and right
    c = T.make_attention_case(23, 5, (32, 8), 128, 16, O.DT_F16, kv_fp8, False, max_seq_len=900,
                              num_blocks=320, seq_lens=[1, 17, 512, 513, 900])
    want = T.oracle_attention(c, version)
    kvd = "fp8" if kv_fp8 else "auto"
    outs = []
    try:
        for policy in (1, 2, 0):
            ops.set_kv_load_policy(policy)
            out = torch.full_like(c.t_q, float("nan"))
            if version == "v1":
                ops.paged_attention_v1(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl,
                                       c.bs, c.max_seq_len, None, kvd, 1.0)
            else:
                S, H, D = out.shape
                tmp = torch.empty((S, H, 2, D), dtype=out.dtype, device="cuda")
                es = torch.empty((S, H, 2), dtype=torch.float32, device="cuda")
                ml = torch.empty_like(es)
                ops.paged_attention_v2(out, es, ml, tmp, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale,
                                       c.t_bt, c.t_sl, c.bs, c.max_seq_len, None, kvd, 1.0)
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        ops.set_kv_load_policy(0)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    _check(outs[0], want, O.DT_F16, kv_fp8)
    with pytest.raises(Exception):
        ops.set_kv_load_policy(3)
