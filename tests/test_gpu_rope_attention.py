"""GPU parity tests of the launch that does rotary embedding, the cache write and paged attention of a
decode step at once (vla_rope_paged_attention): its cache bytes and its output must equal those of
vla_rotary_reshape_and_cache followed by vla_paged_attention_v1 / _v2 bit for bit -- the separate
entry points are the ones checked against the oracle (tests/test_gpu_fused.py,
tests/test_gpu_attention.py).  The oracle checks the composition once more on a small case."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vllm_legacy_amd import _custom_ops
    return _custom_ops


def _cos_sin(rot_dim, max_pos, dt):
    inv = 1.0 / (10000.0**(np.arange(0, rot_dim, 2, dtype=np.float32) / rot_dim))
    f = np.outer(np.arange(max_pos, dtype=np.float32), inv)
    return O.to_bits(np.concatenate([np.cos(f), np.sin(f)], axis=-1).astype(np.float32), dt)


def _case(rng, seq_lens, hq, hkv, block_size, kv, dt, num_slabs, pad_row=None):
    """Inputs of one decode step: every sequence's new token sits at position seq_len - 1, whose
    slot is inside the sequence's last block."""
    d = 128
    tokens = len(seq_lens)
    n = (hq + 2 * hkv) * d
    x = 16 if kv == "fp8" else 8
    max_len = int(max(seq_lens))
    blocks_per_seq = (max_len + block_size - 1) // block_size
    nb = tokens * blocks_per_seq + 3
    bt = rng.permutation(nb)[:tokens * blocks_per_seq].astype(np.int32).reshape(tokens, blocks_per_seq)
    lens = np.asarray(seq_lens, dtype=np.int32)
    pos = (lens - 1).astype(np.int64)
    slots = np.array([int(bt[i, (l - 1) // block_size]) * block_size + (l - 1) % block_size
                      for i, l in enumerate(lens)], dtype=np.int64)
    if pad_row is not None:
        slots[pad_row] = -1
    cdt = np.uint8 if kv == "fp8" else np.uint16
    if kv == "fp8":
        kc = rng.integers(0, 120, (nb, hkv, d // x, block_size, x)).astype(cdt)
        vc = rng.integers(0, 120, (nb, hkv, d, block_size)).astype(cdt)
    else:
        kc = O.to_bits(rng.uniform(-1, 1, (nb, hkv, d // x, block_size, x)).astype(np.float32), dt)
        vc = O.to_bits(rng.uniform(-1, 1, (nb, hkv, d, block_size)).astype(np.float32), dt)
    if num_slabs:
        slabs = (rng.standard_normal((num_slabs, tokens, n)) * 0.5).astype(np.float32)
        qkv = None
    else:
        slabs = None
        qkv = O.to_bits((rng.standard_normal((tokens, n)) * 0.7).astype(np.float32), dt)
    return dict(d=d, tokens=tokens, n=n, hq=hq, hkv=hkv, bs=block_size, kv=kv, dt=dt, bt=bt, lens=lens,
                pos=pos, slots=slots, kc=kc, vc=vc, slabs=slabs, qkv=qkv, max_len=max_len,
                cache=_cos_sin(d, max(4096, max_len + 1), dt), kv_scale=0.5 if kv == "fp8" else 1.0)


def _run(ops, c, fused, use_v1, seq_order=None, write_kv=True):
    dt, kv = c["dt"], c["kv"]

    def cache_t(a):
        return torch.from_numpy(a.copy()).cuda() if kv == "fp8" else T.from_bits(a, dt, "cuda")

    kc, vc = cache_t(c["kc"]), cache_t(c["vc"])
    if c["slabs"] is not None:
        t_qkv = torch.zeros((c["tokens"], c["n"]), dtype=T.TORCH_DT[dt], device="cuda")
        ws = torch.from_numpy(c["slabs"]).cuda().view(torch.uint8).view(-1)
        sl = ops.SplitKSlabs(ws, c["slabs"].shape[0], c["tokens"], c["n"])
    else:
        t_qkv = T.from_bits(c["qkv"], dt, "cuda")
        sl = None
    hq, hkv, d = c["hq"], c["hkv"], c["d"]
    q, k, v = t_qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    pos, slots = torch.from_numpy(c["pos"]).cuda(), torch.from_numpy(c["slots"]).cuda()
    cos = T.from_bits(c["cache"], dt, "cuda")
    bt, lens = torch.from_numpy(c["bt"]).cuda(), torch.from_numpy(c["lens"]).cuda()
    scale = d ** -0.5
    if fused:
        if not write_kv:      # key and value in a launch of their own, the query untouched
            ops.rotary_reshape_and_cache(pos, q, k, v, d, cos, True, kc, vc, slots, kv, c["kv_scale"], sl,
                                         skip_query=True)
        out = ops.rope_paged_attention(pos, q, k, v, d, cos, True, kc, vc, slots, kv, c["kv_scale"], hkv,
                                       scale, bt, lens, c["max_len"], slabs=sl, seq_order=seq_order,
                                       use_v1=use_v1, write_kv=write_kv)
    else:
        ops.rotary_reshape_and_cache(pos, q, k, v, d, cos, True, kc, vc, slots, kv, c["kv_scale"], sl)
        q3 = q.reshape(c["tokens"], hq, d).contiguous()
        out = torch.empty_like(q3)
        if use_v1:
            ops.paged_attention_v1(out, q3, kc, vc, hkv, scale, bt, lens, c["bs"], c["max_len"], None, kv,
                                   c["kv_scale"], seq_order=seq_order)
        else:
            parts = (c["max_len"] + 511) // 512
            tmp = torch.empty((c["tokens"], hq, parts, d), dtype=out.dtype, device="cuda")
            es = torch.empty((c["tokens"], hq, parts), dtype=torch.float32, device="cuda")
            ops.paged_attention_v2(out, es, torch.empty_like(es), tmp, q3, kc, vc, hkv, scale, bt, lens,
                                   c["bs"], c["max_len"], None, kv, c["kv_scale"], seq_order=seq_order)
        out = out.view(c["tokens"], hq * d)
    torch.cuda.synchronize()
    return T.bits(out), T.bits(kc), T.bits(vc)


# sequence lengths: a single token, lengths inside the first step, at and around the step and the
# partition boundaries (64 tokens per step with a byte cache, 32 with a 16-bit one; 512 per
# partition), and long ones whose last step belongs to each of the four waves in turn
LENS = [1, 2, 16, 17, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 193, 257, 511, 512, 513, 577, 641, 1025, 1500]


@pytest.mark.parametrize("write_kv", [True, False])
@pytest.mark.parametrize("use_v1", [True, False])
@pytest.mark.parametrize("num_slabs", [0, 2])
@pytest.mark.parametrize("kv", ["fp8", "auto"])
@pytest.mark.parametrize("hq,hkv,block_size", [(8, 2, 16), (4, 4, 32), (16, 1, 8), (32, 8, 16)])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
def test_fused_launch_equals_the_separate_launches(ops, write_kv, use_v1, num_slabs, kv, hq, hkv, block_size,
                                                    dt):
    if not ops.rope_paged_attention_supported(hq, hkv, 128, 128, True, block_size, T.TORCH_DT[dt]):
        pytest.skip("shape not taken by the fused launch")
    rng = np.random.default_rng(hq * 31 + hkv * 7 + block_size + num_slabs)
    c = _case(rng, LENS, hq, hkv, block_size, kv, dt, num_slabs, pad_row=4)
    want = _run(ops, c, False, use_v1)
    got = _run(ops, c, True, use_v1, write_kv=write_kv)
    assert np.array_equal(got[1], want[1]), "key cache"
    assert np.array_equal(got[2], want[2]), "value cache"
    # the padding row's token is not cached: both runs attend to whatever the slot held
    assert np.array_equal(got[0], want[0]), "attention output"


def test_fused_launch_with_a_sequence_order(ops):
    rng = np.random.default_rng(11)
    c = _case(rng, [300, 5, 1200, 64, 700, 33], 8, 2, 16, "fp8", O.DT_F16, 2)
    order = ops.make_seq_order(torch.from_numpy(c["lens"]).cuda())
    want = _run(ops, c, False, True)
    got = _run(ops, c, True, True, seq_order=order)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_fused_launch_against_the_oracle(ops):
    """The composition itself once more against the CPU restatements: rotary_embedding, then
    reshape_and_cache, then paged attention over the cache that holds the new token."""
    rng = np.random.default_rng(5)
    dt, kv = O.DT_F16, "fp8"
    c = _case(rng, [70, 1, 200, 513], 8, 2, 16, kv, dt, 0)
    hq, hkv, d = c["hq"], c["hkv"], c["d"]
    got_out, got_kc, got_vc = _run(ops, c, True, True)
    wq, wk = O.rotary_embedding(c["pos"], c["qkv"][:, :hq * d], c["qkv"][:, hq * d:(hq + hkv) * d], d,
                                c["cache"], True, dt)
    wv = np.ascontiguousarray(c["qkv"][:, (hq + hkv) * d:])
    wkc, wvc = c["kc"].copy(), c["vc"].copy()
    O.reshape_and_cache(wk.reshape(-1, hkv, d), wv.reshape(-1, hkv, d), wkc, wvc, c["slots"], dt,
                        O.KV_FP8, c["kv_scale"])
    assert np.array_equal(got_kc, wkc) and np.array_equal(got_vc, wvc)
    want = O.paged_attention_v1(np.ascontiguousarray(wq.reshape(-1, hq, d)), wkc, wvc, c["bt"], c["lens"],
                                d ** -0.5, dt, O.KV_FP8, c["kv_scale"])
    np.testing.assert_allclose(O.from_bits(got_out.reshape(-1, hq, d), dt), O.from_bits(want, dt),
                               atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("kv", ["fp8", "auto"])
def test_decode_step_is_the_same_with_and_without_the_fused_launch(ops, kv, monkeypatch):
    """A whole decode step of a two-layer Llama-3-8B-wide stack (AWQ linears, so the qkv product
    may arrive as split-K slabs): hidden state and caches bit-identical either way."""
    from tests import e2e_gpu as G
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(num_hidden_layers=2, vocab_size=2048)
    model = LlamaDecodeModel(cfg, quant="awq", kv_cache_dtype=kv, device=dev, seed=3)
    G.normalise_linears(model)
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64, 513, 1030], dtype=np.int32)
    num_blocks, block_tables, positions, slot_mapping, input_ids = G.make_inputs(cfg, seq_lens)
    meta = G.make_meta(dev, seq_lens, block_tables, slot_mapping)
    results = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("VLA_FUSE_ROPE_ATTN", fuse)
        kv_caches = G.make_caches(model, ops, num_blocks, kv, dev)
        with torch.inference_mode():
            hidden = model(torch.from_numpy(input_ids).to(dev), torch.from_numpy(positions).to(dev),
                           kv_caches, meta)
        torch.cuda.synchronize()
        results.append((T.bits(hidden), [T.bits(c) for c in kv_caches]))
    assert np.array_equal(results[0][0], results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert np.array_equal(a, b)
