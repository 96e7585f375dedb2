"""GPU parity tests of the fused decode-step kernels: rotary + cache write in one launch, and
add + RMSNorm fed by unreduced split-K slabs.  Expectations come from the oracle's restatements of
the separate reference kernels run back to back (rotary_embedding then reshape_and_cache;
sum-and-round then fused_add_rms_norm); integer/byte results (cache contents, rotated q/k, the
residual sum) are compared bit for bit, the normalised output within one unit of the last place."""
import os
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


@pytest.mark.parametrize("use_slabs", [False, True])
@pytest.mark.parametrize("kv", ["fp8", "auto"])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("head_size,rot_dim,block_size", [(128, 128, 16), (64, 64, 16), (80, 80, 8), (128, 64, 32)])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
def test_rotary_reshape_and_cache(ops, use_slabs, kv, is_neox, head_size, rot_dim, block_size, dt):
    rng = np.random.default_rng(head_size * 7 + rot_dim + block_size)
    tokens, hq, hkv, max_pos, nb = 37, 8, 2, 4096, 64
    n = (hq + 2 * hkv) * head_size
    x = 16 if kv == "fp8" else 8
    kvc = O.KV_FP8 if kv == "fp8" else O.KV_AUTO
    kv_scale = 0.5 if kv == "fp8" else 1.0
    cache = _cos_sin(rot_dim, max_pos, dt)
    pos = rng.integers(0, max_pos, tokens).astype(np.int64)
    slots = rng.permutation(nb * block_size)[:tokens].astype(np.int64)
    slots[3] = -1                                             # a padding token
    if use_slabs:
        slabs = rng.standard_normal((3, tokens, n)).astype(np.float32)
        qkv = O.to_bits(slabs[0] + slabs[1] + slabs[2], dt)   # slab order, then one rounding
    else:
        slabs = None
        qkv = O.to_bits(rng.standard_normal((tokens, n)).astype(np.float32), dt)
    cdt = np.uint8 if kv == "fp8" else np.uint16
    kc0 = rng.integers(0, 120, (nb, hkv, head_size // x, block_size, x)).astype(cdt)
    vc0 = rng.integers(0, 120, (nb, hkv, head_size, block_size)).astype(cdt)

    # oracle: the two reference kernels back to back
    wq, wk = O.rotary_embedding(pos, qkv[:, :hq * head_size],
                                qkv[:, hq * head_size:(hq + hkv) * head_size], head_size, cache,
                                is_neox, dt)
    wv = np.ascontiguousarray(qkv[:, (hq + hkv) * head_size:])
    wkc, wvc = kc0.copy(), vc0.copy()
    O.reshape_and_cache(wk.reshape(tokens, hkv, head_size), wv.reshape(tokens, hkv, head_size), wkc,
                        wvc, slots, dt, kvc, kv_scale)

    def cache_t(a):
        return torch.from_numpy(a).cuda() if kv == "fp8" else T.from_bits(a, dt, "cuda")

    tkc, tvc = cache_t(kc0), cache_t(vc0)
    if use_slabs:
        t_qkv = torch.zeros((tokens, n), dtype=T.TORCH_DT[dt], device="cuda")
        ws = torch.from_numpy(slabs).cuda().view(torch.uint8).view(-1)
        sl = ops.SplitKSlabs(ws, 3, tokens, n)
    else:
        t_qkv = T.from_bits(qkv, dt, "cuda")
        sl = None
    q, k, v = t_qkv.split([hq * head_size, hkv * head_size, hkv * head_size], dim=-1)
    ops.rotary_reshape_and_cache(torch.from_numpy(pos).cuda(), q, k, v, head_size,
                                 T.from_bits(cache, dt, "cuda"), is_neox, tkc, tvc,
                                 torch.from_numpy(slots).cuda(), kv, kv_scale, sl)
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(q), wq)
    assert np.array_equal(T.bits(k), wk)
    assert np.array_equal(T.bits(v), wv)
    assert np.array_equal(T.bits(tkc), wkc)
    assert np.array_equal(T.bits(tvc), wvc)


@pytest.mark.parametrize("num_slabs", [1, 2, 8])
@pytest.mark.parametrize("tokens,hidden", [(7, 768), (256, 4096), (33, 8192)])
@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
def test_fused_add_rms_norm_splitk(ops, num_slabs, tokens, hidden, dt):
    rng = np.random.default_rng(hidden + num_slabs)
    slabs = rng.standard_normal((num_slabs, tokens, hidden)).astype(np.float32)
    acc = slabs[0].copy()
    for p in range(1, num_slabs):
        acc += slabs[p]
    x = O.to_bits(acc, dt)
    r = O.to_bits(rng.standard_normal((tokens, hidden)).astype(np.float32), dt)
    w = O.to_bits((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32), dt)
    want_out, want_res = O.fused_add_rms_norm(x, r, w, 1e-5, dt)
    tr, tw = T.from_bits(r, dt, "cuda"), T.from_bits(w, dt, "cuda")
    out = torch.empty_like(tr)
    sl = ops.SplitKSlabs(torch.from_numpy(slabs).cuda().view(torch.uint8).view(-1), num_slabs, tokens,
                         hidden)
    ops.fused_add_rms_norm_splitk(sl, out, tr, tw, 1e-5)
    torch.cuda.synchronize()
    assert np.array_equal(T.bits(tr), want_res)
    ulp = 2.0**-10 if dt == O.DT_F16 else 2.0**-7
    np.testing.assert_allclose(out.float().cpu().numpy(), O.from_bits(want_out, dt), rtol=2 * ulp,
                               atol=1e-6)


@pytest.mark.parametrize("k,n", [(4096, 4096), (4096, 6144), (14336, 4096), (512, 256)])
def test_partial_gemm_slabs_reduce_to_the_full_result(ops, k, n):
    g = torch.Generator(device="cuda").manual_seed(k + n)
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (k // 128, n // 8), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    sc = (torch.empty((k // 128, n), device="cuda").uniform_(0.5, 1.5, generator=g) * 1e-2).half()
    x = torch.empty((256, k), device="cuda").uniform_(-1, 1, generator=g).half()
    pw, pz = ops.awq_repack(qw, qz)
    full = ops.q4_gemm_packed(x, pw, pz, sc, 0)
    part = ops.q4_gemm_packed_partial(x, pw, pz, sc, 0)
    got = part.reduce() if isinstance(part, ops.SplitKSlabs) else part
    torch.cuda.synchronize()
    assert torch.equal(got, full)


def _run_llama_tp(tmp_path, world, xgmi, fuse_norm=True):
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / f"llama_tp_{int(xgmi)}{int(fuse_norm)}.npz")
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(here, "llama_tp_worker.py"), str(r), str(world),
                               str(port), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, VLA_XGMI_ALLREDUCE="1" if xgmi else "0",
                                       VLA_XGMI_FUSE_NORM="1" if fuse_norm else "0",
                                       HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(world)]
    try:
        res = [p.communicate(timeout=150) + (p.returncode, ) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for so, se, rc in res:
        assert rc == 0, se[-3000:]
    return np.load(out)


def test_tensor_parallel_step_through_the_peer_memory_all_reduce(tmp_path):
    """Two ranks on one GPU, one decode step of a two-layer stack, three ways.  (a) Gloo: the
    row-parallel linears reduce their split-K slabs, the sum is staged through the host.  (b) the
    peer-memory all-reduce fed with the unreduced slabs: each rank's contribution is rounded to
    fp16 once either way and a two-rank sum has a single rounding, so the logits equal (a) bit for
    bit.  (c) the all-reduce also doing the add + RMSNorm that follows it: the sum of squares is
    accumulated in another order, so normalised values may move in the last place and the logits
    agree to a few fp16 units."""
    a = _run_llama_tp(tmp_path, 2, xgmi=False)
    b = _run_llama_tp(tmp_path, 2, xgmi=True, fuse_norm=False)
    c = _run_llama_tp(tmp_path, 2, xgmi=True, fuse_norm=True)
    assert int(b["slabs"]) > 0 and int(b["norm"]) == 0 and int(c["norm"]) == 4
    assert np.isfinite(a["logits"]).all() and float(np.abs(a["logits"]).max()) > 0.1
    assert np.array_equal(a["logits"], b["logits"])
    np.testing.assert_allclose(c["logits"], a["logits"], atol=2e-2, rtol=1e-2)
    assert float(np.mean(c["logits"] != a["logits"])) < 0.5


@pytest.mark.parametrize("dt", [O.DT_F16, O.DT_BF16])
@pytest.mark.parametrize("tokens,d,ns", [(1, 64, 2), (37, 448, 3), (256, 3584, 4), (64, 14336, 2)])
def test_silu_and_mul_fed_by_slabs(ops, dt, tokens, d, ns):
    """SiLU·mul on unreduced split-K slabs against reduce-then-activate: the same sums, the same
    rounding, the same arithmetic -- equal bits; and against the oracle's activation."""
    from vllm_legacy_amd._custom_ops import SplitKSlabs
    g = torch.Generator(device="cuda").manual_seed(tokens + d)
    ws = torch.empty((ns, tokens, 2 * d), dtype=torch.float32, device="cuda").uniform_(-2, 2, generator=g)
    slabs = SplitKSlabs(ws.view(-1).view(torch.uint8), ns, tokens, 2 * d)
    tdt = torch.float16 if dt == O.DT_F16 else torch.bfloat16
    got = torch.empty((tokens, d), dtype=tdt, device="cuda")
    ops.silu_and_mul_splitk(slabs, got)
    reduced = slabs.reduce(tdt).contiguous()
    want = torch.empty_like(got)
    ops.silu_and_mul(want, reduced)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    ref = O.silu_and_mul(T.bits(reduced), dt)
    np.testing.assert_allclose(got.float().cpu().numpy(), O.from_bits(ref, dt), rtol=2.0**-7 if dt == O.DT_BF16 else 2.0**-10,
                               atol=1e-6)
