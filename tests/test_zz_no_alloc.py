"""GPU test of vla_paged_attention_v2_noalloc: the V2 decode attention whose scratch tensors live
in a static device array inside the library, for callers that cannot allocate.

Checked through the C-ABI against vla_paged_attention_v2 (caller-owned scratch) bit for bit, on two
calls in a row, the second with a smaller input (so stale contents of the shared array from the
first call are in place when the second runs), with the scratch poisoned in between (a call whose
query is all NaN leaves NaN in every partial output, sum and maximum it writes), for a first call
that arrives under stream capture in a fresh process, and for the size limit's error.
"""
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


def _v2(ops, c):
    return T.run_hip_attention(ops, c, "v2")


def _v2_noalloc(ops, c):
    out = torch.empty_like(c.t_q)
    kvd = "fp8" if c.kv == O.KV_FP8 else "auto"
    ops.paged_attention_v2_noalloc(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl,
                                   c.bs, c.max_seq_len, c.t_alibi, kvd, 1.0)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("kv_fp8", [True, False])
def test_noalloc_equals_v2_on_two_calls_second_smaller(ops, kv_fp8):
    big = T.make_attention_case(41, 9, (32, 8), 128, 16, O.DT_F16, kv_fp8, True, max_seq_len=3000,
                                num_blocks=512)
    small = T.make_attention_case(42, 3, (32, 8), 128, 16, O.DT_F16, kv_fp8, False,
                                  max_seq_len=700, num_blocks=128)
    want_big, want_small = _v2(ops, big), _v2(ops, small)
    before = torch.cuda.memory_allocated()
    got_big = _v2_noalloc(ops, big)
    got_small = _v2_noalloc(ops, small)
    assert torch.cuda.memory_allocated() - before == got_big.nbytes + got_small.nbytes
    assert torch.equal(got_big.view(torch.int16), want_big.view(torch.int16))
    assert torch.equal(got_small.view(torch.int16), want_small.view(torch.int16))
    # and against the oracle, to the tolerance of the V2 parity tests
    tol = 1e-2 if kv_fp8 else 1e-3
    np.testing.assert_allclose(got_small.float().cpu().numpy(),
                               T.to_f32(T.oracle_attention(small, "v2"), O.DT_F16), atol=tol,
                               rtol=1e-5)
    # once more in the other order: small, then big
    assert torch.equal(_v2_noalloc(ops, small).view(torch.int16), want_small.view(torch.int16))
    assert torch.equal(_v2_noalloc(ops, big).view(torch.int16), want_big.view(torch.int16))
    # poison the shared array: the big problem with a NaN query fills every partial output, sum
    # and maximum of its (larger) footprint with NaN; the next calls must not see any of it
    q_keep = big.t_q.clone()
    big.t_q.fill_(float("nan"))
    assert torch.isnan(_v2_noalloc(ops, big)).all()
    big.t_q.copy_(q_keep)
    assert torch.equal(_v2_noalloc(ops, small).view(torch.int16), want_small.view(torch.int16))
    assert torch.equal(_v2_noalloc(ops, big).view(torch.int16), want_big.view(torch.int16))


def test_noalloc_other_dtypes_and_generic_kernel(ops):
    # bf16 on the MFMA kernel, fp32 with block size 8 on the generic kernel
    for dt, hs, bs in ((O.DT_BF16, 96, 32), (O.DT_F32, 64, 8)):
        c = T.make_attention_case(43, 4, (8, 2), hs, bs, dt, False, False, max_seq_len=1100,
                                  num_blocks=160)
        want, got = _v2(ops, c), _v2_noalloc(ops, c)
        assert torch.equal(got.view(torch.int32) if dt == O.DT_F32 else got.view(torch.int16),
                           want.view(torch.int32) if dt == O.DT_F32 else want.view(torch.int16))


def test_noalloc_under_graph_capture(ops):
    c = T.make_attention_case(44, 5, (32, 8), 128, 16, O.DT_F16, True, False, max_seq_len=1300,
                              num_blocks=256)
    want = _v2(ops, c)
    out = torch.empty_like(c.t_q)
    ops.paged_attention_v2_noalloc(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl,
                                   c.bs, c.max_seq_len, None, "fp8", 1.0)   # first call: address
    torch.cuda.synchronize()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.paged_attention_v2_noalloc(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt,
                                       c.t_sl, c.bs, c.max_seq_len, None, "fp8", 1.0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


def test_noalloc_rejects_problems_beyond_its_scratch(ops):
    from vllm_legacy_amd import _lib
    # 64 sequences x 32 heads x 64 partitions x 128 x 2 bytes = 32 MiB of tmp_out alone
    s, hq, d, max_len = 64, 32, 128, 64 * 512
    need = s * hq * 64 * (d * 2 + 8)
    assert need > _lib.PA_V2_SCRATCH_BYTES
    q = torch.zeros((s, hq, d), dtype=torch.float16, device="cuda")
    kc = torch.zeros((4, 8, d // 8, 16, 8), dtype=torch.float16, device="cuda")
    vc = torch.zeros((4, 8, d, 16), dtype=torch.float16, device="cuda")
    bt = torch.zeros((s, max_len // 16), dtype=torch.int32, device="cuda")
    sl = torch.ones((s, ), dtype=torch.int32, device="cuda")
    out = torch.full_like(q, 7.0)
    with pytest.raises(RuntimeError, match="static scratch"):
        ops.paged_attention_v2_noalloc(out, q, kc, vc, 8, 0.1, bt, sl, 16, max_len, None, "auto",
                                       1.0)
    torch.cuda.synchronize()
    assert (out == 7.0).all()      # nothing was launched


_FRESH = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
from tests import pa_testlib as T
from vllm_legacy_amd import _custom_ops as ops
c = T.make_attention_case(45, 4, (32, 8), 128, 16, O.DT_F16, True, False, max_seq_len=1500, num_blocks=256)
want = T.run_hip_attention(ops, c, "v2")
out = torch.zeros_like(c.t_q)
def call():
    ops.paged_attention_v2_noalloc(out, c.t_q, c.t_kc, c.t_vc, c.hkv, c.scale, c.t_bt, c.t_sl, c.bs,
                                   c.max_seq_len, c.t_alibi, "fp8", 1.0)
mode = sys.argv[2]
if mode == "prepared":
    ops.paged_attention_v2_noalloc_prepare()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
refused = False
with torch.cuda.stream(s):
    try:
        with torch.cuda.graph(g, stream=s):
            call()                  # the first call of this process on this device
    except RuntimeError as e:
        refused = "prepare" in str(e)
        if not refused:
            raise
torch.cuda.synchronize()
if mode == "prepared":
    assert not refused
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
else:
    assert refused, "an unprepared first call under capture must be refused"
    # the refusal launched nothing and left the library usable
    with torch.cuda.stream(torch.cuda.Stream()):
        call()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
print("OK")
"""


@pytest.mark.parametrize("mode", ["prepared", "unprepared"])
def test_first_call_of_a_process_under_capture(mode):
    """A fresh process whose first noalloc call comes from inside a stream capture: after
    paged_attention_v2_noalloc_prepare() the captured call replays V2's bits; without it the call
    is refused with a message naming the prepare call (the address lookup is a HIP API call that
    must stay out of a capture) and nothing is launched."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FRESH, root, mode], capture_output=True, text=True,
                       timeout=150)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
