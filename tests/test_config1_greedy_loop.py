"""Configuration 1 as a loop: OPT-125M's attention shape (12 heads of 64, bfloat16, block 16, one
sequence) through a prompt and 16 greedy decode steps in which step t's cache write feeds step
t + 1's attention.

The fixture tests/golden/loop_bf16_opt125m.npz was produced by the reference's CPU executor path
(vllm/attention/backends/torch_sdpa.py:159-228: reshape_and_cache and paged_attention_v1 from
csrc/cpu, torch's scaled_dot_product_attention for the prompt) inside a one-layer stand-in model
that closes the greedy loop (tests/golden/make_golden.py::greedy_loop_case).  It stores every
step's token, q / k / v, attention output and the final cache.

Checked here:
 * the oracle run through the same loop (CPU test),
 * Gfx950PagedAttentionBackend fed the fixture's q / k / v step by step (the cache is the one the
   earlier GPU steps wrote): every step's attention output, the sampled token, the final cache;
 * the same loop free-running on the GPU (its own tokens -> its own q / k / v): the greedy token
   sequence.

Tolerance: the cache is byte work, exact.  Attention outputs are bfloat16.  The reference's CPU
kernel keeps probabilities in fp32; the GPU kernel and the oracle (as the reference's CUDA kernel,
csrc/attention/attention_kernels.cu:398-431) round them to bfloat16 before P.V, a relative error of
at most 2^-9 per probability, so an output moves by at most 2^-9 * max|v| (the probabilities sum to
one), plus the rounding of the output itself (one bfloat16 ulp, rtol 2^-8).  The values of this
loop reach |v| ~ 4.5: atol = 2^-9 * max|v| ~ 9e-3 on outputs of magnitude up to 1.2.  Tokens must
be identical.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

FIX = os.path.join(os.path.dirname(__file__), "golden", "loop_bf16_opt125m.npz")
HEADS, D, BS, HIDDEN, X = 12, 64, 16, 768, 8
DT = O.DT_BF16
RTOL = 2.0**-8


def _atol(d):
    vmax = max(float(np.abs(_f(d["prompt_v"])).max()), float(np.abs(_f(d["step_v"])).max()))
    return 2.0**-9 * vmax


def _f(bits):
    return O.from_bits(np.ascontiguousarray(bits), DT)


def _check_out(d, got_bits, want_bits, what):
    np.testing.assert_allclose(_f(got_bits), _f(want_bits), atol=_atol(d), rtol=RTOL, err_msg=what)


def test_oracle_reproduces_the_reference_cpu_loop():
    d = np.load(FIX)
    scale = float(d["scale"])
    bt = d["block_table"].reshape(1, -1)
    nb = d["key_cache_final"].shape[0]
    kc = np.zeros((nb, HEADS, D // X, BS, X), dtype=np.uint16)
    vc = np.zeros((nb, HEADS, D, BS), dtype=np.uint16)
    w_out = _f(d["w_out"])
    p = len(d["prompt"])
    O.reshape_and_cache(d["prompt_k"].reshape(p, HEADS, D), d["prompt_v"].reshape(p, HEADS, D), kc, vc,
                        d["prompt_slots"], DT)
    out = O.prefill_attention(d["prompt_q"].reshape(p, HEADS, D), d["prompt_k"].reshape(p, HEADS, D),
                              d["prompt_v"].reshape(p, HEADS, D), [p], scale, DT)
    _check_out(d, out.reshape(p, HIDDEN), d["prompt_out"], "prompt")
    assert int(np.argmax(_f(out.reshape(p, HIDDEN))[-1] @ w_out)) == int(d["tokens"][0])
    for t in range(len(d["step_q"])):
        pos = p + t
        slot = int(bt[0, pos // BS]) * BS + pos % BS
        O.reshape_and_cache(d["step_k"][t].reshape(1, HEADS, D), d["step_v"][t].reshape(1, HEADS, D),
                            kc, vc, np.array([slot]), DT)
        o = O.paged_attention_v1(np.ascontiguousarray(d["step_q"][t].reshape(1, HEADS, D)), kc, vc, bt,
                                 np.array([pos + 1], dtype=np.int32), scale, DT)
        _check_out(d, o.reshape(HIDDEN), d["step_out"][t], f"step {t}")
        assert int(np.argmax(_f(o.reshape(HIDDEN)) @ w_out)) == int(d["tokens"][t + 1]), t
    assert np.array_equal(kc, d["key_cache_final"]) and np.array_equal(vc, d["value_cache_final"])


def _bf(bits, dev):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(dev).view(torch.bfloat16)


def _bits(t):
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def _meta(dev, *, prompt_len=0, pos=None, slots, block_table):
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    if prompt_len:
        return Gfx950PagedAttentionMetadata(
            num_prefills=1, num_prefill_tokens=prompt_len, num_decode_tokens=0, slot_mapping=slots,
            seq_lens=[prompt_len], seq_lens_tensor=torch.tensor([prompt_len], dtype=torch.int32, device=dev),
            max_query_len=prompt_len, max_prefill_seq_len=prompt_len, max_decode_seq_len=0,
            query_start_loc=torch.tensor([0, prompt_len], dtype=torch.int32, device=dev),
            seq_start_loc=torch.tensor([0, prompt_len], dtype=torch.int32, device=dev),
            use_cuda_graph=False, context_lens_tensor=torch.zeros(1, dtype=torch.int32, device=dev),
            block_tables=torch.empty((1, 0), dtype=torch.int32, device=dev))
    return Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=1, slot_mapping=slots, seq_lens=None,
        seq_lens_tensor=torch.tensor([pos + 1], dtype=torch.int32, device=dev), max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=pos + 1, query_start_loc=None, seq_start_loc=None,
        use_cuda_graph=False, context_lens_tensor=None, block_tables=block_table)


@pytest.mark.gpu
@pytest.mark.parametrize("free_running", [False, True])
def test_backend_runs_the_reference_cpu_loop(free_running):
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionBackend
    dev = torch.device("cuda", 0)
    d = np.load(FIX)
    scale = float(d["scale"])
    nb = d["key_cache_final"].shape[0]
    p = len(d["prompt"])
    steps = len(d["step_q"])
    impl = Gfx950PagedAttentionBackend.get_impl_cls()(HEADS, D, scale, HEADS, None, None, "auto", None)
    kv_cache = torch.zeros(Gfx950PagedAttentionBackend.get_kv_cache_shape(nb, BS, HEADS, D),
                           dtype=torch.bfloat16, device=dev)
    bt = torch.from_numpy(d["block_table"].reshape(1, -1).astype(np.int32)).to(dev)
    embed, w_qkv, w_out = _bf(d["embed"], dev), _bf(d["w_qkv"], dev), _bf(d["w_out"], dev)

    def qkv_of(tokens):
        h = embed[torch.as_tensor(tokens, device=dev)]
        return [t.contiguous() for t in (h.float() @ w_qkv.float()).to(torch.bfloat16).split(HIDDEN, dim=-1)]

    def next_token(row):
        return int(torch.argmax(row.float() @ w_out.float()).item())

    slot = lambda pos: int(d["block_table"][pos // BS]) * BS + pos % BS          # noqa: E731
    if free_running:
        q, k, v = qkv_of(d["prompt"])
    else:
        q, k, v = (_bf(d[n], dev) for n in ("prompt_q", "prompt_k", "prompt_v"))
    slots = torch.from_numpy(d["prompt_slots"]).to(dev)
    out = impl.forward(q, k, v, kv_cache, _meta(dev, prompt_len=p, slots=slots, block_table=bt))
    torch.cuda.synchronize()
    _check_out(d, _bits(out), d["prompt_out"], "prompt")
    tokens = [next_token(out[-1])]
    for t in range(steps):
        pos = p + t
        if free_running:
            q, k, v = qkv_of([tokens[-1]])
        else:
            q, k, v = (_bf(d[n][t:t + 1], dev) for n in ("step_q", "step_k", "step_v"))
        out = impl.forward(q, k, v, kv_cache,
                           _meta(dev, pos=pos, slots=torch.tensor([slot(pos)], dtype=torch.int64, device=dev),
                                 block_table=bt))
        torch.cuda.synchronize()
        _check_out(d, _bits(out).reshape(HIDDEN), d["step_out"][t], f"step {t}")
        tokens.append(next_token(out[0]))
    assert tokens == [int(x) for x in d["tokens"]], (tokens, d["tokens"].tolist())
    if not free_running:
        # the cache holds exactly what the reference's reshape_and_cache wrote, step by step
        from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
        kc, vc = PagedAttention.split_kv_cache(kv_cache, HEADS, D)
        assert np.array_equal(_bits(kc), d["key_cache_final"])
        assert np.array_equal(_bits(vc), d["value_cache_final"])
