"""Test infrastructure shared by the end-to-end GPU parity tests: building inputs and caches for
one decode step of vllm_legacy_amd.model.llama_decode.LlamaDecodeModel, turning the model into the
oracle's dense-weight stack (tests/e2e_oracle.py), and the comparison with its fixed bars
(explained in tests/test_gpu_e2e_logits.py)."""
import numpy as np
import torch

from oracle import oracle as O
from tests import e2e_oracle as E
from tests import pa_testlib as T

DT = O.DT_F16

# Fixed bars (none of them is computed from the data).  The normed hidden state is O(1) and reaches
# |h| = 4.7 in these tests; an fp16 ulp is 2^-8 there: 2^-6 is four ulps of the largest values.
HIDDEN_ATOL = 2.0**-6
HIDDEN_REL_L2 = 1.5e-3
LOGIT32_ATOL = 2.0**-7
FP16_OUT_RTOL, FP16_OUT_ATOL = 2.0**-11 * 1.01, 2.0**-14


def normalise_linears(model, seed=17):
    g = torch.Generator(device=model.lm_head.device).manual_seed(seed)
    for layer in model.layers:
        for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
            k = lin.qweight.shape[0]
            s = torch.empty(lin.scales.shape, dtype=torch.float32, device=lin.scales.device)
            s.uniform_(0.5, 1.5, generator=g)
            lin.scales.data = (s * (k**-0.5 / 42.5**0.5)).half()


def oracle_stack(model, deq):
    cfg = model.cfg
    layers = [E.OracleLayer(T.bits(l.input_layernorm.weight), T.bits(l.post_attention_layernorm.weight),
                            deq(l.qkv_proj), deq(l.o_proj), deq(l.gate_up_proj), deq(l.down_proj),
                            float(l.attn._kv_scale)) for l in model.layers]
    return E.OracleStack(T.bits(model.embed_tokens), T.bits(model.lm_head), T.bits(model.norm.weight),
                         layers, T.bits(model.rotary_emb.cos_sin_cache), model.layers[0].hq,
                         model.layers[0].hkv, cfg.head_dim, cfg.rms_norm_eps)


def awq_dense(lin):
    return O.awq_dequantize(T.bits(lin.qweight), T.bits(lin.scales), T.bits(lin.qzeros))


def make_inputs(cfg, seq_lens, bs=16, seed=11):
    B = len(seq_lens)
    max_blocks = (int(seq_lens.max()) + bs - 1) // bs
    num_blocks = B * max_blocks + 3
    rng = np.random.default_rng(seed)
    block_tables = rng.permutation(num_blocks)[:B * max_blocks].reshape(B, max_blocks).astype(np.int32)
    positions = (seq_lens - 1).astype(np.int64)
    slot_mapping = np.array([block_tables[b, p // bs] * bs + p % bs for b, p in enumerate(positions)],
                            dtype=np.int64)
    input_ids = rng.integers(0, cfg.vocab_size, B).astype(np.int64)
    return num_blocks, block_tables, positions, slot_mapping, input_ids


def make_meta(dev, seq_lens, block_tables, slot_mapping):
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    return Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=len(seq_lens),
        slot_mapping=torch.from_numpy(slot_mapping).to(dev), seq_lens=None,
        seq_lens_tensor=torch.from_numpy(seq_lens).to(dev), max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=int(seq_lens.max()), query_start_loc=None,
        seq_start_loc=None, use_cuda_graph=False, context_lens_tensor=None,
        block_tables=torch.from_numpy(block_tables).to(dev))


def make_caches(model, ops, num_blocks, kv, dev, bs=16, seed=5, amplitude=1.0):
    shape = model.kv_cache_shape(num_blocks, bs)
    g = torch.Generator(device=dev).manual_seed(seed)
    caches = []
    for layer in model.layers:
        src = torch.empty(shape, dtype=torch.float16, device=dev).uniform_(-amplitude, amplitude, generator=g)
        if kv == "fp8":
            c = torch.empty(shape, dtype=torch.uint8, device=dev)
            ops.convert_fp8(c, src, float(layer.attn._kv_scale))
        else:
            c = src
        caches.append(c)
    return caches


def trace_hooks(model):
    """Forward hooks that keep what the linears and the attention of every layer returned (split-K
    slabs are summed the way their fused consumer sums them)."""
    from vllm_legacy_amd import _custom_ops as ops
    got, handles = {}, []

    def keep(name):
        def hook(_m, _inp, out):
            if isinstance(out, ops.SplitKSlabs):
                out = out.reduce(torch.float16)
            got[name] = T.bits(out.detach().clone().contiguous())
        return hook

    for li, layer in enumerate(model.layers):
        for mod, tag in ((layer.attn, "attn"), (layer.o_proj, "o"), (layer.down_proj, "down")):
            handles.append(mod.register_forward_hook(keep(f"{li}.{tag}")))
    return got, handles


def compare_step(model, hidden, logits16, want: E.StepResult, label, got_trace=None,
                 hidden_atol=HIDDEN_ATOL, hidden_rel_l2=HIDDEN_REL_L2, logit32_atol=LOGIT32_ATOL,
                 min_clear=0.75):
    """The bars of tests/test_gpu_e2e_logits.py's docstring; returns the measured errors."""
    h_got = O.from_bits(T.bits(hidden), DT).astype(np.float64)
    h_want = O.from_bits(want.hidden, DT).astype(np.float64)
    head = O.from_bits(T.bits(model.lm_head), DT).astype(np.float64)
    l_got, l_want = h_got @ head.T, h_want @ head.T
    h_err = float(np.abs(h_got - h_want).max())
    h_rel = float(np.linalg.norm(h_got - h_want) / np.linalg.norm(h_want))
    l_err = float(np.abs(l_got - l_want).max())
    out_q = float(np.abs(logits16 - l_got).max())
    print(f"{label}: hidden max|err| {h_err:.3e} (|h| <= {np.abs(h_want).max():.2f}), rel L2 {h_rel:.3e}; "
          f"fp32 logits max|err| {l_err:.3e} (|logit| <= {np.abs(l_want).max():.2f}, rms "
          f"{np.sqrt(np.mean(l_want**2)):.2f}); fp16 output rounding {out_q:.3e}")
    if got_trace:
        for name in sorted(got_trace, key=lambda n: (int(n.split('.')[0]), n)):
            if name not in want.trace:
                continue
            a = O.from_bits(got_trace[name], DT).astype(np.float64)
            b = O.from_bits(want.trace[name], DT).astype(np.float64)
            print(f"    drift at {name:8s} max|err| {np.abs(a - b).max():.3e}  rel L2 "
                  f"{np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30):.3e}  "
                  f"elements that differ {np.mean(a != b):.3f}")
    assert np.isfinite(logits16).all()
    assert h_err <= hidden_atol, (label, "hidden", h_err)
    assert h_rel <= hidden_rel_l2, (label, "hidden rel L2", h_rel)
    assert l_err <= logit32_atol, (label, "fp32 logits", l_err)
    np.testing.assert_allclose(logits16, l_got, rtol=FP16_OUT_RTOL, atol=FP16_OUT_ATOL)
    # greedy choice: identical wherever the expected winner leads by more than the logit bars
    # allow two candidates to move towards each other; a closer race is a tie within the bars
    top2 = np.sort(l_want, axis=-1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * (logit32_atol + FP16_OUT_RTOL * np.abs(top2[:, 1]))
    assert clear.mean() >= min_clear, "too many near-ties for the argmax check to mean anything"
    assert (logits16.argmax(-1)[clear] == l_want.argmax(-1)[clear]).all()
    return h_err, h_rel, l_err, out_q


def check_written_cache(kv_caches, want: E.StepResult, hkv, d, kv, kv_scales=None, drift_atol=4e-3,
                        max_frac=0.1):
    """The cache write is index / byte work, tested bit for bit elsewhere on identical inputs.
    Here its inputs already differ in the last place (GPU rsqrt against libm), so: nothing but
    the new slots may change, and a new element may differ by one step of the cache format."""
    from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
    for li, cache in enumerate(kv_caches):
        kc, vc = want.caches[li]
        k, v = want.new_kv[li]
        got_k, got_v = PagedAttention.split_kv_cache(cache, hkv, d)
        for got, want_c, new in ((T.bits(got_k), kc, k), (T.bits(got_v), vc, v)):
            diff = got != want_c
            assert int(diff.sum()) <= new.size, (li, int(diff.sum()))   # only the new slots
            if kv == "fp8":
                # a last-place difference of an fp16 input only shows where it crosses an fp8
                # rounding boundary: few elements, and then by one fp8 step (1/8 relative; near
                # zero by a few subnormal steps of 2^-9)
                frac = float(diff.sum()) / new.size
                print(f"layer {li}: {frac:.4f} of the new fp8 cache bytes differ")
                assert frac <= max_frac, (li, frac)
                sc = 1.0 if kv_scales is None else float(kv_scales[li])
                gd = O.from_bits(O.convert_fp8(got, DT, False, sc), DT)
                wd = O.from_bits(O.convert_fp8(want_c, DT, False, sc), DT)
                np.testing.assert_allclose(gd, wd, rtol=0.13, atol=max(3 * 2.0**-9, drift_atol))
            else:
                # (a rotated key is a difference of two products of the magnitude of k: where it
                # cancels, the error is an ulp of the products, not of the result)
                np.testing.assert_allclose(O.from_bits(got, DT), O.from_bits(want_c, DT),
                                           rtol=2.0**-8, atol=drift_atol)


def capture_hooks(model):
    """Hooks that keep, for every layer, what went into and came out of each step of the real
    forward pass (norms, the four linears, attention after the fused rotary + cache write), as
    fp16 bits; split-K slabs are summed the way their fused consumer sums them."""
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.distributed.communication_op import PendingAllReduce
    cap, handles = {}, []

    def val(x):
        if isinstance(x, PendingAllReduce):
            x = x.value
        if isinstance(x, ops.SplitKSlabs):
            x = x.reduce(torch.float16)
        return T.bits(x.detach().clone().contiguous()) if isinstance(x, torch.Tensor) else x

    def pre(name):
        def hook(_m, args, kwargs):
            cap[name + ".in"] = [val(a) for a in args if isinstance(a, (torch.Tensor, ops.SplitKSlabs, PendingAllReduce))]
        return hook

    def post(name):
        def hook(_m, _args, out):
            cap[name + ".out"] = [val(o) for o in out] if isinstance(out, tuple) else [val(out)]
        return hook

    for li, layer in enumerate(model.layers):
        for mod, tag in ((layer.input_layernorm, "ln1"), (layer.qkv_proj, "qkv"), (layer.attn, "attn"),
                         (layer.o_proj, "o"), (layer.post_attention_layernorm, "ln2"),
                         (layer.down_proj, "down")):
            handles.append(mod.register_forward_pre_hook(pre(f"{li}.{tag}"), with_kwargs=True))
            handles.append(mod.register_forward_hook(post(f"{li}.{tag}")))
    handles.append(model.norm.register_forward_pre_hook(pre("final.ln"), with_kwargs=True))
    handles.append(model.norm.register_forward_hook(post("final.ln")))
    return cap, handles


def op_local_table(cap, st: E.OracleStack, positions, block_tables, seq_lens, gpu_caches_after, kv):
    """Every step of the GPU's forward pass against the oracle's restatement of the same step ON
    THE GPU'S OWN INPUT: what each op contributes by itself, free of what it inherited.  Returns
    {step: (max ulps, fraction of elements that differ, max |err|)} and prints the table."""
    kvc = O.KV_FP8 if kv == "fp8" else O.KV_AUTO
    x = 16 if kv == "fp8" else 8
    hq, hkv, d = st.hq, st.hkv, st.d
    rows = {}

    def put(name, got, want):
        """max error in units of 2^-10 (an fp16 ulp is 2^-11 .. 2^-10 of its value) of the
        expected value or, for values that are small because their terms cancelled, of the rms of
        their row; fraction of elements that differ at all; max |err|."""
        g = O.from_bits(got, DT).astype(np.float64).reshape(got.shape[0], -1)
        w = O.from_bits(want, DT).astype(np.float64).reshape(got.shape[0], -1)
        err = np.abs(g - w)
        scale = np.maximum(np.abs(w), np.sqrt(np.mean(w * w, axis=1, keepdims=True)))
        rows[name] = (float((err / (2.0**-10 * np.maximum(scale, 1e-30))).max()),
                      float(np.mean(got.reshape(-1) != want.reshape(-1))), float(err.max()))

    def norm(name, weight):
        i, o = cap[name + ".in"], cap[name + ".out"]
        if len(i) == 1:
            put(name, o[0], O.rms_norm(i[0], weight, st.eps, DT))
        else:
            want_h, want_r = O.fused_add_rms_norm(i[0], i[1], weight, st.eps, DT)
            put(name, o[0], want_h)
            put(name + " (residual)", o[1], want_r)

    for li, L in enumerate(st.layers):
        norm(f"{li}.ln1", L.ln1)
        h = cap[f"{li}.qkv.in"][0]
        qkv = cap[f"{li}.qkv.out"][0]
        put(f"{li}.qkv gemm", qkv, O.gemm_f16(h, L.qkv))
        B = qkv.shape[0]
        q = np.ascontiguousarray(qkv[:, :hq * d])
        k = np.ascontiguousarray(qkv[:, hq * d:(hq + hkv) * d])
        q_rot, k_rot = O.rotary_embedding(positions, q, k, d, st.cos_sin, True, DT)
        a_in = cap[f"{li}.attn.in"]
        put(f"{li}.rotary q", np.ascontiguousarray(a_in[0]), q_rot)
        put(f"{li}.rotary k", np.ascontiguousarray(a_in[1]), k_rot)
        cache = gpu_caches_after[li]
        nb = cache.shape[1]
        kc = np.ascontiguousarray(cache[0]).reshape(nb, hkv, d // x, 16, x)
        vc = np.ascontiguousarray(cache[1]).reshape(nb, hkv, d, 16)
        want = O.paged_attention_v1(np.ascontiguousarray(a_in[0]).reshape(B, hq, d), kc, vc, block_tables,
                                    seq_lens, d**-0.5, DT, kvc, L.kv_scale)
        put(f"{li}.attention", cap[f"{li}.attn.out"][0].reshape(B, hq * d), want.reshape(B, hq * d))
        put(f"{li}.o gemm", cap[f"{li}.o.out"][0], O.gemm_f16(cap[f"{li}.o.in"][0], L.o))
        norm(f"{li}.ln2", L.ln2)
        act = cap[f"{li}.down.in"][0]
        put(f"{li}.gate_up gemm + silu", act,
            O.silu_and_mul(O.gemm_f16(cap[f"{li}.ln2.out"][0], L.gate_up), DT))
        put(f"{li}.down gemm", cap[f"{li}.down.out"][0], O.gemm_f16(act, L.down))
    norm("final.ln", st.norm)
    print("    step on the GPU's own input        max err / 2^-10 of scale   differ    max|err|")
    for name, (u, frac, err) in rows.items():
        print(f"    {name:34s} {u:24.2f} {frac:8.4f} {err:11.3e}")
    return rows
