"""From a checkpoint to logits on the GPU: AWQ / GPTQ safetensors files with Hugging Face names are
loaded through LlamaDecodeModel.load_weights (the role of LlamaForCausalLM.load_weights,
vllm/model_executor/models/llama.py:411-469: q/k/v and gate/up stacked, every linear cut to the
rank's shard) and kv_cache_scales.json through load_kv_cache_scales (llama.py:474-493,
model_loader/weight_utils.py:365), and one decode step runs on what was loaded.

 * one rank: final hidden state and logits against the oracle pipeline on the dense matrices the
   checkpoint must dequantise to -- for GPTQ the reference packer's own `w_ref` (tests/ckpt_lib.py);
 * two tensor-parallel ranks (two processes on one GPU, all-reduce staged through Gloo) from the
   same file: the unsharded result within the bars below.  Not bit-identical: a row-parallel linear's
   two partial products are each rounded to fp16 and then added, where one rank rounds one fp32 sum;
 * the per-layer factors of the reference's own tests/fp8_kv/llama2-7b-fp8-kv/kv_cache_scales.json
   (kept as data under tests/golden/fp8_kv/) land on the attention layers, change the bytes the
   cache write stores and the attention output exactly as the oracle says, and the whole 32-layer
   step with them matches the oracle pipeline.

Bars: tests/e2e_gpu.py (hidden state atol 2^-6 and relative L2 1.5e-3, fp32 logits atol 2^-7,
identical argmax; explained in tests/test_gpu_e2e_logits.py).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import ckpt_lib
from tests import e2e_oracle as E
from tests import pa_testlib as T
from tests.e2e_gpu import check_written_cache, compare_step, make_caches, make_inputs, make_meta

pytestmark = pytest.mark.gpu
DT = O.DT_F16
HERE = os.path.dirname(os.path.abspath(__file__))
SCALES_JSON = os.path.join(HERE, "golden", "fp8_kv", "llama2-7b-kv_cache_scales.json")


def _cfg():
    from vllm_legacy_amd.model.llama_decode import LlamaConfig
    return LlamaConfig(hidden_size=1024, intermediate_size=1536, num_hidden_layers=2,
                       num_attention_heads=8, num_key_value_heads=2, head_dim=128, vocab_size=1024,
                       max_position_embeddings=512)


def _oracle_stack(model, dense, tensors):
    cfg = model.cfg
    layers = []
    for li, dl in enumerate(dense):
        m = ckpt_lib.stacked_dense(dl)
        layers.append(E.OracleLayer(T.bits(tensors[f"model.layers.{li}.input_layernorm.weight"]),
                                    T.bits(tensors[f"model.layers.{li}.post_attention_layernorm.weight"]),
                                    m["qkv"], m["o"], m["gate_up"], m["down"],
                                    float(model.layers[li].attn._kv_scale)))
    return E.OracleStack(T.bits(tensors["model.embed_tokens.weight"]), T.bits(tensors["lm_head.weight"]),
                         T.bits(tensors["model.norm.weight"]), layers,
                         T.bits(model.rotary_emb.cos_sin_cache), cfg.num_attention_heads,
                         cfg.num_key_value_heads, cfg.head_dim, cfg.rms_norm_eps)


def _run_tp(tmp_path, world, ckpt, inputs, method, kv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / f"tp{world}_{method}_{kv}.npz")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "ckpt_tp_worker.py"), str(r),
                               str(world), str(port), ckpt, inputs, out, method, kv],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, VLA_XGMI_ALLREDUCE="0", HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(world)]
    try:
        res = [p.communicate(timeout=150) + (p.returncode, ) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for _so, se, rc in res:
        assert rc == 0, se[-3000:]
    shards = [np.load(f"{out}.rank{r}.npz") for r in range(world)]
    return np.load(out), shards


@pytest.mark.parametrize("method,kv", [("gptq", "auto"), ("gptq", "fp8"), ("awq", "fp8")])
def test_safetensors_checkpoint_to_logits_one_and_two_ranks(tmp_path, method, kv):
    from safetensors.torch import save_file
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.model.llama_decode import LlamaDecodeModel
    dev = torch.device("cuda", 0)
    cfg = _cfg()
    tensors, dense = ckpt_lib.build_checkpoint(method, cfg, seed=5)
    ckpt = str(tmp_path / f"{method}.safetensors")
    save_file({k: v.contiguous() for k, v in tensors.items()}, ckpt)

    model = LlamaDecodeModel(cfg, method, kv, device=dev, seed=3)
    n = model.load_weights(ckpt_lib.read_safetensors(ckpt), method)
    per_linear = 4 if method == "gptq" else 3
    assert n == cfg.num_hidden_layers * (4 * per_linear + 2) + 3
    if method == "gptq":
        # what was loaded dequantises to the reference packer's matrix, bit for bit (before the
        # first forward pass shuffles it)
        lin = model.layers[1].down_proj
        got = ops.gptq_dequantize(lin.qweight, lin.qzeros, lin.scales, lin.g_idx)
        assert np.array_equal(T.bits(got), dense[1]["mlp.down_proj"])

    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64] * 3, dtype=np.int32)
    num_blocks, block_tables, positions, slot_mapping, input_ids = make_inputs(cfg, seq_lens)
    kv_caches = make_caches(model, ops, num_blocks, kv, dev)
    caches_before = [T.bits(c.clone()) for c in kv_caches]
    meta = make_meta(dev, seq_lens, block_tables, slot_mapping)
    with torch.inference_mode():
        hidden = model(torch.from_numpy(input_ids).to(dev), torch.from_numpy(positions).to(dev),
                       kv_caches, meta)
        logits = model.compute_logits(hidden).float().cpu().numpy()
    torch.cuda.synchronize()
    stack = _oracle_stack(model, dense, tensors)
    hkv, d = cfg.num_key_value_heads, cfg.head_dim
    # fp8 cache: the oracle's attention reads the bytes the GPU wrote, after they were checked
    # against its own write (tests/e2e_oracle.py::decode_step)
    after = [T.bits(c) for c in kv_caches] if kv == "fp8" else None
    want = E.decode_step(stack, input_ids, positions, block_tables, seq_lens, slot_mapping,
                         caches_before, kv, attend_to=after)
    check_written_cache(kv_caches, want, hkv, d, kv)
    compare_step(model, hidden, logits, want, f"{method} kv={kv} one rank")

    # two ranks from the same file
    keys = ["hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
            "num_key_value_heads", "head_dim", "vocab_size", "rms_norm_eps", "rope_theta",
            "max_position_embeddings"]
    inputs = str(tmp_path / "inputs.npz")
    store = {f"cache{li}": (c if kv == "fp8" else c.view(np.int16)) for li, c in enumerate(caches_before)}
    np.savez(inputs, cfg_keys=np.array(keys), cfg_vals=np.array([float(getattr(cfg, k)) for k in keys]),
             seq_lens=seq_lens, block_tables=block_tables, positions=positions,
             slot_mapping=slot_mapping, input_ids=input_ids, **store)
    tp2, shards = _run_tp(tmp_path, 2, ckpt, inputs, method, kv)
    hidden2 = T.from_bits(tp2["hidden"], DT, dev)
    # the ranks' cache shards put back together by KV head: [2, NB, Hkv, BS * D]
    after2 = []
    for li in range(cfg.num_hidden_layers):
        parts = [sh[f"cache{li}"].reshape(2, num_blocks, hkv // 2, -1) for sh in shards]
        after2.append(np.ascontiguousarray(np.concatenate(parts, axis=2)).reshape(2, num_blocks, -1))
    want2 = want if kv != "fp8" else E.decode_step(stack, input_ids, positions, block_tables, seq_lens,
                                                   slot_mapping, caches_before, kv, attend_to=after2)
    check_written_cache([T.from_bits(c, DT, dev) if kv != "fp8" else torch.from_numpy(c).to(dev)
                         for c in after2], want2, hkv, d, kv)
    compare_step(model, hidden2, tp2["logits"], want2, f"{method} kv={kv} two ranks vs oracle")
    # and against the one-rank run itself
    h1 = O.from_bits(T.bits(hidden), DT).astype(np.float64)
    h2 = O.from_bits(tp2["hidden"], DT).astype(np.float64)
    print(f"two ranks vs one rank: hidden max|diff| {np.abs(h1 - h2).max():.3e}, logits "
          f"{np.abs(tp2['logits'] - logits).max():.3e}")
    if kv != "fp8":     # (with an fp8 cache the two runs may have stored different fp8 steps)
        assert np.abs(h1 - h2).max() <= 2.0**-7
        assert (tp2["logits"].argmax(-1) == logits.argmax(-1)).all()


def test_kv_cache_scales_json_reaches_the_kernels():
    """The reference's llama2-7b kv_cache_scales.json on a 32-layer stack (small widths)."""
    import json
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
    from tests.e2e_gpu import awq_dense, normalise_linears, oracle_stack
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=32,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=64, vocab_size=512,
                      max_position_embeddings=512)
    model = LlamaDecodeModel(cfg, "awq", "fp8", device=dev, seed=9)
    normalise_linears(model)
    assert model.load_kv_cache_scales(SCALES_JSON) == 32
    table = json.load(open(SCALES_JSON))["kv_cache"]["scaling_factor"]["0"]
    for li, layer in enumerate(model.layers):
        # used as written: the cache format is e4m3fn, the one the file declares (no doubling)
        assert layer.attn._kv_scale == pytest.approx(float(table[str(li)]), rel=0, abs=0)
    assert len({l.attn._kv_scale for l in model.layers}) > 16

    hq, hkv, d, bs = 4, 2, 64, 16
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64], dtype=np.int32)
    B = len(seq_lens)
    num_blocks, block_tables, positions, slot_mapping, input_ids = make_inputs(cfg, seq_lens)
    meta = make_meta(dev, seq_lens, block_tables, slot_mapping)

    # ---- layer level, identical inputs on both sides: the fused rotary + cache write and the
    # attention of three layers with their loaded factor; bytes exact, outputs to fp8 tolerance
    g = torch.Generator(device=dev).manual_seed(2)
    cos_sin = T.bits(model.rotary_emb.cos_sin_cache)
    for li in (0, 5, 31):
        layer = model.layers[li]
        s = layer.attn._kv_scale
        assert s != 1.0
        # values a calibrated factor is meant for: amax ~ 448 * s
        amp = 300.0 * s
        cache = make_caches(model, ops, num_blocks, "fp8", dev, seed=li, amplitude=amp)[li]
        cache0 = T.bits(cache.clone())
        qkv = torch.empty((B, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-amp, amp, generator=g)
        q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
        q_in, k_in, v_in = (T.bits(t.contiguous()) for t in (q, k, v))
        kc_t, vc_t = PagedAttention.split_kv_cache(cache, hkv, d)
        ops.rotary_reshape_and_cache(torch.from_numpy(positions).to(dev), q, k, v, d,
                                     model.rotary_emb.cos_sin_cache, True, kc_t, vc_t,
                                     meta.slot_mapping, "fp8", s, None)
        out = layer.attn(q, k, v, cache, meta, kv_cached=True)
        torch.cuda.synchronize()
        qr, kr = O.rotary_embedding(positions, q_in, k_in, d, cos_sin, True, DT)
        kc = np.ascontiguousarray(cache0[0]).reshape(num_blocks, hkv, d // 16, bs, 16)
        vc = np.ascontiguousarray(cache0[1]).reshape(num_blocks, hkv, d, bs)
        kc1, vc1 = kc.copy(), vc.copy()
        O.reshape_and_cache(kr.reshape(B, hkv, d), v_in.reshape(B, hkv, d), kc, vc, slot_mapping, DT,
                            O.KV_FP8, s)
        O.reshape_and_cache(kr.reshape(B, hkv, d), v_in.reshape(B, hkv, d), kc1, vc1, slot_mapping, DT,
                            O.KV_FP8, 1.0)
        assert np.array_equal(T.bits(kc_t), kc) and np.array_equal(T.bits(vc_t), vc), li
        assert not np.array_equal(kc, kc1) and not np.array_equal(vc, vc1)    # the factor matters
        want = O.paged_attention_v1(qr.reshape(B, hq, d), kc, vc, block_tables, seq_lens, d**-0.5, DT,
                                    O.KV_FP8, s)
        wrong = O.paged_attention_v1(qr.reshape(B, hq, d), kc, vc, block_tables, seq_lens, d**-0.5, DT,
                                     O.KV_FP8, 1.0)
        got = O.from_bits(T.bits(out), DT).reshape(B, hq, d)
        # the reference's fp8 attention tolerance (atol 1e-2 on values of magnitude ~ 1 / sqrt(D),
        # tests/kernels/test_attention.py:271-273), relative to this test's value range
        np.testing.assert_allclose(got, O.from_bits(want, DT), atol=1e-2 * amp * d**0.5, rtol=1e-5)
        assert np.abs(got - O.from_bits(wrong, DT)).max() > 10 * np.abs(got - O.from_bits(want, DT)).max()

    # ---- the whole 32-layer step with the loaded factors against the oracle pipeline
    kv_caches = make_caches(model, ops, num_blocks, "fp8", dev, amplitude=1.0)
    caches_before = [T.bits(c.clone()) for c in kv_caches]
    with torch.inference_mode():
        hidden = model(torch.from_numpy(input_ids).to(dev), torch.from_numpy(positions).to(dev),
                       kv_caches, meta)
        logits = model.compute_logits(hidden).float().cpu().numpy()
    torch.cuda.synchronize()
    want = E.decode_step(oracle_stack(model, awq_dense), input_ids, positions, block_tables, seq_lens,
                         slot_mapping, caches_before, "fp8", attend_to=[T.bits(c) for c in kv_caches])
    # sixteen times the depth of the two-layer stacks the default bars are set for: last-place
    # differences add up as a random walk over the layers, four times as far
    check_written_cache(kv_caches, want, hkv, d, "fp8", [l.attn._kv_scale for l in model.layers],
                        drift_atol=1.6e-2, max_frac=0.3)
    compare_step(model, hidden, logits, want, "32 layers with kv_cache_scales.json",
                 hidden_atol=2.0**-5, hidden_rel_l2=6e-3, logit32_atol=2.0**-6, min_clear=0.25)
