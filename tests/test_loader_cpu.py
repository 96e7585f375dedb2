"""Host-side wire formats: kv_cache_scales.json (checked against the two fixture files the
reference's own tests hold, tests/fp8_kv/*/kv_cache_scales.json, copied as data under
tests/golden/fp8_kv/) and AWQ / GPTQ tensor-parallel sharding (checked through the oracle's
dequantisation: a rank's shard must dequantise to the matching slice of the full matrix)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from vllm_legacy_amd.model_loader import (kv_cache_scales_loader, load_kv_cache_scales,
                                          load_llama_layer_weights, param_specs, shard_column,
                                          shard_row)

FIX = os.path.join(os.path.dirname(__file__), "golden", "fp8_kv")


def test_kv_scales_reference_fixtures():
    got = dict(kv_cache_scales_loader(os.path.join(FIX, "llama2-7b-kv_cache_scales.json"), 0, 1, 32,
                                      "llama"))
    assert len(got) == 32 and got[0] == pytest.approx(0.0152239128947258)
    assert all(isinstance(k, int) and isinstance(v, float) for k, v in got.items())
    got70 = dict(kv_cache_scales_loader(os.path.join(FIX, "llama2-70b-kv_cache_scales.json"), 0, 1,
                                        80, "llama"))
    assert sorted(got70) == list(range(80))


@pytest.mark.parametrize("mutate,why", [
    (lambda d: d.update(model_type="opt"), "model type"),
    (lambda d: d["kv_cache"].update(dtype="float8_e5m2"), "dtype"),
    (lambda d: d["kv_cache"]["scaling_factor"]["0"].pop("3"), "missing layer"),
    (lambda d: d["kv_cache"]["scaling_factor"].update({"1": {}}), "tp size"),
])
def test_kv_scales_rejections_fall_back_to_empty(tmp_path, mutate, why):
    d = json.load(open(os.path.join(FIX, "llama2-7b-kv_cache_scales.json")))
    mutate(d)
    p = tmp_path / "bad.json"
    p.write_text(json.dumps(d))
    assert list(kv_cache_scales_loader(str(p), 0, 1, 32, "llama")) == [], why
    assert list(kv_cache_scales_loader(str(tmp_path / "absent.json"), 0, 1, 32, "llama")) == []
    (tmp_path / "garbage.json").write_text("{not json")
    assert list(kv_cache_scales_loader(str(tmp_path / "garbage.json"), 0, 1, 32, "llama")) == []


def test_kv_scales_reach_the_attention_layers():
    class A:
        _kv_scale = 1.0

    class L:
        def __init__(self):
            self.attn = A()

    layers = [L() for _ in range(32)]
    n = load_kv_cache_scales(layers, os.path.join(FIX, "llama2-7b-kv_cache_scales.json"), 0, 1)
    assert n == 32 and layers[5].attn._kv_scale == pytest.approx(0.0433175228536129)
    # the factor is used as written (no doubling: the cache format is e4m3fn, as the file says)
    assert layers[0].attn._kv_scale == pytest.approx(0.0152239128947258)


def _rand_i32(rng, shape):
    return torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=shape, dtype=np.int64).astype(np.int32))


def _awq(rng, k, n, g=128):
    return {"qweight": _rand_i32(rng, (k, n // 8)), "qzeros": _rand_i32(rng, (k // g, n // 8)),
            "scales": torch.from_numpy(rng.uniform(0.005, 0.015, (k // g, n)).astype(np.float16))}


def _gptq(rng, k, n, g=128):
    return {"qweight": _rand_i32(rng, (k // 8, n)), "qzeros": _rand_i32(rng, (k // g, n // 8)),
            "scales": torch.from_numpy(rng.uniform(0.005, 0.015, (k // g, n)).astype(np.float16)),
            "g_idx": torch.arange(k, dtype=torch.int32) // g}


def _deq(method, t):
    s = t["scales"].numpy().view(np.uint16)
    if method == "awq":
        return O.awq_dequantize(t["qweight"].numpy(), s, t["qzeros"].numpy())
    return O.gptq_dequantize(t["qweight"].numpy(), t["qzeros"].numpy(), s, t["g_idx"].numpy())


@pytest.mark.parametrize("method", ["awq", "gptq"])
@pytest.mark.parametrize("tp", [2, 4])
def test_column_and_row_shards_dequantise_to_slices(method, tp):
    rng = np.random.default_rng(tp)
    k, n = 512, 256
    full = (_awq if method == "awq" else _gptq)(rng, k, n)
    w = _deq(method, full)
    for rank in range(tp):
        col = {p: shard_column(s, full[p], rank, tp) for p, s in param_specs(method).items()}
        assert np.array_equal(_deq(method, col), w[:, rank * n // tp:(rank + 1) * n // tp])
        row = {p: shard_row(s, full[p], rank, tp)
               for p, s in param_specs(method, row_parallel=True).items()}
        if method == "gptq":
            row["g_idx"] = row["g_idx"] - row["g_idx"].min()
        assert np.array_equal(_deq(method, row), w[rank * k // tp:(rank + 1) * k // tp])


@pytest.mark.parametrize("method", ["awq", "gptq"])
@pytest.mark.parametrize("tp,num_kv_heads", [(2, 2), (4, 2), (1, 2)])
def test_llama_layer_from_safetensors(tmp_path, method, tp, num_kv_heads):
    from safetensors.torch import save_file, safe_open
    rng = np.random.default_rng(7)
    hidden, inter, hq, d = 256, 512, 8, 64
    make = _awq if method == "awq" else _gptq
    shapes = {"self_attn.q_proj": (hidden, hq * d), "self_attn.k_proj": (hidden, num_kv_heads * d),
              "self_attn.v_proj": (hidden, num_kv_heads * d), "self_attn.o_proj": (hq * d, hidden),
              "mlp.gate_proj": (hidden, inter), "mlp.up_proj": (hidden, inter),
              "mlp.down_proj": (inter, hidden)}
    ckpt, dense = {}, {}
    for name, (k, n) in shapes.items():
        t = make(rng, k, n)
        dense[name] = O.from_bits(_deq(method, t), O.DT_F16)
        for p, v in t.items():
            ckpt[f"model.layers.3.{name}.{p}"] = v
    path = str(tmp_path / "layer.safetensors")
    save_file(ckpt, path)
    with safe_open(path, framework="pt") as f:
        tensors = {key: f.get_tensor(key) for key in f.keys()}
    for rank in range(tp):
        got = load_llama_layer_weights(tensors, 3, method, rank, tp, hq, num_kv_heads, d)
        w = {name: O.from_bits(_deq(method, t), O.DT_F16) for name, t in got.items()}
        hq_l = hq // tp
        kv_l = max(1, num_kv_heads // tp)
        kv_id = rank // max(1, tp // num_kv_heads)
        want_qkv = np.concatenate([
            dense["self_attn.q_proj"][:, rank * hq_l * d:(rank + 1) * hq_l * d],
            dense["self_attn.k_proj"][:, kv_id * kv_l * d:(kv_id + 1) * kv_l * d],
            dense["self_attn.v_proj"][:, kv_id * kv_l * d:(kv_id + 1) * kv_l * d]], axis=1)
        assert np.array_equal(w["qkv_proj"], want_qkv)
        i_l = inter // tp
        assert np.array_equal(w["gate_up_proj"], np.concatenate(
            [dense["mlp.gate_proj"][:, rank * i_l:(rank + 1) * i_l],
             dense["mlp.up_proj"][:, rank * i_l:(rank + 1) * i_l]], axis=1))
        assert np.array_equal(w["o_proj"], dense["self_attn.o_proj"][rank * hq_l * d:(rank + 1) * hq_l * d])
        assert np.array_equal(w["down_proj"], dense["mlp.down_proj"][rank * i_l:(rank + 1) * i_l])
