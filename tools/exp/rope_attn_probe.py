#!/usr/bin/env python3
"""Kernel study: the decode step's rotary + cache write + paged attention as two launches and as
one (ops.rope_paged_attention), Llama-3-8B heads, fp8 cache, unique blocks; HIP events, eager.

    python3 tools/exp/rope_attn_probe.py [--batch 256] [--seq 4096] [--slabs 2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--slabs", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, S, BS, D, hq, hkv = a.batch, a.seq, 16, 128, 32, 8
    g = torch.Generator(device=dev).manual_seed(0)
    nb = B * (S // BS)
    caches = []
    for _ in range(2):
        kc = torch.randint(0, 120, (nb, hkv, D // 16, BS, 16), dtype=torch.uint8, device=dev, generator=g)
        vc = torch.randint(0, 120, (nb, hkv, D, BS), dtype=torch.uint8, device=dev, generator=g)
        caches.append((kc, vc))
    bt = torch.randperm(nb, generator=g, device=dev).to(torch.int32).view(B, S // BS).contiguous()
    sl = torch.full((B, ), S, dtype=torch.int32, device=dev)
    pos = torch.full((B, ), S - 1, dtype=torch.int64, device=dev)
    slots = bt[:, (S - 1) // BS].to(torch.int64) * BS + (S - 1) % BS
    n = (hq + 2 * hkv) * D
    cos = torch.randn((S, D), device=dev, generator=g).half()
    qkv = torch.randn((B, n), device=dev, generator=g).half()
    slabs = None
    if a.slabs:
        ws = (torch.randn((a.slabs, B, n), device=dev, generator=g) * 0.5).contiguous()
        slabs = ops.SplitKSlabs(ws.view(torch.uint8).view(-1), a.slabs, B, n)
    q, k, v = qkv.split([hq * D, hkv * D, hkv * D], dim=-1)
    out = torch.empty((B, hq, D), dtype=torch.float16, device=dev)
    scale = D ** -0.5

    def separate(i):
        kc, vc = caches[i % 2]
        ops.rotary_reshape_and_cache(pos, q, k, v, D, cos, True, kc, vc, slots, "fp8", 1.0, slabs)
        ops.paged_attention_v1(out, q.reshape(B, hq, D), kc, vc, hkv, scale, bt, sl, BS, S, None, "fp8", 1.0)

    def fused(i):
        kc, vc = caches[i % 2]
        ops.rope_paged_attention(pos, q, k, v, D, cos, True, kc, vc, slots, "fp8", 1.0, hkv, scale, bt, sl, S,
                                 slabs=slabs)

    def split(i):
        kc, vc = caches[i % 2]
        ops.rotary_reshape_and_cache(pos, q, k, v, D, cos, True, kc, vc, slots, "fp8", 1.0, slabs, skip_query=True)
        ops.rope_paged_attention(pos, q, k, v, D, cos, True, kc, vc, slots, "fp8", 1.0, hkv, scale, bt, sl, S,
                                 slabs=slabs, write_kv=False)

    res = {}
    for name, fn in (("separate_us", separate), ("fused_us", fused), ("kv_launch_plus_fused_q_us", split)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for i in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[name] = round(ts[len(ts) // 2], 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
