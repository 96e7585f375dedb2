#!/usr/bin/env python3
"""Kernel study: HIP prompt attention vs torch SDPA on Llama-3-8B head shapes (HIP events)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def timed(fn, iters=15, warmup=10):
    # untimed launches first: the first ones of a process run 10-20 % slow (clock ramp)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda", 0)
    hq, hkv, d = 32, 8, 128
    res = {}
    for nseq, s in ((8, 4096), (32, 1024), (4, 8192)):
        t = nseq * s
        qkv = torch.empty((t, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1)
        q, k, v = [x.view(t, -1, d) for x in qkv.split([hq * d, hkv * d, hkv * d], dim=-1)]
        out = torch.empty((t, hq, d), dtype=torch.float16, device=dev)
        cu = torch.arange(0, t + 1, s, dtype=torch.int32, device=dev)
        ms = timed(lambda: ops.prefill_attention(out, q, k, v, cu, s, d**-0.5))
        flops = 4.0 * nseq * hq * d * s * (s + 1) / 2

        def sdpa():
            for i in range(nseq):
                qs = q[i * s:(i + 1) * s].transpose(0, 1)
                ks = k[i * s:(i + 1) * s].transpose(0, 1).repeat_interleave(hq // hkv, dim=0)
                vs = v[i * s:(i + 1) * s].transpose(0, 1).repeat_interleave(hq // hkv, dim=0)
                torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=True, scale=d**-0.5)
        ms2 = timed(sdpa, iters=3, warmup=1)
        res[f"{nseq}x{s}"] = {"hip_ms": round(ms, 3), "hip_TFLOPs": round(flops / ms / 1e9, 1),
                              "sdpa_ms": round(ms2, 3), "sdpa_TFLOPs": round(flops / ms2 / 1e9, 1)}
    # prompt continuing a cached prefix: `s` new tokens after `ctx` cached ones
    bs = 16
    for nseq, ctx, s, kvd in ((8, 3072, 1024, "auto"), (8, 3072, 1024, "fp8"), (32, 4032, 64, "fp8"),
                              (4, 0, 8192, "auto")):
        t = nseq * s
        qkv = torch.empty((t, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1)
        q, k, v = [x.view(t, -1, d) for x in qkv.split([hq * d, hkv * d, hkv * d], dim=-1)]
        out = torch.empty((t, hq, d), dtype=torch.float16, device=dev)
        mb = max(1, (ctx + bs - 1) // bs)
        nb = nseq * mb
        if kvd == "fp8":
            kc = torch.randint(0, 120, (nb, hkv, d // 16, bs, 16), dtype=torch.uint8, device=dev)
            vc = torch.randint(0, 120, (nb, hkv, d, bs), dtype=torch.uint8, device=dev)
        else:
            kc = torch.empty((nb, hkv, d // 8, bs, 8), dtype=torch.float16, device=dev).uniform_(-1, 1)
            vc = torch.empty((nb, hkv, d, bs), dtype=torch.float16, device=dev).uniform_(-1, 1)
        bt = torch.randperm(nb, device=dev).to(torch.int32).view(nseq, mb)
        qs = torch.arange(0, t + 1, s, dtype=torch.int32, device=dev)
        sl = torch.full((nseq,), ctx + s, dtype=torch.int32, device=dev)
        cl = torch.full((nseq,), ctx, dtype=torch.int32, device=dev)
        ms = timed(lambda: ops.prefix_prefill_attention(out, q, k, v, kc, vc, bt, qs, sl, cl, s, d**-0.5,
                                                        None, None, kvd, 1.0))
        flops = 4.0 * nseq * hq * d * (s * ctx + s * (s + 1) / 2)
        res[f"prefix {nseq}x({ctx}+{s}) {kvd}"] = {"hip_ms": round(ms, 3),
                                                   "hip_TFLOPs": round(flops / ms / 1e9, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
