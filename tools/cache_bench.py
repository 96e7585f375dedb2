#!/usr/bin/env python3
"""Kernel study: time the cache ops on Llama-3-8B shapes (graph replay, HIP events), print JSON.

    python3 tools/cache_bench.py

reshape_and_cache at 256 decode tokens (fp16 -> fp8), copy_blocks of 512 block pairs over 32
layers, swap_blocks of 512 blocks device -> pinned host and back, convert_fp8 over 1 GiB.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def timed(fn, iters=20, graph=True):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(iters):
                    fn()
            run = g.replay
        else:
            def run():
                for _ in range(iters):
                    fn()
        run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            run()
            e1.record(s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2] * 1e3  # us


def main():
    dev = torch.device("cuda", 0)
    hkv, d, bs, nb, layers = 8, 128, 16, 8192, 32
    res = {}
    kc = torch.zeros((nb, hkv, d // 16, bs, 16), dtype=torch.uint8, device=dev)
    vc = torch.zeros((nb, hkv, d, bs), dtype=torch.uint8, device=dev)
    t = 256
    key = torch.randn((t, hkv, d), dtype=torch.float16, device=dev)
    val = torch.randn((t, hkv, d), dtype=torch.float16, device=dev)
    slots = torch.randperm(nb * bs, device=dev)[:t].to(torch.int64)
    us = timed(lambda: ops.reshape_and_cache(key, val, kc, vc, slots, "fp8", 1.0))
    b = 2 * t * hkv * d * 3 + 8 * t
    res["reshape_and_cache"] = {"us": round(us, 2), "bytes": b, "GBps": round(b / us / 1e3, 1)}

    kcs = [torch.zeros((nb, hkv * d * bs), dtype=torch.uint8, device=dev) for _ in range(layers)]
    vcs = [torch.zeros((nb, hkv * d * bs), dtype=torch.uint8, device=dev) for _ in range(layers)]
    pairs = 512
    perm = torch.randperm(nb, device=dev)
    mapping = torch.stack([perm[:pairs], perm[pairs:2 * pairs]], dim=1).to(torch.int64).contiguous()
    us = timed(lambda: ops.copy_blocks(kcs, vcs, mapping), iters=5)
    b = 2 * 2 * layers * pairs * hkv * d * bs
    res["copy_blocks"] = {"us": round(us, 1), "bytes": b, "GBps": round(b / us / 1e3, 1)}

    host = torch.empty((nb, hkv * d * bs), dtype=torch.uint8).pin_memory()
    runs = torch.arange(pairs, dtype=torch.int64)
    contiguous = torch.stack([runs, runs], dim=1)
    scattered = torch.stack([torch.randperm(nb)[:pairs], torch.randperm(nb)[:pairs]], dim=1).to(torch.int64)
    b = pairs * hkv * d * bs
    for name, m in (("contiguous", contiguous), ("scattered", scattered)):
        us = timed(lambda: ops.swap_blocks(kcs[0], host, m), iters=3, graph=False)
        res[f"swap_out_{name}"] = {"us": round(us, 1), "bytes": b, "GBps": round(b / us / 1e3, 1)}
        us = timed(lambda: ops.swap_blocks(host, kcs[0], m), iters=3, graph=False)
        res[f"swap_in_{name}"] = {"us": round(us, 1), "bytes": b, "GBps": round(b / us / 1e3, 1)}

    n = 1 << 30
    src = torch.empty(n, dtype=torch.float16, device=dev).uniform_(-1, 1)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    us = timed(lambda: ops.convert_fp8(dst, src), iters=3)
    res["convert_fp8_encode"] = {"us": round(us, 1), "bytes": 3 * n, "GBps": round(3 * n / us / 1e3, 1)}
    us = timed(lambda: ops.convert_fp8(src, dst), iters=3)
    res["convert_fp8_decode"] = {"us": round(us, 1), "bytes": 3 * n, "GBps": round(3 * n / us / 1e3, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
