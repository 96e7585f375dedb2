#!/usr/bin/env python3
"""Kernel study: time paged_attention_v1/v2 alone on the metric shape (HIP events), print JSON.

    python3 tools/attn_bench.py [--kv fp8|auto] [--batch 256] [--seq 4096] [--iters 20] [--layers 2]

`--layers` caches (2 GiB each at the default shape) are cycled so that no launch finds its data
in the 256 MiB Infinity Cache.  Used with rocprofv3 for the per-kernel traces and PMC passes.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kv", default="fp8")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--version", default="v1")
    ap.add_argument("--ragged", action="store_true", help="sequence lengths U(1, seq)")
    ap.add_argument("--sorted", action="store_true", help="with --ragged: longest sequences first")
    ap.add_argument("--order", action="store_true", help="pass ops.make_seq_order(seq_lens) to the kernel")
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--generic", action="store_true", help="force the coverage kernel")
    ap.add_argument("--graph", action="store_true",
                    help="time the launches replayed from one HIP graph (no host launch cost)")
    ap.add_argument("--kv-policy", type=int, default=0,
                    help="K/V load policy: 0 by size, 1 nontemporal, 2 default cache policy")
    ap.add_argument("--pool", type=int, default=0,
                    help="the reference harness's set-up (benchmarks/kernels/benchmark_paged_attention.py"
                         ":11, :59-63): a cache of this many blocks, block tables drawn from it with "
                         "repeats, so the data stays cache resident (0: unique blocks, real HBM traffic)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    B, S, BS, D = a.batch, a.seq, a.block, a.head_size
    if a.generic:
        ops.set_attention_path(1)
    if a.kv_policy:
        ops.set_kv_load_policy(a.kv_policy)
    nb = a.pool if a.pool else B * (S // BS)
    g = torch.Generator(device=dev).manual_seed(0)
    scale = D**-0.5
    fp8 = a.kv == "fp8"
    x = 16 if fp8 else 8
    caches = []
    for _ in range(a.layers):
        if fp8:
            kc = torch.empty((nb, a.hkv, D // x, BS, x), dtype=torch.uint8, device=dev)
            vc = torch.empty((nb, a.hkv, D, BS), dtype=torch.uint8, device=dev)
            for c in (kc, vc):
                flat = c.view(-1)
                for i in range(0, flat.numel(), 1 << 28):
                    n = min(1 << 28, flat.numel() - i)
                    t = torch.empty(n, dtype=torch.float16, device=dev).uniform_(-scale, scale, generator=g)
                    ops.convert_fp8(flat[i:i + n], t)
        else:
            kc = torch.empty((nb, a.hkv, D // x, BS, x), dtype=dt, device=dev).uniform_(-scale, scale, generator=g)
            vc = torch.empty((nb, a.hkv, D, BS), dtype=dt, device=dev).uniform_(-scale, scale, generator=g)
        caches.append((kc, vc))
    q = torch.empty((B, a.hq, D), dtype=dt, device=dev).uniform_(-scale, scale, generator=g)
    out = torch.empty_like(q)
    if a.pool:
        bt = torch.randint(0, nb, (B, S // BS), generator=g, device=dev, dtype=torch.int32)
    else:
        bt = torch.randperm(nb, generator=g, device=dev).to(torch.int32).view(B, S // BS).contiguous()
    if a.ragged:
        sl = torch.randint(1, S + 1, (B, ), generator=g, device=dev, dtype=torch.int32)
        if a.sorted:
            sl = torch.sort(sl, descending=True).values.contiguous()
    else:
        sl = torch.full((B, ), S, dtype=torch.int32, device=dev)
    P = (S + 511) // 512
    tmp = torch.empty((B, a.hq, P, D), dtype=dt, device=dev)
    es = torch.empty((B, a.hq, P), dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    kvd = "fp8" if fp8 else "auto"
    order = ops.make_seq_order(sl) if a.order else None

    def run(i):
        kc, vc = caches[i % a.layers]
        if a.version == "v1":
            ops.paged_attention_v1(out, q, kc, vc, a.hkv, scale, bt, sl, BS, S, None, kvd, 1.0, seq_order=order)
        else:
            ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, a.hkv, scale, bt, sl, BS, S, None, kvd, 1.0, seq_order=order)

    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    if a.graph:
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for i in range(a.iters):
                run(i)
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g_.replay()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / a.iters)
        a.iters = 0
    evs = []
    for i in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run(i)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(ts if a.graph else [s.elapsed_time(e) for s, e in evs])
    e_kv = 1 if fp8 else 2
    tokens = int(sl.sum().item())
    algo = 2 * a.hkv * D * tokens * e_kv + 4 * B * (S // BS) + 2 * B * a.hq * D * 2 + 8 * B
    med = ts[len(ts) // 2]
    print(json.dumps({"version": a.version, "kv": a.kv, "block": BS, "head_size": D, "generic": a.generic, "batch": B, "seq": S, "ragged": a.ragged, "pool": a.pool,
                      "ms_median": round(med, 4), "ms_min": round(ts[0], 4), "ms_mean": round(sum(ts) / len(ts), 4),
                      "algo_bytes": algo, "GBps_median": round(algo / med / 1e6, 1),
                      "frac_of_8TBps": round(algo / med / 1e6 / 8000, 4)}))


if __name__ == "__main__":
    main()
