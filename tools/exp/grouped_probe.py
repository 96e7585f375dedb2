#!/usr/bin/env python3
"""Kernel study: one grouped int4 launch at the shapes of a Mixtral-8x7B expert projection (w13:
4096 -> 28672, w2: 14336 -> 4096; 8 sets), rows spread over the sets like top-2 routing of
`--tokens` tokens, the row gather on.  Prints the launch time (HIP events, eager) as JSON.

    python3 tools/exp/grouped_probe.py [--tokens 256] [--which w13] [--iters 20]
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
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--which", default="w13", choices=["w13", "w2"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--copies", type=int, default=2, help="weight copies cycled (HBM-resident weights)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    k, n = (4096, 28672) if a.which == "w13" else (14336, 4096)
    copies = []
    for _ in range(a.copies):
        qw = torch.randint(-2**31, 2**31 - 1, (a.sets, k // 8, n), dtype=torch.int64, device=dev,
                           generator=g).to(torch.int32)
        qz = torch.randint(-2**31, 2**31 - 1, (a.sets, k // 128, n // 8), dtype=torch.int64, device=dev,
                           generator=g).to(torch.int32)
        sc = (torch.empty((a.sets, k // 128, n), device=dev).uniform_(0.5, 1.5, generator=g) * 1e-2).half()
        copies.append((qw, qz, sc))
    # top-2 routing of `tokens` tokens over the sets
    logits = torch.randn((a.tokens, a.sets), device=dev, generator=g)
    sel = torch.topk(logits, 2, dim=-1).indices.reshape(-1)
    order = torch.argsort(sel, stable=True)
    counts = torch.bincount(sel, minlength=a.sets)
    offsets = torch.zeros(a.sets + 1, dtype=torch.int32, device=dev)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    x_rows = (order // 2).to(torch.int32).contiguous()
    x = torch.randn((a.tokens, k), device=dev, generator=g).half()
    bound = min(2 * a.tokens, a.tokens)

    def run(i):
        qw, qz, sc = copies[i % a.copies]
        return ops.q4_gemm_grouped(x, qw, qz, sc, offsets, 0, bound, x_rows=x_rows)

    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(json.dumps({"which": a.which, "tokens": a.tokens, "rows": counts.tolist(),
                      "us_median": round(ts[len(ts) // 2], 1), "us_min": round(ts[0], 1)}))


if __name__ == "__main__":
    main()
